"""One ARRSAC sample: the device estimators against the oracle's (5..7 points: run5Point on m points; 8..14: the 8-point fit)."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import pose, synth
import oracle_lib
ora = oracle_lib.load(); ctx = mpa.Context(0); ctx.set_option("solver_polish", int(sys.argv[4]) if len(sys.argv) > 4 else 0)
n, frac, seed = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])
noise = float(sys.argv[5]) if len(sys.argv) > 5 else 0.3
idx = np.array([int(x) for x in sys.argv[6].split(",")], np.int32)
p1, p2, R, t, truth, th = synth.pose_scene(n, frac, seed=seed, noise_px=noise)
Eg, vg = pose.arrsac_sample_models(p1, p2, idx, 0 if len(idx) < 8 else 1, ctx=ctx)
Eo = [np.asarray(e).reshape(3, 3) for e in ora.run5point(p1[idx], p2[idx])] if len(idx) < 8 else [ora.cv_fm_8point(p1[idx], p2[idx])[1]]
A = np.array([[a[0] * b[0], a[1] * b[0], b[0], a[0] * b[1], a[1] * b[1], b[1], a[0], a[1], 1.0] for a, b in zip(p1[idx], p2[idx])])
U, S, Vt = np.linalg.svd(A); N = Vt[-4:].T      # numpy's 4-dimensional subspace
def off_subspace(E):
    e = E.reshape(9) / np.linalg.norm(E); return np.linalg.norm(e - N @ (N.T @ e))
def cubic(E):
    E = E / np.linalg.norm(E); return max(np.abs(2 * E @ E.T @ E - np.trace(E @ E.T) * E).max(), abs(np.linalg.det(E)))
d = lambda a, b: min(np.abs(a / np.linalg.norm(a) - b / np.linalg.norm(b)).max(), np.abs(a / np.linalg.norm(a) + b / np.linalg.norm(b)).max())
print("gpu models", len(Eg), "oracle models", len(Eo))
for e in Eg:
    print("  gpu   : off the subspace %.2e, constraint residual %.2e, nearest oracle %.2e" % (off_subspace(e), cubic(e), min(d(e, x) for x in Eo)))
for e in Eo:
    print("  oracle: off the subspace %.2e, constraint residual %.2e, nearest gpu    %.2e" % (off_subspace(e), cubic(e), min(d(e, x) for x in Eg) if len(Eg) else -1))
