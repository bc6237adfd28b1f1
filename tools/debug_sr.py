"""Diagnostics: where does a StereoRefine sequence's GPU run leave the CPU state machine?  Replays the CPU state machine up to a frame,
then runs the pool's robust estimation on both sides and compares the winning models."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
import test_gpu_stereo_refine as T
from stereo_refine_oracle import StereoRefineOracle
from matchinglib_poselib_amd import pose

name, upto = sys.argv[1], int(sys.argv[2])
o = oracle_lib.load()
cfg, method, dist, frames = T.sequence(name)
sr = StereoRefineOracle(o, cfg, T.K, T.K, np.zeros(8), np.zeros(8), 777)
calls = []
orig = sr.robust
def spy(a, b):
    calls.append((a.copy(), b.copy()))
    return orig(a, b)
sr.robust = spy
for kp1, kp2, dd in frames[: upto + 1]:
    sr.add(kp1, kp2, dd)
a, b = calls[-1]
print("last robust call on", len(a), "correspondences")
r = o.ransac_essential(a, b, sr.th, 0.999, 1000, False, 777, trace=True)
g = pose.ransac_essential(a, b, sr.th, 0.999, 1000, refit=False, seed=777)
print("oracle inl", r["n_inliers"], "iters", r["iters"], " gpu inl", g["n_inliers"], "iters", g["iters"])
Eg = g["E"]
print("E diff", min(np.abs(Eg - r["E"]).max(), np.abs(Eg + r["E"]).max()))
cnt_o, m_o, e_o = o.get_inliers_strict(a, b, r["E"], sr.th2)
cnt_g, m_g, e_g = o.get_inliers_strict(a, b, Eg, sr.th2)
d = np.flatnonzero(m_o != m_g)
print("mask differences at", d, "errors/th2 oracle", e_o[d] / sr.th2, "gpu", e_g[d] / sr.th2)
from test_gpu_baseline_configs import cubic_residual
print("constraint residual: oracle", cubic_residual(r["E"]), "gpu", cubic_residual(Eg))
from matchinglib_poselib_amd._lib import default_context
ctx = default_context()
ctx.lib.mlpl_set_option(ctx.handle, b"solver_polish", 0)
g0 = pose.ransac_essential(a, b, sr.th, 0.999, 1000, refit=False, seed=777)
print("polish off: inl", g0["n_inliers"], "E diff to oracle", np.abs(g0["E"] - r["E"]).max(), "residual", cubic_residual(g0["E"]))
