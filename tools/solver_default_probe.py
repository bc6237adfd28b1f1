"""Distribution of |E_device - E_oracle| per minimal sample at the library default (solver_polish = 0): 10 000 samples of the C3 scene."""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import pose, synth

def e_dist(a, b):
    a, b = np.ravel(a) / np.linalg.norm(a), np.ravel(b) / np.linalg.norm(b)
    return min(np.abs(a - b).max(), np.abs(a + b).max())

def cubic_residual(E):
    return max(np.abs(2 * E @ E.T @ E - np.trace(E @ E.T) * E).max(), abs(np.linalg.det(E)))

ora = oracle_lib.load(); ctx = mpa.Context(0)
print("default solver_polish =", ctx.get_option("solver_polish"))
p1, p2, R, t, mask, th = synth.pose_scene(5000, seed=20260103)
samples = ora.sample_table(12345, p1, p2, 10000)
E, nm = pose.solve_5pt(p1, p2, samples, ctx=ctx)
ds, rows, cm = [], [], 0
for s in range(len(samples)):
    Eo = ora.run5point(p1[samples[s]], p2[samples[s]])
    Eg = E[s, :nm[s]]
    if len(Eo) != len(Eg):
        cm += 1; continue
    for e in Eo:
        d = min(e_dist(e, x) for x in Eg) if len(Eg) else np.inf
        ds.append(d); 
        if d > 1e-8: rows.append((s, d, cubic_residual(e), min(cubic_residual(x) for x in Eg)))
ds = np.array(ds)
print("models", len(ds), "count mismatches", cm, "max", ds.max(), "quantiles 50/99/99.9:", np.quantile(ds, [0.5, 0.99, 0.999]))
for thr in (1e-12, 1e-10, 1e-9, 1e-8, 1e-7, 1e-6, 1e-5): print("  >", thr, int((ds > thr).sum()))
for r in rows[:40]: print("  sample %d d %.2e oracle residual %.2e device residual %.2e" % r)
