"""Latency of mlpl_arrsac_essential (host pointers) on the C3-shaped scene and a few others; the oracle's time beside it."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import pose, synth
import oracle_lib
ora = oracle_lib.load(); ctx = mpa.Context(0)
for n, frac, seed in [(5000, 0.5, 20260103), (5000, 0.3, 20260104), (5000, 0.8, 20260105), (2000, 0.95, 20260106), (8192, 0.4, 20260112)]:
    p1, p2, R, t, truth, th = synth.pose_scene(n, frac, seed=seed)
    for refine in (False, True):
        for _ in range(3):
            g = pose.arrsac_essential(p1, p2, th, refine=refine, rng_state=np.array(pose.ARRSAC_RNG_FRESH, np.uint64), ctx=ctx)
        t0 = time.perf_counter(); K = 20
        for _ in range(K):
            g = pose.arrsac_essential(p1, p2, th, refine=refine, rng_state=np.array(pose.ARRSAC_RNG_FRESH, np.uint64), ctx=ctx)
        dt = (time.perf_counter() - t0) / K
        t0 = time.perf_counter(); o = ora.arrsac_essential(p1, p2, th, refine=refine); do = time.perf_counter() - t0
        print(f"n={n} inliers={frac} refine={refine}: GPU {dt*1e3:.2f} ms  (batches {g['stats'][8]}, samples solved {g['stats'][9]}, used {g['stats'][10]}, refine status {g['stats'][11]}), "
              f"oracle {do*1e3:.1f} ms, inliers {g['n_inliers']}")
