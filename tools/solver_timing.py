"""Solver kernels at scale: mlpl_solve_5pt on N samples of the C3 scene (kernel times from `rocprofv3 --kernel-trace`), plus the C3
RANSAC call, for both forms of the elimination kernel.   python tools/solver_timing.py [n_samples]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import matchinglib_poselib_amd as mpa  # noqa: E402
from matchinglib_poselib_amd import pose, synth  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 41472
ctx = mpa.Context(0)
dev = torch.device("cuda", 0)
p1, p2, R, t, mask, th = synth.pose_scene(5000, seed=20260103)
d1, d2 = torch.from_numpy(p1).to(dev), torch.from_numpy(p2).to(dev)
dm = torch.empty(5000, dtype=torch.uint8, device=dev)
rng = np.random.default_rng(1)
samples = np.stack([rng.choice(5000, 5, replace=False) for _ in range(ns)]).astype(np.int32)
for w3 in (0, 1):
    ctx.set_option("solver_wave3", w3)
    for _ in range(6):
        E, nm = pose.solve_5pt(p1, p2, samples, ctx=ctx)
    call = lambda: pose.ransac_essential_device(d1, d2, th, confidence=1.0, max_iters=20000, refit=False, seed=12345, ctx=ctx, mask_out=dm)  # noqa: E731
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        r = call()
        ts.append(time.perf_counter() - t0)
    print(f"solver_wave3={w3}: C3 call median {np.median(ts) * 1e3:.3f} ms, min {min(ts) * 1e3:.3f} ms; iterations {r['iters']}, inliers {r['n_inliers']}; "
          f"models of {ns} samples {nm.sum()}", flush=True)
