"""Wall time per RANSAC call (device-resident points) at the C3 throughput settings and at the reference's own settings.
Usage: python tools/ransac_timeline.py   (run under `rocprofv3 --kernel-trace --stats` for the kernel timeline)"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402,F401
import torch  # noqa: E402

import matchinglib_poselib_amd as mpa  # noqa: E402
from matchinglib_poselib_amd import pose, synth  # noqa: E402

ctx = mpa.Context(0)
dev = torch.device("cuda", 0)
p1, p2, R, t, mask, th = synth.pose_scene(5000, seed=20260103)
d1 = torch.from_numpy(p1).to(dev)
d2 = torch.from_numpy(p2).to(dev)
dm = torch.empty(5000, dtype=torch.uint8, device=dev)
for name, conf, iters, refit in (("C3 throughput (20000, 1.0)", 1.0, 20000, False), ("reference settings (1000, 0.999)", 0.999, 1000, False),
                                 ("reference settings + refit", 0.999, 1000, True)):
    call = lambda: pose.ransac_essential_device(d1, d2, th, confidence=conf, max_iters=iters, refit=refit, seed=12345, ctx=ctx,  # noqa: E731
                                                mask_out=dm)
    for _ in range(3):
        call()
    torch.cuda.synchronize()
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        r = call()
        ts.append(time.perf_counter() - t0)
    print(f"{name}: median {np.median(ts) * 1e3:.3f} ms, min {min(ts) * 1e3:.3f} ms; iterations {r['iters']}, inliers {r['n_inliers']}", flush=True)
pose.ransac_essential(p1, p2, th, confidence=0.999, max_iters=1000, refit=True, seed=12345, ctx=ctx)  # first call: workspace growth
for refit in (False, True):
    ts = []
    for _ in range(20):
        t0 = time.perf_counter()
        r = pose.ransac_essential(p1, p2, th, confidence=0.999, max_iters=1000, refit=refit, seed=12345, ctx=ctx)
        ts.append(time.perf_counter() - t0)
    print(f"host API refit={refit}: median {np.median(ts) * 1e3:.3f} ms, min {min(ts) * 1e3:.3f}, max {max(ts) * 1e3:.3f}")
