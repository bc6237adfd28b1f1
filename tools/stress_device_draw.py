"""Soak: RANSAC runs whose sample tables are drawn on the device (draw_scan / draw_chain / draw_fill_kernel) against the same runs with the
host drawing them -- random correspondence counts, iteration counts (one to three passes), seeds and inlier ratios; results must be equal bit
for bit, and the device path must not fall back.   Usage: python tools/stress_device_draw.py [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import matchinglib_poselib_amd as mpa  # noqa: E402
from matchinglib_poselib_amd import pose, synth  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
ctx = mpa.Context(0)
rng = np.random.default_rng(4242)
t0 = time.time()
cases = used_dev = 0
o = np.zeros(2, np.int64)
while time.time() - t0 < budget:
    iters = int(rng.choice([4096, 5000, 8191, 12000, 20000, 32768, 32769, 40000, 70000]))
    n_min = int(np.ceil(min(iters, 32768) * 60.0 / 3072)) + 1        # the eligibility rule of the driver
    n = int(rng.integers(max(64, n_min), 9000))
    p1, p2, R, t, mask, th = synth.pose_scene(n, inlier_frac=float(rng.choice([0.15, 0.3, 0.5, 0.8])), seed=int(rng.integers(1, 1 << 30)))
    seed = int(rng.integers(0, 1 << 32))
    res = []
    for dd in (0, 1):
        ctx.set_option("ransac_device_draw", dd)
        res.append(pose.ransac_essential(p1, p2, th, confidence=1.0, max_iters=iters, refit=False, seed=seed, ctx=ctx))
        ctx.lib.mlpl_debug_ransac_draw(ctx.handle, o.ctypes.data)
        if dd:
            used_dev += int(o[1])
    a, b = res
    same = a["iters"] == b["iters"] and a["n_inliers"] == b["n_inliers"] and np.array_equal(a["mask"], b["mask"]) and \
        np.array_equal(np.asarray(a["E"]).view(np.uint64), np.asarray(b["E"]).view(np.uint64))
    if not same:
        print("MISMATCH", n, iters, seed, a["iters"], b["iters"], a["n_inliers"], b["n_inliers"], flush=True)
        sys.exit(1)
    cases += 1
ctx.set_option("ransac_device_draw", 1)
print(f"{cases} random runs in {time.time() - t0:.0f} s: device-drawn == host-drawn; device path used in {used_dev}, fallbacks {int(o[0])}")
