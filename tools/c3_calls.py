"""C3 (5000 correspondences, 20 000 iterations, confidence 1.0) calls one at a time with a pause between them, for a kernel timeline:
    rocprofv3 --kernel-trace --output-format csv -d DIR -- python3 tools/c3_calls.py && python tools/kernel_timeline.py DIR -3 300"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import pose, synth

ctx = mpa.Context(0)
dev = torch.device("cuda", 0)
p1, p2, R, t, mask, th = synth.pose_scene(5000, seed=20260103)
d1, d2 = torch.from_numpy(p1).to(dev), torch.from_numpy(p2).to(dev)
dm = torch.empty(5000, dtype=torch.uint8, device=dev)
ts = []
for i in range(40):
    torch.cuda.synchronize()
    time.sleep(0.002)
    t0 = time.perf_counter()
    r = pose.ransac_essential_device(d1, d2, th, confidence=1.0, max_iters=20000, refit=False, seed=12345, ctx=ctx, mask_out=dm)
    ts.append(time.perf_counter() - t0)
print(f"C3 call: median {np.median(ts[5:]) * 1e3:.3f} ms, min {min(ts[5:]) * 1e3:.3f} ms; iterations {r['iters']}, inliers {r['n_inliers']}")
