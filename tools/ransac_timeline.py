import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import pose, synth
ctx = mpa.Context(0)
dev = torch.device("cuda", 0)
p1, p2, R, t, mask, th = synth.pose_scene(5000, seed=20260103)
d1 = torch.from_numpy(p1).to(dev); d2 = torch.from_numpy(p2).to(dev)
dm = torch.empty(5000, dtype=torch.uint8, device=dev)
for _ in range(3):
    pose.ransac_essential_device(d1, d2, th, confidence=1.0, max_iters=20000, refit=False, seed=12345, ctx=ctx, mask_out=dm)
torch.cuda.synchronize()
ts = []
for _ in range(10):
    t0 = time.perf_counter()
    r = pose.ransac_essential_device(d1, d2, th, confidence=1.0, max_iters=20000, refit=False, seed=12345, ctx=ctx, mask_out=dm)
    ts.append(time.perf_counter() - t0)
print("wall ms per call:", [round(x * 1e3, 3) for x in ts], r["iters"], r["n_inliers"])
