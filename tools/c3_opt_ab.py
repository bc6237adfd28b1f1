"""C3 call (5000 correspondences, 20 000 iterations, confidence 1.0, device-resident points): same-process A/B of one integer option
(mlpl_set_option), values alternately and three times; results must be identical.
python tools/c3_opt_ab.py option v0,v1[,...]"""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import pose, synth

opt, vals = sys.argv[1], [int(v) for v in sys.argv[2].split(",")]
ctx = mpa.Context(0)
dev = torch.device("cuda", 0)
p1, p2, R, t, mask, th = synth.pose_scene(5000, seed=20260103)
d1, d2 = torch.from_numpy(p1).to(dev), torch.from_numpy(p2).to(dev)
dm = torch.empty(5000, dtype=torch.uint8, device=dev)
call = lambda: pose.ransac_essential_device(d1, d2, th, confidence=1.0, max_iters=20000, refit=False, seed=12345, ctx=ctx, mask_out=dm)
ref = None
for rnd in range(3):
    for v in vals:
        ctx.set_option(opt, v)
        for _ in range(10):
            r = call()
        torch.cuda.synchronize()
        ts = []
        for _ in range(50):
            t0 = time.perf_counter(); r = call(); ts.append(time.perf_counter() - t0)
        key = (r["iters"], r["n_inliers"], r["E"].tobytes(), dm.cpu().numpy().tobytes())
        ref = ref or key
        print(f"round {rnd} {opt}={v}: C3 call median {np.median(ts) * 1e3:.4f} ms, min {min(ts) * 1e3:.4f} ms ({20000 / np.median(ts) / 1e6:.2f} M hyp/s); same result: {key == ref}", flush=True)
ctx.close()
