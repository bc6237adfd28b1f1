"""C5 with RANSAC (512 pairs of 8192 keypoints through mlpl_pair_pose_batch_dev, one call at a time): same-process A/B of one integer option,
values alternately and three times; the records must be identical.  Prints the step time and the counting kernels' time per step (HIP events
inside the library).   python tools/c5_ransac_opt_ab.py option v0,v1[,...]"""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import batch, synth

opt, vals = sys.argv[1], [int(v) for v in sys.argv[2].split(",")]
total, nk = 512, 8192
dev = torch.device("cuda:0")
ctx = mpa.Context(0)
lib = ctx.lib
sps = [synth.stereo_pair(nk, seed=20260200 + i, unmatched_frac=0.30 + 0.02 * (i % 8)) for i in range(8)]
K = sps[0]["K"]
stk = [torch.from_numpy(np.stack([sps[i % 8][k] for i in range(total)])).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")]
seeds = [100 + i for i in range(total)]
call = lambda: batch.process_pairs_batched(ctx, *stk, K, K, seeds)
ref = None
for rnd in range(3):
    for v in vals:
        ctx.set_option(opt, v)
        for _ in range(3):
            r = call()
        torch.cuda.synchronize()
        ts = []
        for _ in range(8):
            t0 = time.perf_counter(); r = call(); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        lib.mlpl_profile_reset(ctx.handle); lib.mlpl_profile_enable(ctx.handle, 1)
        for _ in range(2):
            call()
        torch.cuda.synchronize(); lib.mlpl_profile_enable(ctx.handle, 0)
        ms, cnt = C.c_double(0), C.c_int(0)
        lib.mlpl_profile_read(ctx.handle, 3, C.byref(ms), C.byref(cnt))
        key = r.tobytes()
        ref = ref or key
        print(f"round {rnd} {opt}={v}: step median {np.median(ts) * 1e3:.3f} ms, min {min(ts) * 1e3:.3f} ms; counting kernels {ms.value / 2:.3f} ms per step ({cnt.value // 2} launches); same records: {key == ref}", flush=True)
ctx.close()
