"""mlpl_arrsac_essential_batch_dev: wall time per call for B problems of the C3 shape beside B single-problem calls, and C5 with ARRSAC
(mlpl_pair_pose_batch_arrsac_dev).  usage: python tools/arrsac_batch_timing.py [B=512] [reps=3]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import batch, pose, synth
import make_golden

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
n = 5000
ctx = mpa.Context(0)
distinct = 16
sc = [make_golden.usac_scene(n, 0.5, 20260103 + i) for i in range(distinct)]
th = sc[0][2]
p1 = np.stack([sc[b % distinct][0] for b in range(B)])
p2 = np.stack([sc[b % distinct][1] for b in range(B)])
dev = torch.device("cuda:0")
d1, d2 = torch.from_numpy(p1).to(dev), torch.from_numpy(p2).to(dev)
masks = torch.zeros((B, n), dtype=torch.uint8, device=dev)
for refine in (True, False):
    states = np.array([[0xFFFFFFFF + 7 * b, 0xFFFFFFFF + 3 * b] for b in range(B)], np.uint64)
    got = pose.arrsac_essential_batch(d1, d2, [n] * B, th, refine=refine, rng_states=states.copy(), masks_out=masks, ctx=ctx)
    ts = []
    for _ in range(reps):
        st = states.copy()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        got = pose.arrsac_essential_batch(d1, d2, [n] * B, th, refine=refine, rng_states=st, masks_out=masks, ctx=ctx)
        torch.cuda.synchronize()
        ts.append((time.perf_counter() - t0) * 1e3)
    k = min(B, 32)
    t0 = time.perf_counter()
    for b in range(k):
        pose.arrsac_essential(p1[b], p2[b], th, refine=refine, rng_state=states[b].copy(), ctx=ctx)
    single = (time.perf_counter() - t0) * 1e3 / k
    stats = np.zeros(12, np.int64)
    ctx.lib.mlpl_arrsac_last_stats(ctx.handle, stats.ctypes.data)
    print(f"B {B} refine {refine}: batch {min(ts):.2f} ms (runs {[round(t, 1) for t in ts]}) = {min(ts) / B * 1e3:.1f} us per problem; one at a time {single:.3f} ms per problem "
          f"({single * B:.0f} ms for B); ok {sum(g['ok'] for g in got)} mean inliers {np.mean([g['n_inliers'] for g in got]):.0f}", flush=True)
# C5 with ARRSAC
nk, total = 8192, B
sps = [synth.stereo_pair(nk, seed=20260200 + i, unmatched_frac=0.30 + 0.02 * (i % 8)) for i in range(8)]
K = sps[0]["K"]
stk = [torch.from_numpy(np.stack([sps[i % 8][k] for i in range(total)])).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")]
rec, raw = batch.process_pairs_batched_arrsac(ctx, *stk, K, K)
ts = []
for _ in range(reps):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    rec, raw = batch.process_pairs_batched_arrsac(ctx, *stk, K, K)
    torch.cuda.synchronize()
    ts.append((time.perf_counter() - t0) * 1e3)
print(f"C5 with ARRSAC + refine: {total} pairs in {min(ts):.2f} ms (runs {[round(t, 1) for t in ts]}) = {total / min(ts) * 1e3:.0f} pairs/s; status ok {(raw['status'] == 0).sum()} "
      f"mean matches {raw['n_matches'].mean():.0f} inliers {raw['n_inliers'].mean():.0f}", flush=True)
