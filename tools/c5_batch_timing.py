"""C5 through mlpl_pair_pose_batch_dev on one GPU: pairs/s for a few batch sizes (tools; the bench line is bench.py --workload c5)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import batch, synth

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
total = int(sys.argv[2]) if len(sys.argv) > 2 else 128
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 5
dev = torch.device("cuda:0")
ctx = mpa.Context(0)
distinct = 8
sps = [synth.stereo_pair(n, seed=20260200 + i, unmatched_frac=0.30 + 0.02 * (i % 8)) for i in range(distinct)]
K = sps[0]["K"]
st = {k: torch.from_numpy(np.stack([sps[i % distinct][k] for i in range(total)])).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")}
seeds = [100 + i for i in range(total)]
for pb in (64, 32, 128):
    ctx.set_option("pair_batch", pb)
    rec = batch.process_pairs_batched(ctx, st["desc1"], st["desc2"], st["kp1"], st["kp2"], K, K, seeds)
    torch.cuda.synchronize()
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter()
        rec = batch.process_pairs_batched(ctx, st["desc1"], st["desc2"], st["kp1"], st["kp2"], K, K, seeds)
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    stats = np.zeros(8, np.int64)   # mlpl_pair_batch_last_stats writes eight values
    ctx.lib.mlpl_pair_batch_last_stats(ctx.handle, stats.ctypes.data)
    print(f"pair_batch {pb}: {total} pairs, ms per pass {[round(t * 1e3, 2) for t in ts]}, best {total / min(ts):.0f} pairs/s = {min(ts) / total * 1e6:.1f} us/pair, "
          f"status ok {(rec['status'] == 0).sum()}, mean matches {rec['n_matches'].mean():.0f} inliers {rec['n_inliers'].mean():.0f}, stats {stats}", flush=True)
