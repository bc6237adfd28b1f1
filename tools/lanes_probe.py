"""C5 on one GPU: one batched call at a time against batch.BatchLanes (two calls in flight), step by step in ONE fresh process -- the
question of VERDICT r3 weak 4 (the driver's box measured two lanes SLOWER than one).  Prints the wall time of every step of alternating
blocks (lanes, single, lanes, single ...), each lane's start / end offsets inside a step, and the library's arena sizes before and after.
usage: python tools/lanes_probe.py [pairs=512] [steps per block=6] [blocks=3] [n=8192]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import batch, synth

total = int(sys.argv[1]) if len(sys.argv) > 1 else 512
steps = int(sys.argv[2]) if len(sys.argv) > 2 else 6
blocks = int(sys.argv[3]) if len(sys.argv) > 3 else 3
n = int(sys.argv[4]) if len(sys.argv) > 4 else 8192
dev = torch.device("cuda:0")
t_start = time.perf_counter()
ctx = mpa.Context(0)
distinct = 8
sps = [synth.stereo_pair(n, seed=20260200 + i, unmatched_frac=0.30 + 0.02 * (i % 8)) for i in range(distinct)]
K = sps[0]["K"]
st = {k: torch.from_numpy(np.stack([sps[i % distinct][k] for i in range(total)])).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")}
seeds = [100 + i for i in range(total)]
d_matches = torch.zeros((total, n, 4), dtype=torch.int32, device=dev)
torch.cuda.synchronize()
print(f"setup {time.perf_counter() - t_start:.1f} s", flush=True)
lanes = batch.BatchLanes(0, lanes=2, first_ctx=ctx)


def one(mode):
    t0 = time.perf_counter()
    if mode == "lanes":
        rec = lanes.process(st["desc1"], st["desc2"], st["kp1"], st["kp2"], K, K, seeds, matches_out=d_matches)
    else:
        rec = batch.process_pairs_batched(ctx, st["desc1"], st["desc2"], st["kp1"], st["kp2"], K, K, seeds, matches_out=d_matches)
    torch.cuda.synchronize()
    t1 = time.perf_counter()
    extra = ""
    if mode == "lanes":
        extra = " lanes " + " ".join(f"[{(a - t0) * 1e3:.2f}..{(b - t0) * 1e3:.2f}]" for a, b in lanes.last_lane_span)
    return (t1 - t0) * 1e3, extra, rec


ref = None
for blk in range(blocks):
    for mode in ("lanes", "single"):
        ts = []
        for k in range(steps):
            ms, extra, rec = one(mode)
            if ref is None:
                ref = rec.tobytes()
            assert rec.tobytes() == ref
            ts.append(ms)
            print(f"block {blk} {mode:6s} step {k}: {ms:7.2f} ms{extra}", flush=True)
        print(f"block {blk} {mode:6s}: median {np.median(ts):.2f} min {min(ts):.2f} max {max(ts):.2f} ms -> {total / np.median(ts) * 1e3:.0f} pairs/s", flush=True)
lanes.close()
