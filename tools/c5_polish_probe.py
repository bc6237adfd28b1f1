import os, sys, time
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
import numpy as np, torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import batch, synth
dev = torch.device("cuda:0"); ctx = mpa.Context(0)
n, total = 8192, 128
sps = [synth.stereo_pair(n, seed=20260200 + i, unmatched_frac=0.30 + 0.02 * (i % 8)) for i in range(8)]
K = sps[0]["K"]
st = {k: torch.from_numpy(np.stack([sps[i % 8][k] for i in range(total)])).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")}
seeds = [100 + i for i in range(total)]
for polish in (1, 0):
    ctx.set_option("solver_polish", polish)
    for _ in range(3):
        rec = batch.process_pairs_batched(ctx, st["desc1"], st["desc2"], st["kp1"], st["kp2"], K, K, seeds)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5):
        rec = batch.process_pairs_batched(ctx, st["desc1"], st["desc2"], st["kp1"], st["kp2"], K, K, seeds)
    torch.cuda.synchronize()
    print("polish", polish, (time.perf_counter() - t0) / 5 * 1e3, "ms per 128 pairs")
