"""Kernel timeline of one image pair's pipeline (match -> gather -> RANSAC -> cheirality), device-resident inputs.
Run under `rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/pair_timeline.py`."""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch  # noqa: E402

import matchinglib_poselib_amd as mpa  # noqa: E402
from matchinglib_poselib_amd import batch, synth  # noqa: E402

ctx = mpa.Context(0)
dev = torch.device("cuda:0")
sps = [synth.stereo_pair(8192, seed=20260200 + i, unmatched_frac=0.30 + 0.02 * i) for i in range(4)]
ins = [tuple(torch.from_numpy(sp[k]).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")) for sp in sps]
K = sps[0]["K"]
scratch = {}
for i in range(4):
    batch.process_pair_on_device(ctx, *ins[i], K, K, seed=100 + i, pair_id=i, scratch=scratch)
torch.cuda.synchronize()
t0 = time.perf_counter()
for rep in range(5):
    for i in range(4):
        r = batch.process_pair_on_device(ctx, *ins[i], K, K, seed=100 + i, pair_id=i, scratch=scratch)
torch.cuda.synchronize()
print("ms per pair", (time.perf_counter() - t0) / 20 * 1e3, "inliers", int(r["n_inliers"][0]))
