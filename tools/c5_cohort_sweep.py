"""C5 with USAC / ARRSAC: runs per cohort (option hub_cohort) with the cohort feed on -- smaller cohorts let every lane start earlier and
take a second (third) cohort when it is done.  512 pairs of 8192 keypoints, same process, records must be identical.
python tools/c5_cohort_sweep.py [steps=5]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import batch, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
total, nk = 512, 8192
dev = torch.device("cuda:0")
ctx = mpa.Context(0)
sps = [synth.stereo_pair(nk, seed=20260200 + i, unmatched_frac=0.30 + 0.02 * (i % 8)) for i in range(8)]
K = sps[0]["K"]
stk = [torch.from_numpy(np.stack([sps[i % 8][k] for i in range(total)])).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")]
seeds = [100 + i for i in range(total)]
cases = {"usac_uniform": lambda: batch.process_pairs_batched_usac(ctx, *stk, K, K, seeds, prosac=False)[1],
         "usac_prosac": lambda: batch.process_pairs_batched_usac(ctx, *stk, K, K, seeds, prosac=True)[1],
         "arrsac": lambda: batch.process_pairs_batched_arrsac(ctx, *stk, K, K, refine=True)[1]}
for name, fn in cases.items():
    ref = None
    for rnd in range(2):
        for cohort in (128, 96, 64, 48, 32):
            ctx.set_option("hub_cohort", cohort)
            fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(steps):
                t0 = time.perf_counter(); raw = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            key = raw.tobytes()
            ref = ref or key
            print(json.dumps({"case": name, "round": rnd, "hub_cohort": cohort, "ms_min": round(min(ts), 2), "ms_median": round(float(np.median(ts)), 2),
                              "same_records": key == ref}), flush=True)
ctx.close()
