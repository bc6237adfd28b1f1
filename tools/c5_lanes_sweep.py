"""C5 with USAC / ARRSAC: cohorts in flight (option hub_lanes, up to 8) x runs per cohort (hub_cohort) x worker threads per cohort (hub_workers),
cohort feed on.  512 pairs of 8192 keypoints, same process, alternating, records must be identical.
python tools/c5_lanes_sweep.py [steps=5] [cases=usac_uniform,usac_prosac,usac_default_refine,arrsac]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import batch, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
want = sys.argv[2].split(",") if len(sys.argv) > 2 else ["usac_uniform", "usac_prosac", "usac_default_refine", "arrsac"]
total, nk = 512, 8192
dev = torch.device("cuda:0")
ctx = mpa.Context(0)
sps = [synth.stereo_pair(nk, seed=20260200 + i, unmatched_frac=0.30 + 0.02 * (i % 8)) for i in range(8)]
K = sps[0]["K"]
stk = [torch.from_numpy(np.stack([sps[i % 8][k] for i in range(total)])).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")]
seeds = [100 + i for i in range(total)]
cases = {"usac_uniform": lambda: batch.process_pairs_batched_usac(ctx, *stk, K, K, seeds, prosac=False)[1],
         "usac_prosac": lambda: batch.process_pairs_batched_usac(ctx, *stk, K, K, seeds, prosac=True)[1],
         "usac_default_refine": lambda: batch.process_pairs_batched_usac(ctx, *stk, K, K, seeds, prosac=True, refine=5)[1],
         "arrsac": lambda: batch.process_pairs_batched_arrsac(ctx, *stk, K, K, refine=True)[1]}
# (lanes, cohort, workers): the default first
configs = [(4, 128, 0), (8, 64, 0), (8, 64, 8), (8, 64, 4), (6, 96, 0), (8, 128, 0)]
for name in want:
    fn = cases[name]
    ref = None
    for rnd in range(2):
        for lanes, cohort, workers in configs:
            ctx.set_option("hub_lanes", lanes), ctx.set_option("hub_cohort", cohort), ctx.set_option("hub_workers", workers)
            fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(steps):
                t0 = time.perf_counter(); raw = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            key = raw.tobytes()
            ref = ref or key
            print(json.dumps({"case": name, "round": rnd, "hub_lanes": lanes, "hub_cohort": cohort, "hub_workers": workers, "ms_min": round(min(ts), 2),
                              "ms_median": round(float(np.median(ts)), 2), "same_records": key == ref}), flush=True)
ctx.close()
