"""C5 with USAC / ARRSAC: same-process A/B of one integer option (mlpl_set_option), values alternately and twice; 512 pairs of 8192 keypoints;
the records must be identical.
python tools/c5_opt_ab.py option v0,v1[,...] [cases=usac_uniform,usac_prosac,usac_default_refine,arrsac] [steps=5]"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import batch, synth

opt, vals = sys.argv[1], [int(v) for v in sys.argv[2].split(",")]
want = sys.argv[3].split(",") if len(sys.argv) > 3 else ["usac_uniform", "usac_prosac", "usac_default_refine", "arrsac"]
steps = int(sys.argv[4]) if len(sys.argv) > 4 else 5
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
for name in want:
    fn = cases[name]
    ref = None
    for rnd in range(2):
        for v in vals:
            ctx.set_option(opt, v)
            fn(); torch.cuda.synchronize()
            ts = []
            for _ in range(steps):
                t0 = time.perf_counter(); raw = fn(); torch.cuda.synchronize(); ts.append((time.perf_counter() - t0) * 1e3)
            key = raw.tobytes()
            ref = ref or key
            sa = np.zeros(8, np.int64)
            ctx.lib.mlpl_usac_last_stats(ctx.handle, sa.ctypes.data)   # (lanes summed; ARRSAC keeps its own)
            st = {"rounds": int(sa[0]), "merged_launches": int(sa[1]), "hub_waiting_for_host_ms": sa[2] / 1e3, "device_ms": sa[3] / 1e3}
            print(json.dumps({"case": name, "round": rnd, opt: v, "ms_min": round(min(ts), 2), "ms_median": round(float(np.median(ts)), 2),
                              "same_records": key == ref, "records_sha1": __import__("hashlib").sha1(key).hexdigest()[:12], "hub": st}), flush=True)
ctx.close()
