"""The latency shape (ONE C2 image pair per call): same-process A/B of one integer option, values alternately and three times; step time by HIP
events over 300 back-to-back calls; match lists must be identical.   python tools/single_pair_opt_ab.py option v0,v1[,...]"""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import synth
from matchinglib_poselib_amd.matching import match_hamming_device

opt, vals = sys.argv[1], [int(v) for v in sys.argv[2].split(",")]
n = 8192
dev = torch.device("cuda:0")
ctx = mpa.Context(0)
q, t = synth.orb_pair(n, n, seed=20260102)
dq, dt = torch.from_numpy(q[None]).to(dev), torch.from_numpy(t[None]).to(dev)
ref = None
for rnd in range(3):
    for v in vals:
        ctx.set_option(opt, v)
        out = match_hamming_device(dq, dt, ctx=ctx)
        for _ in range(100):
            out = match_hamming_device(dq, dt, ctx=ctx, out=out)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300):
            out = match_hamming_device(dq, dt, ctx=ctx, out=out)
        e1.record()
        torch.cuda.synchronize()
        cnt = int(out["count"][0].item())
        key = (cnt, out["matches"][0, :cnt].cpu().numpy().tobytes(), out["idx"].cpu().numpy().tobytes(), out["dist"].cpu().numpy().tobytes())
        ref = ref or key
        print(json.dumps({"round": rnd, opt: v, "us_per_pair": round(e0.elapsed_time(e1) / 300 * 1e3, 2), "matches": cnt, "same_outputs": key == ref}), flush=True)
ctx.close()
