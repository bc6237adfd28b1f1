"""A/B of the fused merge epilogue of the static LDS-ring Hamming kernel (option hamming_fused_merge): the full matching step
(mlpl_match_hamming_dev: expand + kernel [+ merge] + ratio_write) for 1 / 8 / 64 image pairs of 8192 x 8192 ORB-256 per launch; outputs
must be identical."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import synth
from matchinglib_poselib_amd.matching import match_hamming_device

ctx = mpa.Context(0)
dev = torch.device("cuda:0")
n = 8192
for P in (1, 8, 64):
    qs, ts = zip(*[synth.orb_pair(n, n, seed=20260102 + p) for p in range(P)])
    dq, dt = torch.from_numpy(np.stack(qs)).to(dev), torch.from_numpy(np.stack(ts)).to(dev)
    ref = None
    for rnd in range(2):
        for fused in (0, 1):
            ctx.set_option("hamming_fused_merge", fused)
            out = match_hamming_device(dq, dt, ctx=ctx)
            torch.cuda.synchronize()
            key = tuple(out[k].cpu().numpy().tobytes() for k in ("idx", "dist", "count")) + (out["matches"][0, : int(out["count"][0])].cpu().numpy().tobytes(),)
            if ref is None:
                ref = key
            assert key == ref, (P, fused)
            reps = 200 if P < 64 else 50
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                out = match_hamming_device(dq, dt, ctx=ctx, out=out)
            e1.record()
            torch.cuda.synchronize()
            ms = e0.elapsed_time(e1) / reps
            print(f"{P} pairs per launch, fused merge {fused}: {ms * 1e3:.1f} us per step = {P * n * n / (ms * 1e-3) / 1e12:.2f} T pairs/s", flush=True)
