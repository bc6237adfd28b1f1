"""Workload for rocprofv3 --kernel-trace --stats: the matching step for P pairs with the fused merge epilogue on or off.
usage: python tools/hamming_fuse_prof.py P fused reps"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import synth
from matchinglib_poselib_amd.matching import match_hamming_device
P, fused, reps = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
ctx = mpa.Context(0)
dev = torch.device("cuda:0")
n = 8192
qs, ts = zip(*[synth.orb_pair(n, n, seed=20260102 + p) for p in range(P)])
dq, dt = torch.from_numpy(np.stack(qs)).to(dev), torch.from_numpy(np.stack(ts)).to(dev)
ctx.set_option("hamming_fused_merge", fused)
out = match_hamming_device(dq, dt, ctx=ctx)
for _ in range(reps):
    out = match_hamming_device(dq, dt, ctx=ctx, out=out)
torch.cuda.synchronize()
