"""The latency shape (ONE C2 image pair per call): step time by HIP events over back-to-back calls, with the merge kernel emitting the matches
(option hamming_merge_emit = 1, 3 launches per step) and without (0, 4 launches), and -- to tell host submission from device time -- the same
steps issued in bursts behind a long blocker kernel (the device then finds the whole queue waiting: what it needs per step is device time).
Run under `rocprofv3 --kernel-trace` + tools/kernel_timeline.py for the kernel-by-kernel picture.   python tools/single_pair_probe.py"""
import json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import synth
from matchinglib_poselib_amd.matching import match_hamming_device

n = 8192
dev = torch.device("cuda:0")
ctx = mpa.Context(0)
q, t = synth.orb_pair(n, n, seed=20260102)
dq, dt = torch.from_numpy(q[None]).to(dev), torch.from_numpy(t[None]).to(dev)
big = torch.empty((8192, 8192), device=dev)
for emit in (1, 0, 1, 0):
    ctx.set_option("hamming_merge_emit", emit)
    out = match_hamming_device(dq, dt, ctx=ctx)
    for _ in range(50):
        out = match_hamming_device(dq, dt, ctx=ctx, out=out)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    t0 = time.perf_counter()
    e0.record()
    for _ in range(200):
        out = match_hamming_device(dq, dt, ctx=ctx, out=out)
    e1.record()
    host_us = (time.perf_counter() - t0) / 200 * 1e6
    torch.cuda.synchronize()
    back_to_back = e0.elapsed_time(e1) / 200 * 1e3
    # queue 100 steps behind ~2 ms of other work: when the blocker ends, the device runs them with nothing to wait for on the host side
    for _ in range(3):
        big.normal_()
    e2, e3 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    for _ in range(12):
        big.normal_()
    e2.record()
    for _ in range(100):
        out = match_hamming_device(dq, dt, ctx=ctx, out=out)
    e3.record()
    torch.cuda.synchronize()
    print(json.dumps({"hamming_merge_emit": emit, "launches_per_step": 3 if emit else 4, "us_per_step_back_to_back": round(back_to_back, 2),
                      "host_us_per_call": round(host_us, 2), "us_per_step_queued_behind_a_blocker": round(e2.elapsed_time(e3) / 100 * 1e3, 2),
                      "matches": int(out["count"][0].item())}), flush=True)
ctx.close()
