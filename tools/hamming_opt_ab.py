"""Same-box A/B of one integer option of the matrix-core Hamming path (mlpl_set_option), values alternately and twice: step time (HIP events over 200 steps of 64 C2 pairs after 100 warm-up steps), kernel time (HIP events
inside the library, every 4th launch), and the shader clock inside the kernel (clock ring, option hamming_stamps = 2).
    python tools/hamming_opt_ab.py option v0,v1[,v2...] [pairs] [steps]      e.g. hamming_split_rows 4096,8192"""
import ctypes as C
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch  # noqa: E402
import matchinglib_poselib_amd as mpa  # noqa: E402
from matchinglib_poselib_amd import synth  # noqa: E402
from matchinglib_poselib_amd.matching import match_hamming_device  # noqa: E402

OPT, VALS = sys.argv[1], [int(v) for v in sys.argv[2].split(",")]
P = int(sys.argv[3]) if len(sys.argv) > 3 else 64
STEPS = int(sys.argv[4]) if len(sys.argv) > 4 else 200
n = 8192
dev = torch.device("cuda", 0)
ctx = mpa.Context(0)
lib = ctx.lib
qs, ts = zip(*[synth.orb_pair(n, n, seed=20260102 + p) for p in range(P)])
dq, dt = torch.from_numpy(np.stack(qs)).to(dev), torch.from_numpy(np.stack(ts)).to(dev)
ctx.set_option("hamming_stamps", 2)
out = match_hamming_device(dq, dt, ctx=ctx)
ref = None
res = []
for rnd in range(2):
    for enc in VALS:
        ctx.set_option(OPT, enc)
        for _ in range(100):
            out = match_hamming_device(dq, dt, ctx=ctx, out=out)
        torch.cuda.synchronize()
        lib.mlpl_profile_reset(ctx.handle)
        lib.mlpl_profile_enable(ctx.handle, 4)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(STEPS):
            out = match_hamming_device(dq, dt, ctx=ctx, out=out)
        e1.record()
        torch.cuda.synchronize()
        lib.mlpl_profile_enable(ctx.handle, 0)
        ms, cnt = C.c_double(0), C.c_int(0)
        lib.mlpl_profile_read(ctx.handle, 0, C.byref(ms), C.byref(cnt))
        buf = np.zeros((256, 4), np.uint64)
        m = lib.mlpl_debug_hamming_clock(ctx.handle, buf.ctypes.data, min(STEPS, 256))
        r = buf[:m].astype(np.float64)
        ghz = r[:, 0] / np.maximum(r[:, 1], 1) * 0.1
        key = (out["idx"].cpu().numpy().tobytes(), out["dist"].cpu().numpy().tobytes(), out["count"].cpu().numpy().tobytes())
        if ref is None:
            ref = key
        rec = {"round": rnd, OPT: enc, "step_ms": e0.elapsed_time(e1) / STEPS, "kernel_ms": ms.value / max(cnt.value, 1), "launches_timed": cnt.value,
               "clock_GHz_median": float(np.median(ghz)), "clock_GHz_min_max": [float(ghz.min()), float(ghz.max())],
               "first_workgroup_us_median": float(np.median(r[:, 1]) / 100.0), "same_outputs_as_first_run": key == ref,
               "T_pairs_per_s": P * n * n / (e0.elapsed_time(e1) / STEPS * 1e-3) / 1e12}
        print(json.dumps(rec), flush=True)
        res.append(rec)
ctx.close()
