"""A/B the Hamming k-NN kernel variants in ONE process (interleaved rounds, guide rule 24)."""
import ctypes as C
import itertools
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np
import torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import _lib, synth
from matchinglib_poselib_amd.matching import match_hamming_device

ctx = mpa.Context(0)
lib = ctx.lib
n = int(sys.argv[1]) if len(sys.argv) > 1 else 8192
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1
qs, ts = zip(*[synth.orb_pair(n, n, seed=20260102 + b) for b in range(B)])
dq = torch.from_numpy(np.stack(qs)).cuda()
dt = torch.from_numpy(np.stack(ts)).cuda()
stream = None
configs = [(0, 1, 32), (2, 1, 4), (2, 1, 8), (2, 1, 16), (2, 1, 32), (2, 1, 64)]
res = {c: [] for c in configs}
ref = None
for rnd in range(5):
    for c in configs:
        for name, v in zip(("hamming_variant", "hamming_qpl", "hamming_blocks_per_cu"), c):
            _lib.check(lib.mlpl_set_option(ctx.handle, name.encode(), v), name)
        out = match_hamming_device(dq, dt, ctx=ctx, stream=stream)
        torch.cuda.synchronize()
        if ref is None:
            ref = out["idx"].clone()
        assert torch.equal(ref, out["idx"])
        lib.mlpl_profile_reset(ctx.handle); lib.mlpl_profile_enable(ctx.handle, 1)
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(50):
            out = match_hamming_device(dq, dt, ctx=ctx, out=out, stream=stream)
        e1.record(); torch.cuda.synchronize()
        lib.mlpl_profile_enable(ctx.handle, 0)
        ms, cnt = C.c_double(0), C.c_int(0)
        lib.mlpl_profile_read(ctx.handle, 0, C.byref(ms), C.byref(cnt))
        res[c].append((ms.value / cnt.value * 1e3, e0.elapsed_time(e1) / 50 * 1e3))
for c in configs:
    k = np.array(res[c])
    print(f"variant={c[0]} qpl={c[1]} blocks/cu={c[2]:2d}: partial kernel med {np.median(k[:,0]):7.1f} us min {k[:,0].min():7.1f} | step med {np.median(k[:,1]):7.1f} us")
