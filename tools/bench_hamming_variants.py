"""Times the selectable Hamming kernels on the C2 shape (8 pairs per launch and a single pair) on one GPU.
Usage: python tools/bench_hamming_variants.py [variant ...] [--batch=N] [--bpc=N]"""
import json
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402
import torch  # noqa: E402

import matchinglib_poselib_amd as mpa  # noqa: E402
from matchinglib_poselib_amd import _lib, synth  # noqa: E402


def main():
    args = [a for a in sys.argv[1:] if not a.startswith("--")]
    opts = dict(a[2:].split("=") for a in sys.argv[1:] if a.startswith("--"))
    variants = [int(v) for v in args] or [0, 3]
    batches = [int(opts["batch"])] if "batch" in opts else [8, 1]
    bpcs = [int(opts["bpc"])] if "bpc" in opts else None
    ctx_opts = {k: int(v) for k, v in opts.items() if k not in ("batch", "bpc")}
    ctx = mpa.Context(0)
    for k, v in ctx_opts.items():
        ctx.set_option(k, v)
    dev = torch.device("cuda:0")
    out = {}
    for batch in batches:
        qs, ts = [], []
        for b in range(batch):
            q, t = synth.orb_pair(8192, 8192, seed=100 + b)
            qs.append(q)
            ts.append(t)
        dq = torch.from_numpy(np.stack(qs)).to(dev)
        dt = torch.from_numpy(np.stack(ts)).to(dev)
        idx = torch.empty((batch, 8192, 2), dtype=torch.int32, device=dev)
        dist = torch.empty_like(idx)
        st = torch.cuda.current_stream().cuda_stream
        ref = None
        for v in variants:
            for bpc in (bpcs or ((2, 3, 4, 6) if v == 3 else (32,))):
                ctx.set_option("hamming_variant", v)
                ctx.set_option("hamming_mfma_blocks_per_cu" if v == 3 else "hamming_blocks_per_cu", bpc)

                def call():
                    _lib.check(ctx.lib.mlpl_knn2_hamming_dev(ctx.handle, dq.data_ptr(), 8192, 32, 8192 * 32, dt.data_ptr(), 8192, 32,
                                                             8192 * 32, 32, 2, batch, idx.data_ptr(), dist.data_ptr(), st), "knn")
                call()
                torch.cuda.synchronize()
                res = (idx.cpu().numpy().copy(), dist.cpu().numpy().copy())
                if ref is None:
                    ref = res
                same = bool(np.array_equal(ref[0], res[0]) and np.array_equal(ref[1], res[1]))
                reps = 20
                t0 = time.perf_counter()
                for _ in range(reps):
                    call()
                torch.cuda.synchronize()
                dt_ = (time.perf_counter() - t0) / reps
                ctx.lib.mlpl_profile_reset(ctx.handle)
                ctx.lib.mlpl_profile_enable(ctx.handle, 1)
                for _ in range(5):
                    call()
                torch.cuda.synchronize()
                ctx.lib.mlpl_profile_enable(ctx.handle, 0)
                import ctypes as C
                ms, n = C.c_double(0), C.c_int(0)
                ctx.lib.mlpl_profile_read(ctx.handle, 0, C.byref(ms), C.byref(n))
                out[f"batch{batch}_variant{v}_bpc{bpc}"] = {"us_per_call": dt_ * 1e6, "kernel_us": ms.value / max(n.value, 1) * 1e3,
                                                           "Tpairs_per_s": batch * 8192 * 8192 / dt_ / 1e12, "same_as_first": same}
                print(f"batch{batch} variant{v} bpc{bpc}", out[f"batch{batch}_variant{v}_bpc{bpc}"], flush=True)
    print(json.dumps(out))


if __name__ == "__main__":
    main()
