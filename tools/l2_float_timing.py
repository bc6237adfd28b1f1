"""Squared-L2 2-NN on float descriptors that are NOT integer-valued: whole-call time of the exact fp32 kernel (mode 1), the forced fp16
matrix-core candidate path (mode 3) and the auto path (mode 0, hint-driven) on normal and RootSIFT-like data.  Under
`rocprofv3 --kernel-trace` the per-kernel times; candidate statistics via the re-rank overflow counter are not exposed."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from matchinglib_poselib_amd import _lib  # noqa: E402

ctx = _lib.default_context()
dev = torch.device("cuda:0")


def rootsift_pair(n, dim, seed):
    rng = np.random.default_rng(seed)
    tr = rng.gamma(0.6, 1.0, size=(n, dim))
    qr = np.abs(tr + 0.5 * tr.mean() * rng.gamma(0.6, 1.0, size=tr.shape))
    return np.sqrt(qr / qr.sum(1, keepdims=True)).astype(np.float32), np.sqrt(tr / tr.sum(1, keepdims=True)).astype(np.float32)


for kind in ("rootsift", "normal"):
    for n, dim in ((4096, 128), (4096, 64), (8192, 128), (2048, 32)):
        if kind == "normal":
            rng = np.random.default_rng(1)
            q, t = rng.normal(size=(n, dim)).astype(np.float32), rng.normal(size=(n, dim)).astype(np.float32)
        else:
            q, t = rootsift_pair(n, dim, 4)
        dq, dt = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
        idx = torch.empty((n, 2), dtype=torch.int32, device=dev)
        dist = torch.empty((n, 2), dtype=torch.float32, device=dev)
        st = torch.cuda.current_stream().cuda_stream
        call = lambda: _lib.check(ctx.lib.mlpl_knn2_l2sq_f32_dev(ctx.handle, dq.data_ptr(), n, dim, 0, dt.data_ptr(), n, dim, 0, dim, 2, 1,  # noqa: E731
                                                                 idx.data_ptr(), dist.data_ptr(), st), "knn_l2")
        ref = None
        for mode, name in ((1, "exact"), (3, "fp16 "), (0, "auto ")):
            ctx.lib.mlpl_set_l2_path(ctx.handle, mode)
            for _ in range(4):
                call()
                torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(30):
                call()
            e1.record()
            torch.cuda.synchronize()
            us = e0.elapsed_time(e1) / 30 * 1e3
            got = (idx.cpu().numpy().copy(), dist.cpu().numpy().copy())
            ref = ref or got
            ok = np.array_equal(got[0], ref[0]) and got[1].tobytes() == ref[1].tobytes()
            print(f"{kind:8s} n={n} dim={dim} {name}: {us:8.1f} us per call, same bits as exact: {ok}", flush=True)
ctx.lib.mlpl_set_l2_path(ctx.handle, 0)
