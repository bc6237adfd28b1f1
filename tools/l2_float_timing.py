"""Squared-L2 2-NN on float descriptors that are NOT integer-valued (unit-variance normals): the auto path has to take the exact fp32
kernel (cvflann's summation order).  Whole-call time, auto path and forced exact path."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from matchinglib_poselib_amd import _lib
ctx = _lib.default_context(); dev = torch.device("cuda:0")
for n, dim in ((4096, 128), (4096, 64), (8192, 128), (2048, 32)):
    rng = np.random.default_rng(1)
    q = rng.normal(size=(n, dim)).astype(np.float32); t = rng.normal(size=(n, dim)).astype(np.float32)
    dq, dt = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
    idx = torch.empty((n, 2), dtype=torch.int32, device=dev); dist = torch.empty((n, 2), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    call = lambda: _lib.check(ctx.lib.mlpl_knn2_l2sq_f32_dev(ctx.handle, dq.data_ptr(), n, dim, 0, dt.data_ptr(), n, dim, 0, dim, 2, 1, idx.data_ptr(), dist.data_ptr(), st), "knn_l2")
    for mode in (0, 1):
        ctx.lib.mlpl_set_l2_path(ctx.handle, mode)
        for _ in range(5): call()
        torch.cuda.synchronize(); e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True); e0.record()
        for _ in range(50): call()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 20
        print(f"n={n} dim={dim} {'auto ' if mode == 0 else 'exact'}: {us:.1f} us per call = {2 * 3 * n * n * dim / us / 1e6:.1f} TFLOP/s fp32 (3 flop per element pair)")
ctx.lib.mlpl_set_l2_path(ctx.handle, 0)
