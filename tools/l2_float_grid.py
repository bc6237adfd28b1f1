"""fp16 candidate path of the squared-L2 2-NN at C4's shape: whole-call time for several grid sizes (option l2_mfma_blocks_per_cu)."""
import os
import sys

import numpy as np
import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from matchinglib_poselib_amd import _lib  # noqa: E402

ctx = _lib.default_context()
dev = torch.device("cuda:0")
rng = np.random.default_rng(4)
n, dim = 4096, 128
tr = rng.gamma(0.6, 1.0, size=(n, dim))
qr = np.abs(tr + 0.5 * tr.mean() * rng.gamma(0.6, 1.0, size=tr.shape))
q, t = np.sqrt(qr / qr.sum(1, keepdims=True)).astype(np.float32), np.sqrt(tr / tr.sum(1, keepdims=True)).astype(np.float32)
dq, dt = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
idx = torch.empty((n, 2), dtype=torch.int32, device=dev)
dist = torch.empty((n, 2), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
call = lambda: _lib.check(ctx.lib.mlpl_knn2_l2sq_f32_dev(ctx.handle, dq.data_ptr(), n, dim, 0, dt.data_ptr(), n, dim, 0, dim, 2, 1,  # noqa: E731
                                                         idx.data_ptr(), dist.data_ptr(), st), "knn_l2")
ctx.set_option("l2_float_mfma", 2)
for per_cu in (1, 2, 3, 4):
    ctx.set_option("l2_mfma_blocks_per_cu", per_cu)
    for _ in range(4):
        call()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(50):
        call()
    e1.record()
    torch.cuda.synchronize()
    print(f"workgroups per CU {per_cu}: {e0.elapsed_time(e1) / 50 * 1e3:.1f} us per call", flush=True)
ctx.set_option("l2_mfma_blocks_per_cu", 0)
ctx.set_option("l2_float_mfma", 1)
