"""C4 (4096 x 4096 SIFT-128f squared-L2 2-NN): whole-call time of the auto path, and the per-kernel breakdown from the library's
own hipEvent brackets."""
import os, sys, time
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from matchinglib_poselib_amd import _lib, synth

ctx = _lib.default_context()
dev = torch.device("cuda:0")
for n in (4096, 2048, 8192):
    q, t = synth.sift_pair(n, n, seed=5)
    dq, dt = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
    idx = torch.empty((n, 2), dtype=torch.int32, device=dev)
    dist = torch.empty((n, 2), dtype=torch.float32, device=dev)
    st = torch.cuda.current_stream().cuda_stream
    for waves, bpc in ((0, 2), (0, 3), (0, 4), (0, 0), (0, 2), (0, 3)):
        mode = 0
        ctx.lib.mlpl_set_option(ctx.handle, b"l2_mfma_waves", waves)
        ctx.lib.mlpl_set_option(ctx.handle, b"l2_mfma_blocks_per_cu", bpc)
        def call():
            _lib.check(ctx.lib.mlpl_knn2_l2sq_f32_dev(ctx.handle, dq.data_ptr(), n, 128, 0, dt.data_ptr(), n, 128, 0, 128, 2, 1,
                                                      idx.data_ptr(), dist.data_ptr(), st), "knn_l2_dev")
        for _ in range(20): call()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(200): call()
        e1.record(); torch.cuda.synchronize()
        print(f"n={n} waves={waves} blocks/CU={bpc}: {e0.elapsed_time(e1) / 200 * 1000:.1f} us per call")
ctx.lib.mlpl_set_l2_path(ctx.handle, 0)
