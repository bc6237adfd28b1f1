"""One workload for rocprofv3: the C4 squared-L2 2-NN call (auto path), 200 times."""
import os, sys
import torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from matchinglib_poselib_amd import _lib, synth
n = int(sys.argv[1]) if len(sys.argv) > 1 else 4096
ctx = _lib.default_context()
dev = torch.device("cuda:0")
q, t = synth.sift_pair(n, n, seed=5)
dq, dt = torch.from_numpy(q).to(dev), torch.from_numpy(t).to(dev)
idx = torch.empty((n, 2), dtype=torch.int32, device=dev)
dist = torch.empty((n, 2), dtype=torch.float32, device=dev)
st = torch.cuda.current_stream().cuda_stream
for _ in range(200):
    _lib.check(ctx.lib.mlpl_knn2_l2sq_f32_dev(ctx.handle, dq.data_ptr(), n, 128, 0, dt.data_ptr(), n, 128, 0, 128, 2, 1,
                                              idx.data_ptr(), dist.data_ptr(), st), "knn_l2_dev")
torch.cuda.synchronize()
