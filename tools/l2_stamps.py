"""Diagnostics: per-workgroup start/end ticks of the L2 MFMA kernel on C4 -- how long is a workgroup alive, how long does the grid take to start."""
import os, sys
import numpy as np, torch, ctypes as C
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
ctx.lib.mlpl_set_option(ctx.handle, b"hamming_stamps", 1)
WAVES = int(sys.argv[2]) if len(sys.argv) > 2 else 0
BPC = int(sys.argv[3]) if len(sys.argv) > 3 else 0
ctx.lib.mlpl_set_option(ctx.handle, b"l2_mfma_waves", WAVES)
ctx.lib.mlpl_set_option(ctx.handle, b"l2_mfma_blocks_per_cu", BPC)
for _ in range(30):
    _lib.check(ctx.lib.mlpl_knn2_l2sq_f32_dev(ctx.handle, dq.data_ptr(), n, 128, 0, dt.data_ptr(), n, 128, 0, 128, 2, 1,
                                              idx.data_ptr(), dist.data_ptr(), st), "knn_l2_dev")
torch.cuda.synchronize()
buf = np.zeros((65536, 4), np.uint64)
cnt = ctx.lib.mlpl_debug_hamming_stamps(ctx.handle, buf.ctypes.data, 65536)
s = buf[:cnt]
t0 = s[:, 1].min()
start = (s[:, 1] - t0).astype(np.int64) * 10  # ns
end = (s[:, 2] - t0).astype(np.int64) * 10
cyc = s[:, 0].astype(np.int64)
print(f"{cnt} workgroups; kernel span {end.max()} ns; starts: median {np.median(start):.0f} ns, 90% {np.percentile(start, 90):.0f}, max {start.max()} ns")
print(f"workgroup lifetime: median {np.median(end - start):.0f} ns, max {(end - start).max()} ns; cycles median {np.median(cyc):.0f} -> clock {np.median(cyc) / max(1, np.median(end - start)):.2f} GHz")
order = np.argsort(start)
print("start times of every 64th workgroup (ns):", start[order][::64].tolist())
xcc = (s[:, 3] >> np.uint64(32)).astype(int)
print("workgroups per XCC:", np.bincount(xcc).tolist())
