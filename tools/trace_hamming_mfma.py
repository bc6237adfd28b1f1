"""Temporary: per-wave timeline of knn_hamming_mfma_kernel from the MLPL_DEBUG_TRACE dump."""
import os, sys, collections
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import _lib, synth

bpc = int(sys.argv[1]) if len(sys.argv) > 1 else 2
batch = 8
ctx = mpa.Context(0)
dev = torch.device("cuda:0")
qs, ts = zip(*[synth.orb_pair(8192, 8192, seed=100 + b) for b in range(batch)])
dq = torch.from_numpy(np.stack(qs)).to(dev); dt = torch.from_numpy(np.stack(ts)).to(dev)
idx = torch.empty((batch, 8192, 2), dtype=torch.int32, device=dev); dist = torch.empty_like(idx)
st = torch.cuda.current_stream().cuda_stream
ctx.set_option("hamming_variant", 3); ctx.set_option("hamming_mfma_blocks_per_cu", bpc)
def call():
    _lib.check(ctx.lib.mlpl_knn2_hamming_dev(ctx.handle, dq.data_ptr(), 8192, 32, 8192 * 32, dt.data_ptr(), 8192, 32, 8192 * 32, 32, 2, batch, idx.data_ptr(), dist.data_ptr(), st), "knn")
os.environ.pop("MLPL_DEBUG_TRACE", None)
for _ in range(5): call()
torch.cuda.synchronize()
path = f"/tmp/trace_{bpc}.bin"
os.environ["MLPL_DEBUG_TRACE"] = path
import ctypes
ctypes.CDLL(None).setenv(b"MLPL_DEBUG_TRACE", path.encode(), 1)
call(); torch.cuda.synchronize()
a = np.fromfile(path, dtype=np.int64).reshape(-1, 4)
a = a[a[:, 1] > 0]
t0 = a[:, 0].min()
start = (a[:, 0] - t0) * 10e-3; end = (a[:, 1] - t0) * 10e-3   # us
hw = a[:, 3] & 0xFFFFFFFF; xcc = (a[:, 3] >> 32) & 0xF
simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
key = (((xcc * 8 + se) * 2 + sh) * 16 + cu) * 4 + simd
print(f"bpc={bpc} waves={len(a)} span={end.max():.1f} us; wave duration us: min {np.min(end-start):.1f} med {np.median(end-start):.1f} max {np.max(end-start):.1f}")
print("wave start us percentiles", np.percentile(start, [0, 50, 90, 99, 100]).round(1))
print("wave end us percentiles", np.percentile(end, [0, 10, 50, 90, 100]).round(1))
print("cycles per wave (memtime) med", np.median(a[:, 2]), " -> clock GHz", np.median(a[:, 2] / ((a[:, 1] - a[:, 0]) * 10e-9)) / 1e9)
cnt = collections.Counter(key.tolist())
print("distinct SIMDs used", len(cnt), "waves per SIMD histogram", sorted(collections.Counter(cnt.values()).items()))
print("distinct xcc", sorted(set(xcc.tolist())), "se", sorted(set(se.tolist())), "cu", sorted(set(cu.tolist())))
# concurrency per SIMD over time: average number of resident waves while busy
tot = 0; busy = 0
for k in cnt:
    m = key == k
    ev = sorted([(s, 1) for s in start[m]] + [(e, -1) for e in end[m]])
    cur = 0; last = ev[0][0]
    for t, d in ev:
        if cur > 0: tot += cur * (t - last); busy += (t - last)
        cur += d; last = t
print("avg resident waves per busy SIMD", tot / busy, "avg busy us per SIMD", busy / len(cnt))
