"""Per-tile clock trace of the LDS-ring Hamming kernel (static splits): cycles per tile iteration per wave, by progress and finish order."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import synth
from matchinglib_poselib_amd.matching import match_hamming_device
ctx = mpa.Context(0)
dev = torch.device("cuda", 0)
P, n = 8, 8192
qs, ts = zip(*[synth.orb_pair(n, n, seed=20260102 + p) for p in range(P)])
dq = torch.from_numpy(np.stack(qs)).to(dev); dt = torch.from_numpy(np.stack(ts)).to(dev)
ctx.set_option("hamming_mfma_blocks_per_cu", 4); ctx.set_option("hamming_mfma_lds", 1)
out = match_hamming_device(dq, dt, ctx=ctx)
t0 = time.perf_counter()
while time.perf_counter() - t0 < 1.0:
    for _ in range(50):
        out = match_hamming_device(dq, dt, ctx=ctx, out=out)
    torch.cuda.synchronize()
ctx.set_option("hamming_stamps", 1)
for _ in range(5):
    out = match_hamming_device(dq, dt, ctx=ctx, out=out)
torch.cuda.synchronize()
ctx.lib.mlpl_debug_hamming_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
rec = np.zeros((1 << 16, 4), np.uint64)
m = ctx.lib.mlpl_debug_hamming_stamps(ctx.handle, rec.ctypes.data, len(rec))
tr = np.zeros((m, 48), np.uint64)
m2 = ctx.lib.mlpl_debug_hamming_stamps(ctx.handle, tr.ctypes.data, -m)
ctx.set_option("hamming_stamps", 0)
rec = rec[:m]; tr = tr[:m2].astype(np.int64)
ntile = int((tr > 0).sum(axis=1).max())
d = np.diff(tr[:, :ntile], axis=1)          # cycles per tile iteration
dur = rec[:, 0].astype(np.float64)
order = np.argsort(dur)
q = len(order) // 4
print("waves", m, "tiles/wave", ntile)
for name, sel in (("fastest quarter", order[:q]), ("second", order[q:2*q]), ("third", order[2*q:3*q]), ("slowest quarter", order[3*q:])):
    dd = d[sel]
    print(f"{name}: total cycles median {np.median(dur[sel]):.0f}; cycles/tile by position: first 4 {np.median(dd[:, :4], axis=0).round(0)}, "
          f"middle {np.median(dd[:, ntile//2-2:ntile//2+2], axis=0).round(0)}, last 4 {np.median(dd[:, -4:], axis=0).round(0)}; overall median {np.median(dd):.0f}")
# does the finish order follow the dispatch order?  item -> blockIdx (inverse of the XCD-aware mapping), 4 waves per workgroup
nblk = m // 4
per_xcd = nblk // 8
items = np.arange(nblk)
bid = (items % per_xcd) * 8 + items // per_xcd
blk_dur = dur.reshape(nblk, 4).mean(axis=1)
for lo in range(0, nblk, nblk // 8):
    selb = (bid >= lo) & (bid < lo + nblk // 8)
    print(f"blockIdx [{lo},{lo + nblk // 8}): mean duration {blk_dur[selb].mean():.0f} cycles (min {blk_dur[selb].min():.0f}, max {blk_dur[selb].max():.0f})")
hw = ((rec[:, 2] >> np.uint64(32)) & np.uint64(0xFFFFF)).astype(np.int64)
xcc = (rec[:, 2] >> np.uint64(56)).astype(np.int64)
cu = ((xcc * 8 + ((hw >> 13) & 7)) * 2 + ((hw >> 12) & 1)) * 16 + ((hw >> 8) & 15)
wid = hw & 15
simd = (hw >> 4) & 3
# per (cu, simd): the 4 waves sorted by duration: which wave slot ids / block ids?
key = cu * 4 + simd
o = np.lexsort((dur, key))
k4 = key[o].reshape(-1, 4); b4 = np.repeat(bid, 4)[o].reshape(-1, 4); w4 = wid[o].reshape(-1, 4)
print("per SIMD, waves sorted fastest->slowest: mean blockIdx", b4.mean(axis=0).round(0), " mean hw wave slot", w4.mean(axis=0).round(2))
print("fraction of SIMDs where blockIdx increases with duration:", np.mean((np.diff(b4, axis=1) > 0).all(axis=1)))
first_start = tr[:, 0].min()
print("time of first tile stamp relative to earliest: pct", np.percentile(tr[:, 0] - first_start, [0, 50, 90, 100]))
ctx.close()
