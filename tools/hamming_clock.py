"""In-kernel clock and cycles per 32x32 unit of the matrix-core Hamming kernel (diagnostic build path: option hamming_stamps).
Runs >= 2 s of back-to-back launches on random data first (DVFS settles), then reads the per-wave stamps of one launch."""
import ctypes as C, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import synth
from matchinglib_poselib_amd.matching import match_hamming_device

def run(ctx, P, n, bpc, qt, zero=False, secs=2.0, lds=1, prio=1, weighted=1):
    dev = torch.device("cuda", 0)
    qs, ts = zip(*[synth.orb_pair(n, n, seed=20260102 + p) for p in range(P)])
    dq = torch.from_numpy(np.stack(qs)).to(dev); dt = torch.from_numpy(np.stack(ts)).to(dev)
    if zero:
        dq.zero_(); dt.zero_()
    ctx.set_option("hamming_mfma_blocks_per_cu", bpc); ctx.set_option("hamming_mfma_qt", qt); ctx.set_option("hamming_mfma_lds", lds); ctx.set_option("hamming_mfma_prio", prio); ctx.set_option("hamming_mfma_weighted", weighted)
    out = match_hamming_device(dq, dt, ctx=ctx)
    torch.cuda.synchronize()
    t0 = time.perf_counter(); k = 0
    while time.perf_counter() - t0 < secs:
        for _ in range(50):
            out = match_hamming_device(dq, dt, ctx=ctx, out=out)
        torch.cuda.synchronize(); k += 50
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(100):
        out = match_hamming_device(dq, dt, ctx=ctx, out=out)
    e1.record(); torch.cuda.synchronize()
    step_us = e0.elapsed_time(e1) * 10
    ctx.set_option("hamming_stamps", 1)
    for _ in range(20):
        out = match_hamming_device(dq, dt, ctx=ctx, out=out)
    torch.cuda.synchronize()
    buf = np.zeros((1 << 17, 4), np.uint64)
    ctx.lib.mlpl_debug_hamming_stamps.argtypes = [C.c_void_p, C.c_void_p, C.c_int]
    m = ctx.lib.mlpl_debug_hamming_stamps(ctx.handle, buf.ctypes.data, len(buf))
    ctx.set_option("hamming_stamps", 0)
    raw = buf[:m]
    units = (raw[:, 2] & np.uint64(0xFFFFFFFF)).astype(np.float64)
    hw = ((raw[:, 2] >> np.uint64(32)) & np.uint64(0xFFFFF)).astype(np.int64)
    xcc = (raw[:, 2] >> np.uint64(56)).astype(np.int64)
    s = raw.astype(np.float64)
    s[:, 2] = units
    clk = np.median(s[:, 0] / s[:, 1]) * 100e6
    cyc_unit = np.median(s[:, 0] / s[:, 2])
    start = (s[:, 3] - s[:, 3].min()) / 100.0
    dur = s[:, 1] / 100.0
    end = start + dur
    # gfx9 HW_ID: wave_id[3:0] simd_id[5:4] pipe[7:6] cu_id[11:8] sh_id[12] se_id[15:13]
    simd = (hw >> 4) & 3; cu = (hw >> 8) & 15; sh = (hw >> 12) & 1; se = (hw >> 13) & 7
    key = ((xcc * 8 + se) * 2 + sh) * 16 + cu
    ncu = len(np.unique(key))
    per_cu = np.bincount(np.unique(key, return_inverse=True)[1])
    skey = key * 4 + simd
    per_simd = np.bincount(np.unique(skey, return_inverse=True)[1])
    print(f"   start pct [0,50,90,99,100] us: {np.percentile(start,[0,50,90,99,100]).round(1)}  dur pct: {np.percentile(dur,[0,10,50,90,100]).round(1)}  "
          f"end pct [0,10,25,50,75,90,100]: {np.percentile(end,[0,10,25,50,75,90,100]).round(1)}  CUs seen {ncu}, waves/CU min/max {per_cu.min()}/{per_cu.max()}, "
          f"waves/SIMD min/max {per_simd.min()}/{per_simd.max()} (SIMDs seen {len(per_simd)}), per-XCD waves {np.bincount(xcc).tolist()}")
    span = (s[:, 3].max() + s[:, 1].max() - s[:, 3].min()) / 100.0   # us, launch start -> last wave end (approx)
    units_total = P * (n // 32) ** 2
    waves = m
    print(f"P={P} n={n} bpc={bpc} qt={qt} zero={zero} lds={lds} prio={prio} weighted={weighted}: step {step_us:.1f} us, waves {waves}, in-kernel clock {clk/1e9:.3f} GHz, "
          f"wave-cycles/unit {cyc_unit:.0f}, kernel span ~{span:.1f} us, units/SIMD {units_total/1024:.0f}, "
          f"SIMD-cycles/unit (span) {span*1e-6*clk/(units_total/1024):.0f}")

ctx = mpa.Context(0)
CASES = [(8, 4, 0, 1, 0), (8, 4, 0, 1, 1), (1, 4, 0, 1, 0), (1, 4, 0, 1, 1), (4, 4, 0, 1, 0), (4, 4, 0, 1, 1)]
if len(sys.argv) > 1 and sys.argv[1] == 'c64':  # the headline launch (64 pairs): where the waves start and end
    CASES = [(64, 4, 0, 1, 1), (64, 4, 0, 1, 1)]
if len(sys.argv) > 1 and sys.argv[1] == 'ab':  # same-box A/B of the 8-pair step: register-prefetch kernel (r1) vs LDS ring, twice
    CASES = [(8, 3, 0, 0, 0), (8, 4, 0, 1, 0), (8, 4, 0, 1, 1), (8, 3, 0, 0, 0), (8, 4, 0, 1, 1)]
for P, bpc, qt, lds, wt in CASES:
    run(ctx, P, 8192, bpc, qt, False, 1.0, lds, 0, wt)
ctx.close()
