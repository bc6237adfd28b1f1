#!/usr/bin/env python3
"""bench.py -- descriptor-pairs/s of the brute-force Hamming 2-NN + ratio hot path on MI355X.

A step = one pass of getMatches("LINEAR") device path (knn2_hamming partial + merge + ratio/compaction) over one
batch of `--pairs-per-gpu` (default 8) synthetic image pairs of BASELINE config C2 (8192 x 8192 ORB-256 each), one batched
launch per kernel, inputs resident in HBM.  The single-pair (latency) figure is reported under extras.  One process per GPU; image pairs shard across ranks with no data-path collective (weak scaling); with N > 1
the fixed-size per-pair result records (match counts) of every 8 steps (= a rank's 64-pair share of a C5 batch) are gathered by one
asynchronous RCCL all_gather that overlaps the next block's kernels.

Prints ONE JSON line (rank 0) with `roofline` (dominant kernel, measured live with HIP events inside the library,
see mlpl_profile_*) and `cpu_baseline` (the oracle's single-thread LUT port timed on this host).
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md)
BYTES_PER_PAIR = 64              # streaming-equivalent algorithmic bytes: two 32-byte operands
FP4_MFMA_PEAK_TFLOPS = 10000.0   # dense FP4 MFMA peak (MI355X_MICROARCH.md, Matrix cores table: ~10 PF dense)
FLOP_PER_PAIR = 2 * 256          # one +-1 multiply-add per descriptor bit: the distance table is a K = 256 GEMM
N_SIMD = 256 * 4
# --- matrix-core kernel (default): per unit = one 32 x 32 distance tile = 1024 pairs the kernel issues 4 fp4 MFMAs and 22 VALU
# ops (20 for the running top-2, 2 re-base adds).  tools/mfma_unit_probe.hip measures that instruction stream with operands in
# registers and no memory traffic: 226 / 138 / 111 / 100 cycles per unit per SIMD at 1 / 2 / 3 / 4 resident waves, at the
# 1.85 GHz the chip holds under this load.  The 4-wave figure is the vector-issue floor (22 x ~4.3 cycles + MFMA issue).
MFMA_UNIT_PAIRS = 1024
MFMA_UNIT_FLOOR_CYCLES = 100.0
MFMA_CLOCK_HZ = 1.85e9
# --- VALU kernels (--hamming-variant 0/1/2): 8 v_xor + 8 v_bcnt(acc) + v_lshl_or + v_med3 + v_min per descriptor pair.
# Issue cost per wave and train row from tools/valu_peak.hip (8 waves/SIMD, cycles per wave-instruction per SIMD at 2.4 GHz).
OPS_PER_PAIR = 19
MIX_CYCLES_PER_WAVE_ROW = 8 * 3.24 + 8 * 4.73 + 4.8 + 4.8 + 3.24
CLOCK_HZ = 2.4e9


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--pairs-per-gpu", type=int, default=8, help="image pairs per rank per step (one batched launch)")
    ap.add_argument("--n", type=int, default=8192, help="descriptors per image")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-queries", type=int, default=8192)
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket the dominant kernel with HIP events")
    ap.add_argument("--kernel-event-every", type=int, default=8,
                    help="bracket every Nth launch of the dominant kernel inside the timed region (two event records cost ~10 us)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N > 1)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--gather-every", type=int, default=8, help="steps per record gather (N > 1): 8 steps x 8 pairs = a C5 shard of 64 pairs")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary (RANSAC / L2) measurements")
    ap.add_argument("--hamming-variant", type=int, default=3,
                    help="3 = fp4 matrix-core kernel (library default), 0/1/2 = the integer VALU kernels")
    args = ap.parse_args()

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus}"
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if world > 1:
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)

    import matchinglib_poselib_amd as mpa
    from matchinglib_poselib_amd import _lib, synth
    from matchinglib_poselib_amd.matching import match_hamming_device

    ctx = mpa.Context(local_rank)
    lib = ctx.lib
    ctx.set_option("hamming_variant", args.hamming_variant)
    P, n = args.pairs_per_gpu, args.n
    # synthetic C2 inputs, one distinct pair per (rank, slot); resident in HBM before the timed region
    qs, ts = [], []
    for p in range(P):
        q, t = synth.orb_pair(n, n, seed=20260102 + rank * P + p)
        qs.append(q)
        ts.append(t)
    d_q = torch.from_numpy(np.stack(qs)).to(dev)
    d_t = torch.from_numpy(np.stack(ts)).to(dev)
    stream = None  # = torch's current stream, so torch/RCCL work is ordered after our kernels
    # Result records (here: the match counts) are gathered per SHARD BLOCK of G steps = G * P image pairs per rank (G = 8, P = 8:
    # the 64 pairs a rank owns of BASELINE's C5 batch of 512 on 8 GPUs), not per step: the kernels of a step write their counts
    # straight into their row of the block's record buffer, and one asynchronous RCCL all_gather per block runs on RCCL's stream
    # while the next block computes (two record buffers; a buffer is reused only after its gather has completed).
    G = max(1, args.gather_every)
    first = match_hamming_device(d_q, d_t, ratio_test=True, ratio=0.75, ctx=ctx, stream=stream)   # allocates idx/dist/matches
    records = [torch.zeros((G * P,), dtype=torch.int32, device=dev) for _ in range(2)]
    outs = [[dict(first, count=records[b][g * P:(g + 1) * P]) for g in range(G)] for b in range(2)]
    gathered = [torch.empty((world * G * P,), dtype=torch.int32, device=dev) for _ in range(2)] if world > 1 else None
    pending = [None, None]
    step_no = 0
    out = None

    def flush(b):
        if world > 1:
            pending[b] = dist.all_gather_into_tensor(gathered[b], records[b], async_op=True)

    def step():
        nonlocal out, step_no
        b, g = (step_no // G) & 1, step_no % G
        if g == 0 and pending[b] is not None:
            pending[b].wait()
            pending[b] = None
        out = match_hamming_device(d_q, d_t, ratio_test=True, ratio=0.75, ctx=ctx, out=outs[b][g], stream=stream)
        step_no += 1
        if step_no % G == 0:
            flush(b)

    def barrier():
        if step_no % G:  # a partial block at the end of a phase
            flush((step_no // G) & 1)
        for i in (0, 1):
            if pending[i] is not None:
                pending[i].wait()
                pending[i] = None
        if world > 1:
            dist.barrier()
        torch.cuda.synchronize()

    for _ in range(args.warmup):
        step()
    barrier()
    _lib.check(lib.mlpl_profile_reset(ctx.handle), "profile_reset")
    _lib.check(lib.mlpl_profile_enable(ctx.handle, 0 if args.no_kernel_events else max(1, args.kernel_event_every)), "profile_enable")
    barrier()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    barrier()
    elapsed = time.perf_counter() - t0
    _lib.check(lib.mlpl_profile_enable(ctx.handle, 0), "profile_enable")
    if world > 1:
        tt = torch.tensor([elapsed], dtype=torch.float64, device=dev)
        dist.all_reduce(tt, op=dist.ReduceOp.MAX)
        elapsed = float(tt.item())

    tot_ms, launches = C.c_double(0), C.c_int(0)
    _lib.check(lib.mlpl_profile_read(ctx.handle, 0, C.byref(tot_ms), C.byref(launches)), "profile_read")
    kern_ms = tot_ms.value / max(launches.value, 1)

    pairs_per_step_rank = P * n * n
    value = world * pairs_per_step_rank * args.steps / elapsed
    counts = out["count"].cpu().numpy().tolist()
    if world > 1:  # every rank must hold every rank's records after the last gather
        lastb = ((step_no - 1) // G) & 1
        g = gathered[lastb].cpu().numpy().reshape(world, G * P)
        mine = records[lastb].cpu().numpy()
        assert (g[rank] == mine).all() and (g[:, :P] > 0).all(), "gathered records are wrong"
    if rank == 0:
        kern_ms = kern_ms if kern_ms > 0 else float('nan')
        hbm_equiv = pairs_per_step_rank * BYTES_PER_PAIR / (kern_ms * 1e-3) / 1e9   # north_star's "HBM-roofline GB/s" reading
        mfma_path = args.hamming_variant == 3
        kernel_name = "knn_hamming_mfma_kernel<4, 4>" if mfma_path else "knn_hamming_partial_kernel<8>"
        traffic = None
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tf):
            try:
                traffic = json.load(open(tf)).get("knn_hamming_mfma_bytes_per_launch" if mfma_path
                                                  else "knn_hamming_partial_bytes_per_launch")
            except Exception:
                traffic = None
        if mfma_path:
            flops = pairs_per_step_rank * FLOP_PER_PAIR          # algorithmic FLOP per launch of the dominant kernel
            achieved = flops / (kern_ms * 1e-3) / 1e12
            floor_ms = pairs_per_step_rank / MFMA_UNIT_PAIRS / N_SIMD * MFMA_UNIT_FLOOR_CYCLES / MFMA_CLOCK_HZ * 1e3
            roofline = {
                "kernel": kernel_name,
                "bound": "mfma",
                "achieved": achieved,
                "peak": FP4_MFMA_PEAK_TFLOPS,
                "unit": "TFLOP/s",
                "frac": achieved / FP4_MFMA_PEAK_TFLOPS,
                "traffic": traffic,
                "kernel_ms_avg": kern_ms,
                "launches_timed": launches.value,
                "note": "the all-pairs Hamming table as a +-1 GEMM on v_mfma_scale_f32_32x32x64_f8f6f4 (fp4 x fp4, fp32 accumulate, "
                        "exact): 2*256 FLOP per descriptor pair against the dense FP4 peak.  The kernel is bound by VECTOR ISSUE, not "
                        "by the matrix pipe: per 32x32 tile it issues 4 MFMAs and 22 VALU ops (running top-2); issue_floor_ms is that "
                        "stream's measured cost with operands in registers (tools/mfma_unit_probe.hip, 4 waves/SIMD)",
                "flop_per_pair": FLOP_PER_PAIR,
                "issue_floor_ms": floor_ms,
                "issue_frac": floor_ms / kern_ms,
                "hbm_equiv_GBps": hbm_equiv,
                "hbm_equiv_frac": hbm_equiv / HBM_PEAK_GBS,
            }
            dtype = "fp4 (E2M1 +-1 per descriptor bit) MFMA, fp32 accumulate -- exact integer distances"
        else:
            roofline = {
                "kernel": kernel_name,
                "bound": "hbm",
                "achieved": hbm_equiv,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": hbm_equiv / HBM_PEAK_GBS,
                "traffic": traffic,
                "kernel_ms_avg": kern_ms,
                "launches_timed": launches.value,
                "note": "streaming-equivalent bytes (64 B per descriptor pair, SURVEY 8(d)); operands are LDS/L2-"
                        "resident so the kernel is integer-VALU bound: valu_frac = measured instruction-issue floor of the xor/bcnt/"
                        "med3 mix (tools/valu_peak.hip) / kernel time",
                "valu_ops_per_pair": OPS_PER_PAIR,
                "valu_achieved_Tops": pairs_per_step_rank * OPS_PER_PAIR / (kern_ms * 1e-3) / 1e12,
                "valu_issue_floor_ms": pairs_per_step_rank / 64 * MIX_CYCLES_PER_WAVE_ROW / N_SIMD / CLOCK_HZ * 1e3,
                "valu_frac": (pairs_per_step_rank / 64 * MIX_CYCLES_PER_WAVE_ROW / N_SIMD / CLOCK_HZ * 1e3) / kern_ms,
            }
            dtype = "u32 (xor + popcount on packed 256-bit descriptors)"
        rec = {
            "metric": "descriptor-pairs/s (8k x 8k ORB BF-Hamming kNN=2 + ratio)",
            "value": value,
            "unit": "descriptor-pairs/s",
            "n_gpus": world,
            "steps": args.steps,
            "warmup": args.warmup,
            "ms_per_step": elapsed / args.steps * 1e3,
            "higher_is_better": True,
            "scaling": "weak",
            "vs_baseline": None,
            "dtype": dtype,
            "data": "synthetic",
            "config": {
                "workload": f"C2: {n}x{n} ORB-256 BF-Hamming kNN=2 + 0.75 ratio + DMatch compaction, "
                            f"{P} image pair(s) per GPU per step",
                "pairs_per_gpu": P,
                "matches_first_pair": counts[0],
                "parallelism": f"shard{world}",
                "records_gathered_every_steps": G,
                "hamming_kernel": kernel_name,
            },
            "roofline": roofline,
        }
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib
            ora = oracle_lib.load()
            nqs = min(args.cpu_sample_queries, n)
            tc = time.perf_counter()
            ora.knn_hamming(qs[0][:nqs], ts[0])
            tc = time.perf_counter() - tc
            rec["cpu_baseline"] = {
                "value": nqs * n / tc,
                "unit": "descriptor-pairs/s",
                "cores": 1,
                "kind": "port",
                "sample": f"{nqs} of {n} queries x {n} train rows of the same C2 pair (byte-LUT popcount, serial, "
                          f"{tc:.2f} s)",
                "host_cores_available": os.cpu_count(),
            }
            # BASELINE.md "CPU-best" tier: same results with hardware popcount + OpenMP on the host cores we may use
            nthreads = min(16, os.cpu_count() or 1)
            tb = time.perf_counter()
            _, _, used = ora.knn_hamming_fast(qs[0], ts[0], threads=nthreads)
            tb = time.perf_counter() - tb
            rec["cpu_best"] = {"value": n * n / tb, "unit": "descriptor-pairs/s", "cores": used, "kind": "port",
                               "sample": f"one full C2 pair, popcnt64 + OpenMP ({tb:.3f} s)"}
        if not args.no_extras:
            try:
                import bench_extras
                rec["extras"] = bench_extras.run(ctx, dev, cpu_baseline=not args.no_cpu_baseline)
            except ImportError:
                pass
        print(json.dumps(rec), flush=True)
    if world > 1:
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
