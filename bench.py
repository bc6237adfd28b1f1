#!/usr/bin/env python3
"""bench.py -- descriptor-pairs/s of the brute-force Hamming 2-NN + ratio hot path on MI355X.

A step = one pass of getMatches("LINEAR") device path (knn2_hamming partial + merge + ratio/compaction) over one
batch of `--pairs-per-gpu` (default 64 = a rank's share of BASELINE's 512-pair batch on 8 GPUs; rounds 1-2 used 8, still reported as
value_8_pairs_per_launch) synthetic image pairs of BASELINE config C2 (8192 x 8192 ORB-256 each), one batched
launch per kernel, inputs resident in HBM.  The single-pair (latency) figure is reported beside it.  One process per GPU; image pairs shard across ranks with no data-path collective (weak scaling); with N > 1
the fixed-size per-pair result records (match counts) of every step (= a rank's 64-pair share of a C5 batch) are gathered by one
asynchronous RCCL all_gather that overlaps the next step's kernels.

Rank 0 prints, LAST, one JSON line shorter than 4 KB with `roofline` (dominant kernel, measured live with HIP events inside the library,
see mlpl_profile_*) and `cpu_baseline` (the oracle's single-thread LUT port timed on this host).  Before it: one `bench_detail {...}` line
(everything: extras, per-step arrays, notes; also written to bench_detail.json) and one `bench_secondary {...}` line (C3 RANSAC hyp/s and
config 5, each with its own roofline and cpu_baseline) -- see bench_record.py.  The C5 measurement itself lives in bench_c5.py, the other
secondary measurements in bench_extras.py.

`python bench.py --gpus N` works by itself: when N > 1 and no torch.distributed environment is present, this process starts the N
rank processes (python -m torch.distributed.run on 127.0.0.1) BEFORE it touches the GPU and relays their output; under an external
launcher (RANK / WORLD_SIZE set) it is a rank.  `--workload c5` times BASELINE config 5 instead: 512 stereo pairs, each through the
whole per-pair pipeline (8k ORB match -> gather -> 5-pt RANSAC -> cheirality), dealt to the ranks by batch.pair_shard, a rank's share
as batches with the pair as a grid dimension (mlpl_pair_pose_batch_dev), records gathered by one RCCL all_gather per step and the
padded match lists sent to rank 0 by grouped send / recv; the line carries `roofline` (dominant kernel of the pipeline) and
`cpu_baseline` (the oracle pipeline on a sample of the pairs).  The default (C2) line carries the same C5 measurement under
extras.c5_batch_sharded at every N.
"""
import argparse
import ctypes as C
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)
from bench_c5 import FORCE_DIST, dist_on, measure_c5   # noqa: E402  (config 5 as a measurement; --force-dist = every collective of the N > 1 path with ONE rank)

HBM_PEAK_GBS = 8000.0            # MI355X HBM3E spec (MI355X_MICROARCH.md)
BYTES_PER_PAIR = 64              # streaming-equivalent algorithmic bytes: two 32-byte operands
FP4_MFMA_PEAK_TFLOPS = 10000.0   # dense FP4 MFMA peak (MI355X_MICROARCH.md, Matrix cores table: ~10 PF dense)
FLOP_PER_PAIR = 2 * 256          # one +-1 multiply-add per descriptor bit: the distance table is a K = 256 GEMM
N_SIMD = 256 * 4
# --- matrix-core kernel (default): per unit = one 32 x 32 distance tile = 1024 pairs the kernel issues 4 fp4 MFMAs and 22 VALU
# ops (20 for the running top-2, 2 re-base adds).  tools/hamming_unit_probe3.hip (occupancy-controlled, AGGREGATE throughput = all
# units / wall time, in-kernel shader clock; round 5: the UNSCALED opcode): the 4 MFMAs alone cost 140 cycles per unit per SIMD at
# 4 waves (LDS-fed operands; nominal 4 x 32 = 128), MFMA + top-2 update 159 with the LDS ring at 8 waves per workgroup (they do not
# overlap fully on a SIMD: part of the VALU work adds), at the ~1.9 GHz the chip holds on random descriptors (2.3 GHz on zeros).
# Rounds 2-4 (scaled opcode, 4-wave workgroups): 141 / 175.
MFMA_UNIT_PAIRS = 1024
MFMA_UNIT_FLOOR_CYCLES = 140.0       # round 5, unscaled opcode: 139.8 (the scaled form of rounds 1-4: 140.8 -- the scale prefix costs in the MIXED stream)
MFMA_UNIT_WITH_TOP2_CYCLES = 159.0   # round 5: 159.1 for 8-wave workgroups with the LDS ring (`ringnw`), = the stream without synchronisation (rounds 2-4, scaled: 175)
MFMA_CLOCK_HZ = 1.9e9
# --- VALU kernels (--hamming-variant 0/1/2): 8 v_xor + 8 v_bcnt(acc) + v_lshl_or + v_med3 + v_min per descriptor pair.
# Issue cost per wave and train row from tools/valu_peak.hip (8 waves/SIMD, cycles per wave-instruction per SIMD at 2.4 GHz).
OPS_PER_PAIR = 19
MIX_CYCLES_PER_WAVE_ROW = 8 * 3.24 + 8 * 4.73 + 4.8 + 4.8 + 3.24
CLOCK_HZ = 2.4e9


def spawn_ranks(n):
    """Parent of a self-launched multi-GPU run: start n rank processes and relay their output.  Nothing here may touch the GPU
    (a process that has initialised HIP must not be replaced or forked into ranks), so torch is not even imported."""
    import socket
    import subprocess

    with socket.socket() as sock:
        sock.bind(("127.0.0.1", 0))
        port = sock.getsockname()[1]
    env = dict(os.environ)
    env.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    env.setdefault("OMP_NUM_THREADS", "4")
    cmd = [sys.executable, "-m", "torch.distributed.run", "--nnodes=1", f"--nproc-per-node={n}", "--master-addr", "127.0.0.1",
           "--master-port", str(port), "--", os.path.abspath(__file__)] + sys.argv[1:]   # "--": the launcher's parser must not read bench.py's options (--n is a prefix of its --nnodes)
    return subprocess.run(cmd, env=env).returncode


def run_c5(args, rank, local_rank, world, dev, ctx):
    rec = measure_c5(args, rank, local_rank, world, dev, ctx, args.steps, args.warmup, not args.no_cpu_baseline)
    if rank == 0:
        import bench_record
        rec["c5"] = {k: v for k, v in rec.items() if k != "c5"}   # the same object as the secondary line's `c5`
        if dist_on(world):
            rec["rccl_ranks_seen"] = RCCL_RANKS_SEEN.get("n")
        bench_record.emit(rec, os.path.join(ROOT, "bench_detail.json"))


RCCL_RANKS_SEEN = {}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=200)
    ap.add_argument("--warmup", type=int, default=20)
    ap.add_argument("--pairs-per-gpu", type=int, default=64, help="image pairs per rank per step (one batched launch per kernel)")
    ap.add_argument("--n", type=int, default=8192, help="descriptors per image")
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--cpu-sample-queries", type=int, default=8192)
    ap.add_argument("--c5-cpu-pairs", type=int, default=8, help="pairs of the C5 batch the oracle pipeline is timed (and checked) on")
    ap.add_argument("--cpu-sample-pairs", type=int, default=16, help="C2 pairs of the step the serial CPU port is timed (and checked) on")
    ap.add_argument("--no-kernel-events", action="store_true", help="do not bracket the dominant kernel with HIP events")
    ap.add_argument("--kernel-event-every", type=int, default=8,
                    help="bracket every Nth launch of the dominant kernel inside the timed region (two event records cost ~10 us)")
    ap.add_argument("--backend", default="nccl", help="torch.distributed backend (nccl = RCCL; gloo only to rehearse N > 1)")
    ap.add_argument("--share-gpu", action="store_true", help="rehearsal only: every rank uses cuda:0")
    ap.add_argument("--force-dist", action="store_true",
                    help="rehearsal only: with --gpus 1, initialise the process group (world size 1) and run every collective of the N > 1 path")
    ap.add_argument("--gather-every", type=int, default=1, help="steps per record gather (N > 1): one step of 64 pairs = a C5 shard")
    ap.add_argument("--no-extras", action="store_true", help="skip the secondary (RANSAC / L2) measurements")
    ap.add_argument("--hamming-variant", type=int, default=3,
                    help="3 = fp4 matrix-core kernel (library default), 0/1/2 = the integer VALU kernels")
    ap.add_argument("--workload", default="c2", choices=["c2", "c5"],
                    help="c2 (default, the headline metric): batched 8k x 8k Hamming matching; c5: 512 stereo pairs through the whole per-pair pipeline")
    ap.add_argument("--c5-pairs", type=int, default=512)
    ap.add_argument("--c5-distinct", type=int, default=8, help="distinct synthetic inputs generated per rank (cycled over its shard)")
    ap.add_argument("--c5-steps", type=int, default=3, help="timed C5 batches of the extras block of the default (C2) line")
    ap.add_argument("--dump-records", default=None, help="c5 workload, rank 0: write the gathered records and match lists of the last step to this .npz (tests)")
    ap.add_argument("--steady-steps", type=int, default=200,
                    help="steps of the steady_state measurement that follows the timed region when --steps < this (0 = none)")
    ap.add_argument("--settle-steps", type=int, default=400,
                    help="untimed steps BEFORE the --warmup steps: the power controller needs ~100-300 launches to settle after the GPU sat idle "
                         "through the input generation (clock 1.87 -> 1.50 -> 1.9 GHz); the metric is steady-state throughput, the transient is "
                         "reported separately (after_idle).  0 = rounds 1-5: warm-up steps only")
    ap.add_argument("--step-events", type=int, default=1, help="1 = one HIP event per timed step (roofline.ms_steps_gpu); 0 = none inside the timed region")
    ap.add_argument("--after-idle-launches", type=int, default=24,
                    help="launches whose clock / duration are recorded after the GPU sat idle during the CPU baseline (0 = none)")
    ap.add_argument("--hamming-train01", type=int, default=-1, help="encoding of the train operand of the matrix-core kernel: 1 = {0, +1}, 0 = +-1, -1 = library default")
    ap.add_argument("--estimator", default="ransac", choices=["ransac", "usac", "usac_prosac", "usac_default_refine", "arrsac"],
                    help="robust estimator of the C5 pipeline (--workload c5 and the c5 object of the default line)")
    args = ap.parse_args()

    if args.gpus > 1 and "WORLD_SIZE" not in os.environ:
        sys.exit(spawn_ranks(args.gpus))          # before any GPU call: this process only launches and relays

    import numpy as np
    import torch
    import torch.distributed as dist

    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    assert world == args.gpus, f"WORLD_SIZE={world} but --gpus {args.gpus} (under an external launcher pass --gpus = its world size)"
    FORCE_DIST["on"] = bool(args.force_dist)
    if args.force_dist and "MASTER_PORT" not in os.environ:
        import socket
        with socket.socket() as sock:
            sock.bind(("127.0.0.1", 0))
            os.environ["MASTER_PORT"] = str(sock.getsockname()[1])
    if args.share_gpu:
        local_rank = 0
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    if dist_on(world):
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        if args.backend == "nccl":
            dist.init_process_group("nccl", rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
        # proof in the record that the collective library saw N ranks: the world size after init and one all_reduce of ones
        ones = torch.ones(1, dtype=torch.int32, device=dev if args.backend == "nccl" else None)
        dist.all_reduce(ones)
        RCCL_RANKS_SEEN["n"] = {"world_size": dist.get_world_size(), "all_reduce_of_ones": int(ones.item()), "backend": dist.get_backend()}
        assert RCCL_RANKS_SEEN["n"]["all_reduce_of_ones"] == world

    import matchinglib_poselib_amd as mpa
    from matchinglib_poselib_amd import _lib, synth
    from matchinglib_poselib_amd.matching import match_hamming_device

    ctx = mpa.Context(local_rank)
    lib = ctx.lib
    ctx.set_option("hamming_variant", args.hamming_variant)
    if args.hamming_train01 >= 0:
        ctx.set_option("hamming_train01", args.hamming_train01)
    # (the timed region and the steady-state pass run the PRODUCTION kernel: no clock stamps; a separate stamped pass behind them records
    #  the shader clock -- hamming_stamps = 2 adds two clock reads and one 32-byte store to the kernel's first workgroup)
    if args.workload == "c5":
        if args.steps == 200 and args.warmup == 20:   # the defaults are sized for the C2 step; a C5 step is a whole batch
            args.steps, args.warmup = 5, 2
        run_c5(args, rank, local_rank, world, dev, ctx)
        if dist_on(world):
            dist.barrier()
            dist.destroy_process_group()
        ctx.close()
        return
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
    gathered = [torch.empty((world * G * P,), dtype=torch.int32, device=dev) for _ in range(2)] if dist_on(world) else None
    pending = [None, None]
    step_no = 0
    out = None

    def flush(b):
        if dist_on(world):
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
        if dist_on(world):
            dist.barrier()
        torch.cuda.synchronize()

    def read_clock(nlast):
        """(GHz, first-workgroup lifetime in us, start in us since the first record) of the last `nlast` launches of the matrix-core kernel:
        s_memtime (shader clock) and s_memrealtime (100 MHz) deltas of its work item 0, recorded by the kernel itself (hamming_stamps = 2)"""
        buf = np.zeros((256, 4), np.uint64)
        m = lib.mlpl_debug_hamming_clock(ctx.handle, buf.ctypes.data, int(min(nlast, 256)))
        if m <= 0:
            return np.zeros(0), np.zeros(0), np.zeros(0)
        r = buf[:m].astype(np.float64)
        return r[:, 0] / np.maximum(r[:, 1], 1.0) * 0.1, r[:, 1] / 100.0, (r[:, 2] - r[0, 2]) / 100.0

    def timed_region(nsteps, per_step_events):
        """EXACTLY nsteps steps between two barriers (+ device synchronisation); returns (wall seconds, max over ranks; kernel ms by HIP
        events; launches bracketed; per-step GPU ms from one event per step on the launch stream; per-launch clock records)"""
        _lib.check(lib.mlpl_profile_reset(ctx.handle), "profile_reset")
        _lib.check(lib.mlpl_profile_enable(ctx.handle, 0 if args.no_kernel_events else max(1, args.kernel_event_every)), "profile_enable")
        evs = [torch.cuda.Event(enable_timing=True) for _ in range(nsteps + 1)] if per_step_events else None
        barrier()
        t0 = time.perf_counter()
        if evs:
            evs[0].record()
        for i in range(nsteps):
            step()
            if evs:
                evs[i + 1].record()
        barrier()
        el = time.perf_counter() - t0
        _lib.check(lib.mlpl_profile_enable(ctx.handle, 0), "profile_enable")
        if dist_on(world):
            tt = torch.tensor([el], dtype=torch.float64, device=dev)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        tot_ms, launches = C.c_double(0), C.c_int(0)
        _lib.check(lib.mlpl_profile_read(ctx.handle, 0, C.byref(tot_ms), C.byref(launches)), "profile_read")
        ms_steps = [evs[i].elapsed_time(evs[i + 1]) for i in range(nsteps)] if evs else None
        return el, tot_ms.value / max(launches.value, 1), launches.value, ms_steps, (read_clock(nsteps) if stamped["on"] else (np.zeros(0),) * 3)

    stamped = {"on": False}

    def set_stamps(on):
        ctx.set_option("hamming_stamps", 2 if on else 0)
        stamped["on"] = bool(on)

    for _ in range(max(0, args.settle_steps)):   # untimed: out of the idle -> load transient of the power controller (see --settle-steps)
        step()
    barrier()
    for _ in range(args.warmup):
        step()
    elapsed, kern_ms, n_bracketed, ms_steps_gpu, clk_timed = timed_region(args.steps, bool(args.step_events))
    # steady state (VERDICT r4 #1b): the driver's protocol (20 steps after 5 warm-ups) times the first ~11 ms of GPU work after the input
    # upload; the same step in a long run is measured right behind it, same buffers, same verification below.  `value` stays the former.
    steady = None
    if args.steady_steps > args.steps and args.steady_steps > 0:
        st_el, st_kern, st_n, _, _ = timed_region(args.steady_steps, False)
        steady = {"steps": args.steady_steps, "ms_per_step": st_el / args.steady_steps * 1e3, "kernel_ms_avg": st_kern, "launches_timed": st_n,
                  "value": world * P * n * n * args.steady_steps / st_el, "clock_GHz_median": None}
        if args.hamming_variant == 3:
            # the same steps once more WITH the per-launch clock record: the shader clock of the steady state, and the A/B that the stamp is free
            set_stamps(True)
            sk_el, sk_kern, sk_n, _, st_clk = timed_region(args.steady_steps, False)
            set_stamps(False)
            steady["clock_GHz_median"] = float(np.median(st_clk[0])) if len(st_clk[0]) else None
            steady["stamped_pass"] = {"steps": args.steady_steps, "ms_per_step": sk_el / args.steady_steps * 1e3, "kernel_ms_avg": sk_kern,
                                      "launches_timed": sk_n, "what": "hamming_stamps = 2 (one clock record per launch); the figures above are without it"}

    class _L:   # (the name the code below reads the number of bracketed launches from)
        value = n_bracketed
    launches = _L

    pairs_per_step_rank = P * n * n
    value = world * pairs_per_step_rank * args.steps / elapsed
    counts = out["count"].cpu().numpy().tolist()
    # a wrong kernel must not produce a fast number silently: half of the queries of every synthetic pair are true matches that pass
    # the ratio test, the rest (almost) never do
    assert all(abs(c - n // 2) <= max(8, n // 100) for c in counts), f"match counts {counts} are not ~{n // 2}"
    # the literal config 2 (ONE image pair per launch, latency shape) beside the batched headline
    single_ms = eight_ms = None
    if rank == 0:
        o1 = match_hamming_device(d_q[:1], d_t[:1], ratio_test=True, ratio=0.75, ctx=ctx, stream=stream)
        for _ in range(50):   # (the first calls of a new launch shape size its workspaces and split tables)
            o1 = match_hamming_device(d_q[:1], d_t[:1], ratio_test=True, ratio=0.75, ctx=ctx, out=o1, stream=stream)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(300):
            o1 = match_hamming_device(d_q[:1], d_t[:1], ratio_test=True, ratio=0.75, ctx=ctx, out=o1, stream=stream)
        e1.record()
        torch.cuda.synchronize()
        single_ms = e0.elapsed_time(e1) / 300
        assert int(o1["count"][0].item()) == counts[0] or P == 0
        eight_ms = None
        if P >= 8:   # the launch shape of rounds 1-2 (8 pairs per launch), for continuity
            o8 = match_hamming_device(d_q[:8], d_t[:8], ratio_test=True, ratio=0.75, ctx=ctx, stream=stream)
            torch.cuda.synchronize()
            e0.record()
            for _ in range(50):
                o8 = match_hamming_device(d_q[:8], d_t[:8], ratio_test=True, ratio=0.75, ctx=ctx, out=o8, stream=stream)
            e1.record()
            torch.cuda.synchronize()
            eight_ms = e0.elapsed_time(e1) / 50
    if dist_on(world):  # every rank must hold every rank's records after the last gather
        lastb = ((step_no - 1) // G) & 1
        g = gathered[lastb].cpu().numpy().reshape(world, G * P)
        mine = records[lastb].cpu().numpy()
        assert (g[rank] == mine).all() and (g[:, :P] > 0).all(), "gathered records are wrong"
    if rank == 0:
        kern_ms = kern_ms if kern_ms > 0 else float('nan')
        hbm_equiv = pairs_per_step_rank * BYTES_PER_PAIR / (kern_ms * 1e-3) / 1e9   # north_star's "HBM-roofline GB/s" reading
        mfma_path = args.hamming_variant == 3
        kernel_name = "knn_hamming_mfma_lds_kernel<4, 0, 8>" if mfma_path else "knn_hamming_partial_kernel<8>"
        # traffic was measured with 8 pairs per launch: scale the per-launch figure to this run's launch size (it is per-pair work)
        traffic, traffic_src = None, None
        prof = {}   # counter-derived figures of the same launch shape from the committed profile (not measured in this run)
        tf = os.path.join(ROOT, "profiles", "pmc_traffic.json")
        if os.path.exists(tf):
            try:
                tj = json.load(open(tf))
                traffic = tj.get("knn_hamming_mfma_bytes_per_launch" if mfma_path else "knn_hamming_partial_bytes_per_launch")
                if traffic and tj.get("pairs_per_launch"):
                    traffic = traffic * P / float(tj["pairs_per_launch"])
                traffic_src = f"from profiles/pmc_traffic.json (round {tj.get('round', '?')}, {tj.get('pairs_per_launch', '?')} pairs per " \
                              "launch; rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE in separate passes) -- not measured in this run"
                if mfma_path and tj.get("pairs_per_launch") == P:
                    prof = {k: tj[k] for k in ("mfma_busy_frac", "mfma_busy_how", "shader_clock_GHz_under_pmc", "kernel_us_rocprof_trace") if k in tj}
            except Exception:
                traffic = None
        def fmt(a, nd=3):   # a list of figures as ONE short string (the driver's parsed record keeps scalars and strings, <= 128 characters)
            return " ".join((f"{x:.{nd}f}".lstrip("0") if 0 <= x < 1 else f"{x:.{nd}f}") for x in a)

        if mfma_path:
            flops = pairs_per_step_rank * FLOP_PER_PAIR          # algorithmic FLOP per launch of the dominant kernel
            achieved = flops / (kern_ms * 1e-3) / 1e12
            floor_ms = pairs_per_step_rank / MFMA_UNIT_PAIRS / N_SIMD * MFMA_UNIT_FLOOR_CYCLES / MFMA_CLOCK_HZ * 1e3
            # the first ten keys are the final line's `roofline` (bench_record.ROOFLINE_KEYS); the rest stays in bench_detail
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
                "mfma_busy_frac": prof.get("mfma_busy_frac"),   # from profiles/pmc_traffic.json (same launch shape; counters are not read in this run)
                # the protocol dependence in the record itself: GPU time of every timed step (one event per step on the launch stream)
                "ms_steps_gpu": fmt(ms_steps_gpu[:24]) if ms_steps_gpu else None,
                "steady_frac": (flops / (steady["kernel_ms_avg"] * 1e-3) / 1e12 / FP4_MFMA_PEAK_TFLOPS) if steady and steady["kernel_ms_avg"] > 0 else None,
                "steady_kernel_ms_avg": steady["kernel_ms_avg"] if steady else None,
                "steady_value": steady["value"] if steady else None,
                "steady_ms_per_step": steady["ms_per_step"] if steady else None,
                "steady_clock_GHz_median": steady["clock_GHz_median"] if steady else None,
                "frac_of_mfma_plus_top2": floor_ms * MFMA_UNIT_WITH_TOP2_CYCLES / MFMA_UNIT_FLOOR_CYCLES / kern_ms,
                "hbm_equiv_GBps": hbm_equiv,
                "traffic_source": traffic_src,
                "from_profiles": prof or None,
                "note": "the all-pairs Hamming table as a GEMM on v_mfma_f32_32x32x64_f8f6f4 (fp4 x fp4, fp32 accumulate, "
                        "exact): 2*256 FLOP per descriptor pair against the dense FP4 peak at the nominal 2.4 GHz.  Per 32x32 tile "
                        "the kernel issues 4 MFMAs and the VALU ops of the running top-2, which overlap only partly on a SIMD: "
                        "mfma_only_floor_ms is the measured cost of the MFMAs alone (140 cycles per tile per SIMD at 4 waves, "
                        "tools/hamming_unit_probe3.hip) at the ~1.9 GHz the chip holds on random descriptors, mfma_plus_top2_ms the "
                        "measured cost of MFMAs + top-2 update with the LDS ring at 8 waves per workgroup (159 cycles; the unscaled opcode).  steady_* = the same "
                        "step over --steady-steps further steps right behind the timed region",
                "flop_per_pair": FLOP_PER_PAIR,
                "mfma_only_floor_ms": floor_ms,
                "mfma_plus_top2_ms": floor_ms * MFMA_UNIT_WITH_TOP2_CYCLES / MFMA_UNIT_FLOOR_CYCLES,
                "frac_of_mfma_only_floor": floor_ms / kern_ms,
                "hbm_equiv_frac": hbm_equiv / HBM_PEAK_GBS,
            }
            dtype = "fp4 (E2M1 MFMA, f32 accumulate: exact)"
        else:
            roofline = {
                "kernel": kernel_name,
                "bound": "hbm",
                "achieved": hbm_equiv,
                "peak": HBM_PEAK_GBS,
                "unit": "GB/s",
                "frac": hbm_equiv / HBM_PEAK_GBS,
                "traffic": traffic,
                "traffic_source": traffic_src,
                "kernel_ms_avg": kern_ms,
                "launches_timed": launches.value,
                "ms_steps_gpu": fmt(ms_steps_gpu[:24]) if ms_steps_gpu else None,
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
            # `config`: bench_record.CONFIG_KEYS (<= 20 scalars) go to the final line, everything else stays in bench_detail
            "config": {
                "workload": f"C2: {n}x{n} ORB-256 BF-Hamming kNN=2 + 0.75 ratio + DMatch rows, {P} image pair(s) per GPU per step",
                "pairs_per_gpu": P,
                "value_single_pair": (n * n / (single_ms * 1e-3)) if single_ms else None,
                "ms_single_pair": single_ms,
                "value_8_pairs_per_launch": (8 * n * n / (eight_ms * 1e-3)) if eight_ms else None,
                "ms_8_pairs_per_launch": eight_ms,
                "steady_pairs_per_s": steady["value"] if steady else None,
                "steady_frac": roofline.get("steady_frac"),
                "steady_kernel_ms": steady["kernel_ms_avg"] if steady else None,
                "clock_GHz_steady_median": steady["clock_GHz_median"] if steady else None,
                "solver_polish": ctx.get_option("solver_polish"),
                "settle_steps": args.settle_steps,
                "gc": "default (the C2 steps are one library call each; the C5 steps freeze + disable it, see c5.config.host_threads)",
            },
            "roofline": roofline,
            "steady_state": steady,
        }
        rec["config"]["hamming_train_operand"] = "{0,+1}" if ctx.get_option("hamming_train01") == 1 else "+-1"
        if not args.no_cpu_baseline:
            sys.path.insert(0, os.path.join(ROOT, "tests"))
            import oracle_lib
            ora = oracle_lib.load()
            nqs = min(args.cpu_sample_queries, n)
            npairs_cpu = max(1, min(args.cpu_sample_pairs, P))   # ~0.56 s per C2 pair on one core: 16 pairs ~ 9 s
            gi_all, gd_all = out["idx"][:npairs_cpu, :nqs].cpu().numpy(), out["dist"][:npairs_cpu, :nqs].cpu().numpy()
            tc = 0.0
            for pp in range(npairs_cpu):
                t1 = time.perf_counter()
                oi, od = ora.knn_hamming(qs[pp][:nqs], ts[pp])
                tc += time.perf_counter() - t1
                assert np.array_equal(gi_all[pp], oi) and np.array_equal(gd_all[pp], od), f"the timed step's (idx, dist) of pair {pp} differ from the CPU path"
            rec["config"]["verified"] = f"(idx, dist) of the last timed step, pairs 0..{npairs_cpu - 1}, all queries: bit-exact vs the CPU port" if nqs == n else \
                                        f"(idx, dist) of the last timed step, pairs 0..{npairs_cpu - 1}, queries 0..{nqs - 1}: bit-exact vs the CPU port"
            rec["cpu_baseline"] = {
                "value": npairs_cpu * nqs * n / tc,
                "unit": "descriptor-pairs/s",
                "cores": 1,
                "kind": "port",
                "sample": f"{npairs_cpu} of the step's {P} C2 pairs, {nqs} of {n} queries each; byte-LUT popcount, serial, {tc:.1f} s",
                "host_cores_available": os.cpu_count(),
            }
            # BASELINE.md "CPU-best" tier: same results with hardware popcount + OpenMP on the host cores we may use
            nthreads = min(16, os.cpu_count() or 1)
            tb = time.perf_counter()
            _, _, used = ora.knn_hamming_fast(qs[0], ts[0], threads=nthreads)
            tb = time.perf_counter() - tb
            rec["cpu_best"] = {"value": n * n / tb, "unit": "descriptor-pairs/s", "cores": used, "kind": "port",
                               "sample": f"one full C2 pair, popcnt64 + OpenMP ({tb:.3f} s)"}
        # After idle (VERDICT r4 #1a): the GPU has just sat idle for the seconds of the CPU baseline, as it does before the driver's run
        # while the inputs are generated and uploaded.  The clock and duration of the first launches from that state, launch by launch,
        # tell a clock / power ramp (clock rises or falls over the launches) from first-touch effects (only launch 0 is slow) and from a
        # slow box (every figure low, steady state included).  Same buffers as the timed steps; not part of any reported rate.
        if args.after_idle_launches > 0 and mfma_path:
            if args.no_cpu_baseline:
                time.sleep(2.0)
            k = min(args.after_idle_launches, 200)
            evs = [torch.cuda.Event(enable_timing=True) for _ in range(k + 1)]
            set_stamps(True)
            torch.cuda.synchronize()
            evs[0].record()
            for i in range(k):
                match_hamming_device(d_q, d_t, ratio_test=True, ratio=0.75, ctx=ctx, out=outs[0][0], stream=stream)
                evs[i + 1].record()
            torch.cuda.synchronize()
            ai_ms = [evs[i].elapsed_time(evs[i + 1]) for i in range(k)]
            ai_ghz, ai_dur, _ = read_clock(k)
            set_stamps(False)
            rec["after_idle"] = {"launches": k, "ms_steps_gpu": [round(x, 4) for x in ai_ms], "clock_GHz": [round(float(x), 3) for x in ai_ghz],
                                 "first_workgroup_us": [round(float(x), 1) for x in ai_dur],
                                 "what": "the first launches after the GPU idled through the CPU baseline: GPU time per step (events) and the "
                                         "shader clock inside each launch"}
            rec["timed_region"] = {"ms_steps_gpu": [round(x, 4) for x in ms_steps_gpu] if ms_steps_gpu else None}
            rec["config"]["clock_GHz_stamped_median"] = float(np.median(ai_ghz)) if len(ai_ghz) else None   # of the after-idle launches
        if not args.no_extras and world == 1:
            try:
                import bench_extras
                rec["extras"] = bench_extras.run(ctx, dev, cpu_baseline=not args.no_cpu_baseline)
            except ImportError:
                pass
    # north_star's config 5 (the whole per-pair pipeline, sharded over the ranks, records + match lists gathered) beside the headline at
    # EVERY N, so that a scaling run measures the RANSAC / cheirality half too -- not part of `value`, its own barrier-bracketed regions
    c5 = None
    if not args.no_extras:
        c5 = measure_c5(args, rank, local_rank, world, dev, ctx, max(args.c5_steps, 5), 2, cpu_baseline=not args.no_cpu_baseline)
    if rank == 0:
        # BASELINE's metric is "descriptor-pairs/s + RANSAC hyp/s; 1/2/4/8 GPU": the second half (C3) and the batch of config 5 are
        # TOP-LEVEL objects of the line, each with its own roofline and cpu_baseline, and their headline scalars are repeated in `config`
        # (a reader of the driver's `parsed` record, which keeps `config`, can check BASELINE.md section 4 from it alone)
        cfg = {}
        ransac = rec.get("extras", {}).pop("ransac_c3", None)
        if ransac is not None:
            rec["ransac"] = ransac
            cfg["ransac_c3_hyp_per_s"] = ransac["value"]
            cfg["ransac_c3_ms_per_call"] = ransac["ms_per_call"]
            cfg["ransac_c3_count_frac_fp32_peak"] = ransac["roofline"]["frac"]
            cfg["ransac_c3_solver_frac_fp64_peak"] = ransac["roofline"]["solver_frac_of_fp64_vector_peak"]
            if ransac.get("cpu_baseline"):
                cfg["ransac_c3_cpu_hyp_per_s_1_core"] = ransac["cpu_baseline"]["value"]
            if ransac.get("polish_changed_frac") is not None:
                cfg["ransac_c3_polish_changed_frac"] = ransac["polish_changed_frac"]
        if c5 is not None:
            rec["c5"] = c5
            cfg["c5_image_pairs_per_s"] = c5["value"]
            cfg["c5_ms_per_step"] = c5["ms_per_step"]
            cfg["c5_estimator"] = c5["config"]["estimator"]
            cfg["c5_mode"] = c5["mode"]
            other = [v for k, v in c5.items() if k.startswith("ms_per_step_")]
            cfg["c5_ms_per_step_other_mode"] = other[0] if other else None
            cfg["c5_dominant_kernel"] = c5["roofline"]["kernel"]
            cfg["c5_dominant_kernel_frac"] = c5["roofline"]["frac"]
            if c5.get("cpu_baseline"):
                cfg["c5_cpu_image_pairs_per_s_1_core"] = c5["cpu_baseline"]["value"]
        ex = rec.get("extras", {})
        for name, key in (("c5_usac_batch_uniform", "c5_usac_uniform_ms_per_512"), ("c5_usac_batch_prosac", "c5_usac_prosac_ms_per_512"),
                          ("c5_usac_batch_default_refinement_prosac", "c5_usac_default_refine_ms_per_512")):
            if isinstance(ex.get(name), dict) and "ms_per_step" in ex[name]:
                cfg[key] = ex[name]["ms_per_step"]
        for name, key in (("usac_uniform", "usac_call_ms"), ("arrsac_default_method", "arrsac_call_ms"), ("hamming_c2_single_pair", "hamming_single_pair_ms"),
                          ("hamming_c2_8pairs_mfma_kernel", "hamming_8_pairs_ms")):
            if isinstance(ex.get(name), dict) and "ms_per_call" in ex[name]:
                cfg[key] = ex[name]["ms_per_call"]
        import bench_record
        merged = dict(rec["config"])
        merged.update(cfg)
        merged.update({"matches_first_pair": counts[0], "hamming_kernel": kernel_name, "world_size": world, "backend": args.backend if dist_on(world) else None,
                       "parallelism": f"shard{world}", "records_gathered_every_steps": G,
                       "value_is": f"the BATCHED rate: {P} independent C2 image pairs per launch per GPU (a rank's share of BASELINE's 512-pair batch "
                                   "on 8 GPUs) under the given --steps / --warmup; steady_* = the same step in a long run; value_single_pair = "
                                   "ONE pair per launch (the literal config 2, latency shape)"})
        # the final line's keys first, in its order; everything else behind them (bench_detail only)
        rec["config"] = {k: merged[k] for k in bench_record.CONFIG_KEYS if k in merged}
        rec["config"].update({k: v for k, v in merged.items() if k not in rec["config"]})
        if dist_on(world):
            rec["rccl_ranks_seen"] = RCCL_RANKS_SEEN.get("n")
        bench_record.emit(rec, os.path.join(ROOT, "bench_detail.json"))
    if dist_on(world):
        dist.barrier()
        dist.destroy_process_group()
    ctx.close()


if __name__ == "__main__":
    main()
