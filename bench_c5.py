"""bench_c5.py -- BASELINE config 5 as a measurement (moved out of bench.py in round 6: that file had grown to 55 KB).

512 stereo pairs, each through the whole per-pair pipeline (8k ORB match -> gather -> robust estimator -> cheirality), dealt to the ranks by
batch.pair_shard; used by `bench.py --workload c5` (headline) and by the default line (its `c5` object, printed in the `bench_secondary` line).
"""
import ctypes as C
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
FP4_MFMA_PEAK_TFLOPS = 10000.0   # dense FP4 MFMA peak (MI355X_MICROARCH.md)
FLOP_PER_PAIR = 2 * 256          # one +-1 multiply-add per descriptor bit
FORCE_DIST = {"on": False}       # bench.py --force-dist: run every collective of the N > 1 path with ONE rank (rehearsal on a one-GPU box)


def dist_on(world):
    return world > 1 or FORCE_DIST["on"]


FP32_VALU_PEAK_TFLOPS = 157.3    # MI355X vector fp32 (MI355X_MICROARCH.md): the counting kernel decides in packed fp32
FP64_VALU_PEAK_TFLOPS = 78.6     # MI355X vector fp64
FLOP_PER_HYPOTHESIS = 15e3       # fp64 FLOP of one 5-point solve (null space, 10 x 20 elimination, degree-10 roots, <= 10 models)
FLOP_PER_SAMPSON = 39            # fp64 FLOP per (model, correspondence) evaluation


def measure_c5(args, rank, local_rank, world, dev, ctx, steps, warmup, cpu_baseline):
    """BASELINE config 5: `--c5-pairs` (512) stereo pairs x (8192-keypoint ORB match + ratio -> gather/ImgToCamCoordTrans -> RANSAC
    1000 it / 0.999 -> cheirality), pairs dealt to ranks in contiguous blocks; a rank's share goes through mlpl_pair_pose_batch_dev (the
    pair is a grid dimension of every launch, no host threads); per step one all_gather of the 184-byte records and the padded match
    lists to rank 0 by grouped send / recv.  Total work is fixed: strong scaling.  Returns the fields of the JSON line (rank 0: complete)."""
    import numpy as np
    import torch
    import torch.distributed as dist
    from matchinglib_poselib_amd import _lib, batch, synth

    total = args.c5_pairs
    begin, end = batch.pair_shard(total, rank, world)
    mine = end - begin
    cap = batch.shard_capacity(total, world)
    distinct = max(1, min(mine, args.c5_distinct))
    sps = [synth.stereo_pair(args.n, seed=20260200 + begin + i, unmatched_frac=0.30 + 0.02 * (i % 8)) for i in range(distinct)]
    K = sps[0]["K"]
    stacked = {k: torch.from_numpy(np.stack([sps[i % distinct][k] for i in range(mine)])).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")}
    seeds = [100 + begin + i for i in range(mine)]
    ids = list(range(begin, end))
    d_matches = torch.zeros((cap, args.n, 4), dtype=torch.int32, device=dev)
    state = {}
    use_rccl = args.backend == "nccl" or not dist_on(world)

    # The robust estimator of the pipeline: "ransac" = estimateEssentialMat(..., "RANSAC") at (1000, 0.999), mlpl_pair_pose_batch_dev;
    # "usac" / "usac_prosac" = the harness' cfgUSAC (POSE_STEWENIUS + REF_WEIGHTS + SPRT + LO; T/poselib-test/main.cpp:734, 1135-1162), uniform /
    # PROSAC sampling; "usac_default_refine" = ConfigUSAC's own default refinement REF_STEWENIUS_WEIGHTS (pose_estim.h:99-100), PROSAC;
    # "arrsac" = estimateEssentialMat's default method (pose_estim.h:207) + robustEssentialRefine.  The sequential estimators run as
    # fibers on a few host threads per rank behind one launch hub (csrc/batch_hub.h): the rank's host-thread budget is its share of the
    # CPUs this process may run on.
    est = getattr(args, "estimator", "ransac")
    cpus = len(os.sched_getaffinity(0)) if hasattr(os, "sched_getaffinity") else (os.cpu_count() or 1)
    local_world = int(os.environ.get("LOCAL_WORLD_SIZE", world))
    budget = batch.host_thread_budget(cpus, local_world, est)   # lanes x workers within the rank's share of the node's cores (batch.py)
    host_threads = {"cpus_visible": cpus, "local_world_size": local_world, "hub_workers_per_cohort": budget["hub_workers"],
                    "hub_lanes_option": budget["hub_lanes"],   # 0 = the estimator's own choice; the lanes a call used: hub_last_internal_call.lanes_used
                    "batch_lanes": budget["batch_lanes"], "threads_bound": budget["threads"],
                    "gc": "gc.freeze() + gc.disable() around the timed steps (a generation-2 collection stalls every thread for 50-90 ms)"}
    if est != "ransac":
        ctx.set_option("hub_workers", budget["hub_workers"])
        if budget["hub_lanes"]:
            ctx.set_option("hub_lanes", budget["hub_lanes"])
    usac_kw = {"usac": dict(prosac=False, refine=0), "usac_prosac": dict(prosac=True, refine=0), "usac_default_refine": dict(prosac=True, refine=5)}.get(est)

    lanes = batch.BatchLanes(local_rank, lanes=2, first_ctx=ctx) if est == "ransac" else None   # two batched calls in flight, half of the rank's share each

    def one_call():
        a = (stacked["desc1"], stacked["desc2"], stacked["kp1"], stacked["kp2"], K, K)
        if est == "ransac":
            return batch.process_pairs_batched(ctx, *a, seeds, pair_ids=ids, matches_out=d_matches[:mine])
        if est == "arrsac":
            r, raw = batch.process_pairs_batched_arrsac(ctx, *a, matches_out=d_matches[:mine])
            r["pair_id"] = ids
        else:
            r, raw = batch.process_pairs_batched_usac(ctx, *a, seeds, pair_ids=ids, matches_out=d_matches[:mine], **usac_kw)
        state["raw"] = raw
        return r

    def step(mode):
        if mode == "two_calls_in_flight":
            recs = lanes.process(stacked["desc1"], stacked["desc2"], stacked["kp1"], stacked["kp2"], K, K, seeds, pair_ids=ids,
                                 matches_out=d_matches[:mine])
        else:
            recs = one_call()
        state["rec"] = batch.gather_records(recs, total, rank, world, device=dev if use_rccl else None, force_collective=FORCE_DIST["on"])
        state["matches"] = batch.gather_match_lists(d_matches if use_rccl else d_matches.cpu(), total, rank, world, root=0)

    def barrier():
        if dist_on(world):
            dist.barrier()
        torch.cuda.synchronize()

    # Both ways of driving the rank's share are timed, each with its own warm-up and EXACTLY `steps` barrier-bracketed steps; the line's
    # value is the faster one (on one box two calls in flight were slower than one call at a time; the per-step wall times of
    # both stay in the line so that a slow first step or a slow lane is visible in the record itself).
    lib = ctx.lib
    timed = {}
    # A step is ~10 ms of wall time with Python threads in the loop: a generation-2 garbage collection of this process' ~10^6 objects
    # (torch, numpy) stalls every thread for 50-90 ms when it strikes (tools/lanes_tail_probe.py caught one: a 93 ms step among 9.4 ms ones).
    # As a long-running service would, freeze what exists and keep the collector out of the timed steps.
    import gc
    gc.collect()
    gc.freeze()
    gc.disable()
    for mode in (("two_calls_in_flight", "one_call_at_a_time") if est == "ransac" else ("one_call_at_a_time",)):
        for _ in range(max(warmup, 2)):
            step(mode)
        barrier()
        per_step = []
        t0 = time.perf_counter()
        for _ in range(steps):
            ts = time.perf_counter()
            step(mode)
            per_step.append((time.perf_counter() - ts) * 1e3)
            if mode == "two_calls_in_flight":   # when each lane's call started and ended inside the step (which lane stalled, if one did)
                state.setdefault("lane_spans", []).append([[round((a - ts) * 1e3, 2), round((b - ts) * 1e3, 2)] for a, b in lanes.last_lane_span])
        barrier()
        el = time.perf_counter() - t0
        if dist_on(world):
            tt = torch.tensor([el], dtype=torch.float64, device=dev if use_rccl else None)
            dist.all_reduce(tt, op=dist.ReduceOp.MAX)
            el = float(tt.item())
        timed[mode] = (el, per_step)
    gc.enable()
    best_mode = min(timed, key=lambda m: timed[m][0])   # the same on every rank: the elapsed times are the maxima over the ranks
    other_mode = ([m for m in timed if m != best_mode] or [None])[0]
    elapsed = timed[best_mode][0]
    # Kernel durations: with two calls in flight the HIP events around a kernel also bracket what runs beside it, so the per-kernel
    # figures come from `prof_steps` further steps through ONE call at a time (not part of `elapsed`), every launch bracketed.
    prof_steps = 0 if args.no_kernel_events else 2
    _lib.check(lib.mlpl_profile_reset(ctx.handle), "profile_reset")
    _lib.check(lib.mlpl_profile_enable(ctx.handle, 1 if prof_steps else 0), "profile_enable")
    for _ in range(prof_steps):
        one_call()
    torch.cuda.synchronize()
    _lib.check(lib.mlpl_profile_enable(ctx.handle, 0), "profile_enable")
    allrec = state["rec"]
    assert len(allrec) == total and (allrec["status"] == 0).all(), "a pair failed"
    if rank != 0:
        if lanes:
            lanes.close()
        return None
    # the gathered match lists are the lists the poses were computed from
    m = state["matches"]
    assert m.shape[0] == total
    first = m[0, : int(allrec["n_matches"][0])].cpu().numpy()
    assert (np.diff(first[:, 0]) > 0).all() and (first[:, 2] == -1).all(), "gathered match list of pair 0 is not a DMatch list"
    if getattr(args, "dump_records", None):   # tests: the gathered records and the valid rows of every gathered match list
        mh = m.cpu().numpy()
        np.savez(args.dump_records, records=allrec.view(np.uint8), mode=np.array([0]),
                 matches=np.concatenate([mh[i, : int(allrec["n_matches"][i])] for i in range(total)]))
    stats = np.zeros(8, np.int64)
    hub = None
    if est == "ransac":
        lib.mlpl_pair_batch_last_stats(ctx.handle, stats.ctypes.data)   # of the last single-call step: the rank's whole share
    elif est != "arrsac":
        hs = np.zeros(8, np.int64)
        lib.mlpl_usac_last_stats(ctx.handle, hs.ctypes.data)
        hub = {"rounds": int(hs[0]), "merged_launches": int(hs[1]), "hub_waiting_for_host_ms": hs[2] / 1e3, "device_ms": hs[3] / 1e3,
               "lanes_used": int(hs[6]), "cohorts": int(hs[7])}
        stats[6] = int(state["raw"]["iters"].sum())   # hypotheses of the rank's share
    prof = {}
    for name, kid in (("knn_hamming_mfma_lds_kernel<4, 0, 8>", 0), ("solve5pt3_kernel + roots_kernel_t<true>", 2),
                      ("count_models_f32_kernel<256, 512, 2, true>", 3), ("decompose / triangulate / select (batch)", 4)):
        ms, cnt = C.c_double(0), C.c_int(0)
        lib.mlpl_profile_read(ctx.handle, kid, C.byref(ms), C.byref(cnt))
        prof[name] = (ms.value, cnt.value)
    per_step = {k: v[0] / max(prof_steps, 1) for k, v in prof.items()}
    dom = max(per_step, key=per_step.get)
    evals = float(stats[5])  # Sampson evaluations of this rank's last step
    score_ms = per_step["count_models_f32_kernel<256, 512, 2, true>"]
    ham_ms = per_step["knn_hamming_mfma_lds_kernel<4, 0, 8>"]
    out = {
        "metric": "image-pairs/s (C5: stereo pairs x (8k ORB BF-Hamming match + 5-pt RANSAC + cheirality))",
        "value": total * steps / elapsed, "unit": "image-pairs/s", "n_gpus": world, "steps": steps, "warmup": warmup,
        "ms_per_step": elapsed / steps * 1e3, "mode": best_mode,
        **({"ms_per_step_" + other_mode: timed[other_mode][0] / steps * 1e3} if other_mode else {}),
        "ms_steps_rank0": {m: [round(x, 3) for x in timed[m][1]] for m in timed},
        "lane_spans_ms_rank0": state.get("lane_spans"),
        "higher_is_better": True, "scaling": "strong",
        "vs_baseline": None,
        "dtype": "fp4 MFMA (Hamming) + f64 / banded packed f32", "data": "synthetic",
        "config": {"workload": f"C5: {total} stereo pairs x ({args.n} ORB-256 keypoints per image: Hamming 2-NN + 0.75 ratio -> "
                               "ImgToCamCoordTrans gather -> " + {"ransac": "RANSAC 1000 it / 0.999", "usac": "USAC (cfgUSAC, uniform sampling)",
                                                                  "usac_prosac": "USAC (cfgUSAC, PROSAC by matching cost)",
                                                                  "usac_default_refine": "USAC (REF_STEWENIUS_WEIGHTS, PROSAC)",
                                                                  "arrsac": "ARRSAC + robustEssentialRefine"}[est] + " -> getPoseTriangPts), one step = the whole batch",
                   "estimator": est, "host_threads": host_threads, "hub_last_internal_call": hub,
                   "pairs_total": total, "pairs_this_rank": mine, "distinct_inputs_per_rank": distinct,
                   "entry": ("mlpl_pair_pose_batch_dev (pair = grid dimension; internal batches of 256 pairs); timed both with two calls in flight "
                             "per rank (batch.BatchLanes: two library contexts, half of the rank's share each) and one call at a time -- "
                             "`mode` names the faster one, which `value` is") if est == "ransac" else
                            ("mlpl_pair_pose_batch_arrsac_dev" if est == "arrsac" else "mlpl_pair_pose_batch_usac_dev") +
                            " (every pair's sequential estimator = a fiber behind the launch hub, launches merged over the pairs; one call per step)",
                   "parallelism": f"shard{world}",
                   "world_size": world, "backend": args.backend if dist_on(world) else None,
                   "gathered_per_step": "184-byte records by all_gather + the padded match lists to rank 0 by grouped send / recv",
                   "mean_matches": float(allrec["n_matches"].mean()), "mean_inliers": float(allrec["n_inliers"].mean()),
                   "ransac_passes_rank0": int(stats[0]), "pair_slots_rank0": int(stats[1]), "iterations_rank0": int(stats[6]),
                   "host_rand_stream_ms_per_step_rank0": float(stats[3]) / 1e3},
        "kernel_ms_per_step_rank0": per_step,
        "kernel_ms_measured": "HIP events around every launch in steps that run one call at a time, after the timed region",
        "roofline": {
            "kernel": dom, "bound": "valu-fp32" if dom.startswith("count") else ("mfma" if dom.startswith("knn") else "valu-fp64 (issue / latency bound)"),
            "kernel_ms_per_step": per_step[dom], "launches_timed": prof[dom][1],
            "achieved": (evals * FLOP_PER_SAMPSON / (score_ms * 1e-3) / 1e12) if dom.startswith("count") else
                        ((mine * args.n * args.n * FLOP_PER_PAIR / (ham_ms * 1e-3) / 1e12) if dom.startswith("knn") else
                         (float(stats[6]) * FLOP_PER_HYPOTHESIS / (per_step[dom] * 1e-3) / 1e12)),
            "peak": FP32_VALU_PEAK_TFLOPS if dom.startswith("count") else (FP4_MFMA_PEAK_TFLOPS if dom.startswith("knn") else FP64_VALU_PEAK_TFLOPS),
            "unit": "TFLOP/s",
            "traffic": None,
            "note": "dominant kernel of the pipeline by HIP events inside the library (every launch bracketed).  count_models: 39 FLOP per "
                    "(model, correspondence) evaluation x the evaluations of the step, decided in packed fp32 (two per instruction) inside "
                    "a rigorous error band, priced against the fp32 vector peak; Hamming: 2 x 256 FLOP per descriptor pair against the "
                    "dense FP4 peak; solver kernels: ~15 kFLOP (fp64) per 5-point hypothesis x the iterations of the step against the fp64 vector peak -- "
                    "three hypotheses per wavefront (elimination) and six (roots), issue / latency bound, far from that roofline by construction (SURVEY 8(d))",
            "sampson_evaluations_per_step": evals, "score_kernel_TFLOPs_equiv": evals * FLOP_PER_SAMPSON / (score_ms * 1e-3) / 1e12 if score_ms > 0 else None,
            "hamming_kernel_frac_of_fp4_peak": mine * args.n * args.n * FLOP_PER_PAIR / (ham_ms * 1e-3) / 1e12 / FP4_MFMA_PEAK_TFLOPS if ham_ms > 0 else None,
        },
        "cpu_baseline": None,
    }
    r = out["roofline"]
    r["frac"] = (r["achieved"] / r["peak"]) if r["achieved"] and r["peak"] else None
    if cpu_baseline:
        sys.path.insert(0, os.path.join(ROOT, "tests"))
        import oracle_lib
        ora = oracle_lib.load()
        ncpu = max(1, min(args.c5_cpu_pairs, mine))   # ~0.75 s per pair on one core
        tc = time.perf_counter()
        for i in range(ncpu):
            sp = sps[i % distinct]
            n = len(sp["desc1"])
            rc, mm = ora.get_matches_linear(n, n, sp["desc1"], sp["desc2"])
            a, b = sp["kp1"][mm["queryIdx"]], sp["kp2"][mm["trainIdx"]]
            cam = lambda p: np.stack([((p[:, 0].astype(np.float64) - K[2]) / K[0]).astype(np.float32),  # noqa: E731
                                      ((p[:, 1].astype(np.float64) - K[3]) / K[1]).astype(np.float32)], axis=1).astype(np.float64)
            p1, p2 = cam(a), cam(b)
            th = 0.8 * 4.0 / (np.sqrt(2.0) * (2 * K[0] + 2 * K[1]))
            if est == "ransac":
                o = ora.ransac_essential(p1, p2, th, confidence=0.999, max_iters=1000, lesqu=False, seed=seeds[i])
                good, R, t, Q, mk = ora.recover_pose(o["E"], p1, p2, 50.0, o["mask"])
                ninl = o["n_inliers"]
            elif est == "arrsac":
                o = ora.arrsac_essential(p1, p2, th, refine=True)
                good, R, t, Q, mk = ora.recover_pose(o["E"], p1, p2, 50.0, o["mask"])
                ninl = o["n_inliers"]
            else:
                order = None
                if usac_kw["prosac"]:   # poselib::getSortedMatchIdx: std::sort by the matching cost (the library's own host helper: a sort, no GPU)
                    order = np.zeros(len(mm), np.uint32)
                    mmc = np.ascontiguousarray(mm)
                    assert lib.mlpl_sorted_match_idx(mmc.ctypes.data, len(mm), order.ctypes.data) == 0
                o = ora.usac_essential(p1, p2, th, seeds[i], refine=usac_kw["refine"], sorted_idx=order, prosac_beta=0.05, sprt_ms=6.0, sprt_tm=2736.0)
                good, R, t, Q, mk = ora.recover_pose(o["E"], p1, p2, 50.0, o["flags"])
                ninl = int(o["final"][5])
            if i < mine:   # the timed records are the CPU path's records
                assert len(mm) == allrec["n_matches"][i] and ninl == allrec["n_inliers"][i], "pair record differs from the CPU path"
                assert np.abs(allrec["R"][i].reshape(3, 3) - np.asarray(R).reshape(3, 3)).max() < 1e-6 and np.abs(allrec["t"][i] - np.asarray(t).ravel()).max() < 1e-6
        tc = time.perf_counter() - tc
        out["cpu_baseline"] = {"value": ncpu / tc, "unit": "image-pairs/s", "cores": 1, "kind": "port",
                               "sample": f"the first {ncpu} pairs of the batch through the oracle pipeline (LINEAR matching, {est} oracle, "
                                         f"recoverPose), {tc:.1f} s; their records equal the timed step's (counts exact, R, t to 1e-6)",
                               "host_cores_available": os.cpu_count()}
    if lanes:
        lanes.close()
    return out
