"""Batches of independent image pairs sharded over the GPUs of one node (BASELINE config C5).

Every image pair is an independent unit (reference harness loop, tests/poselib-test/main.cpp:1440-2072), so pairs are
dealt to ranks in contiguous blocks, each rank runs match -> gather -> RANSAC -> pose on its own GPU with no data-path
collective, and the fixed-size per-pair result records are gathered once (torch.distributed all_gather: RCCL over xGMI
on GPUs, gloo in the CPU tests).
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from ._lib import Context, check

# {pair_id, n_matches, n_inliers, status, E[9], R[9], t[3]} = 184 bytes
RECORD_DTYPE = np.dtype([("pair_id", np.int32), ("n_matches", np.int32), ("n_inliers", np.int32), ("status", np.int32),
                         ("E", np.float64, (9,)), ("R", np.float64, (9,)), ("t", np.float64, (3,))])
assert RECORD_DTYPE.itemsize == 184


def pair_shard(num_pairs: int, rank: int, world: int) -> Tuple[int, int]:
    """Contiguous block [begin, end) of pair ids owned by `rank`; sizes differ by at most one."""
    base, rem = divmod(num_pairs, world)
    begin = rank * base + min(rank, rem)
    return begin, begin + base + (1 if rank < rem else 0)


def shard_capacity(num_pairs: int, world: int) -> int:
    return (num_pairs + world - 1) // world


# Host-thread budget of one rank (VERDICT r5 #7a).  A rank that runs a batch of SEQUENTIAL estimators (USAC, ARRSAC) serves up to `lanes`
# cohorts side by side, each with `hub_workers` worker threads its runs are fibers on (csrc/batch_hub.h); with RANSAC it drives two batched
# calls from two Python threads (BatchLanes).  Most of those threads sleep on the device most of the time, so the rank may oversubscribe its
# share of the node's cores -- by at most kThreadOversubscription -- but 8 ranks x (6 lanes x 16 workers) = 768 threads on a 64-core host
# is not a plan.  Measured (gpurun_out/r5/c5_workers_ab.log): 8 workers per cohort instead of 16 cost 4-7 %, 32 or 64 gain nothing.
kThreadOversubscription = 2
kHubLanesDefault = {"usac": 6, "usac_prosac": 6, "usac_default_refine": 4, "arrsac": 4}   # csrc/batch_hub.h: hub_usac_lanes_default / kHubLanesDefault


def host_thread_budget(cpus: int, local_world: int, estimator: str) -> dict:
    """How many host threads a rank of `local_world` ranks on a node with `cpus` usable cores gives the estimator of its C5 share:
    {"cpus_per_rank", "hub_lanes" (0 = leave the library's default), "hub_workers" (per cohort), "batch_lanes", "threads"} with
    threads = everything the rank keeps runnable at once (lanes x workers + the lanes' own threads + the Python threads)."""
    per_rank = max(1, cpus // max(1, local_world))
    if estimator == "ransac":
        return {"cpus_per_rank": per_rank, "hub_lanes": 0, "hub_workers": 0, "batch_lanes": 2, "threads": 3}
    lanes = kHubLanesDefault[estimator]
    bound = max(4, kThreadOversubscription * per_rank)           # runnable threads this rank may own
    capped = lanes
    while capped > 1 and capped * 2 + capped + 1 > bound:        # every lane needs its own thread and at least two workers
        capped -= 1
    workers = max(2, min(16, (bound - capped - 1) // capped))
    return {"cpus_per_rank": per_rank, "hub_lanes": 0 if capped == lanes else capped, "hub_workers": workers, "batch_lanes": 0,
            "threads": capped * workers + capped + 1}


_gather_stage = {}   # (cap, world, device) -> (pinned host block, device block, device gather buffer, pinned host gather buffer)


def gather_records(local: np.ndarray, num_pairs: int, rank: int, world: int, device=None, group=None, force_collective: bool = False) -> Optional[np.ndarray]:
    """All-gathers the ranks' record blocks (padded to equal size) and returns the num_pairs records in pair order on
    every rank.  `device` = torch device the collective runs on (cuda for RCCL, cpu / None for gloo).  On a GPU the blocks travel through
    PERSISTENT pinned and device staging buffers (one asynchronous H2D, the all_gather on the device, one D2H, one synchronisation per
    call; no allocation after the first call -- the records themselves are final only on the host, where the sequential replay of the
    estimator ends: include/mlpl_c.h, mlpl_pair_pose_batch_dev)."""
    import torch
    import torch.distributed as dist

    cap = shard_capacity(num_pairs, world)
    isz = RECORD_DTYPE.itemsize
    if world == 1 and not force_collective:   # (force_collective: rehearsal of the collective path with one rank)
        buf = np.zeros(cap, RECORD_DTYPE)
        buf["pair_id"] = -1
        buf[: len(local)] = local
        allr = buf.view(np.uint8).reshape(cap, isz)   # nothing to exchange
    elif device is not None and torch.device(device).type == "cuda":
        key = (cap, world, str(device))
        st = _gather_stage.get(key)
        if st is None:
            st = (torch.empty((cap, isz), dtype=torch.uint8).pin_memory(), torch.empty((cap, isz), dtype=torch.uint8, device=device),
                  torch.empty((world * cap, isz), dtype=torch.uint8, device=device), torch.empty((world * cap, isz), dtype=torch.uint8).pin_memory())
            _gather_stage[key] = st
        h_in, d_in, d_out, h_out = st
        blk = h_in.numpy().view(RECORD_DTYPE).reshape(-1)
        blk[:] = np.zeros((), RECORD_DTYPE)
        blk["pair_id"] = -1
        blk[: len(local)] = local
        d_in.copy_(h_in, non_blocking=True)
        dist.all_gather_into_tensor(d_out, d_in, group=group)
        h_out.copy_(d_out, non_blocking=True)
        torch.cuda.current_stream(device).synchronize()
        allr = h_out.numpy()
    else:
        buf = np.zeros(cap, RECORD_DTYPE)
        buf["pair_id"] = -1
        buf[: len(local)] = local
        t = torch.from_numpy(buf.view(np.uint8).reshape(cap, isz).copy())
        out = torch.empty((world * cap, isz), dtype=torch.uint8)
        dist.all_gather_into_tensor(out, t, group=group)
        allr = out.numpy()
    rec = np.ascontiguousarray(allr).view(RECORD_DTYPE).reshape(-1).copy()
    rec = rec[rec["pair_id"] >= 0]
    order = np.argsort(rec["pair_id"], kind="stable")
    rec = rec[order]
    assert len(rec) == num_pairs and np.array_equal(rec["pair_id"], np.arange(num_pairs)), "lost or duplicated pairs"
    return rec


def gather_match_lists(local, num_pairs: int, rank: int, world: int, root: int = 0, group=None):
    """The padded match lists of every rank's pairs on `root` (RCCL has no gather -- grouped send / recv to the root; over
    xGMI every peer -> root transfer rides its own link).  local: torch tensor [shard, nq, 4] int32 (cv::DMatch rows; the number of valid
    rows per pair travels with the records).  Returns [num_pairs, nq, 4] in pair order on the root, None elsewhere."""
    import torch
    import torch.distributed as dist

    if world == 1:
        return local[:num_pairs]
    nq = local.shape[1]
    sizes = [pair_shard(num_pairs, r, world) for r in range(world)]
    if rank == root:
        out = torch.empty((num_pairs, nq, 4), dtype=local.dtype, device=local.device)
        b, e = sizes[root]
        out[b:e] = local[: e - b]
        ops = [dist.P2POp(dist.irecv, out[sizes[r][0]:sizes[r][1]], r, group=group) for r in range(world) if r != root and sizes[r][1] > sizes[r][0]]
    else:
        out = None
        b, e = sizes[rank]
        ops = [dist.P2POp(dist.isend, local[: e - b].contiguous(), root, group=group)] if e > b else []
    if ops:
        for w in dist.batch_isend_irecv(ops):
            w.wait()
    return out


class _PairResult(C.Structure):
    _fields_ = [("n_matches", C.c_int32), ("n_inliers", C.c_int32), ("n_good", C.c_int32), ("status", C.c_int32), ("iters", C.c_int32),
                ("pad", C.c_int32), ("E", C.c_double * 9), ("R", C.c_double * 9), ("t", C.c_double * 3)]


_PAIR_RESULT_DTYPE = np.dtype([("n_matches", np.int32), ("n_inliers", np.int32), ("n_good", np.int32), ("status", np.int32), ("iters", np.int32),
                               ("pad", np.int32), ("E", np.float64, (9,)), ("R", np.float64, (9,)), ("t", np.float64, (3,))])
assert _PAIR_RESULT_DTYPE.itemsize == C.sizeof(_PairResult)


def process_pair_on_device(ctx: Context, d_q, d_t, d_kp1, d_kp2, K0, K1, th_pix: float = 0.8, max_iters: int = 1000,
                           confidence: float = 0.999, refit: bool = False, seed: int = 0, dist: float = 50.0, pair_id: int = 0,
                           scratch: Optional[dict] = None) -> np.ndarray:
    """One image pair, device-resident inputs (torch CUDA tensors): Hamming 2-NN + ratio -> gather matched keypoints
    (ImgToCamCoordTrans) -> RANSAC essential matrix -> cheirality, in one library call (mlpl_pair_pose_dev: two host hops,
    nothing else leaves the device).  Returns one RECORD_DTYPE record.  `scratch` is accepted for compatibility and unused."""
    import torch

    assert d_q.is_cuda and d_q.dtype == torch.uint8 and d_q.is_contiguous() and d_t.is_contiguous()
    assert d_kp1.dtype == torch.float32 and d_kp2.dtype == torch.float32 and d_kp1.is_contiguous() and d_kp2.is_contiguous()
    rec = np.zeros(1, RECORD_DTYPE)
    rec["pair_id"] = pair_id
    k0 = (C.c_double * 4)(*K0)
    k1 = (C.c_double * 4)(*K1)
    th = th_pix * 4.0 / (np.sqrt(2.0) * (K0[0] + K0[1] + K1[0] + K1[1]))  # stereo_pose_refinement.h:280-286
    res = _PairResult()
    st = torch.cuda.current_stream(d_q.device).cuda_stream
    check(ctx.lib.mlpl_pair_pose_dev(ctx.handle, d_q.data_ptr(), d_q.shape[0], d_t.data_ptr(), d_t.shape[0], d_q.shape[1],
                                     d_kp1.data_ptr(), d_kp2.data_ptr(), k0, k1, float(th), int(max_iters), float(confidence),
                                     1 if refit else 0, int(seed) & 0xFFFFFFFF, float(dist), C.addressof(res), st), "mlpl_pair_pose_dev")
    rec["n_matches"] = res.n_matches
    rec["status"] = res.status
    if res.status == 0:
        rec["n_inliers"] = res.n_inliers
        rec["E"] = np.frombuffer(res.E, np.float64)
        rec["R"] = np.frombuffer(res.R, np.float64)
        rec["t"] = np.frombuffer(res.t, np.float64)
    return rec


def process_pairs_batched(ctx: Context, d_q, d_t, d_kp1, d_kp2, K0, K1, seeds, th_pix: float = 0.8, max_iters: int = 1000,
                          confidence: float = 0.999, refit: bool = False, dist: float = 50.0, pair_ids=None, matches_out=None) -> np.ndarray:
    """A batch of image pairs in ONE library call (mlpl_pair_pose_batch_dev): device tensors d_q [B, nq, nbytes] uint8, d_t [B, nt, nbytes],
    d_kp1 [B, nq, 2] float32, d_kp2 [B, nt, 2]; seeds: B RANSAC seeds.  The pair is a grid dimension of every launch -- no host threads, a
    handful of host hops per 256 pairs -- and every record equals process_pair_on_device's for that pair.  matches_out: optional int32
    CUDA tensor [B, nq, 4] that receives the match lists (cv::DMatch rows, n_matches valid per pair).  Returns B RECORD_DTYPE records."""
    import torch

    B = d_q.shape[0]
    assert d_q.is_cuda and d_q.dtype == torch.uint8 and d_q.dim() == 3 and d_t.dim() == 3 and d_t.shape[0] == B
    assert d_kp1.dtype == torch.float32 and d_kp2.dtype == torch.float32 and d_kp1.shape == (B, d_q.shape[1], 2) and d_kp2.shape == (B, d_t.shape[1], 2)
    assert d_q.is_contiguous() and d_t.is_contiguous() and d_kp1.is_contiguous() and d_kp2.is_contiguous()
    k0 = (C.c_double * 4)(*K0)
    k1 = (C.c_double * 4)(*K1)
    th = th_pix * 4.0 / (np.sqrt(2.0) * (K0[0] + K0[1] + K1[0] + K1[1]))  # stereo_pose_refinement.h:280-286
    sd = np.ascontiguousarray(np.asarray(seeds, np.int64) & 0xFFFFFFFF, np.uint32)
    assert len(sd) == B
    res = (_PairResult * B)()
    st = torch.cuda.current_stream(d_q.device).cuda_stream
    if matches_out is not None:
        assert matches_out.is_cuda and matches_out.dtype == torch.int32 and matches_out.shape == (B, d_q.shape[1], 4) and matches_out.is_contiguous()
    check(ctx.lib.mlpl_pair_pose_batch_dev(ctx.handle, B, d_q.data_ptr(), d_q.shape[1], d_t.data_ptr(), d_t.shape[1], d_q.shape[2], d_kp1.data_ptr(),
                                           d_kp2.data_ptr(), k0, k1, float(th), int(max_iters), float(confidence), 1 if refit else 0, sd.ctypes.data,
                                           float(dist), C.addressof(res), matches_out.data_ptr() if matches_out is not None else None, st),
          "mlpl_pair_pose_batch_dev")
    raw = np.frombuffer(res, _PAIR_RESULT_DTYPE, count=B)   # the library's result block, field by field (no per-pair Python)
    rec = np.zeros(B, RECORD_DTYPE)
    rec["pair_id"] = np.arange(B) if pair_ids is None else np.asarray(pair_ids)
    rec["n_matches"], rec["status"] = raw["n_matches"], raw["status"]
    ok = raw["status"] == 0
    for f in ("n_inliers", "E", "R", "t"):
        rec[f][ok] = raw[f][ok]
    return rec


def process_pairs_batched_usac(ctx: Context, d_q, d_t, d_kp1, d_kp2, K0, K1, seeds, th_pix: float = 0.8, prosac: bool = False, estimator: int = 2,
                               refine: int = 0, check_degeneracy: int = 0, sprt_delta: float = 0.05, sprt_epsilon: float = 0.15, sprt_ms: float = 6.0,
                               sprt_tm: float = 2736.0, max_hyp: int = 50000, dist: float = 50.0, pair_ids=None, matches_out=None) -> Tuple[np.ndarray, np.ndarray]:
    """process_pairs_batched with USAC (the reference harness' default RobMethod; defaults = its cfgUSAC: POSE_STEWENIUS + REF_WEIGHTS) as the
    robust estimator: mlpl_pair_pose_batch_usac_dev.  prosac: PROSAC sampling in the order of the matching costs.
    Returns (records [B] RECORD_DTYPE, the library's raw result block [B]: n_matches, n_inliers, n_good, status, iters, E, R, t)."""
    import torch
    from .pose import UsacParams

    B = d_q.shape[0]
    assert d_q.is_cuda and d_q.dtype == torch.uint8 and d_q.dim() == 3 and d_t.dim() == 3 and d_t.shape[0] == B
    assert d_kp1.dtype == torch.float32 and d_kp2.dtype == torch.float32 and d_kp1.shape == (B, d_q.shape[1], 2) and d_kp2.shape == (B, d_t.shape[1], 2)
    assert d_q.is_contiguous() and d_t.is_contiguous() and d_kp1.is_contiguous() and d_kp2.is_contiguous()
    k0 = (C.c_double * 4)(*K0)
    k1 = (C.c_double * 4)(*K1)
    th = th_pix * 4.0 / (np.sqrt(2.0) * (K0[0] + K0[1] + K1[0] + K1[1]))
    P = UsacParams()
    ctx.lib.mlpl_usac_default_params(C.addressof(P), float(th))
    P.max_hyp, P.estimator, P.refine, P.check_degeneracy = int(max_hyp), int(estimator), int(refine), int(check_degeneracy)
    P.sprt_delta, P.sprt_epsilon, P.sprt_mS, P.sprt_tM, P.prosac_beta = float(sprt_delta), float(sprt_epsilon), float(sprt_ms), float(sprt_tm), float(sprt_delta)
    P.th_pixels, P.focal_length = float(th_pix), float((K0[0] + K0[1] + K1[0] + K1[1]) / 4.0)
    sd = np.ascontiguousarray(np.asarray(seeds, np.int64) & 0xFFFFFFFF, np.uint32)
    assert len(sd) == B
    res = (_PairResult * B)()
    st = torch.cuda.current_stream(d_q.device).cuda_stream
    if matches_out is not None:
        assert matches_out.is_cuda and matches_out.dtype == torch.int32 and matches_out.shape == (B, d_q.shape[1], 4) and matches_out.is_contiguous()
    check(ctx.lib.mlpl_pair_pose_batch_usac_dev(ctx.handle, B, d_q.data_ptr(), d_q.shape[1], d_t.data_ptr(), d_t.shape[1], d_q.shape[2], d_kp1.data_ptr(),
                                                d_kp2.data_ptr(), k0, k1, C.addressof(P), 1 if prosac else 0, sd.ctypes.data, float(dist), C.addressof(res),
                                                matches_out.data_ptr() if matches_out is not None else None, st), "mlpl_pair_pose_batch_usac_dev")
    raw = np.frombuffer(res, _PAIR_RESULT_DTYPE, count=B).copy()
    rec = np.zeros(B, RECORD_DTYPE)
    rec["pair_id"] = np.arange(B) if pair_ids is None else np.asarray(pair_ids)
    rec["n_matches"], rec["status"] = raw["n_matches"], raw["status"]
    ok = raw["status"] == 0
    for f in ("n_inliers", "E", "R", "t"):
        rec[f][ok] = raw[f][ok]
    return rec, raw


def process_pairs_batched_arrsac(ctx: Context, d_q, d_t, d_kp1, d_kp2, K0, K1, th_pix: float = 0.8, refine: bool = True, rng_states=None, dist: float = 50.0,
                                 matches_out=None):
    """process_pairs_batched with ARRSAC (estimateEssentialMat's default method) as the robust estimator: mlpl_pair_pose_batch_arrsac_dev.
    rng_states: uint64 [B, 2], advanced in place (default: fresh cv::RNG streams for every pair).  Returns (records, raw result block)."""
    import torch
    from .pose import ARRSAC_RNG_FRESH

    B = d_q.shape[0]
    assert d_q.is_cuda and d_q.dtype == torch.uint8 and d_q.dim() == 3 and d_t.dim() == 3 and d_t.shape[0] == B
    assert d_q.is_contiguous() and d_t.is_contiguous() and d_kp1.is_contiguous() and d_kp2.is_contiguous()
    k0 = (C.c_double * 4)(*K0)
    k1 = (C.c_double * 4)(*K1)
    th = th_pix * 4.0 / (np.sqrt(2.0) * (K0[0] + K0[1] + K1[0] + K1[1]))
    st = np.tile(np.array(ARRSAC_RNG_FRESH, np.uint64), (B, 1)) if rng_states is None else rng_states
    assert st.dtype == np.uint64 and st.shape == (B, 2) and st.flags.c_contiguous
    res = (_PairResult * B)()
    stream = torch.cuda.current_stream(d_q.device).cuda_stream
    check(ctx.lib.mlpl_pair_pose_batch_arrsac_dev(ctx.handle, B, d_q.data_ptr(), d_q.shape[1], d_t.data_ptr(), d_t.shape[1], d_q.shape[2], d_kp1.data_ptr(),
                                                  d_kp2.data_ptr(), k0, k1, float(th), 1 if refine else 0, st.ctypes.data, float(dist), C.addressof(res),
                                                  matches_out.data_ptr() if matches_out is not None else None, stream), "mlpl_pair_pose_batch_arrsac_dev")
    raw = np.frombuffer(res, _PAIR_RESULT_DTYPE, count=B).copy()
    rec = np.zeros(B, RECORD_DTYPE)
    rec["pair_id"] = np.arange(B)
    rec["n_matches"], rec["status"] = raw["n_matches"], raw["status"]
    ok = raw["status"] == 0
    for f in ("n_inliers", "E", "R", "t"):
        rec[f][ok] = raw[f][ok]
    return rec, raw


def ransac_pose_batched(ctx: Context, d_p1, d_p2, counts, seeds, thresh: float, max_iters: int = 1000, confidence: float = 0.999,
                        recover_pose: bool = True, dist: float = 50.0, masks_out=None) -> list:
    """A batch of correspondence sets in ONE library call (mlpl_ransac_essential_batch_dev): d_p1, d_p2 float64 CUDA tensors [B, stride, 2]
    (camera coordinates), counts[b] <= stride valid rows per problem, seeds[b] its RANSAC seed; masks_out: optional uint8 CUDA tensor
    [B, stride].  Returns one dict per problem (status, n, n_inliers, n_good, iters, E, R, t): what ransac_essential_device(refit=False)
    followed by getPoseTriangPts_device returns for it."""
    import torch

    B, stride = d_p1.shape[0], d_p1.shape[1]
    assert d_p1.is_cuda and d_p1.dtype == torch.float64 and d_p1.shape == d_p2.shape == (B, stride, 2) and d_p1.is_contiguous() and d_p2.is_contiguous()
    cn = np.ascontiguousarray(counts, np.int32)
    sd = np.ascontiguousarray(np.asarray(seeds, np.int64) & 0xFFFFFFFF, np.uint32)
    assert len(cn) == B and len(sd) == B
    if masks_out is not None:
        assert masks_out.is_cuda and masks_out.dtype == torch.uint8 and masks_out.shape == (B, stride) and masks_out.is_contiguous()
    res = (_PairResult * B)()
    st = torch.cuda.current_stream(d_p1.device).cuda_stream
    check(ctx.lib.mlpl_ransac_essential_batch_dev(ctx.handle, B, d_p1.data_ptr(), d_p2.data_ptr(), stride, cn.ctypes.data, float(thresh), int(max_iters),
                                                  float(confidence), sd.ctypes.data, 1 if recover_pose else 0, float(dist), C.addressof(res),
                                                  masks_out.data_ptr() if masks_out is not None else None, st), "mlpl_ransac_essential_batch_dev")
    return [dict(status=r.status, n=r.n_matches, n_inliers=r.n_inliers, n_good=r.n_good, iters=r.iters, E=np.frombuffer(r.E, np.float64).reshape(3, 3).copy(),
                 R=np.frombuffer(r.R, np.float64).reshape(3, 3).copy(), t=np.frombuffer(r.t, np.float64).copy()) for r in res]


class BatchLanes:
    """`lanes` (2) independent (library context, torch stream, host thread) triples on one GPU, each running mlpl_pair_pose_batch_dev on its
    share of a batch.  A batched call has ~5 host hops per 256 pairs (match counts, one per RANSAC pass, the pose) during which its stream
    is idle, and its solver kernels are latency-bound: a second call in flight fills both (512 pairs: 11.1 -> 9.8 ms; four lanes: 10.4).
    Records do not depend on the lane count: every pair's result is a function of its inputs and its seed."""

    def __init__(self, device_index: int = 0, lanes: int = 2, first_ctx: Optional[Context] = None):
        import torch
        from concurrent.futures import ThreadPoolExecutor

        self.device = torch.device("cuda", device_index)
        self.ctxs = ([first_ctx] if first_ctx is not None else []) + [Context(device_index) for _ in range(lanes - (1 if first_ctx is not None else 0))]
        self.owned = self.ctxs[1:] if first_ctx is not None else list(self.ctxs)
        self.streams = [torch.cuda.Stream(self.device) for _ in range(lanes)]
        self.pool = ThreadPoolExecutor(max_workers=lanes)
        self.lanes = lanes
        self.last_lane_span = [(0.0, 0.0)] * lanes

    def _run(self, w, b, e, d_q, d_t, d_kp1, d_kp2, K0, K1, seeds, pair_ids, matches_out, kw):
        import torch

        import time

        t0 = time.perf_counter()
        torch.cuda.set_device(self.device)
        with torch.cuda.stream(self.streams[w]):
            r = process_pairs_batched(self.ctxs[w], d_q[b:e], d_t[b:e], d_kp1[b:e], d_kp2[b:e], K0, K1, seeds[b:e],
                                      pair_ids=None if pair_ids is None else pair_ids[b:e],
                                      matches_out=None if matches_out is None else matches_out[b:e], **kw)
            self.streams[w].synchronize()
        self.last_lane_span[w] = (t0, time.perf_counter())   # when this lane's call started and ended (diagnostics: which lane stalled)
        return r

    def process(self, d_q, d_t, d_kp1, d_kp2, K0, K1, seeds, pair_ids=None, matches_out=None, native: bool = True, th_pix: float = 0.8, max_iters: int = 1000,
                confidence: float = 0.999, dist: float = 50.0, **kw) -> np.ndarray:
        """Same arguments and result as process_pairs_batched (without the context).  native (default): ONE library call,
        mlpl_pair_pose_batch_lanes_dev, whose lanes are threads inside the library -- no Python thread hand-off and no interpreter lock
        between the lanes' calls; native = False: the lanes are Python threads (a ThreadPoolExecutor), each making its own call."""
        import torch

        B = d_q.shape[0]
        torch.cuda.current_stream(self.device).synchronize()  # inputs were produced on the caller's stream
        if native and not kw:
            assert d_q.is_cuda and d_q.dtype == torch.uint8 and d_q.dim() == 3 and d_t.dim() == 3 and d_t.shape[0] == B
            assert d_kp1.dtype == torch.float32 and d_kp2.dtype == torch.float32 and d_kp1.shape == (B, d_q.shape[1], 2) and d_kp2.shape == (B, d_t.shape[1], 2)
            assert d_q.is_contiguous() and d_t.is_contiguous() and d_kp1.is_contiguous() and d_kp2.is_contiguous()
            if matches_out is not None:
                assert matches_out.is_cuda and matches_out.dtype == torch.int32 and matches_out.shape == (B, d_q.shape[1], 4) and matches_out.is_contiguous()
            L = self.lanes
            ctxs = (C.c_void_p * L)(*[c.handle for c in self.ctxs])
            strs = (C.c_void_p * L)(*[s.cuda_stream for s in self.streams])
            k0, k1 = (C.c_double * 4)(*K0), (C.c_double * 4)(*K1)
            th = th_pix * 4.0 / (np.sqrt(2.0) * (K0[0] + K0[1] + K1[0] + K1[1]))
            sd = np.ascontiguousarray(np.asarray(seeds, np.int64) & 0xFFFFFFFF, np.uint32)
            assert len(sd) == B
            res = (_PairResult * B)()
            spans = (C.c_double * (2 * L))()
            check(self.ctxs[0].lib.mlpl_pair_pose_batch_lanes_dev(ctxs, strs, L, B, d_q.data_ptr(), d_q.shape[1], d_t.data_ptr(), d_t.shape[1], d_q.shape[2],
                                                                  d_kp1.data_ptr(), d_kp2.data_ptr(), k0, k1, float(th), int(max_iters), float(confidence),
                                                                  sd.ctypes.data, float(dist), C.addressof(res),
                                                                  matches_out.data_ptr() if matches_out is not None else None, spans),
                  "mlpl_pair_pose_batch_lanes_dev")
            import time
            now = time.perf_counter()
            end = max(spans[2 * w + 1] for w in range(L))
            self.last_lane_span = [(now - (end - spans[2 * w]) * 1e-3, now - (end - spans[2 * w + 1]) * 1e-3) for w in range(L)]
            raw = np.frombuffer(res, _PAIR_RESULT_DTYPE, count=B)
            rec = np.zeros(B, RECORD_DTYPE)
            rec["pair_id"] = np.arange(B) if pair_ids is None else np.asarray(pair_ids)
            rec["n_matches"], rec["status"] = raw["n_matches"], raw["status"]
            ok = raw["status"] == 0
            for f in ("n_inliers", "E", "R", "t"):
                rec[f][ok] = raw[f][ok]
            return rec
        kw = dict(kw, th_pix=th_pix, max_iters=max_iters, confidence=confidence, dist=dist)
        lanes = max(1, min(self.lanes, B))
        bounds = [(B * w // lanes, B * (w + 1) // lanes) for w in range(lanes)]
        seeds = list(seeds)
        pair_ids = None if pair_ids is None else list(pair_ids)
        futs = [self.pool.submit(self._run, w, b, e, d_q, d_t, d_kp1, d_kp2, K0, K1, seeds, pair_ids, matches_out, kw) for w, (b, e) in enumerate(bounds)]
        rec = np.concatenate([f.result() for f in futs])
        if pair_ids is None:
            rec["pair_id"] = np.arange(B)
        return rec

    def close(self):
        self.pool.shutdown(wait=True)
        for c in self.owned:
            c.close()


class PairWorkers:
    """`workers` independent (library context, torch stream, host thread) triples on one GPU.  One image pair's pipeline
    is latency-bound (a dozen small launches and two host hops), so a rank overlaps several pairs instead of queueing
    them; every pair still runs exactly process_pair_on_device, so records do not depend on the worker count."""

    def __init__(self, device_index: int = 0, workers: int = 4):
        import torch
        from concurrent.futures import ThreadPoolExecutor

        self.device = torch.device("cuda", device_index)
        self.ctxs = [Context(device_index) for _ in range(workers)]
        self.streams = [torch.cuda.Stream(self.device) for _ in range(workers)]
        self.scratch = [{} for _ in range(workers)]
        self.pool = ThreadPoolExecutor(max_workers=workers)
        self.workers = workers
        self.last_call_ms = [[] for _ in range(workers)]

    def _run(self, w, jobs, K0, K1, kw):
        import torch

        import time

        out, ms = [], []
        torch.cuda.set_device(self.device)
        with torch.cuda.stream(self.streams[w]):
            for pair_id, seed, (d_q, d_t, d_kp1, d_kp2) in jobs:
                t0 = time.perf_counter()
                out.append(process_pair_on_device(self.ctxs[w], d_q, d_t, d_kp1, d_kp2, K0, K1, seed=seed, pair_id=pair_id,
                                                  scratch=self.scratch[w], **kw))
                ms.append(time.perf_counter() - t0)
            self.streams[w].synchronize()
        self.last_call_ms[w] = ms   # wall time of every library call of this worker in the last process(): tells which call stalled
        return out

    def process(self, pairs, K0, K1, seeds=None, pair_ids=None, **kw) -> np.ndarray:
        """pairs: list of (d_q, d_t, d_kp1, d_kp2) device tensors.  Returns the records in input order."""
        import torch

        n = len(pairs)
        seeds = list(range(n)) if seeds is None else list(seeds)
        pair_ids = list(range(n)) if pair_ids is None else list(pair_ids)
        torch.cuda.current_stream(self.device).synchronize()  # inputs were produced on the caller's stream
        jobs = [[(pair_ids[i], seeds[i], pairs[i]) for i in range(w, n, self.workers)] for w in range(self.workers)]
        futs = [self.pool.submit(self._run, w, jobs[w], K0, K1, kw) for w in range(self.workers)]
        recs = np.zeros(n, RECORD_DTYPE)
        for w, f in enumerate(futs):
            for k, r in enumerate(f.result()):
                recs[w + k * self.workers] = r[0]
        return recs

    def close(self):
        self.pool.shutdown(wait=True)
        for c in self.ctxs:
            c.close()
