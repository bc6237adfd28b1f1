"""Host-side mirror of poselib's robust relative-pose entry points (RANSAC path).

Reference interface: poselib/include/poselib/pose_estim.h:192-210 (getPoseTriangPts, estimateEssentialMat),
implementation poselib/source/pose_estim.cpp:857-946 over poselib/source/five-point-nister/*.  Same argument meaning
and failure behaviour; the arithmetic runs in libmlpl_hip.so (HIP, gfx950).  Nothing here computes on the host.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Tuple

import numpy as np

from . import _lib
from ._lib import Context, MlplError, check, default_context

PIX_MIN_GOOD_TH = 1.6  # poselib/include/poselib/pose_estim.h:59 (pixels; callers of the pose API pass camera-unit thresholds explicitly)


def _pts(p) -> np.ndarray:
    a = np.ascontiguousarray(p, dtype=np.float64)
    if a.ndim != 2 or a.shape[1] != 2:
        raise ValueError("points must be an n x 2 array")
    return a


def solve_5pt(p1, p2, samples, ctx: Optional[Context] = None) -> Tuple[np.ndarray, np.ndarray]:
    """Nister 5-point solver on `samples` (n_samples x 5 indices into p1/p2) -> (E [n_samples,10,3,3], n_models).
    Replaces CvEMEstimator::run5Point (five-point.cpp:366-471)."""
    ctx = ctx or default_context()
    p1, p2 = _pts(p1), _pts(p2)
    samples = np.ascontiguousarray(samples, np.int32)
    ns = samples.shape[0]
    E = np.zeros((ns, 10, 3, 3))
    nm = np.zeros(ns, np.int32)
    check(ctx.lib.mlpl_solve_5pt(ctx.handle, p1.ctypes.data, p2.ctypes.data, p1.shape[0], samples.ctypes.data, ns,
                                 E.ctypes.data, nm.ctypes.data), "mlpl_solve_5pt")
    return E, nm


def score_models(p1, p2, E, thresh: float, ctx: Optional[Context] = None) -> Tuple[np.ndarray, np.ndarray]:
    """Sampson inlier count and float-error sum per model (findInliers + cv::sum(err), modelest.cpp:69-83,407)."""
    ctx = ctx or default_context()
    p1, p2 = _pts(p1), _pts(p2)
    E = np.ascontiguousarray(E, np.float64).reshape(-1, 9)
    good = np.zeros(E.shape[0], np.int32)
    esum = np.zeros(E.shape[0], np.float64)
    check(ctx.lib.mlpl_score_models(ctx.handle, p1.ctypes.data, p2.ctypes.data, p1.shape[0], E.ctypes.data, E.shape[0],
                                    float(thresh), good.ctypes.data, esum.ctypes.data), "mlpl_score_models")
    return good, esum


def count_models(p1, p2, E, thresh2: float, shape: int = 0, ctx: Optional[Context] = None) -> np.ndarray:
    """Inlier counts by the division-free predicate of the RANSAC passes; thresh2 = squared threshold (modelest.cpp:79)."""
    ctx = ctx or default_context()
    p1, p2 = _pts(p1), _pts(p2)
    E = np.ascontiguousarray(E, np.float64).reshape(-1, 9)
    good = np.zeros(E.shape[0], np.int32)
    check(ctx.lib.mlpl_count_models(ctx.handle, p1.ctypes.data, p2.ctypes.data, p1.shape[0], E.ctypes.data, E.shape[0], float(thresh2),
                                    int(shape), good.ctypes.data), "mlpl_count_models")
    return good


def median_models(p1, p2, E, ctx: Optional[Context] = None) -> np.ndarray:
    """Median Sampson error per model as runLMeDS takes it (modelest.cpp:540-544)."""
    ctx = ctx or default_context()
    p1, p2 = _pts(p1), _pts(p2)
    E = np.ascontiguousarray(E, np.float64).reshape(-1, 9)
    med = np.zeros(E.shape[0], np.float64)
    check(ctx.lib.mlpl_median_models(ctx.handle, p1.ctypes.data, p2.ctypes.data, p1.shape[0], E.ctypes.data, E.shape[0],
                                     med.ctypes.data), "mlpl_median_models")
    return med


def ransac_essential(p1, p2, thresh: float, confidence: float = 0.999, max_iters: int = 1000, refit: bool = True,
                     seed: int = 0, ctx: Optional[Context] = None) -> dict:
    """CvModelEstimator3::runRANSAC with the 5-point kernel (modelest.cpp:343-474); parameters the reference
    hard-codes (1000 iterations, p = 0.999, srand(time)) are explicit here."""
    ctx = ctx or default_context()
    p1, p2 = _pts(p1), _pts(p2)
    n = p1.shape[0]
    E = np.zeros((3, 3))
    mask = np.zeros(n, np.uint8)
    ninl, iters = C.c_int(0), C.c_int(0)
    rc = ctx.lib.mlpl_ransac_essential(ctx.handle, p1.ctypes.data, p2.ctypes.data, n, float(thresh), float(confidence),
                                       int(max_iters), 1 if refit else 0, int(seed) & 0xFFFFFFFF, E.ctypes.data,
                                       mask.ctypes.data, C.byref(ninl), C.byref(iters))
    if rc not in (0, _lib.MLPL_E_FAILED):
        raise MlplError(rc, "mlpl_ransac_essential", _lib.last_error())
    return dict(ok=(rc == 0), E=E, mask=mask, n_inliers=ninl.value, iters=iters.value)


def lmeds_essential(p1, p2, confidence: float = 0.999, max_iters: int = 2000, seed: int = 0,
                    ctx: Optional[Context] = None) -> dict:
    """CvModelEstimator3::runLMeDS with the 5-point kernel (modelest.cpp:483-564; findEssentialMat passes 2000 iterations,
    five-point.cpp:127)."""
    ctx = ctx or default_context()
    p1, p2 = _pts(p1), _pts(p2)
    n = p1.shape[0]
    E = np.zeros((3, 3))
    mask = np.zeros(n, np.uint8)
    ninl, med = C.c_int(0), C.c_double(0)
    rc = ctx.lib.mlpl_lmeds_essential(ctx.handle, p1.ctypes.data, p2.ctypes.data, n, float(confidence), int(max_iters),
                                      int(seed) & 0xFFFFFFFF, E.ctypes.data, mask.ctypes.data, C.addressof(ninl),
                                      C.addressof(med))
    if rc not in (0, _lib.MLPL_E_FAILED):
        raise MlplError(rc, "mlpl_lmeds_essential", _lib.last_error())
    return dict(ok=(rc == 0), E=E, mask=mask, n_inliers=ninl.value, min_median=med.value)


# The reference's ARRSAC samplers draw from two function-local `static cv::RNG rng;` (prosac_sampler.h:115, random_sampler.h:65) that
# live as long as the process: this pair is their mirror for callers that do not keep their own.
ARRSAC_RNG_FRESH = (0xFFFFFFFF, 0xFFFFFFFF)
_arrsac_rng_state = np.array(ARRSAC_RNG_FRESH, np.uint64)


def arrsac_essential(p1, p2, thresh: float, refine: bool = True, rng_state=None, ctx: Optional[Context] = None) -> dict:
    """CvModelEstimator3::runARRSAC (modelest.cpp:197-341) as findEssentialMat(ARRSAC) drives it.  rng_state: uint64[2] array
    (updated in place) = the two cv::RNG states; None = the process-wide pair, as the reference's statics."""
    ctx = ctx or default_context()
    p1, p2 = _pts(p1), _pts(p2)
    n = p1.shape[0]
    st = _arrsac_rng_state if rng_state is None else rng_state
    assert st.dtype == np.uint64 and st.shape == (2,) and st.flags.c_contiguous
    E = np.zeros((3, 3))
    mask = np.zeros(n, np.uint8)
    ninl = C.c_int(0)
    rc = ctx.lib.mlpl_arrsac_essential(ctx.handle, p1.ctypes.data, p2.ctypes.data, n, float(thresh), 1 if refine else 0, st.ctypes.data,
                                       E.ctypes.data, mask.ctypes.data, C.addressof(ninl))
    if rc not in (0, _lib.MLPL_E_FAILED):
        raise MlplError(rc, "mlpl_arrsac_essential", _lib.last_error())
    stats = np.zeros(12, np.int64)
    ctx.lib.mlpl_arrsac_last_stats(ctx.handle, stats.ctypes.data)
    return dict(ok=(rc == 0), E=E, mask=mask, n_inliers=ninl.value, stats=stats)


class UsacParams(C.Structure):
    """mlpl_usac_params (include/mlpl_c.h): the configuration estimateEssentialMatUsac builds (usac_estimations.cpp:283-470)."""
    _fields_ = [("th", C.c_double), ("conf", C.c_double), ("max_hyp", C.c_int32), ("estimator", C.c_int32), ("refine", C.c_int32),
                ("seed", C.c_uint32), ("prosac_beta", C.c_double), ("sprt_delta", C.c_double), ("sprt_epsilon", C.c_double),
                ("sprt_mS", C.c_double), ("sprt_tM", C.c_double), ("sorted_idx", C.c_void_p), ("check_degeneracy", C.c_int32),
                ("reserved", C.c_int32), ("th_pixels", C.c_double), ("focal_length", C.c_double)]


def usac_essential(p1, p2, th: float, seed: int, sorted_idx=None, max_hyp: int = 50000, conf: float = 0.99, prosac_beta: float = 0.09,
                   sprt_delta: float = 0.05, sprt_epsilon: float = 0.15, sprt_ms: float = 8.5, sprt_tm: float = 2314.0, estimator: int = 0,
                   refine: int = 0, event_cap: int = 0, check_degeneracy: int = 0, th_pixels: float = 0.8, focal_length: float = 800.0,
                   ctx: Optional[Context] = None) -> dict:
    """estimateEssentialMatUsac with the Nister solver and REF_WEIGHTS (usac_estimations.cpp:283-735) on the device; `seed` is the
    reference's srand(time) seed.  sorted_idx (best match first) switches PROSAC on.  event_cap > 0 also returns the decision trace.
    check_degeneracy: 0 none, 1 the rotation-only / no-motion tests + model upgrade after every new best model (DEGEN_USAC_INTERNAL),
    3 also after every local optimisation; the result then carries `degen` = [1, inliers of the rotation, of "no motion", type],
    `R_degen`, `flags_rot`, `flags_nomot`."""
    ctx = ctx or default_context()
    p1, p2 = _pts(p1), _pts(p2)
    n = p1.shape[0]
    P = UsacParams()
    ctx.lib.mlpl_usac_default_params(C.addressof(P), float(th))
    P.conf, P.max_hyp, P.estimator, P.refine, P.seed = float(conf), int(max_hyp), int(estimator), int(refine), int(seed) & 0xFFFFFFFF
    P.prosac_beta, P.sprt_delta, P.sprt_epsilon, P.sprt_mS, P.sprt_tM = float(prosac_beta), float(sprt_delta), float(sprt_epsilon), \
        float(sprt_ms), float(sprt_tm)
    P.check_degeneracy, P.th_pixels, P.focal_length = int(check_degeneracy), float(th_pixels), float(focal_length)
    si = None
    if sorted_idx is not None:
        si = np.ascontiguousarray(sorted_idx, np.uint32)
        P.sorted_idx = si.ctypes.data
    E, mask, res = np.zeros(9), np.zeros(n, np.uint8), np.zeros(12)
    ev = np.zeros((max(event_cap, 1), 16))
    if event_cap:
        ctx.lib.mlpl_debug_usac_trace(ctx.handle, ev.ctypes.data, int(event_cap))
    rc = ctx.lib.mlpl_usac_essential(ctx.handle, p1.ctypes.data, p2.ctypes.data, n, C.addressof(P), E.ctypes.data, mask.ctypes.data,
                                     res.ctypes.data)
    nev = ctx.lib.mlpl_debug_usac_trace(ctx.handle, None, 0) if event_cap else 0
    if rc not in (0, _lib.MLPL_E_FAILED):
        raise MlplError(rc, "mlpl_usac_essential", _lib.last_error())
    stats = np.zeros(8, np.int64)
    ctx.lib.mlpl_usac_last_stats(ctx.handle, stats.ctypes.data)
    out = dict(ok=(rc == 0), E=E, flags=mask, final=res, events=ev[:min(nev, event_cap)], n_events=nev, stats=stats)
    if check_degeneracy and rc == 0:
        info, fr, fn = np.zeros(16), np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        check(ctx.lib.mlpl_usac_last_degeneracy(ctx.handle, info.ctypes.data, fr.ctypes.data, fn.ctypes.data, n), "mlpl_usac_last_degeneracy")
        out.update(degen=info[:4].copy(), R_degen=info[4:13].copy(), flags_rot=fr, flags_nomot=fn)
    return out


def arrsac_essential_batch(d_p1, d_p2, counts, thresh: float, refine: bool = True, rng_states=None, masks_out=None, ctx: Optional[Context] = None) -> list:
    """A batch of ARRSAC problems in ONE library call (mlpl_arrsac_essential_batch_dev): d_p1, d_p2 float64 CUDA tensors [B, stride, 2],
    counts[b] valid rows; rng_states: uint64 [B, 2] (default: fresh streams for every problem), advanced in place.  Returns one dict per problem
    (ok, E, n_inliers, rng_state); masks through masks_out (uint8 CUDA tensor [B, stride])."""
    import torch

    ctx = ctx or default_context()
    B, stride = d_p1.shape[0], d_p1.shape[1]
    assert d_p1.is_cuda and d_p1.dtype == torch.float64 and d_p1.shape == d_p2.shape == (B, stride, 2) and d_p1.is_contiguous() and d_p2.is_contiguous()
    cn = np.ascontiguousarray(counts, np.int32)
    st = np.tile(np.array(ARRSAC_RNG_FRESH, np.uint64), (B, 1)) if rng_states is None else rng_states
    assert st.dtype == np.uint64 and st.shape == (B, 2) and st.flags.c_contiguous
    if masks_out is None:
        masks_out = torch.zeros((B, stride), dtype=torch.uint8, device=d_p1.device)
    assert masks_out.is_cuda and masks_out.dtype == torch.uint8 and masks_out.shape == (B, stride) and masks_out.is_contiguous()
    E, ninl, status = np.zeros((B, 9)), np.zeros(B, np.int32), np.zeros(B, np.int32)
    rc = ctx.lib.mlpl_arrsac_essential_batch_dev(ctx.handle, B, d_p1.data_ptr(), d_p2.data_ptr(), stride, cn.ctypes.data, float(thresh), 1 if refine else 0,
                                                 st.ctypes.data, E.ctypes.data, masks_out.data_ptr(), ninl.ctypes.data, status.ctypes.data,
                                                 torch.cuda.current_stream(d_p1.device).cuda_stream)
    if rc != 0:
        raise MlplError(rc, "mlpl_arrsac_essential_batch_dev", _lib.last_error())
    return [dict(ok=bool(status[b] == 0), E=E[b].copy(), n_inliers=int(ninl[b]), rng_state=st[b].copy()) for b in range(B)]


def usac_essential_batch(d_p1, d_p2, counts, th: float, seeds, sorted_idx=None, max_hyp: int = 50000, conf: float = 0.99, prosac_beta: float = 0.09,
                         sprt_delta: float = 0.05, sprt_epsilon: float = 0.15, sprt_ms: float = 8.5, sprt_tm: float = 2314.0, estimator: int = 0,
                         refine: int = 0, event_cap: int = 0, check_degeneracy: int = 0, th_pixels: float = 0.8, focal_length: float = 800.0,
                         masks_out=None, ctx: Optional[Context] = None) -> list:
    """A batch of USAC problems in ONE library call (mlpl_usac_essential_batch_dev): d_p1, d_p2 float64 CUDA tensors [B, stride, 2],
    counts[b] valid rows, seeds[b]; sorted_idx: None or a list of per-problem PROSAC orders (None entries = uniform sampling).  Returns one
    dict per problem with the keys of usac_essential (ok, E, final, events, n_events, degen ...; flags only through masks_out, a uint8
    CUDA tensor [B, stride])."""
    import torch

    ctx = ctx or default_context()
    B, stride = d_p1.shape[0], d_p1.shape[1]
    assert d_p1.is_cuda and d_p1.dtype == torch.float64 and d_p1.shape == d_p2.shape == (B, stride, 2) and d_p1.is_contiguous() and d_p2.is_contiguous()
    cn = np.ascontiguousarray(counts, np.int32)
    P = (UsacParams * B)()
    keep = []
    for b in range(B):
        ctx.lib.mlpl_usac_default_params(C.addressof(P[b]), float(th))
        P[b].conf, P[b].max_hyp, P[b].estimator, P[b].refine, P[b].seed = float(conf), int(max_hyp), int(estimator), int(refine), int(seeds[b]) & 0xFFFFFFFF
        P[b].prosac_beta, P[b].sprt_delta, P[b].sprt_epsilon, P[b].sprt_mS, P[b].sprt_tM = float(prosac_beta), float(sprt_delta), float(sprt_epsilon), \
            float(sprt_ms), float(sprt_tm)
        P[b].check_degeneracy, P[b].th_pixels, P[b].focal_length = int(check_degeneracy), float(th_pixels), float(focal_length)
        if sorted_idx is not None and sorted_idx[b] is not None:
            si = np.ascontiguousarray(sorted_idx[b], np.uint32)
            keep.append(si)
            P[b].sorted_idx = si.ctypes.data
    E, res, status, degen = np.zeros((B, 9)), np.zeros((B, 12)), np.zeros(B, np.int32), np.zeros((B, 16))
    ev = np.zeros((B, max(event_cap, 1), 16))
    lens = np.zeros(B, np.int32)
    if masks_out is not None:
        assert masks_out.is_cuda and masks_out.dtype == torch.uint8 and masks_out.shape == (B, stride) and masks_out.is_contiguous()
    st = torch.cuda.current_stream(d_p1.device).cuda_stream
    rc = ctx.lib.mlpl_usac_essential_batch_dev(ctx.handle, B, d_p1.data_ptr(), d_p2.data_ptr(), stride, cn.ctypes.data, C.addressof(P), E.ctypes.data,
                                               masks_out.data_ptr() if masks_out is not None else None, res.ctypes.data, status.ctypes.data,
                                               degen.ctypes.data, ev.ctypes.data if event_cap else None, int(event_cap), lens.ctypes.data if event_cap else None, st)
    if rc != 0:
        raise MlplError(rc, "mlpl_usac_essential_batch_dev", _lib.last_error())
    stats = np.zeros(8, np.int64)
    ctx.lib.mlpl_usac_last_stats(ctx.handle, stats.ctypes.data)
    out = []
    for b in range(B):
        if status[b] not in (0, _lib.MLPL_E_FAILED):
            raise MlplError(int(status[b]), "mlpl_usac_essential_batch_dev", f"problem {b}")
        out.append(dict(ok=bool(status[b] == 0), E=E[b].copy(), final=res[b].copy(), events=ev[b, :min(int(lens[b]), event_cap)].copy(), n_events=int(lens[b]),
                        degen=degen[b, :4].copy(), R_degen=degen[b, 4:13].copy(), stats=stats))
    return out


def arrsac_sample_models(p1, p2, idx, kind: int, thresh: float = 1e-3, ctx: Optional[Context] = None):
    """ARRSAC's estimators on one sample (modelest.cpp:111-178) -> (models [k,3,3] before the validity filter, valid flags [k])."""
    ctx = ctx or default_context()
    p1, p2 = _pts(p1), _pts(p2)
    idx = np.ascontiguousarray(idx, np.int32)
    E = np.zeros((10, 3, 3))
    nm = C.c_int(0)
    valid = np.zeros(10, np.uint8)
    check(ctx.lib.mlpl_arrsac_sample_models(ctx.handle, p1.ctypes.data, p2.ctypes.data, p1.shape[0], idx.ctypes.data, len(idx), int(kind),
                                            float(thresh), E.ctypes.data, C.addressof(nm), valid.ctypes.data), "mlpl_arrsac_sample_models")
    return E[: nm.value].copy(), valid[: nm.value].astype(bool)


def robust_essential_refine(p1, p2, E_init, th: float, mask=None, ctx: Optional[Context] = None):
    """poselib::robustEssentialRefine for the essential-matrix model (pose_estim.cpp:337-792) -> (E_refined, rounds, status)."""
    ctx = ctx or default_context()
    p1, p2 = _pts(p1), _pts(p2)
    E0 = np.ascontiguousarray(E_init, np.float64).reshape(3, 3)
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(-1)
    E = np.zeros((3, 3))
    info = np.zeros(2, np.int32)
    check(ctx.lib.mlpl_robust_essential_refine(ctx.handle, p1.ctypes.data, p2.ctypes.data, p1.shape[0], None if m is None else m.ctypes.data,
                                               E0.ctypes.data, float(th), E.ctypes.data, info.ctypes.data), "mlpl_robust_essential_refine")
    return E, int(info[0]), int(info[1])


def ransac_essential_device(p1, p2, thresh: float, confidence: float = 0.999, max_iters: int = 1000, refit: bool = True,
                            seed: int = 0, ctx: Optional[Context] = None, mask_out=None, stream: Optional[int] = None) -> dict:
    """Same as ransac_essential on device-resident torch tensors (float64 [n,2]); the mask stays on the device."""
    import torch

    assert p1.is_cuda and p2.is_cuda and p1.dtype == torch.float64 and p1.is_contiguous() and p2.is_contiguous()
    ctx = ctx or default_context(p1.device.index or 0)
    n = p1.shape[0]
    if mask_out is None:
        mask_out = torch.empty(n, dtype=torch.uint8, device=p1.device)
    E = np.zeros((3, 3))
    ninl, iters = C.c_int(0), C.c_int(0)
    st = torch.cuda.current_stream(p1.device).cuda_stream if stream is None else stream
    rc = ctx.lib.mlpl_ransac_essential_dev(ctx.handle, p1.data_ptr(), p2.data_ptr(), n, float(thresh), float(confidence),
                                           int(max_iters), 1 if refit else 0, int(seed) & 0xFFFFFFFF, E.ctypes.data,
                                           mask_out.data_ptr(), C.byref(ninl), C.byref(iters), st)
    if rc not in (0, _lib.MLPL_E_FAILED):
        raise MlplError(rc, "mlpl_ransac_essential_dev", _lib.last_error())
    return dict(ok=(rc == 0), E=E, mask=mask_out, n_inliers=ninl.value, iters=iters.value)


def arrsac_essential_device(p1, p2, thresh: float, refine: bool = True, rng_state=None, ctx: Optional[Context] = None, mask_out=None,
                            stream: Optional[int] = None) -> dict:
    """Same as arrsac_essential on device-resident torch tensors (float64 [n,2]); the mask stays on the device."""
    import torch

    assert p1.is_cuda and p2.is_cuda and p1.dtype == torch.float64 and p1.is_contiguous() and p2.is_contiguous()
    ctx = ctx or default_context(p1.device.index or 0)
    n = p1.shape[0]
    if mask_out is None:
        mask_out = torch.empty(n, dtype=torch.uint8, device=p1.device)
    st = _arrsac_rng_state if rng_state is None else rng_state
    E = np.zeros((3, 3))
    ninl = C.c_int(0)
    sm = torch.cuda.current_stream(p1.device).cuda_stream if stream is None else stream
    rc = ctx.lib.mlpl_arrsac_essential_dev(ctx.handle, p1.data_ptr(), p2.data_ptr(), n, float(thresh), 1 if refine else 0, st.ctypes.data,
                                           E.ctypes.data, mask_out.data_ptr(), C.addressof(ninl), sm)
    if rc not in (0, _lib.MLPL_E_FAILED):
        raise MlplError(rc, "mlpl_arrsac_essential_dev", _lib.last_error())
    return dict(ok=(rc == 0), E=E, mask=mask_out, n_inliers=ninl.value)


def estimateEssentialMat(p1, p2, method: str = "ARRSAC", threshold: float = PIX_MIN_GOOD_TH, refine: bool = True,
                         seed: Optional[int] = None, ctx: Optional[Context] = None):
    """poselib::estimateEssentialMat (pose_estim.h:204-210, pose_estim.cpp:857-890) -> (ok, E, mask).

    "RANSAC" (the hot path), "LMEDS" and "ARRSAC" (the reference's default) are built in this library.  Like the reference, "USAC"
    and unknown method names are fatal (the reference prints and calls exit(1), pose_estim.cpp:878-887): SystemExit(1) is raised.
    `seed` = None mirrors the reference's std::srand(std::time(nullptr)) (modelest.cpp:58); ARRSAC takes no seed (its cv::RNG
    streams are process-wide, see arrsac_essential)."""
    if method in ("RANSAC", "LMEDS", "ARRSAC"):
        import time

        a, b = _pts(p1), _pts(p2)
        if a.shape[0] == 5:
            # findEssentialMat's minimal case (five-point.cpp:108-114): one solver call, up to 10 stacked 3x3 matrices,
            # mask all true
            Es, nm = solve_5pt(a, b, np.arange(5, dtype=np.int32)[None, :], ctx=ctx)
            return True, Es[0, : nm[0]].reshape(-1, 3).copy(), np.ones(5, np.uint8)
        if method == "ARRSAC":
            r = arrsac_essential(p1, p2, threshold, refine=refine, ctx=ctx)
            return r["ok"], r["E"], r["mask"]
        s = int(time.time()) if seed is None else seed
        if method == "LMEDS":  # no least-squares refit on this branch (pose_estim.cpp:874-877)
            r = lmeds_essential(p1, p2, confidence=0.999, max_iters=2000, seed=s, ctx=ctx)
        else:
            r = ransac_essential(p1, p2, threshold, confidence=0.999, max_iters=1000, refit=refine, seed=s, ctx=ctx)
        return r["ok"], r["E"], r["mask"]
    if method == "USAC":
        print("USAC must be executed by function estimateEssentialOrPoseUSAC as it needs additional paramters! Exiting.")
    else:
        print("Either there is a typo in the specified robust estimation method or the method is not supported. Exiting.")
    raise SystemExit(1)


def getPoseTriangPts(E, p1, p2, mask=None, dist: float = 50.0, translatE: bool = False,
                     ctx: Optional[Context] = None):
    """poselib::getPoseTriangPts (pose_estim.h:192-200, pose_estim.cpp:913-946) -> (n_good, R, t, Q, mask)."""
    ctx = ctx or default_context()
    p1, p2 = _pts(p1), _pts(p2)
    E = np.ascontiguousarray(E, np.float64).reshape(3, 3)
    n = p1.shape[0]
    R, t, Q = np.zeros((3, 3)), np.zeros((3, 1)), np.zeros((n, 3))
    m = None if mask is None else np.ascontiguousarray(mask, np.uint8).reshape(-1).copy()
    if translatE:
        # getTfromTransEssential (pose_helper.cpp:422-433): t = (E12, E20, E01), normalised unless already unit
        tv = np.array([E[1, 2], E[2, 0], E[0, 1]])
        nrm = float(np.sqrt(tv[0] * tv[0] + tv[1] * tv[1] + tv[2] * tv[2]))
        if abs(nrm - 1.0) > 1e-3:
            tv = tv / nrm
        tv = np.ascontiguousarray(tv)
        rc = ctx.lib.mlpl_recover_pose_translation(ctx.handle, tv.ctypes.data, p1.ctypes.data, p2.ctypes.data, n, float(dist),
                                                   R.ctypes.data, t.ctypes.data, Q.ctypes.data,
                                                   None if m is None else m.ctypes.data)
    else:
        rc = ctx.lib.mlpl_recover_pose(ctx.handle, E.ctypes.data, p1.ctypes.data, p2.ctypes.data, n, float(dist),
                                       R.ctypes.data, t.ctypes.data, Q.ctypes.data, None if m is None else m.ctypes.data)
    if rc < 0:
        raise MlplError(rc, "mlpl_recover_pose", _lib.last_error())
    return rc, R, t, Q, m


def getPoseTriangPts_device(E, p1, p2, mask=None, dist: float = 50.0, Q_out=None, ctx: Optional[Context] = None,
                            stream: Optional[int] = None):
    """getPoseTriangPts on device-resident torch tensors (float64 [n,2]; mask uint8 [n], rewritten in place; Q_out float64 [n,3]
    or None) -> (n_good, R, t).  One host hop."""
    import torch

    assert p1.is_cuda and p2.is_cuda and p1.dtype == torch.float64 and p1.is_contiguous() and p2.is_contiguous()
    ctx = ctx or default_context(p1.device.index or 0)
    E = np.ascontiguousarray(E, np.float64).reshape(3, 3)
    R, t = np.zeros((3, 3)), np.zeros((3, 1))
    st = torch.cuda.current_stream(p1.device).cuda_stream if stream is None else stream
    rc = ctx.lib.mlpl_recover_pose_dev(ctx.handle, E.ctypes.data, p1.data_ptr(), p2.data_ptr(), p1.shape[0], float(dist),
                                       R.ctypes.data, t.ctypes.data, None if Q_out is None else Q_out.data_ptr(),
                                       None if mask is None else mask.data_ptr(), st)
    if rc < 0:
        raise MlplError(rc, "mlpl_recover_pose_dev", _lib.last_error())
    return rc, R, t


def ImgToCamCoordTrans(points, K, ctx: Optional[Context] = None) -> np.ndarray:
    """poselib::ImgToCamCoordTrans (pose_helper.cpp:1100-1109): float32 n x 2 pixel points -> camera coordinates."""
    ctx = ctx or default_context()
    pts = np.ascontiguousarray(points, np.float32).copy()
    K = np.asarray(K, np.float64)
    k4 = np.array([K[0, 0], K[1, 1], K[0, 2], K[1, 2]]) if K.shape == (3, 3) else K.astype(np.float64)
    check(ctx.lib.mlpl_img_to_cam(ctx.handle, pts.ctypes.data, pts.shape[0], k4.ctypes.data), "mlpl_img_to_cam")
    return pts


def Remove_LensDist(points1, points2, dist1, dist2, ctx: Optional[Context] = None):
    """poselib::Remove_LensDist (pose_helper.cpp:1169-1223) -> (ok, points1, points2) with failing pairs dropped."""
    ctx = ctx or default_context()
    a = np.ascontiguousarray(points1, np.float32).copy()
    b = np.ascontiguousarray(points2, np.float32).copy()
    d1 = np.ascontiguousarray(np.asarray(dist1, np.float64).reshape(-1))
    d2 = np.ascontiguousarray(np.asarray(dist2, np.float64).reshape(-1))
    assert d1.size == 8 and d2.size == 8 and a.shape == b.shape
    n_out = C.c_int(0)
    rc = ctx.lib.mlpl_remove_lens_dist(ctx.handle, a.ctypes.data, b.ctypes.data, a.shape[0], d1.ctypes.data, d2.ctypes.data,
                                       C.byref(n_out))
    if rc not in (0, _lib.MLPL_E_FAILED):
        raise MlplError(rc, "mlpl_remove_lens_dist", _lib.last_error())
    return rc == 0, a[: n_out.value], b[: n_out.value]


def getInliers(E, p1, p2, th2: float, ctx: Optional[Context] = None):
    """StereoRefine::getInliers (stereo_pose_refinement.cpp:2085-2090) -> (n_inliers, mask, error); strict `<`."""
    ctx = ctx or default_context()
    p1, p2 = _pts(p1), _pts(p2)
    E = np.ascontiguousarray(E, np.float64).reshape(3, 3)
    n = p1.shape[0]
    err = np.zeros(n)
    mask = np.zeros(n, np.uint8)
    rc = ctx.lib.mlpl_get_inliers_strict(ctx.handle, p1.ctypes.data, p2.ctypes.data, n, E.ctypes.data, float(th2),
                                         err.ctypes.data, mask.ctypes.data)
    if rc < 0:
        raise MlplError(rc, "mlpl_get_inliers_strict", _lib.last_error())
    return rc, mask, err


def estimateRelativePose(p1, p2, threshold: float = PIX_MIN_GOOD_TH, refine: bool = True, dist: float = 50.0,
                         seed: Optional[int] = None, ctx: Optional[Context] = None):
    """Convenience named in BASELINE.json: estimateEssentialMat("RANSAC") followed by getPoseTriangPts, as the
    reference's README and harness do (README.md:497-521, tests/poselib-test/main.cpp:1717,1827)."""
    ok, E, mask = estimateEssentialMat(p1, p2, "RANSAC", threshold, refine, seed=seed, ctx=ctx)
    if not ok:
        return False, E, None, None, None, mask
    n_good, R, t, Q, mask2 = getPoseTriangPts(E, p1, p2, mask, dist, ctx=ctx)
    return True, E, R, t, Q, mask2


def smoke_check(ctx, oracle) -> None:
    """Used by __graft_entry__.smoke(): one small RANSAC + pose recovery on the GPU, checked against the CPU oracle."""
    from . import synth

    p1, p2, R, t, mask, th = synth.pose_scene(600, seed=3)
    g = ransac_essential(p1, p2, th, confidence=0.999, max_iters=300, refit=False, seed=99, ctx=ctx)
    o = oracle.ransac_essential(p1, p2, th, confidence=0.999, max_iters=300, lesqu=False, seed=99)
    assert g["ok"] and g["iters"] == o["iters"] and g["n_inliers"] == o["n_inliers"], "RANSAC differs from the oracle"
    d = min(np.abs(g["E"] - o["E"]).max(), np.abs(g["E"] + o["E"]).max())
    assert d < 1e-8, f"essential matrix differs from the oracle by {d}"
    n_good, Rg, tg, Q, m = getPoseTriangPts(g["E"], p1, p2, g["mask"], 50.0, ctx=ctx)
    go, Ro, to, Qo, mo = oracle.recover_pose(o["E"], p1, p2, 50.0, o["mask"])
    assert n_good == go and np.abs(Rg - Ro).max() < 1e-6 and np.abs(tg.ravel() - to).max() < 1e-6
    print(f"smoke: RANSAC 5pt {g['iters']} iters -> {g['n_inliers']} inliers, pose within 1e-6 of the oracle")
    # the sequential estimators as a batch (launch hub: fibers, merged launches, cohorts in flight): four USAC problems in one call, each
    # against the oracle's run of it
    import torch

    seeds = [5, 6, 7, 8]
    d1 = torch.from_numpy(np.repeat(p1[None], 4, 0).copy()).cuda()
    d2 = torch.from_numpy(np.repeat(p2[None], 4, 0).copy()).cuda()
    got = usac_essential_batch(d1, d2, [len(p1)] * 4, th, seeds, ctx=ctx)
    for b, sd in enumerate(seeds):
        ou = oracle.usac_essential(p1, p2, th, sd)
        # (hypotheses, inliers, local optimisations; the model COUNT may differ by the solutions of a sample with a double root)
        assert got[b]["ok"] and np.array_equal(got[b]["final"][[0, 1, 5, 7]], ou["final"][[0, 1, 5, 7]]), "batched USAC differs from the oracle"
        du = min(np.abs(got[b]["E"] - ou["E"]).max(), np.abs(got[b]["E"] + ou["E"]).max())
        assert du < 1e-8, f"batched USAC: essential matrix differs from the oracle by {du}"
    print(f"smoke: USAC x 4 in one batched call -> {int(got[0]['final'][5])} inliers each; hypotheses, inliers, local optimisations and E equal to the oracle's")
