"""Host-side mirror of matchinglib::getMatches for the LINEAR (brute-force) matcher.

Reference interface: matchinglib/include/matchinglib/matchinglib_matchers.h:61-64, implementation
matchinglib/source/matchers.cpp:115-736 (LINEAR branch :525-714).  Same argument meaning, same return
codes; the arithmetic runs in libmlpl_hip.so (HIP, gfx950).  Nothing here computes distances on the host.
"""
from __future__ import annotations

import ctypes as C
from typing import Optional, Sequence, Tuple

import numpy as np

from . import _lib
from ._lib import Context, MlplError, check, default_context

# cv::DMatch {int queryIdx; int trainIdx; int imgIdx; float distance;}
DMATCH_DTYPE = np.dtype(
    [("queryIdx", np.int32), ("trainIdx", np.int32), ("imgIdx", np.int32), ("distance", np.float32)], align=True
)
assert DMATCH_DTYPE.itemsize == 16

CV_8U = 0
CV_32F = 5


def _rows_2d(a: np.ndarray, dtype) -> np.ndarray:
    a = np.asarray(a)
    if a.ndim != 2 or a.dtype != dtype:
        raise ValueError(f"expected a 2-D {np.dtype(dtype).name} array, got {a.dtype} with shape {a.shape}")
    if a.strides[1] != a.itemsize:  # rows must be dense; row stride may be anything (cv::Mat::step)
        a = np.ascontiguousarray(a)
    return a


def knn_hamming(q: np.ndarray, t: np.ndarray, k: int = 2, ctx: Optional[Context] = None) -> Tuple[np.ndarray, np.ndarray]:
    """Exact k-NN (k in {1,2}) of every row of q in t under bit-Hamming distance -> (idx, dist) int32 [nq,k].

    Replaces cvflann::Index<HammingLUT>(LinearIndexParams).knnSearch (matchers.cpp:584-588)."""
    ctx = ctx or default_context()
    q = _rows_2d(q, np.uint8)
    t = _rows_2d(t, np.uint8)
    if q.shape[1] != t.shape[1]:
        raise ValueError("descriptor widths differ")
    nq, nbytes = q.shape
    idx = np.empty((nq, k), np.int32)
    dist = np.empty((nq, k), np.int32)
    rc = ctx.lib.mlpl_knn2_hamming(
        ctx.handle, q.ctypes.data, nq, q.strides[0], t.ctypes.data, t.shape[0], t.strides[0], nbytes, k,
        idx.ctypes.data, dist.ctypes.data)
    check(rc, "mlpl_knn2_hamming")
    return idx, dist


def knn_l2sq(q: np.ndarray, t: np.ndarray, k: int = 2, ctx: Optional[Context] = None) -> Tuple[np.ndarray, np.ndarray]:
    """Exact k-NN under SQUARED L2 on float32 descriptors -> (idx int32 [nq,k], dist float32 [nq,k]).

    Replaces cvflann::Index<L2<float>>(LinearIndexParams).knnSearch (matchers.cpp:660-664)."""
    ctx = ctx or default_context()
    q = _rows_2d(q, np.float32)
    t = _rows_2d(t, np.float32)
    if q.shape[1] != t.shape[1]:
        raise ValueError("descriptor widths differ")
    nq, dim = q.shape
    idx = np.empty((nq, k), np.int32)
    dist = np.empty((nq, k), np.float32)
    rc = ctx.lib.mlpl_knn2_l2sq_f32(
        ctx.handle, q.ctypes.data, nq, q.strides[0] // 4, t.ctypes.data, t.shape[0], t.strides[0] // 4, dim, k,
        idx.ctypes.data, dist.ctypes.data)
    check(rc, "mlpl_knn2_l2sq_f32")
    return idx, dist


def ratio_compact(idx: np.ndarray, dist: np.ndarray, ratio: float = 0.75, ctx: Optional[Context] = None) -> np.ndarray:
    """Ratio test + DMatch emission (matchers.cpp:601-625 / :677-701) -> structured array of DMATCH_DTYPE."""
    ctx = ctx or default_context()
    idx = np.ascontiguousarray(idx, np.int32)
    nq, k = idx.shape
    out = np.empty(nq, DMATCH_DTYPE)
    n = C.c_int(0)
    if dist.dtype == np.float32:
        dist = np.ascontiguousarray(dist)
        rc = ctx.lib.mlpl_ratio_compact_f32(ctx.handle, idx.ctypes.data, dist.ctypes.data, nq, k, ratio,
                                            out.ctypes.data, C.byref(n))
    else:
        dist = np.ascontiguousarray(dist, np.int32)
        rc = ctx.lib.mlpl_ratio_compact_i32(ctx.handle, idx.ctypes.data, dist.ctypes.data, nq, k, ratio,
                                            out.ctypes.data, C.byref(n))
    check(rc, "mlpl_ratio_compact")
    return out[: n.value].copy()


def getMatches(
    keypoints1: Sequence,
    keypoints2: Sequence,
    descriptors1: np.ndarray,
    descriptors2: np.ndarray,
    imgSi=None,
    matcher_name: str = "GMBSOF",
    VFCrefine: bool = False,
    ratioTest: bool = True,
    descriptor_name: str = "",
    idxPars_NMSLIB: str = "",
    queryPars_NMSLIB: str = "",
    nr_threads: int = 0,
    ctx: Optional[Context] = None,
) -> Tuple[int, np.ndarray]:
    """matchinglib::getMatches (matchinglib_matchers.h:61-64).  Returns (err, finalMatches).

    err: 0 ok, -1 wrong input data, -2 matcher not supported, -3 matching failed (< 2 matches),
    -4 too few keypoints (matchers.cpp:109-114).  matcher_name "LINEAR" (cvflann brute force, matchers.cpp:525-714) and
    "BRUTEFORCENMS" (NMSLIB seq_search, matchers.cpp:476-519, with its 240-bit / sqrt-L2 semantics) are built in this
    library; every other name -- including the reference's default "GMBSOF" -- yields -2.
    A dtype mismatch raises ValueError where the reference's CV_Assert throws cv::Exception (matchers.cpp:119).
    """
    d1 = np.asarray(descriptors1)
    d2 = np.asarray(descriptors2)
    empty = np.empty(0, DMATCH_DTYPE)
    if d1.dtype != d2.dtype:
        raise ValueError("descriptors1.type() == descriptors2.type() assertion failed")  # CV_Assert, :119
    n1, n2 = len(keypoints1), len(keypoints2)
    if n1 < 15 or n2 < 15:
        return -4, empty
    if d1.ndim != 2 or d2.ndim != 2 or n1 != d1.shape[0] or n2 != d2.shape[0]:
        return -1, empty
    if matcher_name not in ("LINEAR", "BRUTEFORCENMS"):
        return -2, empty
    if VFCrefine:
        raise NotImplementedError("VFC refinement (matchers.cpp:722-733) is outside the hot path built here")
    if d1.dtype == np.uint8:
        desc_type = CV_8U
    elif d1.dtype == np.float32:
        desc_type = CV_32F
    else:
        return -1, empty  # "Format of descriptors not supported!" (matchers.cpp:540-547)
    if d1.shape[1] != d2.shape[1]:
        return -1, empty
    ctx = ctx or default_context()
    d1 = _rows_2d(d1, d1.dtype)
    d2 = _rows_2d(d2, d2.dtype)
    out = np.empty(max(n1, 1), DMATCH_DTYPE)
    n = C.c_int(0)
    entry = ctx.lib.mlpl_get_matches_linear if matcher_name == "LINEAR" else ctx.lib.mlpl_get_matches_bruteforce_nms
    rc = entry(
        ctx.handle, n1, n2, d1.ctypes.data, d1.shape[0], d1.strides[0], d2.ctypes.data, d2.shape[0], d2.strides[0],
        d1.shape[1], desc_type, 1 if ratioTest else 0, out.ctypes.data, C.byref(n))
    if rc in (0, -3):
        return rc, out[: n.value].copy()
    if rc in (-1, -4):
        return rc, empty
    raise MlplError(rc, "mlpl_get_matches_linear", _lib.last_error())


# ---- device-resident (torch) entry: nothing leaves HBM -------------------------------------------------------

def match_hamming_device(q, t, ratio_test: bool = True, ratio: float = 0.75, ctx: Optional[Context] = None, out=None,
                         stream: Optional[int] = None):
    """Batched knn(+ratio+compaction) on CUDA/HIP torch tensors.

    q: uint8 [B, nq, nbytes] (or [nq, nbytes]), t: uint8 [B, nt, nbytes]; returns dict of torch tensors
    idx [B,nq,k] int32, dist [B,nq,k] int32, matches [B,nq,4] int32 (DMatch rows, .distance bit-cast),
    count [B] int32.  Enqueues on `stream` (a hipStream_t handle; None = torch's current stream) without
    synchronising.
    """
    import torch

    if q.dim() == 2:
        q = q.unsqueeze(0)
    if t.dim() == 2:
        t = t.unsqueeze(0)
    assert q.is_cuda and t.is_cuda and q.dtype == torch.uint8 and t.dtype == torch.uint8
    assert q.stride(2) == 1 and t.stride(2) == 1
    B, nq, nbytes = q.shape
    nt = t.shape[1]
    ctx = ctx or default_context(q.device.index or 0)
    k = 2 if ratio_test else 1
    if out is None:
        out = {
            "idx": torch.empty((B, nq, k), dtype=torch.int32, device=q.device),
            "dist": torch.empty((B, nq, k), dtype=torch.int32, device=q.device),
            "matches": torch.empty((B, nq, 4), dtype=torch.int32, device=q.device),
            "count": torch.empty((B,), dtype=torch.int32, device=q.device),
        }
    rc = ctx.lib.mlpl_match_hamming_dev(
        ctx.handle, q.data_ptr(), nq, q.stride(1), q.stride(0), t.data_ptr(), nt, t.stride(1), t.stride(0), nbytes,
        1 if ratio_test else 0, ratio, B, out["idx"].data_ptr(), out["dist"].data_ptr(), out["matches"].data_ptr(),
        out["count"].data_ptr(), torch.cuda.current_stream(q.device).cuda_stream if stream is None else stream)
    check(rc, "mlpl_match_hamming_dev")
    return out
