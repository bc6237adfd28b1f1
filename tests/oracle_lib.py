"""ctypes binding of oracle/liboracle.so -- the CPU checker.  TEST INFRASTRUCTURE ONLY (see oracle/oracle.h)."""
import ctypes as C
import os
import subprocess

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
ORACLE_DIR = os.path.join(ROOT, "oracle")
LIB = os.path.join(ORACLE_DIR, "liboracle.so")

DMATCH = np.dtype([("queryIdx", np.int32), ("trainIdx", np.int32), ("imgIdx", np.int32), ("distance", np.float32)],
                  align=True)


class TraceRec(C.Structure):
    _fields_ = [("idx", C.c_int32 * 5), ("nmodels", C.c_int32), ("good", C.c_int32 * 10),
                ("err_sum", C.c_double * 10), ("niters_after", C.c_int32), ("best_taken", C.c_int32)]


class GlibcRand(C.Structure):
    _fields_ = [("r", C.c_int32 * 34), ("f", C.c_int), ("b", C.c_int)]


class Oracle:
    def __init__(self, lib):
        self.lib = lib
        vp, ci, cs, cd = C.c_void_p, C.c_int, C.c_size_t, C.c_double
        lib.oracle_knn_hamming.argtypes = [vp, ci, cs, vp, ci, cs, ci, ci, vp, vp]
        lib.oracle_knn_l2sq_f32.argtypes = [vp, ci, cs, vp, ci, cs, ci, ci, vp, vp]
        lib.oracle_ratio_filter_i32.argtypes = [vp, vp, ci, ci, vp]
        lib.oracle_ratio_filter_f32.argtypes = [vp, vp, ci, ci, vp]
        lib.oracle_get_matches_linear.argtypes = [ci, ci, vp, ci, vp, ci, ci, ci, ci, vp, C.POINTER(ci)]

    # ---- matching ----
    def knn_hamming(self, q, t, k=2):
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        idx = np.empty((q.shape[0], k), np.int32)
        dist = np.empty((q.shape[0], k), np.int32)
        rc = self.lib.oracle_knn_hamming(q.ctypes.data, q.shape[0], q.strides[0], t.ctypes.data, t.shape[0],
                                         t.strides[0], q.shape[1], k, idx.ctypes.data, dist.ctypes.data)
        assert rc == 0
        return idx, dist

    def knn_l2sq(self, q, t, k=2):
        q = np.ascontiguousarray(q, np.float32)
        t = np.ascontiguousarray(t, np.float32)
        idx = np.empty((q.shape[0], k), np.int32)
        dist = np.empty((q.shape[0], k), np.float32)
        rc = self.lib.oracle_knn_l2sq_f32(q.ctypes.data, q.shape[0], q.shape[1], t.ctypes.data, t.shape[0],
                                          t.shape[1], q.shape[1], k, idx.ctypes.data, dist.ctypes.data)
        assert rc == 0
        return idx, dist

    def ratio_filter(self, idx, dist):
        idx = np.ascontiguousarray(idx, np.int32)
        nq, k = idx.shape
        out = np.empty(nq, DMATCH)
        if dist.dtype == np.float32:
            n = self.lib.oracle_ratio_filter_f32(idx.ctypes.data, np.ascontiguousarray(dist).ctypes.data, nq, k,
                                                 out.ctypes.data)
        else:
            d = np.ascontiguousarray(dist, np.int32)
            n = self.lib.oracle_ratio_filter_i32(idx.ctypes.data, d.ctypes.data, nq, k, out.ctypes.data)
        return out[:n].copy()

    def get_matches_linear(self, n_kp1, n_kp2, d1, d2, ratio_test=True):
        d1 = np.ascontiguousarray(d1)
        d2 = np.ascontiguousarray(d2)
        desc_type = {np.dtype(np.uint8): 0, np.dtype(np.float32): 5}.get(d1.dtype, 7)
        out = np.empty(max(d1.shape[0], 1), DMATCH)
        n = C.c_int(0)
        rc = self.lib.oracle_get_matches_linear(n_kp1, n_kp2, d1.ctypes.data, d1.shape[0], d2.ctypes.data,
                                                d2.shape[0], d1.shape[1], desc_type, int(ratio_test),
                                                out.ctypes.data, C.byref(n))
        return rc, out[: n.value].copy()


_cached = None


def load() -> Oracle:
    global _cached
    if _cached is None:
        if not os.path.exists(LIB):
            subprocess.run(["make", "-C", ORACLE_DIR], check=True, stdout=subprocess.DEVNULL)
        _cached = Oracle(C.CDLL(LIB))
    return _cached


def ref_tool(name):
    """Path of a checker built from /root/reference (oracle/_ref/<name>), or None when absent."""
    p = os.path.join(ORACLE_DIR, "_ref", name)
    return p if os.path.exists(p) else None


def numpy_knn_hamming(q, t, k=2):
    """Independent brute force: popcount table over xor, lexicographic (dist, idx) selection."""
    lut = np.array([bin(i).count("1") for i in range(256)], np.int32)
    idx = np.empty((q.shape[0], k), np.int32)
    dist = np.empty((q.shape[0], k), np.int32)
    for i in range(q.shape[0]):
        d = lut[np.bitwise_xor(t, q[i][None, :])].sum(axis=1)
        order = np.lexsort((np.arange(t.shape[0]), d))[:k]
        idx[i] = order
        dist[i] = d[order]
    return idx, dist
