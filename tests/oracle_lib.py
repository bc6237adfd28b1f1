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
        self._bind_pose()

    def _bind_pose(self):
        lib = self.lib
        vp, ci, cd, cu = C.c_void_p, C.c_int, C.c_double, C.c_uint
        lib.oracle_srand.argtypes = [C.POINTER(GlibcRand), cu]
        lib.oracle_rand.argtypes = [C.POINTER(GlibcRand)]
        lib.oracle_rand.restype = ci
        lib.oracle_get_subset.argtypes = [C.POINTER(GlibcRand), vp, vp, ci, ci, vp]
        lib.oracle_run5point.argtypes = [vp, vp, ci, vp]
        lib.oracle_sampson_err.argtypes = [vp, vp, ci, vp, vp]
        lib.oracle_find_inliers.argtypes = [vp, vp, ci, vp, cd, vp, vp, C.POINTER(cd)]
        lib.oracle_ransac_update_num_iters.argtypes = [cd, cd, ci, ci]
        lib.oracle_ransac_essential.argtypes = [vp, vp, ci, cd, cd, ci, ci, cu, vp, vp, C.POINTER(ci), C.POINTER(ci), vp]
        lib.oracle_decompose_essential.argtypes = [vp, vp, vp, vp]
        lib.oracle_recover_pose.argtypes = [vp, vp, vp, ci, cd, vp, vp, vp, vp]
        lib.oracle_triangulate_point.argtypes = [vp, vp, vp, vp, vp]
        lib.oracle_jacobi_svd.argtypes = [vp, ci, ci, vp, vp]
        lib.oracle_solve_poly.argtypes = [vp, ci, vp, ci]

    # ---- pose ----
    def rand_stream(self, seed, count):
        st = GlibcRand()
        self.lib.oracle_srand(C.byref(st), seed)
        return np.array([self.lib.oracle_rand(C.byref(st)) for _ in range(count)], np.int64)

    def sample_table(self, seed, p1, p2, iters):
        """The first `iters` 5-point samples getSubset would draw for this seed (no early exit)."""
        p1 = np.ascontiguousarray(p1, np.float64)
        p2 = np.ascontiguousarray(p2, np.float64)
        st = GlibcRand()
        self.lib.oracle_srand(C.byref(st), seed)
        out = np.empty((iters, 5), np.int32)
        tmp = np.empty(5, np.int32)
        for i in range(iters):
            ok = self.lib.oracle_get_subset(C.byref(st), p1.ctypes.data, p2.ctypes.data, p1.shape[0], 300, tmp.ctypes.data)
            assert ok
            out[i] = tmp
        return out

    def run5point(self, q1, q2):
        q1 = np.ascontiguousarray(q1, np.float64)
        q2 = np.ascontiguousarray(q2, np.float64)
        E = np.zeros((10, 3, 3))
        n = self.lib.oracle_run5point(q1.ctypes.data, q2.ctypes.data, q1.shape[0], E.ctypes.data)
        return E[:n].copy()

    def run5point_dbg(self, q1, q2):
        """run5point plus (polynomial c[0..10] ascending, complex roots in solvePoly order, solveZ z-components)."""
        q1 = np.ascontiguousarray(q1, np.float64)
        q2 = np.ascontiguousarray(q2, np.float64)
        E = np.zeros((10, 3, 3))
        c, roots, z = np.zeros(11), np.zeros((10, 2)), np.zeros(10)
        self.lib.oracle_run5point_dbg.argtypes = [C.c_void_p, C.c_void_p, C.c_int] + [C.c_void_p] * 4
        n = self.lib.oracle_run5point_dbg(q1.ctypes.data, q2.ctypes.data, q1.shape[0], E.ctypes.data, c.ctypes.data,
                                          roots.ctypes.data, z.ctypes.data)
        return E[:n].copy(), c, roots[:, 0] + 1j * roots[:, 1], z

    def find_inliers(self, p1, p2, E, thresh):
        p1 = np.ascontiguousarray(p1, np.float64)
        p2 = np.ascontiguousarray(p2, np.float64)
        E = np.ascontiguousarray(E, np.float64)
        n = p1.shape[0]
        err = np.empty(n, np.float32)
        mask = np.empty(n, np.uint8)
        s = C.c_double(0)
        good = self.lib.oracle_find_inliers(p1.ctypes.data, p2.ctypes.data, n, E.ctypes.data, thresh, err.ctypes.data,
                                            mask.ctypes.data, C.byref(s))
        return good, s.value, err, mask

    def sampson_err(self, p1, p2, E):
        return self.find_inliers(p1, p2, E, 1.0)[2]

    def ransac_essential(self, p1, p2, thresh, confidence=0.999, max_iters=1000, lesqu=False, seed=12345, trace=False):
        p1 = np.ascontiguousarray(p1, np.float64)
        p2 = np.ascontiguousarray(p2, np.float64)
        n = p1.shape[0]
        E = np.zeros((3, 3))
        mask = np.zeros(n, np.uint8)
        ninl, iters = C.c_int(0), C.c_int(0)
        tr = (TraceRec * max_iters)() if trace else None
        ok = self.lib.oracle_ransac_essential(p1.ctypes.data, p2.ctypes.data, n, thresh, confidence, max_iters,
                                              int(lesqu), seed, E.ctypes.data, mask.ctypes.data, C.byref(ninl),
                                              C.byref(iters), C.addressof(tr) if trace else None)
        return dict(ok=bool(ok), E=E, mask=mask, n_inliers=ninl.value, iters=iters.value, trace=tr)

    def lmeds_essential(self, p1, p2, confidence=0.999, max_iters=2000, seed=12345):
        p1 = np.ascontiguousarray(p1, np.float64)
        p2 = np.ascontiguousarray(p2, np.float64)
        n = p1.shape[0]
        E = np.zeros((3, 3))
        mask = np.zeros(n, np.uint8)
        ninl, med = C.c_int(0), C.c_double(0)
        self.lib.oracle_lmeds_essential.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_uint, C.c_void_p,
                                                    C.c_void_p, C.POINTER(C.c_int), C.POINTER(C.c_double)]
        ok = self.lib.oracle_lmeds_essential(p1.ctypes.data, p2.ctypes.data, n, confidence, max_iters, seed, E.ctypes.data,
                                             mask.ctypes.data, C.byref(ninl), C.byref(med))
        return dict(ok=bool(ok), E=E, mask=mask, n_inliers=ninl.value, min_median=med.value)

    def recover_pose(self, E, p1, p2, dist=50.0, mask=None):
        p1 = np.ascontiguousarray(p1, np.float64)
        p2 = np.ascontiguousarray(p2, np.float64)
        E = np.ascontiguousarray(E, np.float64)
        n = p1.shape[0]
        R, t, Q = np.zeros((3, 3)), np.zeros(3), np.zeros((n, 3))
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).copy()
        good = self.lib.oracle_recover_pose(E.ctypes.data, p1.ctypes.data, p2.ctypes.data, n, dist, R.ctypes.data,
                                            t.ctypes.data, Q.ctypes.data, None if m is None else m.ctypes.data)
        return good, R, t, Q, m

    def recover_pose_translation(self, Et, p1, p2, dist=50.0, mask=None):
        p1 = np.ascontiguousarray(p1, np.float64)
        p2 = np.ascontiguousarray(p2, np.float64)
        Et = np.ascontiguousarray(Et, np.float64)
        n = p1.shape[0]
        R, t, Q = np.zeros((3, 3)), np.zeros(3), np.zeros((n, 3))
        m = None if mask is None else np.ascontiguousarray(mask, np.uint8).copy()
        self.lib.oracle_recover_pose_translation.argtypes = [C.c_void_p] * 3 + [C.c_int, C.c_double] + [C.c_void_p] * 4
        good = self.lib.oracle_recover_pose_translation(Et.ctypes.data, p1.ctypes.data, p2.ctypes.data, n, dist, R.ctypes.data,
                                                        t.ctypes.data, Q.ctypes.data, None if m is None else m.ctypes.data)
        return good, R, t, Q, m

    def decompose_essential(self, E):
        E = np.ascontiguousarray(E, np.float64)
        R1, R2, t = np.zeros((3, 3)), np.zeros((3, 3)), np.zeros(3)
        self.lib.oracle_decompose_essential(E.ctypes.data, R1.ctypes.data, R2.ctypes.data, t.ctypes.data)
        return R1, R2, t

    def img_to_cam(self, pts, K4):
        a = np.ascontiguousarray(pts, np.float32).copy()
        k = np.ascontiguousarray(K4, np.float64)
        self.lib.oracle_img_to_cam.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self.lib.oracle_img_to_cam(a.ctypes.data, a.shape[0], k.ctypes.data)
        return a

    def remove_lens_dist(self, p1, p2, d1, d2):
        a = np.ascontiguousarray(p1, np.float32).copy()
        b = np.ascontiguousarray(p2, np.float32).copy()
        d1 = np.ascontiguousarray(d1, np.float64)
        d2 = np.ascontiguousarray(d2, np.float64)
        n_out = C.c_int(0)
        self.lib.oracle_remove_lens_dist.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.POINTER(C.c_int)]
        ok = self.lib.oracle_remove_lens_dist(a.ctypes.data, b.ctypes.data, a.shape[0], d1.ctypes.data, d2.ctypes.data,
                                              C.byref(n_out))
        return bool(ok), a[: n_out.value], b[: n_out.value]

    def get_inliers_strict(self, p1, p2, E, th2):
        p1 = np.ascontiguousarray(p1, np.float64)
        p2 = np.ascontiguousarray(p2, np.float64)
        E = np.ascontiguousarray(E, np.float64)
        n = p1.shape[0]
        err, mask = np.zeros(n), np.zeros(n, np.uint8)
        self.lib.oracle_get_inliers_strict.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
        cnt = self.lib.oracle_get_inliers_strict(p1.ctypes.data, p2.ctypes.data, n, E.ctypes.data, th2, err.ctypes.data,
                                                 mask.ctypes.data)
        return cnt, mask, err

    def jacobi_svd(self, A):
        A = np.ascontiguousarray(A, np.float64)
        m, n = A.shape
        w, V = np.zeros(n), np.zeros((n, n))
        self.lib.oracle_jacobi_svd(A.ctypes.data, m, n, w.ctypes.data, V.ctypes.data)
        return w, V

    def solve_poly(self, coeffs):
        c = np.ascontiguousarray(coeffs, np.float64)
        deg = len(c) - 1
        r = np.zeros((deg, 2))
        self.lib.oracle_solve_poly(c.ctypes.data, deg, r.ctypes.data, 0)
        return r[:, 0] + 1j * r[:, 1]

    # ---- ARRSAC (arrsac_oracle.cpp) ----
    def arrsac_essential(self, p1, p2, thresh, refine=True, rng_state=None):
        """runARRSAC as findEssentialMat drives it; rng_state = uint64[2] (updated in place), None = a fresh process."""
        p1 = np.ascontiguousarray(p1, np.float64)
        p2 = np.ascontiguousarray(p2, np.float64)
        n = p1.shape[0]
        st = np.array([0xFFFFFFFF, 0xFFFFFFFF], np.uint64) if rng_state is None else rng_state
        E, mask, ninl, stats = np.zeros((3, 3)), np.zeros(n, np.uint8), C.c_int(0), np.zeros(8, np.int64)
        f = self.lib.oracle_arrsac_essential
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p]
        ok = f(p1.ctypes.data, p2.ctypes.data, n, float(thresh), 1 if refine else 0, st.ctypes.data, E.ctypes.data, mask.ctypes.data,
               C.addressof(ninl), stats.ctypes.data)
        return dict(ok=bool(ok), E=E, mask=mask, n_inliers=ninl.value, stats=stats, rng_state=st)

    # ---- USAC (usac_oracle.cpp) ----
    def usac_essential(self, p1, p2, th, seed, refine=0, sorted_idx=None, max_hyp=50000, conf=0.99, prosac_beta=0.09, sprt_delta=0.05,
                       sprt_epsilon=0.15, sprt_ms=8.5, sprt_tm=2314.0, event_cap=0):
        """estimateEssentialMatUsac (POSE_NISTER, no degeneracy tests) with the seed exposed; same outputs as tests/usac_ref_tool.run."""
        p1 = np.ascontiguousarray(p1, np.float64)
        p2 = np.ascontiguousarray(p2, np.float64)
        n = p1.shape[0]
        E, flags, res, nev = np.zeros(9), np.zeros(n, np.uint8), np.zeros(12), C.c_int(0)
        ev = np.zeros((max(event_cap, 1), 16))
        si = None if sorted_idx is None else np.ascontiguousarray(sorted_idx, np.uint32)
        f = self.lib.oracle_usac_essential
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_uint, C.c_int, C.c_void_p, C.c_int] + [C.c_double] * 6 + \
                     [C.c_void_p] * 4 + [C.c_int, C.c_void_p]
        ok = f(p1.ctypes.data, p2.ctypes.data, n, float(th), int(seed), int(refine), None if si is None else si.ctypes.data, int(max_hyp),
               float(conf), float(prosac_beta), float(sprt_delta), float(sprt_epsilon), float(sprt_ms), float(sprt_tm), E.ctypes.data,
               flags.ctypes.data, res.ctypes.data, ev.ctypes.data if event_cap else None, int(event_cap), C.addressof(nev))
        return dict(ok=bool(ok), E=E, flags=flags, final=res, events=ev[:min(nev.value, event_cap)], n_events=nev.value)

    def usac_essential_degen(self, p1, p2, th, seed, check_degeneracy=1, th_pixels=0.8, focal_length=800.0, refine=0, sorted_idx=None,
                             max_hyp=50000, conf=0.99, prosac_beta=0.09, sprt_delta=0.05, sprt_epsilon=0.15, sprt_ms=8.5, sprt_tm=2314.0,
                             event_cap=0):
        """usac_essential with the degeneracy handling of DEGEN_USAC_INTERNAL; adds degen = [1, inliers of the rotation, of "no motion",
        type], R_degen, flags_rot, flags_nomot (same keys as pose.usac_essential and tests/usac_ref_tool.run)."""
        p1 = np.ascontiguousarray(p1, np.float64)
        p2 = np.ascontiguousarray(p2, np.float64)
        n = p1.shape[0]
        E, flags, res, nev = np.zeros(9), np.zeros(n, np.uint8), np.zeros(12), C.c_int(0)
        info, fr, fn = np.zeros(16), np.zeros(n, np.uint8), np.zeros(n, np.uint8)
        ev = np.zeros((max(event_cap, 1), 16))
        si = None if sorted_idx is None else np.ascontiguousarray(sorted_idx, np.uint32)
        f = self.lib.oracle_usac_essential_degen
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_double, C.c_uint, C.c_int, C.c_void_p, C.c_int] + [C.c_double] * 6 + \
                     [C.c_int, C.c_double, C.c_double] + [C.c_void_p] * 4 + [C.c_int, C.c_void_p] + [C.c_void_p] * 3
        ok = f(p1.ctypes.data, p2.ctypes.data, n, float(th), int(seed), int(refine), None if si is None else si.ctypes.data, int(max_hyp),
               float(conf), float(prosac_beta), float(sprt_delta), float(sprt_epsilon), float(sprt_ms), float(sprt_tm), int(check_degeneracy),
               float(th_pixels), float(focal_length), E.ctypes.data, flags.ctypes.data, res.ctypes.data,
               ev.ctypes.data if event_cap else None, int(event_cap), C.addressof(nev), info.ctypes.data, fr.ctypes.data, fn.ctypes.data)
        return dict(ok=bool(ok), E=E, flags=flags, final=res, events=ev[:min(nev.value, event_cap)], n_events=nev.value,
                    degen=info[:4].copy(), R_degen=info[4:13].copy(), flags_rot=fr, flags_nomot=fn)

    def eigen_svd3(self, M):
        M = np.ascontiguousarray(M, np.float64)
        sv, U, V = np.zeros(3), np.zeros((3, 3)), np.zeros((3, 3))
        self.lib.oracle_eigen_svd3.argtypes = [C.c_void_p] * 4
        self.lib.oracle_eigen_svd3(M.ctypes.data, sv.ctypes.data, U.ctypes.data, V.ctypes.data)
        return sv, U, V

    def valid_model(self, q1, q2, E):
        q1, q2, E = (np.ascontiguousarray(a, np.float64) for a in (q1, q2, E))
        self.lib.oracle_valid_model.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        return bool(self.lib.oracle_valid_model(q1.ctypes.data, q2.ctypes.data, len(q1), E.ctypes.data))

    def cv_fm_8point(self, q1, q2):
        q1, q2 = (np.ascontiguousarray(a, np.float64) for a in (q1, q2))
        F = np.zeros((3, 3))
        self.lib.oracle_cv_fm_8point.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p]
        ok = self.lib.oracle_cv_fm_8point(q1.ctypes.data, q2.ctypes.data, len(q1), F.ctypes.data)
        return bool(ok), F

    def cv_rng_stream(self, state, count):
        st = np.array([state], np.uint64)
        out = np.zeros(count, np.uint32)
        self.lib.oracle_cv_rng_stream.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self.lib.oracle_cv_rng_stream(st.ctypes.data, count, out.ctypes.data)
        return out, int(st[0])

    def robust_essential_refine(self, p1, p2, E_init, th):
        p1, p2, E_init = (np.ascontiguousarray(a, np.float64) for a in (p1, p2, E_init))
        E, err = np.zeros((3, 3)), np.zeros(2)
        f = self.lib.oracle_robust_essential_refine
        f.argtypes = [C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_double, C.c_void_p, C.c_void_p]
        it = f(p1.ctypes.data, p2.ctypes.data, len(p1), E_init.ctypes.data, float(th), E.ctypes.data, err.ctypes.data)
        return it, E, err

    def std_sort_desc(self, score):
        score = np.ascontiguousarray(score, np.float64)
        perm = np.zeros(len(score), np.int32)
        self.lib.oracle_std_sort_desc.argtypes = [C.c_void_p, C.c_int, C.c_void_p]
        self.lib.oracle_std_sort_desc(score.ctypes.data, len(score), perm.ctypes.data)
        return perm

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

    def knn_hamming_fast(self, q, t, threads=0):
        q = np.ascontiguousarray(q, np.uint8)
        t = np.ascontiguousarray(t, np.uint8)
        idx = np.empty((q.shape[0], 2), np.int32)
        dist = np.empty((q.shape[0], 2), np.int32)
        self.lib.oracle_knn_hamming_fast.argtypes = [C.c_void_p, C.c_int, C.c_size_t, C.c_void_p, C.c_int, C.c_size_t, C.c_int,
                                                     C.c_void_p, C.c_void_p, C.c_int]
        used = self.lib.oracle_knn_hamming_fast(q.ctypes.data, q.shape[0], q.strides[0], t.ctypes.data, t.shape[0], t.strides[0],
                                                q.shape[1], idx.ctypes.data, dist.ctypes.data, threads)
        assert used >= 1
        return idx, dist, used

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


def _nms(self, d1, d2, ratio_test=True):
    d1 = np.ascontiguousarray(d1)
    d2 = np.ascontiguousarray(d2)
    dt = 0 if d1.dtype == np.uint8 else 5
    out = np.empty(max(d1.shape[0], 1), DMATCH)
    n = C.c_int(0)
    self.lib.oracle_get_matches_bruteforce_nms.argtypes = [C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int,
                                                           C.c_void_p, C.POINTER(C.c_int)]
    rc = self.lib.oracle_get_matches_bruteforce_nms(d1.ctypes.data, d1.shape[0], d2.ctypes.data, d2.shape[0], d1.shape[1], dt,
                                                    int(ratio_test), out.ctypes.data, C.byref(n))
    return rc, out[: n.value].copy()


Oracle.get_matches_bruteforce_nms = _nms

_cached = None


def load() -> Oracle:
    global _cached
    if _cached is None:
        srcs = [os.path.join(ORACLE_DIR, f) for f in os.listdir(ORACLE_DIR) if f.endswith(("_oracle.c", "_oracle.cpp", ".h"))]
        if not os.path.exists(LIB) or any(os.path.getmtime(f) > os.path.getmtime(LIB) for f in srcs):
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
