"""CPU-only: the robust-pose oracle (oracle/pose_oracle.c) against libc, numpy, analytic ground truth and the golden
E-sets produced by the reference's vendored OpenGV fivept_nister (tests/golden/make_golden.py)."""
import ctypes
import os

import numpy as np
import pytest

from matchinglib_poselib_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def essential_from(R, t):
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    E = tx @ R
    return E / np.linalg.norm(E)


def e_dist(a, b):
    return min(np.linalg.norm(a - b), np.linalg.norm(a + b))


def test_glibc_rand_stream(oracle):
    libc = ctypes.CDLL("libc.so.6")
    for seed in (0, 1, 12345, 20260103, 4294967295):
        libc.srand(ctypes.c_uint(seed))
        ref = [libc.rand() for _ in range(2000)]
        assert oracle.rand_stream(seed, 2000).tolist() == ref


def test_sampler_consumes_one_draw_per_pick(oracle):
    """checkSubset is always true in the reference (modelest.cpp:649 `return i >= i1`), so the sample table is just
    rand() % n with duplicate picks redrawn."""
    p1, p2, *_ = synth.pose_scene(50, seed=3)
    tab = oracle.sample_table(777, p1, p2, 200)
    stream = iter(oracle.rand_stream(777, 5000) % 50)
    for row in tab:
        picked = []
        while len(picked) < 5:
            v = int(next(stream))
            if v in picked:
                continue
            picked.append(v)
        assert row.tolist() == picked


def test_jacobi_svd(oracle):
    rng = np.random.default_rng(0)
    for shape in [(5, 9), (40, 9), (3, 3), (4, 4)]:
        A = rng.normal(size=shape)
        w, V = oracle.jacobi_svd(A)
        k = min(shape)
        assert np.allclose(w[:k], np.linalg.svd(A, compute_uv=False), atol=1e-12)
        assert np.abs(V.T @ V - np.eye(shape[1])).max() < 1e-13
        assert np.allclose(np.linalg.norm(A @ V, axis=0), w, atol=1e-12)


def test_solve_poly(oracle):
    rng = np.random.default_rng(1)
    for _ in range(5):
        c = rng.normal(size=11)
        r = oracle.solve_poly(c)
        rn = np.roots(c[::-1])
        assert max(np.abs(rn - x).min() for x in r) < 1e-10


def test_update_num_iters(oracle):
    f = oracle.lib.oracle_ransac_update_num_iters
    assert f(0.999, 0.5, 5, 1000) == 218              # log(1e-3)/log(1-0.5^5) = 217.6
    assert f(1.0, 0.5, 5, 20000) == 20000             # p = 1: no early exit
    assert f(0.999, 0.0, 5, 1000) == 0
    assert f(0.999, 1.0, 5, 1000) == 1000


def test_5pt_noise_free_contains_truth(oracle):
    p1, p2, R, t, mask, th = synth.pose_scene(400, seed=5, noise_px=0.0)
    Et = essential_from(R, t)
    inl = np.nonzero(mask)[0]
    for s in range(10):
        sel = inl[s * 5:(s + 1) * 5]
        E = oracle.run5point(p1[sel], p2[sel])
        assert 1 <= len(E) <= 10
        assert min(e_dist(e, Et) for e in E) < 1e-6  # conditioning of the minimal problem varies per sample
        x1 = np.c_[p1[sel], np.ones(5)]
        x2 = np.c_[p2[sel], np.ones(5)]
        for e in E:
            assert np.abs(np.einsum("ij,jk,ik->i", x2, e, x1)).max() < 1e-12
            assert abs(np.linalg.norm(e) - 1) < 1e-12


def test_5pt_vs_opengv_golden(oracle):
    """Pin: every self-consistent E of the reference's vendored OpenGV solver is reproduced (up to sign), and the
    solution counts agree."""
    g = np.load(os.path.join(GOLD, "fivept_opengv_120.npz"))
    checked = 0
    for s in range(len(g["pts"])):
        pts = g["pts"][s]
        Eo = oracle.run5point(pts[:, :2], pts[:, 2:])
        assert len(Eo) == g["count"][s]
        for e in g["E"][s, :g["count"][s]]:
            if np.abs(2 * e @ e.T @ e - np.trace(e @ e.T) * e).max() > 1e-12:
                continue  # OpenGV's own Sturm/Newton root was loose on this one
            assert min(e_dist(e, x) for x in Eo) < 1e-8
            checked += 1
    assert checked > 400


def test_sampson_and_inliers(oracle):
    p1, p2, R, t, mask, th = synth.pose_scene(1000, seed=8)
    E = essential_from(R, t)
    good, esum, err, m = oracle.find_inliers(p1, p2, E, th)
    x1 = np.c_[p1, np.ones(len(p1))]
    x2 = np.c_[p2, np.ones(len(p1))]
    Ex1 = x1 @ E.T
    Etx2 = x2 @ E
    ref = (np.sum(x2 * Ex1, axis=1) ** 2 / (Ex1[:, 0] ** 2 + Ex1[:, 1] ** 2 + Etx2[:, 0] ** 2 + Etx2[:, 1] ** 2))
    assert np.allclose(err, ref.astype(np.float32), rtol=1e-6)
    assert good == int((err.astype(np.float64) <= th * th).sum()) == int(m.sum())
    e64 = err.astype(np.float64)
    lanes = [np.add.reduce(e64[j::4]) for j in range(4)]  # order inside a lane does not matter at this tolerance
    assert abs(esum - ((lanes[0] + lanes[2]) + (lanes[1] + lanes[3]))) < 1e-12 * max(1.0, esum)
    assert (m.astype(bool) == mask).mean() > 0.97


@pytest.mark.parametrize("lesqu", [False, True])
def test_ransac_recovers_pose(oracle, lesqu):
    p1, p2, R, t, mask, th = synth.pose_scene(1500, seed=21)
    r = oracle.ransac_essential(p1, p2, th, confidence=0.999, max_iters=1000, lesqu=lesqu, seed=12345, trace=True)
    assert r["ok"] and r["n_inliers"] > 650 and r["iters"] < 1000
    assert int(r["mask"].sum()) == r["n_inliers"]
    # the trace is self-consistent: the stop iteration obeys niters_after
    tr = r["trace"]
    assert tr[r["iters"] - 1].niters_after <= r["iters"]
    good, Rr, tr_, Q, m = oracle.recover_pose(r["E"], p1, p2, 50.0, r["mask"])
    assert good > 600
    assert np.abs(Rr - R).max() < 2e-2 and np.abs(tr_ - t).max() < 2e-2
    assert abs(np.linalg.det(Rr) - 1) < 1e-12 and abs(np.linalg.norm(tr_) - 1) < 1e-12


def test_ransac_same_seed_same_result_and_seed_matters(oracle):
    p1, p2, R, t, mask, th = synth.pose_scene(800, seed=22)
    a = oracle.ransac_essential(p1, p2, th, seed=7)
    b = oracle.ransac_essential(p1, p2, th, seed=7)
    c = oracle.ransac_essential(p1, p2, th, seed=8)
    assert np.array_equal(a["E"], b["E"]) and a["iters"] == b["iters"]
    assert not np.array_equal(a["E"], c["E"])


def test_recover_pose_cheirality(oracle):
    p1, p2, R, t, mask, th = synth.pose_scene(300, inlier_frac=1.0, seed=30, noise_px=0.0)
    E = essential_from(R, t)
    for Ein in (E, -E, 3.7 * E):
        good, Rr, tr, Q, m = oracle.recover_pose(Ein, p1, p2, 50.0, None)
        assert good == 300
        assert np.abs(Rr - R).max() < 1e-9 and np.abs(tr - t).max() < 1e-9
        # triangulated points reproject onto the observations
        assert np.abs(Q[:, :2] / Q[:, 2:3] - p1).max() < 1e-9
        X2 = Q @ R.T + t
        assert np.abs(X2[:, :2] / X2[:, 2:3] - p2).max() < 1e-9
    # far points are cut by `dist`; the incoming mask is ANDed
    good, *_ = oracle.recover_pose(E, p1, p2, 8.0, None)
    zs = None
    good_all, _, _, Q, _ = oracle.recover_pose(E, p1, p2, 50.0, None)
    assert good == int((Q[:, 2] < 8.0).sum())
    inmask = np.zeros(300, np.uint8)
    inmask[:100] = 1
    good, _, _, _, m = oracle.recover_pose(E, p1, p2, 50.0, inmask)
    assert good == 100 and m[:100].all() and not m[100:].any()


def test_decompose_essential(oracle):
    p1, p2, R, t, *_ = synth.pose_scene(10, seed=1)
    R1, R2, tv = oracle.decompose_essential(essential_from(R, t))
    for Rx in (R1, R2):
        assert np.abs(Rx @ Rx.T - np.eye(3)).max() < 1e-12 and abs(np.linalg.det(Rx) - 1) < 1e-12
    assert min(np.abs(R1 - R).max(), np.abs(R2 - R).max()) < 1e-10
    assert min(np.abs(tv - t).max(), np.abs(tv + t).max()) < 1e-10


def test_lmeds_recovers_pose(oracle):
    """runLMeDS (modelest.cpp:483-564): 134 samples at confidence 0.999, no threshold argument."""
    p1, p2, R, t, mask, th = synth.pose_scene(1500, inlier_frac=0.7, seed=23)
    r = oracle.lmeds_essential(p1, p2, seed=12345)
    assert r["ok"] and int(r["mask"].sum()) == r["n_inliers"]
    assert (r["mask"].astype(bool) == mask).mean() > 0.97
    good, Rr, tr_, Q, m = oracle.recover_pose(r["E"], p1, p2, 50.0, r["mask"])
    assert np.abs(Rr - R).max() < 2e-2 and np.abs(tr_ - t).max() < 2e-2
    # same seed, same answer; the sample stream is RANSAC's
    r2 = oracle.lmeds_essential(p1, p2, seed=12345)
    assert np.array_equal(r["E"], r2["E"]) and r["min_median"] == r2["min_median"]
    # breakdown point: with fewer than half inliers the median is an outlier error and the mask degenerates
    p1b, p2b, *_ , maskb, thb = synth.pose_scene(1500, inlier_frac=0.3, seed=23)
    rb = oracle.lmeds_essential(p1b, p2b, seed=12345)
    assert rb["min_median"] > 100 * r["min_median"]


def test_coefficient_arithmetic_equals_the_references_own_expressions(oracle):
    """The oracle derives the 10 x 20 constraint matrix and the degree-10 determinant polynomial by polynomial arithmetic; the
    golden values come from the reference's own generated expressions (getCoeffMat, five-point.cpp:603-824, and c[0..10],
    :418-428), compiled where they lie into oracle/_ref/libfivept_ref.so.  Equal to rounding (1e-14 relative)."""
    import ctypes as C
    g = np.load(os.path.join(GOLD, "coeff_ref.npz"))
    for EE, Aref, b, cref in zip(g["EE"], g["A"], g["b"], g["c"]):
        A, c = np.zeros(200), np.zeros(11)
        EE = np.ascontiguousarray(EE)
        b = np.ascontiguousarray(b)
        oracle.lib.oracle_coeff_matrix(C.c_void_p(EE.ctypes.data), C.c_void_p(A.ctypes.data))
        oracle.lib.oracle_detpoly(C.c_void_p(b.ctypes.data), C.c_void_p(c.ctypes.data))
        assert np.abs(A.reshape(10, 20) - Aref).max() <= 1e-14 * np.abs(Aref).max()
        assert np.abs(c - cref).max() <= 1e-14 * np.abs(cref).max()
