"""Non-integer float descriptors (RootSIFT, SURF, KAZE ...) through the fp16 matrix-core candidate path + exact fp32 re-rank
(knn_l2_f16.hip) against the exact kernel and the oracle's cvflann-order loop: idx and distance BITS must be identical on every path
(reference loop: matchinglib/source/matchers.cpp:634-689)."""
import numpy as np
import pytest

import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import _lib, synth

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    c = mpa.Context(0)
    yield c


@pytest.fixture(scope="module")
def oracle():
    import oracle_lib
    return oracle_lib.load()


def _set_l2(ctx, mode):
    _lib.check(ctx.lib.mlpl_set_l2_path(ctx.handle, mode), "set_l2_path")


def rootsift(n, dim, seed, base=None):
    """L1-normalised, square-rooted gradient-histogram-like rows (unit L2 norm, elements in [0, 1], many small values)."""
    rng = np.random.default_rng(seed)
    x = rng.gamma(0.6, 1.0, size=(n, dim)) if base is None else np.abs(base + 0.5 * base.mean() * rng.gamma(0.6, 1.0, size=base.shape))
    x = np.minimum(x, 0.2 * x.sum(1, keepdims=True) / np.sqrt(dim) * 8)
    x = x / x.sum(1, keepdims=True)
    return np.sqrt(x).astype(np.float32)


def run(ctx, q, t, mode, k=2):
    _set_l2(ctx, mode)
    try:
        return mpa.knn_l2sq(q, t, k=k, ctx=ctx)
    finally:
        _set_l2(ctx, 0)


def full_scans(ctx):
    """Queries the fp16 path has re-ranked against every train row so far (cumulative counter)."""
    f = np.zeros(4, np.int32)
    _lib.check(ctx.lib.mlpl_debug_l2_flags(ctx.handle, f.ctypes.data), "debug_l2_flags")
    return int(f[2])


def same(a, b):
    return np.array_equal(a[0], b[0]) and a[1].tobytes() == b[1].tobytes()


@pytest.mark.parametrize("nq,nt,dim", [(4096, 4096, 128), (300, 1000, 128), (33, 65, 64), (100, 257, 32), (64, 64, 100), (31, 40, 16),
                                       (900, 2100, 7), (1, 2, 128), (2500, 5000, 61), (129, 4097, 128)])
def test_fp16_candidates_plus_rerank_equal_the_exact_kernel(ctx, oracle, nq, nt, dim):
    t = rootsift(nt, dim, 10 + dim)
    q = rootsift(nq, dim, 20 + dim, base=t[np.random.default_rng(nq).integers(0, nt, nq)] ** 2)  # queries near train rows, like a real pair
    if nt > 10:
        t[7] = t[3]   # duplicated train rows: exact ties, the smaller index must win
        q[0] = t[3]
    ex = run(ctx, q, t, 1)
    before = full_scans(ctx)
    f16 = run(ctx, q, t, 3)
    assert same(ex, f16)
    assert full_scans(ctx) == before      # a handful of candidates per query: nobody needed the exhaustive re-rank
    if nq * nt <= 300 * 1000:
        oi, od = oracle.knn_l2sq(q, t)
        assert np.array_equal(f16[0], oi) and f16[1].tobytes() == od.tobytes()
    assert same(run(ctx, q, t, 1, k=1), run(ctx, q, t, 3, k=1))


def test_auto_mode_hint_and_always_modes(ctx):
    """Auto mode: the first call on non-integer data runs the exact kernel and leaves the hint, the following calls take the fp16 path;
    integer-valued data in between go to the int8 path whatever the hint says.  Same bits every time."""
    t = rootsift(3000, 128, 1)
    q = rootsift(2000, 128, 2)
    ex = run(ctx, q, t, 1)
    qi, ti = synth.sift_pair(500, 900, seed=3)
    exi = run(ctx, qi, ti, 1)
    for opt in (1, 2, 0):
        ctx.set_option("l2_float_mfma", opt)
        try:
            for _ in range(3):
                assert same(run(ctx, q, t, 0), ex), opt
                ctx.synchronize()
            assert same(run(ctx, qi, ti, 0), exi), opt
            ctx.synchronize()
            assert same(run(ctx, q, t, 0), ex), opt
        finally:
            ctx.set_option("l2_float_mfma", 1)


def test_general_floats_negative_values_and_scales(ctx):
    """Signed values, rows of very different magnitude (per-row power-of-two scaling), zero rows."""
    rng = np.random.default_rng(5)
    for dim in (128, 64, 33):
        q = rng.normal(size=(700, dim)).astype(np.float32)
        t = rng.normal(size=(1500, dim)).astype(np.float32)
        t[:300] *= 1e-3
        t[300:600] *= 1e4
        q[:100] *= 1e4
        q[100:200] *= 1e-3
        t[10] = 0.0
        q[5] = 0.0
        assert same(run(ctx, q, t, 1), run(ctx, q, t, 3)), dim


def test_many_near_ties_overflow_the_candidate_list(ctx):
    """Copies of a train row plus perturbations far below the resolution of the approximate distances: more than 64 candidates per
    query -> those queries are re-ranked against every train row; large common offset -> distances tiny against the norms."""
    rng = np.random.default_rng(8)
    base = rng.normal(size=(1, 128)).astype(np.float32) + 30.0          # |x|^2 ~ 1e5, distances ~ 1e-2 .. 1: cancellation in |q|^2+|t|^2-2qt
    t = np.repeat(base, 600, axis=0) + (1e-3 * rng.normal(size=(600, 128))).astype(np.float32)
    t[100:140] = t[50]
    q = np.repeat(base, 300, axis=0) + (1e-3 * rng.normal(size=(300, 128))).astype(np.float32)
    q[0] = t[50]
    ex = run(ctx, q, t, 1)
    assert ex[0][0, 0] == 50 and ex[0][0, 1] == 100 and ex[1][0, 0] == 0.0
    before = full_scans(ctx)
    assert same(ex, run(ctx, q, t, 3))
    assert full_scans(ctx) - before == len(q)   # every query sees all 600 rows inside its window
    ctx.set_option("l2_float_mfma", 2)
    try:
        assert same(ex, run(ctx, q, t, 0))
    finally:
        ctx.set_option("l2_float_mfma", 1)


def test_distances_below_the_resolution_of_the_fp16_products(ctx):
    """Train rows on a fine ladder of distances around a large common offset: neighbouring rungs differ by a tenth of the error window, so
    the approximate distances order the first dozens of rows at random and every query has ~20 candidates.  The true top-2 must still be
    among them (the window is a bound, not an estimate) -- and with < 64 candidates nobody falls back to the exhaustive re-rank."""
    rng = np.random.default_rng(21)
    dim = 128
    base = (30.0 + rng.normal(size=dim)).astype(np.float64)
    norm2 = float(base @ base)
    step = 0.1 * 1.0e-3 * 2 * norm2
    u = rng.normal(size=(600, dim))
    u /= np.linalg.norm(u, axis=1, keepdims=True)
    t = (base + np.sqrt(step * np.arange(1, 601))[:, None] * u).astype(np.float32)
    t = t[rng.permutation(600)]
    q = (base + 0.3 * rng.normal(size=(400, dim))).astype(np.float32)
    ex = run(ctx, q, t, 1)
    before = full_scans(ctx)
    assert same(ex, run(ctx, q, t, 3))
    assert full_scans(ctx) == before
    # the same ladder with signs mixed (cancellation inside the dot products) and at a very different scale
    sgn = np.where(rng.random(dim) < 0.5, -1.0, 1.0).astype(np.float32)
    for scale in (1.0, 3e-4, 2e5):
        q2, t2 = (q * sgn * scale).astype(np.float32), (t * sgn * scale).astype(np.float32)
        assert same(run(ctx, q2, t2, 1), run(ctx, q2, t2, 3)), scale


def test_integer_valued_data_through_the_fp16_path(ctx):
    q, t = synth.sift_pair(700, 1300, seed=12)
    assert same(run(ctx, q, t, 1), run(ctx, q, t, 3))


def test_rows_outside_the_range_are_scanned_exhaustively(ctx):
    t = rootsift(500, 128, 31)
    q = rootsift(200, 128, 32)
    for bad in (np.inf, 3e16, 1e-20):
        t2 = t.copy()
        if bad == 1e-20:
            t2[17] = 0.0
            t2[17, 3] = bad          # largest element of the row below 1e-12
        else:
            t2[17, 3] = bad
        ex = run(ctx, q, t2, 1)
        ctx.set_option("l2_float_mfma", 2)
        try:
            assert same(ex, run(ctx, q, t2, 0)), bad
        finally:
            ctx.set_option("l2_float_mfma", 1)
        with pytest.raises(RuntimeError):
            run(ctx, q, t2, 3)
    big = np.random.default_rng(1).random((40, 200)).astype(np.float32)  # dim > 128: not on this path
    with pytest.raises(RuntimeError):
        run(ctx, big, big, 3)
    ctx.set_option("l2_float_mfma", 2)
    try:
        assert same(run(ctx, big, big, 1), run(ctx, big, big, 0))
    finally:
        ctx.set_option("l2_float_mfma", 1)


def test_device_batched_float_descriptors(ctx):
    import torch
    B, nq, nt, dim = 3, 700, 900, 128
    q = np.stack([rootsift(nq, dim, 50 + b) for b in range(B)])
    t = np.stack([rootsift(nt, dim, 60 + b) for b in range(B)])
    dq, dt = torch.from_numpy(q).cuda(), torch.from_numpy(t).cuda()
    idx = torch.empty((B, nq, 2), dtype=torch.int32, device="cuda")
    dist = torch.empty((B, nq, 2), dtype=torch.float32, device="cuda")
    outs = []
    for mode in (1, 3):
        _set_l2(ctx, mode)
        try:
            _lib.check(ctx.lib.mlpl_knn2_l2sq_f32_dev(ctx.handle, dq.data_ptr(), nq, dim, nq * dim, dt.data_ptr(), nt, dim, nt * dim, dim, 2, B,
                                                      idx.data_ptr(), dist.data_ptr(), None), "knn2_l2sq_f32_dev")
            torch.cuda.synchronize()
        finally:
            _set_l2(ctx, 0)
        outs.append((idx.cpu().numpy().copy(), dist.cpu().numpy().copy()))
    assert same(outs[0], outs[1])


def test_get_matches_rootsift_c4_shape(ctx, oracle):
    """BASELINE config 4's shape with non-integer descriptors through the host API (auto mode, second call = fp16 path)."""
    t = rootsift(4096, 128, 71)
    q = rootsift(4096, 128, 72, base=t ** 2)
    sub = np.arange(0, 4096, 37)
    rc_o, mo = oracle.get_matches_linear(len(sub), 4096, q[sub], t)
    for _ in range(2):
        err, m = mpa.getMatches([None] * 4096, [None] * 4096, q, t, matcher_name="LINEAR", ctx=ctx)
        ctx.synchronize()
        assert err == 0
    ms = m[np.isin(m["queryIdx"], sub)]
    assert len(ms) == len(mo)
    assert np.array_equal(ms["trainIdx"], mo["trainIdx"]) and ms["distance"].tobytes() == mo["distance"].tobytes()


def test_property_random_shapes_and_value_distributions(ctx):
    """hypothesis: any shape up to 300 x 400 x 128 and a handful of value distributions (normal, heavy-tailed, sparse, quantised to a few
    levels so that exact ties abound, rows copied between the sets): the fp16 candidate path returns the exact kernel's bits."""
    from hypothesis import given, settings, strategies as st, HealthCheck

    @settings(max_examples=80, deadline=None, suppress_health_check=list(HealthCheck))
    @given(nq=st.integers(1, 300), nt=st.integers(2, 400), dim=st.integers(1, 128), kind=st.integers(0, 4), seed=st.integers(0, 2 ** 31 - 1),
           k=st.integers(1, 2))
    def check(nq, nt, dim, kind, seed, k):
        rng = np.random.default_rng(seed)
        if kind == 0:
            q, t = rng.normal(size=(nq, dim)), rng.normal(size=(nt, dim))
        elif kind == 1:
            q, t = rng.standard_cauchy(size=(nq, dim)) * 1e-2, rng.standard_cauchy(size=(nt, dim)) * 1e-2
        elif kind == 2:
            q, t = rng.normal(size=(nq, dim)) * (rng.random((nq, dim)) < 0.1), rng.normal(size=(nt, dim)) * (rng.random((nt, dim)) < 0.1)
        elif kind == 3:
            q, t = rng.integers(-2, 3, (nq, dim)) * 0.25, rng.integers(-2, 3, (nt, dim)) * 0.25
        else:
            t = rng.normal(size=(nt, dim))
            q = t[rng.integers(0, nt, nq)] + 1e-3 * rng.normal(size=(nq, dim)) * (rng.random((nq, 1)) < 0.5)
        q, t = q.astype(np.float32), t.astype(np.float32)
        assert same(run(ctx, q, t, 1, k=k), run(ctx, q, t, 3, k=k))

    check()
