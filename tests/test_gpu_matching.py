"""GPU parity tests for the matching path: HIP kernels (through the C ABI) vs the CPU oracle, bit-exact."""
import os

import numpy as np
import pytest

import matchinglib_poselib_amd as mpa
import oracle_lib
from matchinglib_poselib_amd import synth

pytestmark = pytest.mark.gpu
GOLD = os.path.join(os.path.dirname(__file__), "golden")


@pytest.mark.parametrize("name", ["hamming_257x263", "hamming_ties_96x80", "hamming_c1_2048"])
def test_hamming_golden(ctx, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    idx, dist = mpa.knn_hamming(g["q"], g["t"], ctx=ctx)
    assert np.array_equal(dist, g["dist"])
    assert np.array_equal(idx, g["idx"])
    m = mpa.ratio_compact(idx, dist, ctx=ctx)
    assert np.array_equal(m["queryIdx"], g["match_q"]) and np.array_equal(m["trainIdx"], g["match_t"])
    assert np.all(m["imgIdx"] == -1)
    assert np.array_equal(m["distance"], dist[g["match_q"], 0].astype(np.float32))


@pytest.mark.parametrize("nq,nt,nbytes,k", [
    (1, 2, 32, 2), (15, 15, 32, 2), (64, 1000, 32, 1), (300, 129, 32, 2), (1000, 5000, 32, 2),
    (77, 333, 64, 2), (50, 200, 16, 2), (40, 90, 61, 2), (33, 70, 24, 2), (20, 40, 1, 2), (10, 600, 128, 2),
])
def test_hamming_shapes_vs_oracle(ctx, oracle, nq, nt, nbytes, k):
    q, t = synth.orb_pair(nq, nt, nbytes=nbytes, seed=1000 + nq + nt + nbytes)
    idx, dist = mpa.knn_hamming(q, t, k=k, ctx=ctx)
    oi, od = oracle.knn_hamming(q, t, k=k)
    assert np.array_equal(dist, od)
    assert np.array_equal(idx, oi)


def test_hamming_strided_rows(ctx, oracle):
    # cv::Mat with step > cols (ROI of a wider matrix)
    big_q = np.random.default_rng(1).integers(0, 256, (100, 48), dtype=np.uint8)
    big_t = np.random.default_rng(2).integers(0, 256, (150, 40), dtype=np.uint8)
    q, t = big_q[:, :32], big_t[:, :32]
    idx, dist = mpa.knn_hamming(q, t, ctx=ctx)
    oi, od = oracle.knn_hamming(np.ascontiguousarray(q), np.ascontiguousarray(t))
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)


def test_hamming_heavy_ties(ctx, oracle):
    # few distinct descriptors -> ties everywhere; the packed-key min must give the smaller train index
    rng = np.random.default_rng(9)
    base = rng.integers(0, 256, (5, 32), dtype=np.uint8)
    t = base[rng.integers(0, 5, 3000)]
    q = base[rng.integers(0, 5, 500)]
    idx, dist = mpa.knn_hamming(q, t, ctx=ctx)
    oi, od = oracle.knn_hamming(q, t)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    assert (dist[:, 0] == 0).all() and (dist[:, 1] == 0).all()


def test_c2_full_size_properties(ctx, oracle):
    """BASELINE config C2 (8192 x 8192 ORB-256): size-independent properties + oracle on a sample of rows."""
    q, t = synth.orb_pair(8192, 8192, seed=20260102)
    idx, dist = mpa.knn_hamming(q, t, ctx=ctx)
    lut = np.array([bin(i).count("1") for i in range(256)], np.int32)
    # (1) reported distances are the true distances of the reported indices
    for j in range(2):
        d = lut[np.bitwise_xor(q, t[idx[:, j]])].sum(axis=1)
        assert np.array_equal(d, dist[:, j])
    # (2) sorted, distinct neighbours
    assert (dist[:, 0] <= dist[:, 1]).all() and (idx[:, 0] != idx[:, 1]).all()
    tie = dist[:, 0] == dist[:, 1]
    assert (idx[tie, 0] < idx[tie, 1]).all()
    # (3) permuting the train rows permutes the indices and keeps the distances (tie-free rows)
    perm = np.random.default_rng(0).permutation(8192)
    idx_p, dist_p = mpa.knn_hamming(q, t[perm], ctx=ctx)
    assert np.array_equal(dist_p, dist)
    sub = np.arange(0, 8192, 37)
    oi, od = oracle.knn_hamming(q[sub], t)
    assert np.array_equal(idx[sub], oi) and np.array_equal(dist[sub], od)
    # (4) query == train: nearest neighbour of row i is i at distance 0
    idx_s, dist_s = mpa.knn_hamming(t, t, ctx=ctx)
    assert (dist_s[:, 0] == 0).all()
    assert np.array_equal(idx_s[:, 0], np.arange(8192)) or (dist_s[:, 1] == 0).any()


def test_ratio_compact_edge_cases(ctx, oracle):
    idx = np.array([[1, 2], [3, 4], [5, 6], [7, 8]], np.int32)
    dist = np.array([[3, 4], [0, 0], [74, 100], [75, 100]], np.int32)
    m = mpa.ratio_compact(idx, dist, ctx=ctx)
    o = oracle.ratio_filter(idx, dist)
    assert m.tobytes() == o.tobytes()
    # nothing passes
    m = mpa.ratio_compact(idx[:2], dist[:2], ctx=ctx)
    assert len(m) == 0
    # k = 1: everything is emitted
    m = mpa.ratio_compact(idx[:, :1].copy(), dist[:, :1].copy(), ctx=ctx)
    assert m["queryIdx"].tolist() == [0, 1, 2, 3]
    # long input crossing many 1024-chunks
    rng = np.random.default_rng(4)
    dist = np.sort(rng.integers(0, 256, (50000, 2)), axis=1).astype(np.int32)
    idx = rng.integers(0, 1 << 20, (50000, 2)).astype(np.int32)
    m = mpa.ratio_compact(idx, dist, ctx=ctx)
    o = oracle.ratio_filter(idx, dist)
    assert m.tobytes() == o.tobytes()


def test_get_matches_linear_u8(ctx, oracle):
    q, t = synth.orb_pair(2048, 2048, seed=20260101)
    kp1, kp2 = [None] * 2048, [None] * 2048
    err, m = mpa.getMatches(kp1, kp2, q, t, matcher_name="LINEAR", ctx=ctx)
    rc, o = oracle.get_matches_linear(2048, 2048, q, t)
    assert err == rc == 0
    assert m.tobytes() == o.tobytes()
    err, m = mpa.getMatches(kp1, kp2, q, t, matcher_name="LINEAR", ratioTest=False, ctx=ctx)
    rc, o = oracle.get_matches_linear(2048, 2048, q, t, ratio_test=False)
    assert err == rc == 0 and len(m) == 2048 and m.tobytes() == o.tobytes()


def test_get_matches_error_codes(ctx):
    q, t = synth.orb_pair(40, 50, seed=2)
    kp = lambda n: [None] * n  # noqa: E731
    assert mpa.getMatches(kp(14), kp(50), q[:14], t, matcher_name="LINEAR", ctx=ctx)[0] == -4
    assert mpa.getMatches(kp(41), kp(50), q, t, matcher_name="LINEAR", ctx=ctx)[0] == -1
    assert mpa.getMatches(kp(40), kp(50), q, t, ctx=ctx)[0] == -2           # default GMBSOF: not built here
    assert mpa.getMatches(kp(40), kp(50), q, t, matcher_name="NOPE", ctx=ctx)[0] == -2
    assert mpa.getMatches(kp(40), kp(50), q.astype(np.int16), t.astype(np.int16), matcher_name="LINEAR", ctx=ctx)[0] == -1
    with pytest.raises(ValueError):
        mpa.getMatches(kp(40), kp(50), q, t.astype(np.float32), matcher_name="LINEAR", ctx=ctx)
    # all queries identical to all trains -> d1 == 0 for everyone -> no match -> -3
    z = np.zeros((20, 32), np.uint8)
    assert mpa.getMatches(kp(20), kp(20), z, z, matcher_name="LINEAR", ctx=ctx)[0] == -3


def test_l2_exact_integer_and_fractional(ctx, oracle):
    from matchinglib_poselib_amd import _lib
    q, t = synth.sift_pair(300, 1000, seed=5)
    _lib.check(ctx.lib.mlpl_set_l2_path(ctx.handle, 1), "set_l2_path")   # exact fp32 VALU kernel
    idx, dist = mpa.knn_l2sq(q, t, ctx=ctx)
    oi, od = oracle.knn_l2sq(q, t)
    assert np.array_equal(idx, oi) and dist.tobytes() == od.tobytes()
    # non-integer descriptors: same fp32 summation order as cvflann::L2<float> -> still bit-exact
    rng = np.random.default_rng(6)
    q = rng.normal(size=(200, 128)).astype(np.float32)
    t = rng.normal(size=(700, 128)).astype(np.float32)
    idx, dist = mpa.knn_l2sq(q, t, ctx=ctx)
    oi, od = oracle.knn_l2sq(q, t)
    assert np.array_equal(idx, oi) and dist.tobytes() == od.tobytes()
    for dim in (64, 32, 30, 7, 130):
        q = rng.normal(size=(50, dim)).astype(np.float32)
        t = rng.normal(size=(90, dim)).astype(np.float32)
        idx, dist = mpa.knn_l2sq(q, t, ctx=ctx)
        oi, od = oracle.knn_l2sq(q, t)
        assert np.array_equal(idx, oi) and dist.tobytes() == od.tobytes(), dim
    _lib.check(ctx.lib.mlpl_set_l2_path(ctx.handle, 0), "set_l2_path")


def test_device_batched_match(ctx, oracle):
    import torch
    from matchinglib_poselib_amd.matching import match_hamming_device
    B = 3
    qs, ts = zip(*[synth.orb_pair(500, 700, seed=40 + b) for b in range(B)])
    q = torch.from_numpy(np.stack(qs)).cuda()
    t = torch.from_numpy(np.stack(ts)).cuda()
    out = match_hamming_device(q, t, ctx=ctx)
    ctx.synchronize()
    torch.cuda.synchronize()
    for b in range(B):
        oi, od = oracle.knn_hamming(qs[b], ts[b])
        assert np.array_equal(out["idx"][b].cpu().numpy(), oi)
        assert np.array_equal(out["dist"][b].cpu().numpy(), od)
        o = oracle.ratio_filter(oi, od)
        n = int(out["count"][b])
        assert n == len(o)
        m = out["matches"][b, :n].cpu().numpy()
        assert np.array_equal(m[:, 0], o["queryIdx"]) and np.array_equal(m[:, 1], o["trainIdx"])
        assert np.array_equal(m[:, 3].view(np.float32), o["distance"])


def _set_l2(ctx, mode):
    from matchinglib_poselib_amd import _lib
    _lib.check(ctx.lib.mlpl_set_l2_path(ctx.handle, mode), "set_l2_path")


@pytest.mark.parametrize("nq,nt,dim", [(4096, 4096, 128), (300, 1000, 128), (33, 65, 64), (100, 257, 32), (64, 64, 100),
                                       (50, 700, 256), (31, 40, 16)])
def test_l2_mfma_integer_descriptors_bit_exact(ctx, oracle, nq, nt, dim):
    """int8 matrix-core distance-GEMM path on integer-valued 0..255 descriptors (OpenCV SIFT layout): bit-exact vs cvflann order."""
    q, t = synth.sift_pair(nq, nt, dim=dim, seed=900 + nq + dim)
    if nt > 10:
        t[7] = t[3]            # duplicated train rows -> exact distance ties -> smaller index must win
        q[0] = t[3]
    _set_l2(ctx, 2)            # force MFMA
    try:
        idx, dist = mpa.knn_l2sq(q, t, ctx=ctx)
    finally:
        _set_l2(ctx, 0)
    if nq * nt <= 300 * 1000:
        oi, od = oracle.knn_l2sq(q, t)
    else:
        sub = np.arange(0, nq, 41)
        oi, od = oracle.knn_l2sq(q[sub], t)
        idx, dist = idx[sub], dist[sub]
    assert np.array_equal(idx, oi)
    assert dist.tobytes() == od.tobytes()


@pytest.mark.parametrize("waves", [4, 8])
@pytest.mark.parametrize("nq,nt,dim", [(16384, 8192, 128), (5000, 9000, 64), (777, 40000, 128), (3000, 3000, 255), (2500, 2500, 129), (900, 900, 7)])
def test_l2_matrix_core_many_groups_per_split_and_extreme_values(ctx, oracle, nq, nt, dim, waves):
    """Sizes at which a workgroup walks several groups of train tiles (the 64-bit fold between groups), both workgroup shapes of the
    forced mode and the fused auto-path kernel; rows of all 0 / all 255 (largest d^2, largest key), K-padding (dim % 32 != 0)."""
    q, t = synth.sift_pair(nq, nt, dim=dim, seed=4000 + nq + dim)
    t[0] = 0.0
    t[1] = 255.0
    q[0] = 255.0            # nearest: t[1] at d^2 = 0; farthest possible: t[0] at 255^2 dim
    q[1] = 0.0
    t[nt - 1] = t[nt // 2]  # a tie across splits: the smaller index must win
    q[2] = t[nt // 2]
    sub = np.unique(np.concatenate([np.arange(0, 8), np.arange(0, nq, max(1, nq // 97)), [nq - 1]]))
    oi, od = oracle.knn_l2sq(q[sub], t)
    assert od[2, 0] == 0.0 and oi[2, 0] == nt // 2 and oi[2, 1] == nt - 1
    ctx.set_option("l2_mfma_waves", waves)
    try:
        for mode in (2, 0):
            _set_l2(ctx, mode)
            idx, dist = mpa.knn_l2sq(q, t, ctx=ctx)
            assert np.array_equal(idx[sub], oi) and dist[sub].tobytes() == od.tobytes(), (mode, waves)
    finally:
        _set_l2(ctx, 0)
        ctx.set_option("l2_mfma_waves", 0)


@pytest.mark.parametrize("dim", [256, 200, 144])
def test_l2_mfma_high_magnitude_above_128_dims(ctx, oracle, dim):
    """dim > 128 with large values: |q|^2 + |t|^2 exceeds 2^24 (not a float), d^2 itself does not.  The advisor's counterexample
    (q = [201, 200, ...], t = [200, ...] -> d^2 = 1, not 0) plus near-duplicate high-valued rows whose d^2 are small integers."""
    rng = np.random.default_rng(dim)
    t = rng.integers(180, 256, size=(300, dim)).astype(np.float32)
    t[0] = 200.0
    q = t[rng.integers(0, 300, 150)].copy()
    q[0] = 200.0
    q[0, 0] = 201.0                               # d^2(q0, t0) = 1
    for i in range(1, 150):                       # 1..3 elements off by one: d^2 in {1, 2, 3}, ties broken by train index
        for c in rng.integers(0, dim, 1 + i % 3):
            q[i, c] = q[i, c] - 1.0 if q[i, c] > 200 else q[i, c] + 1.0
    oi, od = oracle.knn_l2sq(q, t)
    assert od[0, 0] == 1.0 and oi[0, 0] == 0
    for mode in (2, 0):
        _set_l2(ctx, mode)
        try:
            idx, dist = mpa.knn_l2sq(q, t, ctx=ctx)
        finally:
            _set_l2(ctx, 0)
        assert np.array_equal(idx, oi) and dist.tobytes() == od.tobytes(), mode
    err, m = mpa.getMatches([None] * 150, [None] * 300, q, t, matcher_name="LINEAR", ctx=ctx)
    rc, mo = oracle.get_matches_linear(150, 300, q, t)
    assert err == rc and m.tobytes() == mo.tobytes()


def test_l2_auto_path_selection(ctx, oracle):
    import matchinglib_poselib_amd as m
    rng = np.random.default_rng(12)
    q = rng.normal(size=(70, 128)).astype(np.float32)
    t = rng.normal(size=(200, 128)).astype(np.float32)
    idx, dist = mpa.knn_l2sq(q, t, ctx=ctx)      # auto: fractional data -> exact fp32 kernel
    oi, od = oracle.knn_l2sq(q, t)
    assert np.array_equal(idx, oi) and dist.tobytes() == od.tobytes()
    _set_l2(ctx, 2)
    try:
        with pytest.raises(m.MlplError):
            mpa.knn_l2sq(q, t, ctx=ctx)          # forcing MFMA on non-integer data is an error, never a wrong answer
    finally:
        _set_l2(ctx, 0)
    qi, ti = synth.sift_pair(200, 300, seed=77)
    ti[5, 3] = 256.0                              # one value out of range -> auto must fall back and still be exact
    idx, dist = mpa.knn_l2sq(qi, ti, ctx=ctx)
    oi, od = oracle.knn_l2sq(qi, ti)
    assert np.array_equal(idx, oi) and dist.tobytes() == od.tobytes()


def test_get_matches_linear_f32(ctx, oracle):
    q, t = synth.sift_pair(1500, 1800, seed=20260104)
    kp1, kp2 = [None] * 1500, [None] * 1800
    err, m = mpa.getMatches(kp1, kp2, q, t, matcher_name="LINEAR", ctx=ctx)
    rc, o = oracle.get_matches_linear(1500, 1800, q, t)
    assert err == rc == 0 and m.tobytes() == o.tobytes()


@pytest.mark.parametrize("nb", [32, 31, 16, 64, 3])
@pytest.mark.parametrize("ratio", [True, False])
def test_bruteforce_nms_u8(ctx, oracle, nb, ratio):
    q, t = synth.orb_pair(400, 700, nbytes=nb, seed=600 + nb)
    if nb > 3:
        t[9] = t[4]
        q[3] = t[4]          # exact tie between two train rows: without ratio test NMSLIB emits the larger id
    kp = lambda n: [None] * n  # noqa: E731
    err, m = mpa.getMatches(kp(400), kp(700), q, t, matcher_name="BRUTEFORCENMS", ratioTest=ratio, ctx=ctx)
    rc, o = oracle.get_matches_bruteforce_nms(q, t, ratio_test=ratio)
    assert err == rc == 0
    assert m.tobytes() == o.tobytes()
    if not ratio and nb > 3:
        assert m["trainIdx"][3] == 9


@pytest.mark.parametrize("dim", [128, 64, 30, 7])
@pytest.mark.parametrize("ratio", [True, False])
def test_bruteforce_nms_f32(ctx, oracle, dim, ratio):
    rng = np.random.default_rng(70 + dim)
    t = rng.normal(size=(500, dim)).astype(np.float32)
    q = rng.normal(size=(300, dim)).astype(np.float32)
    q[:100] = t[:100] + rng.normal(size=(100, dim)).astype(np.float32) * 0.05
    kp = lambda n: [None] * n  # noqa: E731
    err, m = mpa.getMatches(kp(300), kp(500), q, t, matcher_name="BRUTEFORCENMS", ratioTest=ratio, ctx=ctx)
    rc, o = oracle.get_matches_bruteforce_nms(q, t, ratio_test=ratio)
    assert err == rc == 0
    assert m.tobytes() == o.tobytes()
    if ratio:
        assert len(m) >= 90   # sqrt distances: the ratio test is applied to TRUE distances here


def test_edge_cases_small_and_empty(ctx, oracle):
    import matchinglib_poselib_amd as m
    rng = np.random.default_rng(33)
    t = rng.integers(0, 256, (2, 32), dtype=np.uint8)
    q = rng.integers(0, 256, (3, 32), dtype=np.uint8)
    idx, dist = mpa.knn_hamming(q, t, ctx=ctx)               # nt == k
    oi, od = oracle.knn_hamming(q, t)
    assert np.array_equal(idx, oi) and np.array_equal(dist, od)
    idx, dist = mpa.knn_hamming(q[:0], t, ctx=ctx)            # no queries
    assert idx.shape == (0, 2)
    with pytest.raises(m.MlplError):
        mpa.knn_hamming(q, t[:1], ctx=ctx)                    # nt < k
    # L2: k = 1, strided rows, tiny train set, both paths
    qf, tf = synth.sift_pair(70, 90, dim=128, seed=8)
    big = np.zeros((70, 160), np.float32)
    big[:, :128] = qf
    for mode in (1, 0):
        _set_l2(ctx, mode)
        idx, dist = mpa.knn_l2sq(big[:, :128], tf, k=1, ctx=ctx)
        oi, od = oracle.knn_l2sq(qf, tf, k=1)
        assert np.array_equal(idx, oi) and dist.tobytes() == od.tobytes(), mode
        idx, dist = mpa.knn_l2sq(qf, tf[:2], ctx=ctx)
        oi, od = oracle.knn_l2sq(qf, tf[:2])
        assert np.array_equal(idx, oi) and dist.tobytes() == od.tobytes(), mode
    _set_l2(ctx, 0)


def test_l2_device_batched(ctx, oracle):
    import torch
    from matchinglib_poselib_amd import _lib
    B = 3
    pairs = [synth.sift_pair(200, 333, dim=64, seed=50 + b) for b in range(B)]
    pairs[1][1][7, 3] = 0.5                                    # batch item 1 is not integer-valued -> whole call goes exact
    dq = torch.from_numpy(np.stack([p[0] for p in pairs])).cuda()
    dt = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    idx = torch.empty((B, 200, 2), dtype=torch.int32, device="cuda")
    dist = torch.empty((B, 200, 2), dtype=torch.float32, device="cuda")
    st = torch.cuda.current_stream().cuda_stream
    _lib.check(ctx.lib.mlpl_knn2_l2sq_f32_dev(ctx.handle, dq.data_ptr(), 200, 64, 200 * 64, dt.data_ptr(), 333, 64, 333 * 64, 64,
                                              2, B, idx.data_ptr(), dist.data_ptr(), st), "knn_l2_dev")
    torch.cuda.synchronize()
    for b in range(B):
        oi, od = oracle.knn_l2sq(pairs[b][0], pairs[b][1])
        assert np.array_equal(idx[b].cpu().numpy(), oi) and dist[b].cpu().numpy().tobytes() == od.tobytes(), b
    pairs[1][1][7, 3] = 1.0                                    # now all integer-valued -> MFMA path, same results
    dt = torch.from_numpy(np.stack([p[1] for p in pairs])).cuda()
    _lib.check(ctx.lib.mlpl_knn2_l2sq_f32_dev(ctx.handle, dq.data_ptr(), 200, 64, 200 * 64, dt.data_ptr(), 333, 64, 333 * 64, 64,
                                              2, B, idx.data_ptr(), dist.data_ptr(), st), "knn_l2_dev")
    torch.cuda.synchronize()
    for b in range(B):
        oi, od = oracle.knn_l2sq(pairs[b][0], pairs[b][1])
        assert np.array_equal(idx[b].cpu().numpy(), oi) and dist[b].cpu().numpy().tobytes() == od.tobytes(), b


# (hamming_variant, hamming_mfma_qt[, hamming_train01: the other encoding of the train operand, -1 = the library default])
HAMMING_VARIANTS = {
    "valu_lds_tiled": (0, 0), "valu_scalar_operand": (1, 0), "valu_one_wave_blocks": (2, 0),
    "mfma_fp4": (3, 0), "mfma_fp4_qt4": (3, 4), "mfma_fp4_qt2": (3, 2), "mfma_fp4_qt1": (3, 1),
    "mfma_fp4_train01": (3, 0, 1), "mfma_fp4_qt2_train01": (3, 2, 1), "mfma_fp4_qt1_train01": (3, 1, 1), "mfma_fp4_qt4_train01": (3, 4, 1),
    "mfma_fp4_train_pm1": (3, 0, 0), "mfma_fp4_qt2_train_pm1": (3, 2, 0),
}
HAMMING_DEFAULT = (3, 0)   # the library default: matrix-core kernel, automatic query tiles per wave
HAMMING_TRAIN01_DEFAULT = None   # the library's default encoding of the train operand (1 = {0, +1}; 0 = +-1): read from the context


def _set_hamming(ctx, cfg):
    ctx.set_option("hamming_variant", cfg[0])
    ctx.set_option("hamming_mfma_qt", cfg[1])
    global HAMMING_TRAIN01_DEFAULT
    if HAMMING_TRAIN01_DEFAULT is None:   # the library's default encoding of the train operand, read before the first change
        HAMMING_TRAIN01_DEFAULT = ctx.get_option("hamming_train01")
    ctx.set_option("hamming_train01", cfg[2] if len(cfg) > 2 else HAMMING_TRAIN01_DEFAULT)


@pytest.mark.parametrize("variant", sorted(HAMMING_VARIANTS))
def test_hamming_every_kernel_variant_bit_exact(ctx, oracle, variant):
    """Every selectable Hamming kernel gives the oracle's (distance, index) pairs bit for bit: shapes around the tile sizes
    (32-row MFMA tiles, 128-row LDS tiles), descriptor widths around the K-step (8 bytes), ties, ragged last tiles, k = 1."""
    _set_hamming(ctx, HAMMING_VARIANTS[variant])
    try:
        for nq, nt, nbytes, k in [(1, 2, 32, 2), (15, 15, 32, 2), (64, 1000, 32, 1), (300, 129, 32, 2), (1000, 5000, 32, 2),
                                  (77, 333, 64, 2), (50, 200, 16, 2), (40, 90, 61, 2), (33, 70, 24, 2), (20, 40, 1, 2),
                                  (10, 600, 128, 2), (31, 33, 8, 2), (129, 4097, 32, 2), (513, 31, 32, 2), (2048, 2048, 32, 2),
                                  (100, 9000, 64, 2), (640, 96, 9, 2)]:
            q, t = synth.orb_pair(nq, nt, nbytes=nbytes, seed=2000 + nq + nt + nbytes)
            idx, dist = mpa.knn_hamming(q, t, k=k, ctx=ctx)
            oi, od = oracle.knn_hamming(q, t, k=k)
            assert np.array_equal(dist, od), (variant, nq, nt, nbytes, k)
            assert np.array_equal(idx, oi), (variant, nq, nt, nbytes, k)
        # ties everywhere: 5 distinct descriptors, the smaller train index must win in every merge level
        rng = np.random.default_rng(9)
        base = rng.integers(0, 256, (5, 32), dtype=np.uint8)
        t = base[rng.integers(0, 5, 3000)]
        q = base[rng.integers(0, 5, 500)]
        idx, dist = mpa.knn_hamming(q, t, ctx=ctx)
        oi, od = oracle.knn_hamming(q, t)
        assert np.array_equal(idx, oi) and np.array_equal(dist, od)
        # extremes of the distance range: all-equal and all-different bits
        z = np.zeros((70, 32), np.uint8)
        o = np.full((90, 32), 255, np.uint8)
        for a, b in ((z, o), (z, z[:40]), (o, np.concatenate([z[:45], o[:3]]))):
            idx, dist = mpa.knn_hamming(a, b, ctx=ctx)
            oi, od = oracle.knn_hamming(a, b)
            assert np.array_equal(idx, oi) and np.array_equal(dist, od)
        # the fused getMatches path on this variant
        q, t = synth.orb_pair(700, 900, seed=31)
        err, m = mpa.getMatches([None] * 700, [None] * 900, q, t, matcher_name="LINEAR", ctx=ctx)
        rc, om = oracle.get_matches_linear(700, 900, q, t)
        assert err == rc == 0 and m.tobytes() == om.tobytes()
    finally:
        _set_hamming(ctx, HAMMING_DEFAULT)


@pytest.mark.parametrize("variant", ["mfma_fp4", "mfma_fp4_qt2", "mfma_fp4_train01", "mfma_fp4_train_pm1"])
def test_c2_full_size_matrix_core_equals_valu(ctx, oracle, variant):
    """BASELINE C2 (8192 x 8192 x 256 bit): the matrix-core kernel against the VALU kernel (itself oracle-checked on samples)."""
    q, t = synth.orb_pair(8192, 8192, seed=20260102)
    _set_hamming(ctx, HAMMING_VARIANTS["valu_lds_tiled"])
    idx0, dist0 = mpa.knn_hamming(q, t, ctx=ctx)
    _set_hamming(ctx, HAMMING_VARIANTS[variant])
    try:
        idx3, dist3 = mpa.knn_hamming(q, t, ctx=ctx)
    finally:
        _set_hamming(ctx, HAMMING_DEFAULT)
    assert np.array_equal(idx0, idx3) and np.array_equal(dist0, dist3)
    rows = np.random.default_rng(1).choice(8192, 64, replace=False)
    oi, od = oracle.knn_hamming(q[rows], t)
    assert np.array_equal(idx3[rows], oi) and np.array_equal(dist3[rows], od)


@pytest.mark.parametrize("variant", ["mfma_fp4", "valu_lds_tiled"])
def test_hamming_many_splits_large_train_set(ctx, oracle, variant):
    """Train sets far beyond one split (the matrix-core kernel caps a split at 4096 rows, i.e. 18 splits here, the last one
    ragged) and a query count that is not a multiple of anything; checked on a sample of queries against the oracle."""
    q, t = synth.orb_pair(3001, 70001, seed=77)
    _set_hamming(ctx, HAMMING_VARIANTS[variant])
    try:
        idx, dist = mpa.knn_hamming(q, t, ctx=ctx)
    finally:
        _set_hamming(ctx, HAMMING_DEFAULT)
    rows = np.random.default_rng(5).choice(3001, 48, replace=False)
    oi, od = oracle.knn_hamming(q[rows], t)
    assert np.array_equal(idx[rows], oi) and np.array_equal(dist[rows], od)
    assert idx.min() >= 0 and idx.max() < 70001 and (dist[:, 0] <= dist[:, 1]).all()


def test_hamming_property_random_shapes_and_ties(ctx, oracle):
    """Property test (hypothesis): any small shape, any descriptor width 1..64 bytes, k in {1, 2}, descriptors drawn from a small
    alphabet so that equal distances are the rule -- the default (matrix-core) path returns the oracle's pairs bit for bit."""
    from hypothesis import HealthCheck, given, settings, strategies as st

    @settings(max_examples=60, deadline=None, suppress_health_check=list(HealthCheck), derandomize=True)
    @given(nq=st.integers(1, 200), nt=st.integers(2, 1500), nbytes=st.integers(1, 64), k=st.integers(1, 2),
           alphabet=st.integers(1, 40), seed=st.integers(0, 2**31 - 1))
    def check(nq, nt, nbytes, k, alphabet, seed):
        rng = np.random.default_rng(seed)
        base = rng.integers(0, 256, (alphabet, nbytes), dtype=np.uint8)
        t = base[rng.integers(0, alphabet, nt)]
        q = base[rng.integers(0, alphabet, nq)]
        flip = rng.random(q.shape) < 0.02          # a few perturbed bytes: near-ties as well as exact ties
        q = np.where(flip, rng.integers(0, 256, q.shape, dtype=np.uint8), q)
        idx, dist = mpa.knn_hamming(q, t, k=k, ctx=ctx)
        oi, od = oracle.knn_hamming(q, t, k=k)
        assert np.array_equal(dist, od) and np.array_equal(idx, oi), (nq, nt, nbytes, k, alphabet, seed)

    _set_hamming(ctx, HAMMING_DEFAULT)
    check()


def test_hamming_fused_epilogue_survives_a_dynamic_split_call_between_two_fused_calls(ctx, oracle):
    """the ticket counters of the fused split merge shared a workspace slot with the chunk counters of the dynamic-split
    kernel (hamming_mfma_lds = 2), which leaves them non-zero.  A fused call with several splits, a dynamic-split call, a fused call again:
    all three return the oracle's pairs."""
    q, t = synth.orb_pair(2048, 6000, seed=4242)     # 64 query tiles: several train splits on 256 CUs
    oi, od = oracle.knn_hamming(q[::16], t)
    try:
        for lds in (1, 2, 1, 2, 1):
            ctx.set_option("hamming_mfma_lds", lds)
            idx, dist = mpa.knn_hamming(q, t, ctx=ctx)
            assert np.array_equal(idx[::16], oi) and np.array_equal(dist[::16], od), lds
            err, m = mpa.getMatches([None] * 2048, [None] * 6000, q, t, matcher_name="LINEAR", ctx=ctx)
            assert err == 0 and np.array_equal(m["trainIdx"], idx[m["queryIdx"], 0])
    finally:
        ctx.set_option("hamming_mfma_lds", 1)


@pytest.mark.parametrize("nq,nt,nbytes,B", [(8192, 8192, 32, 1), (1000, 5000, 32, 1), (100, 3000, 32, 1), (1, 40, 32, 1), (8191, 8200, 32, 1), (3000, 4000, 16, 1),
                                           (4096, 4096, 32, 2), (2500, 300, 64, 1), (700, 900, 32, 5)])
def test_matches_emitted_by_the_merge_kernel(ctx, oracle, nq, nt, nbytes, B):
    """Round 5, the latency shape (one image pair per call): the merge kernel writes the DMatch rows itself (option hamming_merge_emit;
    measured no faster, so off by default: workgroups chain their pass counts through a generation-tagged table) instead of a ratio_write launch.  Same idx, dist,
    count and match rows as with the option off and as the oracle; called repeatedly (the table's generation advances) and with shapes
    that take other paths (several pairs, the fused epilogue, wide descriptors)."""
    import torch
    from matchinglib_poselib_amd.matching import match_hamming_device

    qs, ts = zip(*[synth.orb_pair(nq, nt, nbytes=nbytes, seed=7000 + nq + nt + b) for b in range(B)])
    q, t = torch.from_numpy(np.stack(qs)).cuda(), torch.from_numpy(np.stack(ts)).cuda()
    outs = {}
    try:
        for emit in (1, 0, 1):
            ctx.set_option("hamming_merge_emit", emit)
            for rep in range(3):
                o = match_hamming_device(q, t, ctx=ctx)
                torch.cuda.synchronize()
                cur = tuple(o[k].cpu().numpy().copy() for k in ("idx", "dist", "count"))
                rows = [o["matches"][b, : int(cur[2][b])].cpu().numpy().copy() for b in range(B)]
                if emit not in outs:
                    outs[emit] = (cur, rows)
                else:
                    assert all(np.array_equal(a, c) for a, c in zip(outs[emit][0], cur)) and all(np.array_equal(a, c) for a, c in zip(outs[emit][1], rows))
    finally:
        ctx.set_option("hamming_merge_emit", 0)
    assert all(np.array_equal(a, c) for a, c in zip(outs[1][0], outs[0][0])) and all(np.array_equal(a, c) for a, c in zip(outs[1][1], outs[0][1]))
    for b in range(B):
        oi, od = oracle.knn_hamming(qs[b], ts[b])
        o = oracle.ratio_filter(oi, od)
        m = outs[1][1][b]
        assert np.array_equal(outs[1][0][0][b], oi) and np.array_equal(outs[1][0][1][b], od) and len(m) == len(o)
        assert np.array_equal(m[:, 0], o["queryIdx"]) and np.array_equal(m[:, 1], o["trainIdx"]) and (m[:, 2] == -1).all()
        assert np.array_equal(m[:, 3].view(np.float32), o["distance"])


def test_hamming_one_split_of_8192_rows_is_exact(ctx, oracle):
    """Round 5: with the chip full (64 pairs per call) an 8192-row train set is ONE split -- the row rides in the accumulator's fraction
    as (row - frame) * 2^-14, which spans (-1/2, 1/2) over exactly 8192 rows.  Adversarial pairs for that bound: every train row at the same
    distance (the winner is decided by the fraction alone, rows 0 and 1), the two best in the LAST two rows, a tie between the first and the
    last row; the rest random.  Both split caps (option hamming_split_rows 8192 / 4096) give the same tables, equal to the oracle's."""
    import torch
    from matchinglib_poselib_amd.matching import match_hamming_device
    B, n = 64, 8192
    rng = np.random.default_rng(20261005)
    qs, ts = [], []
    for b in range(B):
        q, t = synth.orb_pair(n, n, seed=7000 + b)
        qs.append(q), ts.append(t)
    q0 = rng.integers(0, 256, 32, dtype=np.uint8)

    def flipped(k, salt):  # q0 with k distinct bits flipped
        v = q0.copy()
        for bit in np.random.default_rng(salt).choice(256, k, replace=False):
            v[bit >> 3] ^= np.uint8(1 << (bit & 7))
        return v
    # pair 1: all train rows identical -> (0, 1) for every query
    ts[1] = np.tile(flipped(9, 1), (n, 1))
    # pair 2: identical queries; all rows at distance 8, row 8191 at 1, row 8190 at 2
    qs[2] = np.tile(q0, (n, 1))
    ts[2] = np.stack([flipped(8, 100 + i) for i in range(n)])
    ts[2][n - 1], ts[2][n - 2] = flipped(1, 5), flipped(2, 6)
    # pair 3: identical queries; rows 0 and 8191 tie at distance 3, everything else at 8
    qs[3] = np.tile(q0, (n, 1))
    ts[3] = np.stack([flipped(8, 9000 + i) for i in range(n)])
    ts[3][0], ts[3][n - 1] = flipped(3, 7), flipped(3, 8)
    dq, dt = torch.from_numpy(np.stack(qs)).cuda(), torch.from_numpy(np.stack(ts)).cuda()
    res = {}
    try:
        for cap in (8192, 4096):
            ctx.set_option("hamming_split_rows", cap)
            out = match_hamming_device(dq, dt, ctx=ctx)
            torch.cuda.synchronize()
            res[cap] = (out["idx"].cpu().numpy().copy(), out["dist"].cpu().numpy().copy(), out["count"].cpu().numpy().copy())
    finally:
        ctx.set_option("hamming_split_rows", 0)
    for a, b in zip(res[8192], res[4096]):
        assert np.array_equal(a, b)
    idx, dist, _ = res[8192]
    assert (idx[1] == np.array([0, 1])).all() and (dist[1, :, 0] == dist[1, :, 1]).all()
    assert (idx[2] == np.array([n - 1, n - 2])).all() and (dist[2] == np.array([1, 2])).all()
    assert (idx[3] == np.array([0, n - 1])).all() and (dist[3] == np.array([3, 3])).all()
    rows = rng.choice(n, 24, replace=False)
    for b in (0, 1, 2, 3, 17, 63):
        oi, od = oracle.knn_hamming(qs[b][rows], ts[b])
        assert np.array_equal(idx[b][rows], oi) and np.array_equal(dist[b][rows], od), b


@pytest.mark.parametrize("nbytes", [64, 16, 8, 61])
def test_hamming_long_splits_other_descriptor_widths(ctx, nbytes):
    """The 8192-row split bound for the matrix-core kernels of the OTHER descriptor widths (register-prefetch kernel, 1 / 2 / 8 K-steps; 64
    bytes = integer parts up to +-512): a batch large enough that a whole 8192-row train set is one split, ties everywhere (a small
    alphabet), matrix-core = VALU kernel on every pair, both split caps."""
    import torch
    from matchinglib_poselib_amd.matching import match_hamming_device
    B, nq, nt = 64, 2048, 8192           # with hamming_mfma_blocks_per_cu = 1: 16 wave groups x 64 pairs = the 1024 waves wanted -> ONE split
    rng = np.random.default_rng(500 + nbytes)
    base = rng.integers(0, 256, (7, nbytes), dtype=np.uint8)
    q = base[rng.integers(0, 7, (B, nq))]
    t = base[rng.integers(0, 7, (B, nt))]
    flip = rng.random(t.shape) < 0.02
    t = np.where(flip, rng.integers(0, 256, t.shape, dtype=np.uint8), t)
    t[3, :] = t[3, 0]                      # one pair with every train row equal: (0, 1) everywhere
    dq, dt = torch.from_numpy(np.ascontiguousarray(q)).cuda(), torch.from_numpy(np.ascontiguousarray(t)).cuda()
    res = {}
    try:
        for name, variant, cap in (("valu", 0, 0), ("mfma", 3, 0), ("mfma4096", 3, 4096)):
            ctx.set_option("hamming_variant", variant)
            ctx.set_option("hamming_split_rows", cap)
            ctx.set_option("hamming_mfma_blocks_per_cu", 1)
            out = match_hamming_device(dq, dt, ctx=ctx)
            torch.cuda.synchronize()
            res[name] = [out[k].cpu().numpy().copy() for k in ("idx", "dist", "count")]
    finally:
        ctx.set_option("hamming_variant", 3)
        ctx.set_option("hamming_split_rows", 0)
        ctx.set_option("hamming_mfma_blocks_per_cu", 3)   # the library default
    for name in ("mfma", "mfma4096"):
        for a, b in zip(res["valu"], res[name]):
            assert np.array_equal(a, b), (name, nbytes)
    assert (res["mfma"][0][3] == np.array([0, 1])).all()
