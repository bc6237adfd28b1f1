"""GPU parity tests for the robust-pose path: HIP kernels (through the C ABI) vs the CPU oracle."""
import numpy as np
import pytest

from matchinglib_poselib_amd import pose, synth

pytestmark = pytest.mark.gpu


def e_dist(a, b):
    return min(np.linalg.norm(a - b), np.linalg.norm(a + b))


def match_sets(Eg, Eo, tol):
    """Every oracle model has a GPU counterpart (up to sign) and vice versa."""
    if len(Eg) != len(Eo):
        return False
    for e in Eo:
        if min(e_dist(e, x) for x in Eg) > tol:
            return False
    for e in Eg:
        if min(e_dist(e, x) for x in Eo) > tol:
            return False
    return True


def test_solve_5pt_vs_oracle(ctx, oracle):
    from test_gpu_baseline_configs import compare_solver_with_oracle
    p1, p2, R, t, mask, th = synth.pose_scene(5000, seed=20260103)
    samples = oracle.sample_table(12345, p1, p2, 600)
    cm, unexplained, worst, differ = compare_solver_with_oracle(ctx, oracle, p1, p2, samples)
    # equal solution counts; every CPU model reproduced to 1e-8 unless the CPU model itself violates the essential-matrix constraints
    assert not cm and not unexplained and worst < 1e-12, (cm, unexplained, worst)
    E, nm = pose.solve_5pt(p1, p2, samples, ctx=ctx)
    # constraints hold for every returned model
    for s in range(0, len(samples), 7):
        x1 = np.c_[p1[samples[s]], np.ones(5)]
        x2 = np.c_[p2[samples[s]], np.ones(5)]
        for e in E[s, :nm[s]]:
            assert np.abs(np.einsum("ij,jk,ik->i", x2, e, x1)).max() < 1e-9
            assert abs(np.linalg.norm(e) - 1) < 1e-12


def test_score_models_bit_exact(ctx, oracle):
    """Same E in -> identical inlier counts AND identical double error sums (in-order accumulation, no FMA)."""
    p1, p2, R, t, mask, th = synth.pose_scene(5000, seed=20260103)
    samples = oracle.sample_table(777, p1, p2, 60)
    Es = np.concatenate([oracle.run5point(p1[s], p2[s]) for s in samples])
    good, esum = pose.score_models(p1, p2, Es, th, ctx=ctx)
    for i, e in enumerate(Es):
        g, s, err, m = oracle.find_inliers(p1, p2, e, th)
        assert good[i] == g
        assert esum[i] == s, (i, esum[i], s)
    # odd sizes (tail handling of the 4-point batches)
    for n in (6, 7, 9, 1023):
        g0, s0 = pose.score_models(p1[:n], p2[:n], Es[:5], th, ctx=ctx)
        for i in range(5):
            g, s, _, _ = oracle.find_inliers(p1[:n], p2[:n], Es[i], th)
            assert g0[i] == g and s0[i] == s


@pytest.mark.parametrize("refit", [False, True])
@pytest.mark.parametrize("seed", [12345, 7])
def test_ransac_vs_oracle_reference_settings(ctx, oracle, refit, seed):
    """The reference's own configuration: 1000 iterations, confidence 0.999 (five-point.cpp:117, pose_estim.cpp:872)."""
    p1, p2, R, t, mask, th = synth.pose_scene(5000, seed=20260103)
    g = pose.ransac_essential(p1, p2, th, confidence=0.999, max_iters=1000, refit=refit, seed=seed, ctx=ctx)
    o = oracle.ransac_essential(p1, p2, th, confidence=0.999, max_iters=1000, lesqu=refit, seed=seed)
    assert g["ok"] and o["ok"]
    assert g["iters"] == o["iters"]
    assert g["n_inliers"] == o["n_inliers"]
    assert e_dist(g["E"], o["E"]) < (1e-7 if refit else 1e-8)
    assert (g["mask"] != o["mask"]).sum() <= (2 if refit else 0)
    # north_star bar: R, t within 1e-6 of the CPU path for a fixed seed
    go, Ro, to, Qo, mo = oracle.recover_pose(o["E"], p1, p2, 50.0, o["mask"])
    gg, Rg, tg, Qg, mg = oracle.recover_pose(g["E"], p1, p2, 50.0, g["mask"])
    assert np.abs(Rg - Ro).max() < 1e-6 and np.abs(tg - to).max() < 1e-6


def test_ransac_no_early_exit_c3_shape(ctx, oracle):
    """C3-style run (confidence 1.0 => every hypothesis is evaluated), shortened so that the oracle finishes fast."""
    p1, p2, R, t, mask, th = synth.pose_scene(2000, seed=20260103)
    g = pose.ransac_essential(p1, p2, th, confidence=1.0, max_iters=1500, refit=False, seed=12345, ctx=ctx)
    o = oracle.ransac_essential(p1, p2, th, confidence=1.0, max_iters=1500, lesqu=False, seed=12345)
    assert g["iters"] == o["iters"] == 1500
    assert g["n_inliers"] == o["n_inliers"]
    assert e_dist(g["E"], o["E"]) < 1e-8
    assert np.array_equal(g["mask"], o["mask"])


def test_ransac_failure_and_bad_input(ctx):
    import matchinglib_poselib_amd as mpa
    rng = np.random.default_rng(0)
    p = rng.normal(size=(5, 2))
    with pytest.raises(mpa.MlplError):
        pose.ransac_essential(p, p, 0.01, ctx=ctx)   # n must exceed the model size
    with pytest.raises(SystemExit):
        pose.estimateEssentialMat(p, p, "USAC", ctx=ctx)
    with pytest.raises(SystemExit):
        pose.estimateEssentialMat(p, p, "NOPE", ctx=ctx)


def essential_from(R, t):
    tx = np.array([[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
    E = tx @ R
    return E / np.linalg.norm(E)


def test_recover_pose_vs_oracle(ctx, oracle):
    p1, p2, R, t, mask, th = synth.pose_scene(3000, seed=20260103)
    o = oracle.ransac_essential(p1, p2, th, max_iters=400, seed=5)
    for E, m_in, dist in [(o["E"], o["mask"], 50.0), (-o["E"], None, 50.0), (2.5 * o["E"], o["mask"], 9.0),
                          (essential_from(R, t), None, 50.0)]:
        go, Ro, to, Qo, mo = oracle.recover_pose(E, p1, p2, dist, m_in)
        gg, Rg, tg, Qg, mg = pose.getPoseTriangPts(E, p1, p2, m_in, dist, ctx=ctx)
        assert gg == go
        assert np.abs(Rg - Ro).max() < 1e-12 and np.abs(tg.ravel() - to).max() < 1e-12
        if m_in is not None:
            assert np.array_equal(mg, mo)
        fin = np.isfinite(Qo).all(axis=1) & (np.abs(Qo).max(axis=1) < 1e6)
        assert np.allclose(Qg[fin], Qo[fin], rtol=1e-9, atol=1e-9)
    # R,t close to the ground truth on this scene
    gg, Rg, tg, Qg, mg = pose.getPoseTriangPts(o["E"], p1, p2, o["mask"], 50.0, ctx=ctx)
    assert np.abs(Rg - R).max() < 2e-2 and np.abs(tg.ravel() - t).max() < 2e-2


def test_recover_pose_all_behind(ctx, oracle):
    """Degenerate input: identical points in both views -> whatever the oracle returns, the GPU returns the same."""
    rng = np.random.default_rng(3)
    p = rng.normal(size=(50, 2))
    E = np.array([[0, -1, 0.2], [1, 0, -0.3], [-0.2, 0.3, 0.0]])
    go, Ro, to, Qo, mo = oracle.recover_pose(E, p, p + 0.01, 50.0, None)
    gg, Rg, tg, Qg, mg = pose.getPoseTriangPts(E, p, p + 0.01, None, 50.0, ctx=ctx)
    assert gg == go and np.abs(Rg - Ro).max() < 1e-12 and np.abs(tg.ravel() - to).max() < 1e-12


def test_estimate_relative_pose_end_to_end(ctx):
    p1, p2, R, t, mask, th = synth.pose_scene(4000, seed=77)
    ok, E, Rg, tg, Q, m = pose.estimateRelativePose(p1, p2, threshold=th, refine=True, seed=4242, ctx=ctx)
    assert ok
    assert np.abs(Rg - R).max() < 5e-3 and np.abs(tg.ravel() - t).max() < 2e-2
    assert (m != 0).sum() > 1700


def test_ransac_multi_chunk_replay(ctx, oracle):
    """max_iters above the per-pass capacity: the replay state (best model, niters, iteration count) is carried across
    device passes.  A small pass size forces many passes; results must not depend on it."""
    from matchinglib_poselib_amd import _lib
    p1, p2, R, t, mask, th = synth.pose_scene(1500, seed=41)
    o = oracle.ransac_essential(p1, p2, th, confidence=0.999, max_iters=1000, lesqu=True, seed=9)
    o1 = oracle.ransac_essential(p1, p2, th, confidence=1.0, max_iters=700, lesqu=False, seed=10)
    for chunk in (64, 100, 0):
        _lib.check(ctx.lib.mlpl_set_option(ctx.handle, b"ransac_chunk", chunk), "set_option")
        g = pose.ransac_essential(p1, p2, th, confidence=0.999, max_iters=1000, refit=True, seed=9, ctx=ctx)
        assert g["iters"] == o["iters"] and g["n_inliers"] == o["n_inliers"], chunk
        assert e_dist(g["E"], o["E"]) < 1e-7
        g1 = pose.ransac_essential(p1, p2, th, confidence=1.0, max_iters=700, refit=False, seed=10, ctx=ctx)
        assert g1["iters"] == 700 and g1["n_inliers"] == o1["n_inliers"] and e_dist(g1["E"], o1["E"]) < 1e-8
        assert np.array_equal(g1["mask"], o1["mask"])
    _lib.check(ctx.lib.mlpl_set_option(ctx.handle, b"ransac_chunk", 0), "set_option")


def test_large_shapes_sampled(ctx, oracle):
    """Bigger-than-C2 shapes (many splits, several LDS tiles per block): sampled rows against the oracle."""
    import matchinglib_poselib_amd as mpa
    q, t = synth.orb_pair(20000, 50000, seed=123)
    idx, dist = mpa.knn_hamming(q, t, ctx=ctx)
    sub = np.arange(0, 20000, 397)
    oi, od = oracle.knn_hamming(q[sub], t)
    assert np.array_equal(idx[sub], oi) and np.array_equal(dist[sub], od)
    qf, tf = synth.sift_pair(5000, 20000, seed=124)
    idx, dist = mpa.knn_l2sq(qf, tf, ctx=ctx)
    sub = np.arange(0, 5000, 211)
    oi, od = oracle.knn_l2sq(qf[sub], tf)
    assert np.array_equal(idx[sub], oi) and dist[sub].tobytes() == od.tobytes()


def test_recover_pose_translation_only(ctx, oracle):
    """getPoseTriangPts(..., translatE=true): R = I, candidates [I|t] and [I|-t] only (five-point.cpp:178-193)."""
    rng = np.random.default_rng(8)
    tt = np.array([0.9, -0.1, 0.3])
    tt /= np.linalg.norm(tt)
    X = np.stack([rng.uniform(-2, 2, 400), rng.uniform(-2, 2, 400), rng.uniform(4, 12, 400)], axis=1)
    p1 = X[:, :2] / X[:, 2:3]
    X2 = X + tt
    p2 = X2[:, :2] / X2[:, 2:3]
    Et = np.array([[0, -tt[2], tt[1]], [tt[2], 0, -tt[0]], [-tt[1], tt[0], 0]])
    for Ein in (Et, -Et, 2.0 * Et):
        go, Ro, to, Qo, mo = oracle.recover_pose_translation(Ein, p1, p2, 50.0, None)
        gg, Rg, tg, Qg, mg = pose.getPoseTriangPts(Ein, p1, p2, None, 50.0, translatE=True, ctx=ctx)
        assert gg == go == 400
        assert np.array_equal(Rg, np.eye(3)) and np.abs(tg.ravel() - to).max() < 1e-15 and np.abs(tg.ravel() - tt).max() < 1e-12
        assert np.allclose(Qg, Qo, rtol=1e-9, atol=1e-9)


def test_estimate_essential_minimal_five_points(ctx, oracle):
    p1, p2, R, t, mask, th = synth.pose_scene(50, inlier_frac=1.0, seed=2, noise_px=0.0)
    ok, Es, m = pose.estimateEssentialMat(p1[:5], p2[:5], "RANSAC", th, ctx=ctx)
    Eo = oracle.run5point(p1[:5], p2[:5])
    assert ok and Es.shape == (3 * len(Eo), 3) and m.tolist() == [1] * 5
    assert match_sets(Es.reshape(-1, 3, 3), Eo, 1e-7)


@pytest.mark.parametrize("n", [5000, 4999, 257, 6])
def test_median_models_bit_exact(ctx, oracle, n):
    """The radix select equals the reference's int sort + middle element(s) (modelest.cpp:540-544), odd and even n."""
    p1, p2, R, t, mask, th = synth.pose_scene(max(n, 64), seed=31)
    p1, p2 = p1[:n], p2[:n]
    samples = oracle.sample_table(99, p1, p2, 40)
    Es = np.concatenate([oracle.run5point(p1[s], p2[s]) for s in samples])
    med = pose.median_models(p1, p2, Es, ctx=ctx)
    for k, E in enumerate(Es):
        err = oracle.sampson_err(p1, p2, E)
        srt = np.sort(err.view(np.int32)).view(np.float32)
        ref = float(srt[n // 2]) if n % 2 else float(np.float32(srt[n // 2 - 1] + srt[n // 2])) * 0.5
        assert med[k] == ref, (k, med[k], ref)


def test_median_models_duplicates_and_ties(ctx, oracle):
    """Many identical errors (duplicated correspondences) around the middle: the lower-middle rule must pick the duplicate."""
    p1, p2, R, t, mask, th = synth.pose_scene(64, seed=32)
    p1 = np.repeat(p1[:8], 16, axis=0)
    p2 = np.repeat(p2[:8], 16, axis=0)
    Es = oracle.run5point(p1[::16][:5], p2[::16][:5])
    rng = np.random.default_rng(3)
    Es = np.concatenate([Es, rng.normal(size=(6, 3, 3))])
    med = pose.median_models(p1, p2, Es, ctx=ctx)
    n = p1.shape[0]
    for k, E in enumerate(Es):
        srt = np.sort(oracle.sampson_err(p1, p2, E).view(np.int32)).view(np.float32)
        assert med[k] == float(np.float32(srt[n // 2 - 1] + srt[n // 2])) * 0.5


@pytest.mark.parametrize("seed", [12345, 7])
@pytest.mark.parametrize("n", [5000, 1001])
def test_lmeds_vs_oracle(ctx, oracle, seed, n):
    """estimateEssentialMat(..., "LMEDS") settings: confidence 0.999 => 134 samples (five-point.cpp:127, modelest.cpp:506)."""
    p1, p2, R, t, mask, th = synth.pose_scene(n, inlier_frac=0.7, seed=20260104)
    g = pose.lmeds_essential(p1, p2, confidence=0.999, max_iters=2000, seed=seed, ctx=ctx)
    o = oracle.lmeds_essential(p1, p2, confidence=0.999, max_iters=2000, seed=seed)
    assert g["ok"] and o["ok"]
    assert e_dist(g["E"], o["E"]) < 1e-8
    assert abs(g["min_median"] - o["min_median"]) <= 1e-6 * o["min_median"]
    assert abs(g["n_inliers"] - o["n_inliers"]) <= 2 and (g["mask"] != o["mask"]).sum() <= 2
    assert int(g["mask"].sum()) == g["n_inliers"]
    go, Ro, to, Qo, mo = oracle.recover_pose(o["E"], p1, p2, 50.0, o["mask"])
    gg, Rg, tg, Qg, mg = oracle.recover_pose(g["E"], p1, p2, 50.0, g["mask"])
    assert np.abs(Rg - Ro).max() < 1e-6 and np.abs(tg - to).max() < 1e-6


def test_lmeds_through_estimate_essential_mat(ctx, oracle):
    p1, p2, R, t, mask, th = synth.pose_scene(2000, inlier_frac=0.75, seed=5)
    ok, E, m = pose.estimateEssentialMat(p1, p2, "LMEDS", th, refine=True, seed=3, ctx=ctx)
    o = oracle.lmeds_essential(p1, p2, seed=3)
    assert ok and e_dist(E, o["E"]) < 1e-8 and (m != o["mask"]).sum() <= 2
    import matchinglib_poselib_amd as mpa
    with pytest.raises(mpa.MlplError):
        pose.lmeds_essential(p1[:5], p2[:5], ctx=ctx)
    with pytest.raises(SystemExit):   # the reference prints and calls exit(1) for USAC / unknown names (pose_estim.cpp:878-887)
        pose.estimateEssentialMat(p1, p2, "USAC", ctx=ctx)


def test_recover_pose_device_variant_equals_host_api(ctx, oracle):
    import torch

    p1, p2, R, t, mask, th = synth.pose_scene(3000, seed=61)
    E = essential_from(R, t.reshape(-1))
    m0 = (np.random.default_rng(1).random(3000) < 0.8).astype(np.uint8)
    ng, Rh, th_, Qh, mh = pose.getPoseTriangPts(E, p1, p2, m0, 50.0, ctx=ctx)
    d1, d2 = torch.from_numpy(p1).cuda(), torch.from_numpy(p2).cuda()
    dm = torch.from_numpy(m0.copy()).cuda()
    dQ = torch.empty((3000, 3), dtype=torch.float64, device="cuda")
    ngd, Rd, td = pose.getPoseTriangPts_device(E, d1, d2, dm, 50.0, Q_out=dQ, ctx=ctx)
    torch.cuda.synchronize()
    assert ngd == ng and np.array_equal(Rd, Rh) and np.array_equal(td, th_)
    assert np.array_equal(dm.cpu().numpy(), mh) and np.array_equal(dQ.cpu().numpy(), Qh)
    go, Ro, to, Qo, mo = oracle.recover_pose(E, p1, p2, 50.0, m0)
    assert ngd == go and np.abs(Rd - Ro).max() < 1e-9
    # no mask, no Q
    ng2, R2, t2 = pose.getPoseTriangPts_device(E, d1, d2, ctx=ctx)
    assert ng2 == oracle.recover_pose(E, p1, p2, 50.0, None)[0]


@pytest.mark.parametrize("n", [5000, 4321, 777])
def test_ransac_device_iteration_bounds_equal_host_table(ctx, oracle, n):
    """The adaptive iteration bound is evaluated on the device and verified by the host; forcing the host table must give the
    same iteration count, model and mask -- and both equal the oracle (varying n: the table would be rebuilt per call)."""
    p1, p2, R, t, mask, th = synth.pose_scene(n, seed=500 + n)
    ctx.set_option("ransac_host_table", 1)
    try:
        a = pose.ransac_essential(p1, p2, th, confidence=0.999, max_iters=1000, refit=False, seed=n, ctx=ctx)
    finally:
        ctx.set_option("ransac_host_table", 0)
    b = pose.ransac_essential(p1, p2, th, confidence=0.999, max_iters=1000, refit=False, seed=n, ctx=ctx)
    o = oracle.ransac_essential(p1, p2, th, confidence=0.999, max_iters=1000, lesqu=False, seed=n)
    assert a["iters"] == b["iters"] == o["iters"] and a["n_inliers"] == b["n_inliers"] == o["n_inliers"]
    assert np.array_equal(a["E"], b["E"]) and np.array_equal(a["mask"], b["mask"]) and np.array_equal(b["mask"], o["mask"])


@pytest.mark.parametrize("shape", [1, 2])
def test_division_free_inlier_count_is_the_reference_predicate(ctx, oracle, shape):
    """sampson_inlier: (double)(float)(N / D) <= thresh^2 decided without the division.  Thresholds that sit exactly ON float
    error values (the predicate's <= and the float/double rounding boundaries) and one ulp either side of them."""
    p1, p2, R, t, mask, th = synth.pose_scene(3000, seed=71)
    samples = oracle.sample_table(5, p1, p2, 12)
    Es = np.concatenate([oracle.run5point(p1[s], p2[s]) for s in samples])[:24]
    errs = oracle.sampson_err(p1, p2, Es[0]).astype(np.float64)          # float errors of model 0, widened
    picks = np.sort(errs)[[10, 500, 1500, 2500, 2990]]
    t2s = [th * th]
    for v in picks:
        t2s += [v, np.nextafter(v, 0.0), np.nextafter(v, 1.0), float(np.nextafter(np.float32(v), np.float32(0))),
                0.5 * (v + float(np.nextafter(np.float32(v), np.float32(1))))]   # a float, its double neighbours, the float midpoint
    for t2 in t2s:
        got = pose.count_models(p1, p2, Es, t2, shape=shape, ctx=ctx)
        for k, E in enumerate(Es):
            want = int((oracle.sampson_err(p1, p2, E).astype(np.float64) <= t2).sum())
            assert got[k] == want, (shape, t2, k, got[k], want)
    # degenerate model: all-zero E gives N = D = 0 -> NaN error -> never an inlier
    z = pose.count_models(p1, p2, np.zeros((1, 3, 3)), th * th, shape=shape, ctx=ctx)
    assert z[0] == 0
    # thresholds outside the float range: above FLT_MAX every finite error passes, a negative one passes nothing
    big = pose.count_models(p1, p2, Es, 1e39, shape=shape, ctx=ctx)
    assert (big == 3000).all()
    neg = pose.count_models(p1, p2, Es, -1.0, shape=shape, ctx=ctx)
    assert (neg == 0).all()


@pytest.mark.parametrize("scale,escale", [(1.0, 1.0), (800.0, 1.0), (1e-3, 1.0), (1.0, 1e-9), (3e5, 1e6), (1e13, 1.0), (1.0, 1e-45), (1e-30, 1e30)])
def test_fp32_prefilter_counts_at_every_scale(ctx, oracle, scale, escale):
    """The packed-fp32 pre-filter of the count-only kernels must never change a count: camera units, pixel units (x ~ 10^3), tiny and
    huge coordinates, unnormalised models, and scales at which it has to switch itself off (float overflow / underflow of the inputs).
    Compared with the same kernel with the filter off (pure fp64 predicate) and with the dividing reference arithmetic on the CPU."""
    p1, p2, R, t, mask, th = synth.pose_scene(2600, seed=123)
    samples = oracle.sample_table(7, p1, p2, 10)
    Es = np.concatenate([oracle.run5point(p1[s], p2[s]) for s in samples])[:20] * escale
    # scaling both images by c multiplies the Sampson error by c^2 when E stays fixed only for the bilinear part; simply rescale the
    # data and re-derive the expected counts from the CPU arithmetic on the rescaled inputs
    q1, q2 = p1 * scale, p2 * scale
    ref_err = np.stack([oracle.sampson_err(q1, q2, E).astype(np.float64) for E in Es])
    finite = np.isfinite(ref_err)
    for frac in (0.1, 0.5, 0.9):
        t2 = float(np.quantile(ref_err[finite], frac)) if finite.any() else 1.0
        if not (t2 > 0 and np.isfinite(t2)):
            continue
        want = (ref_err <= t2).sum(axis=1)
        got = {}
        for f in (1, 0):
            ctx.set_option("ransac_f32_filter", f)
            try:
                got[f] = pose.count_models(q1, q2, Es, t2, shape=1, ctx=ctx)
            finally:
                ctx.set_option("ransac_f32_filter", 1)
        assert np.array_equal(got[1], got[0]), (scale, escale, frac, np.nonzero(got[1] != got[0])[0][:5])
        assert np.array_equal(got[1], want), (scale, escale, frac, got[1][:5], want[:5])


@pytest.mark.parametrize("scale,escale", [(1.0, 1.0), (800.0, 1.0), (1.0, 1e-45), (1e13, 1.0)])
def test_count_kernel_deferred_queue_equals_inline_and_the_cpu(ctx, oracle, scale, escale):
    """Round 5: the counting kernel queues its undecided evaluations in LDS and decides them workgroup-wide (option ransac_count_defer,
    default 1) instead of on the spot.  Same counts as the inline form and as the CPU arithmetic: odd and tiny model counts (an idle second
    model of a lane), thresholds ON error values (many evaluations inside the band), scales at which the band is infinite (EVERY evaluation
    undecided: the queue overflows and the inline path takes the rest) and a point count above one tile."""
    p1, p2, R, t, mask, th = synth.pose_scene(2600, seed=321)
    samples = oracle.sample_table(11, p1, p2, 40)
    Es_all = np.concatenate([oracle.run5point(p1[s], p2[s]) for s in samples]) * escale
    q1, q2 = p1 * scale, p2 * scale
    for nm in (1, 3, 77, len(Es_all)):
        Es = Es_all[:nm]
        ref_err = np.stack([oracle.sampson_err(q1, q2, E).astype(np.float64) for E in Es])
        finite = np.isfinite(ref_err)
        t2s = [float(np.quantile(ref_err[finite], f)) for f in (0.2, 0.6)] if finite.any() else [1.0]
        t2s += [float(v) for v in np.sort(ref_err[0][np.isfinite(ref_err[0])])[[5, 1300]]] if np.isfinite(ref_err[0]).sum() > 1400 else []
        for t2 in t2s:
            if not (t2 > 0 and np.isfinite(t2)):
                continue
            want = (ref_err <= t2).sum(axis=1)
            got = {}
            for d in (1, 0):
                ctx.set_option("ransac_count_defer", d)
                try:
                    got[d] = pose.count_models(q1, q2, Es, t2, shape=1, ctx=ctx)
                finally:
                    ctx.set_option("ransac_count_defer", 1)
            assert np.array_equal(got[1], got[0]), (scale, escale, nm, t2)
            assert np.array_equal(got[1], want), (scale, escale, nm, t2, got[1][:5], want[:5])


@pytest.mark.parametrize("seed", [1, 2, 3])
def test_ransac_with_and_without_the_fp32_prefilter(ctx, oracle, seed):
    p1, p2, R, t, mask, th = synth.pose_scene(4000, inlier_frac=0.35, seed=900 + seed)
    runs = []
    for f, d in ((1, 1), (0, 1), (1, 0)):   # packed-fp32 filter with the deferred queue (default), pure fp64, filter with the inline fp64 path
        ctx.set_option("ransac_f32_filter", f)
        ctx.set_option("ransac_count_defer", d)
        try:
            runs.append(pose.ransac_essential(p1, p2, th, confidence=0.999, max_iters=6000, refit=False, seed=seed, ctx=ctx))
        finally:
            ctx.set_option("ransac_f32_filter", 1)
            ctx.set_option("ransac_count_defer", 1)
    a, b, c = runs
    assert a["iters"] == c["iters"] and a["n_inliers"] == c["n_inliers"] and np.array_equal(a["E"], c["E"]) and np.array_equal(a["mask"], c["mask"])
    assert a["iters"] == b["iters"] and a["n_inliers"] == b["n_inliers"] and np.array_equal(a["E"], b["E"]) and np.array_equal(a["mask"], b["mask"])
    o = oracle.ransac_essential(p1, p2, th, confidence=0.999, max_iters=6000, lesqu=False, seed=seed)
    assert a["iters"] == o["iters"] and a["n_inliers"] == o["n_inliers"] and np.array_equal(a["mask"], o["mask"])


@pytest.mark.parametrize("n,chunk", [(5000, 0), (300, 0), (40, 0), (7, 0), (2000, 100)])
def test_ransac_lazy_error_sums_equal_full_sums(ctx, oracle, n, chunk):
    """Counting without the division and summing errors only for the models that can still win must not change anything: same
    iteration count, model, mask as with sums for every model, for adaptive and exhaustive runs, tiny n (ties on the inlier
    count everywhere) and multi-pass replays."""
    p1, p2, R, t, mask, th = synth.pose_scene(n, seed=900 + n)
    ctx.set_option("ransac_chunk", chunk)
    try:
        for conf, iters, seed in ((0.999, 1000, 3), (1.0, 700, 4), (0.99, 300, 5)):
            ctx.set_option("ransac_lazy_sums", 0)
            a = pose.ransac_essential(p1, p2, th, confidence=conf, max_iters=iters, refit=True, seed=seed, ctx=ctx)
            ctx.set_option("ransac_lazy_sums", 1)
            b = pose.ransac_essential(p1, p2, th, confidence=conf, max_iters=iters, refit=True, seed=seed, ctx=ctx)
            assert a["ok"] == b["ok"] and a["iters"] == b["iters"] and a["n_inliers"] == b["n_inliers"], (n, conf)
            assert np.array_equal(a["E"], b["E"]) and np.array_equal(a["mask"], b["mask"])
            o = oracle.ransac_essential(p1, p2, th, confidence=conf, max_iters=iters, lesqu=False, seed=seed)
            c = pose.ransac_essential(p1, p2, th, confidence=conf, max_iters=iters, refit=False, seed=seed, ctx=ctx)
            assert c["iters"] == o["iters"] and c["n_inliers"] == o["n_inliers"] and np.array_equal(c["mask"], o["mask"])
    finally:
        ctx.set_option("ransac_lazy_sums", 1)
        ctx.set_option("ransac_chunk", 0)


def test_ransac_random_small_scenes_vs_oracle(ctx, oracle):
    """Many small random scenes (n = 6..400: inlier counts tie constantly, the adaptive bound moves on almost every record):
    iteration count, inlier count and mask equal the oracle's for every one; E up to the solver tolerance."""
    rng = np.random.default_rng(2026)
    checked = 0
    for case in range(160):
        n = int(rng.integers(6, 400))
        frac = float(rng.uniform(0.3, 0.95))
        p1, p2, R, t, mask, th = synth.pose_scene(n, inlier_frac=frac, seed=int(rng.integers(1, 1 << 30)))
        conf = float(rng.choice([0.9, 0.99, 0.999, 1.0]))
        iters = int(rng.choice([50, 300, 1000]))
        seed = int(rng.integers(0, 1 << 31))
        g = pose.ransac_essential(p1, p2, th, confidence=conf, max_iters=iters, refit=False, seed=seed, ctx=ctx)
        o = oracle.ransac_essential(p1, p2, th, confidence=conf, max_iters=iters, lesqu=False, seed=seed)
        assert g["ok"] == o["ok"], (case, n, conf, iters, seed)
        if not o["ok"]:
            continue
        assert g["iters"] == o["iters"] and g["n_inliers"] == o["n_inliers"], (case, n, conf, iters, seed, g["iters"], o["iters"])
        assert np.array_equal(g["mask"], o["mask"]), (case, n, seed)
        assert e_dist(g["E"], o["E"]) < 1e-6
        checked += 1
    assert checked > 120
