"""The two forms of the solver's elimination kernel (solve5pt_kernel: one hypothesis per wave, matrices in LDS; solve5pt3_kernel: three per
wave, matrices in registers) run the same floating-point operations in the same order: their models must be BIT-identical, and so must
whole RANSAC runs.  Reference arithmetic: five-point.cpp:366-471."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def ctx():
    import matchinglib_poselib_amd as mpa

    c = mpa.Context(0)
    yield c
    c.set_option("solver_wave3", 1)


def _models(ctx, p1, p2, samples, wave3, polish):
    from matchinglib_poselib_amd import pose

    ctx.set_option("solver_wave3", wave3)
    ctx.set_option("solver_polish", polish)
    try:
        return pose.solve_5pt(p1, p2, samples, ctx=ctx)
    finally:
        ctx.set_option("solver_wave3", 1)
        ctx.set_option("solver_polish", 1)


@pytest.mark.parametrize("n_samples", [1, 2, 3, 4, 5, 7, 64, 1000, 4099])
@pytest.mark.parametrize("polish", [0, 1])
def test_models_bit_identical(ctx, n_samples, polish):
    from matchinglib_poselib_amd import synth

    p1, p2, R, t, mask, th = synth.pose_scene(700, inlier_frac=0.6, seed=400 + n_samples)
    rng = np.random.default_rng(n_samples)
    samples = np.stack([rng.choice(len(p1), 5, replace=False) for _ in range(n_samples)]).astype(np.int32)
    E1, n1 = _models(ctx, p1, p2, samples, 0, polish)
    E3, n3 = _models(ctx, p1, p2, samples, 1, polish)
    assert np.array_equal(n1, n3)
    for s in range(n_samples):
        assert np.array_equal(E1[s, :n1[s]].view(np.uint64), E3[s, :n3[s]].view(np.uint64)), s
    assert n1.sum() > 0


def test_degenerate_samples_agree(ctx):
    """Repeated points, collinear points, all-zero coordinates: singular systems must be flagged by both kernels alike."""
    p = np.zeros((12, 2))
    p[:5] = [[0.1, 0.2], [0.1, 0.2], [0.1, 0.2], [0.3, -0.1], [-0.2, 0.05]]
    p[5:10] = [[0.01 * k, 0.02 * k] for k in range(5)]
    q = p * 1.01 + 0.001
    samples = np.array([[0, 1, 2, 3, 4], [5, 6, 7, 8, 9], [10, 11, 10, 11, 10], [0, 5, 3, 9, 4], [3, 4, 8, 9, 7]], np.int32)
    E1, n1 = _models(ctx, p, q, samples, 0, 1)
    E3, n3 = _models(ctx, p, q, samples, 1, 1)
    assert np.array_equal(n1, n3)
    for s in range(len(samples)):
        a, b = E1[s, :n1[s]], E3[s, :n3[s]]
        assert np.array_equal(np.isnan(a), np.isnan(b))
        assert np.array_equal(np.nan_to_num(a).view(np.uint64), np.nan_to_num(b).view(np.uint64))


@pytest.mark.parametrize("refit", [False, True])
def test_ransac_runs_bit_identical(ctx, refit):
    from matchinglib_poselib_amd import pose, synth

    for n, frac, seed, iters in ((5000, 0.5, 3, 20000), (900, 0.3, 5, 1000), (300, 0.7, 9, 1000)):
        p1, p2, R, t, mask, th = synth.pose_scene(n, inlier_frac=frac, seed=seed)
        out = []
        for w3 in (0, 1):
            ctx.set_option("solver_wave3", w3)
            out.append(pose.ransac_essential(p1, p2, th, confidence=0.999 if iters == 1000 else 1.0, max_iters=iters, refit=refit, seed=seed, ctx=ctx))
        ctx.set_option("solver_wave3", 1)
        a, b = out
        assert a["iters"] == b["iters"] and a["n_inliers"] == b["n_inliers"]
        assert np.array_equal(a["mask"], b["mask"])
        assert np.array_equal(np.asarray(a["E"]).view(np.uint64), np.asarray(b["E"]).view(np.uint64))


def _draw_stats(ctx):
    o = np.zeros(2, np.int64)
    ctx.lib.mlpl_debug_ransac_draw(ctx.handle, o.ctypes.data)
    return int(o[0]), int(o[1])


@pytest.mark.parametrize("n,frac,iters", [(5000, 0.5, 20000), (700, 0.4, 9000), (64, 0.6, 5000), (150, 0.5, 40000), (2000, 0.3, 70000)])
def test_device_drawn_samples_equal_the_host_drawn(ctx, n, frac, iters):
    """Large passes draw their sample tables on the device (scan for samples that redraw, chain from event to event, fill in parallel):
    the runs must equal the host-drawn ones bit for bit -- 70 000 iterations walk three passes with the stream position carried on the
    device; with 64 or 150 correspondences too many samples redraw for the candidate list and the host draws."""
    from matchinglib_poselib_amd import pose, synth

    p1, p2, R, t, mask, th = synth.pose_scene(n, inlier_frac=frac, seed=3 * n + 1)
    out = []
    for dd in (0, 1):
        ctx.set_option("ransac_device_draw", dd)
        try:
            f0, _ = _draw_stats(ctx)
            r = pose.ransac_essential(p1, p2, th, confidence=1.0, max_iters=iters, refit=False, seed=77 + n, ctx=ctx)
            f1, used = _draw_stats(ctx)
        finally:
            ctx.set_option("ransac_device_draw", 1)
        eligible = min(iters, 32768) * 60.0 / n < 3072      # few enough redrawing samples for the candidate list (else the host draws)
        assert used == (dd if eligible else 0) and f1 == f0, (dd, used, f0, f1)     # the device path ran (and did not fall back) exactly when asked
        out.append(r)
    a, b = out
    assert a["iters"] == b["iters"] and a["n_inliers"] == b["n_inliers"]
    assert np.array_equal(a["mask"], b["mask"]) and np.array_equal(np.asarray(a["E"]).view(np.uint64), np.asarray(b["E"]).view(np.uint64))


def test_device_drawing_falls_back_when_samples_redraw_too_often(ctx):
    """n = 70 with the stream cache capped below what 6000 iterations need: not eligible -> host path; and a seed change re-uploads."""
    from matchinglib_poselib_amd import pose, synth

    p1, p2, R, t, mask, th = synth.pose_scene(3000, inlier_frac=0.5, seed=5)
    ref = {}
    for seed in (1, 2, 1):
        ctx.set_option("ransac_device_draw", 0)
        h = pose.ransac_essential(p1, p2, th, confidence=1.0, max_iters=6000, refit=False, seed=seed, ctx=ctx)
        ctx.set_option("ransac_device_draw", 1)
        d = pose.ransac_essential(p1, p2, th, confidence=1.0, max_iters=6000, refit=False, seed=seed, ctx=ctx)
        assert _draw_stats(ctx)[1] == 1
        assert h["n_inliers"] == d["n_inliers"] and np.array_equal(h["mask"], d["mask"])
        ref.setdefault(seed, d["n_inliers"])
        assert ref[seed] == d["n_inliers"]
    ctx.set_option("rand_cache_max", 20000)   # fewer values than 6000 samples need: the call is not eligible
    try:
        d = pose.ransac_essential(p1, p2, th, confidence=1.0, max_iters=6000, refit=False, seed=1, ctx=ctx)
        assert _draw_stats(ctx)[1] == 0 and d["n_inliers"] == ref[1]
    finally:
        ctx.set_option("rand_cache_max", 0)
