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
