"""smallest_eigvec9_wave (csrc/ransac_5pt.hip): the inverse iteration that gives the re-weighted 9 x 9 fits (USAC REF_WEIGHTS,
robustEssentialRefine) their one eigenvector, against numpy and against the Jacobi decomposition it replaced -- on covariance matrices of
correspondences (what the kernels feed it), at several scales, from cold and warm starts, and on matrices it must NOT settle on (a double
smallest eigenvalue: the callers then take the Jacobi path)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu


def _run(ctx, G, start=None):
    G = np.ascontiguousarray(G, np.float64)
    n = len(G)
    out, jac = np.zeros((n, 12)), np.zeros((n, 10))
    st = None if start is None else np.ascontiguousarray(start, np.float64)
    rc = ctx.lib.mlpl_debug_eig9(ctx.handle, G.ctypes.data, None if st is None else st.ctypes.data, n, out.ctypes.data, jac.ctypes.data)
    assert rc == 0
    return out, jac


def _cov(rng, n, noise, scale=1.0, inlier_frac=1.0, rows=False):
    """Covariance of the data-matrix rows (x2 x1, x2 y1, x2, y2 x1, ..., 1) of n correspondences of a random essential matrix."""
    from matchinglib_poselib_amd import synth

    p1, p2, R, t, mask, th = synth.pose_scene(n, inlier_frac=inlier_frac, seed=int(rng.integers(1 << 30)), noise_px=noise)
    x1, y1, x2, y2 = p1[:, 0], p1[:, 1], p2[:, 0], p2[:, 1]
    A = np.stack([x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, np.ones(n)], 1) * scale
    return (A.T @ A, A) if rows else A.T @ A


def test_agrees_with_numpy_and_with_the_jacobi_decomposition(ctx):
    rng = np.random.default_rng(5)
    # 200 covariances of inlier sets (what the fits see: it must settle on nearly all) + 100 with 10-50 % outliers in them (a smallest
    # eigenvalue that is not separated: it may refuse -- the callers then take the Jacobi path -- but must be right where it settles)
    G = np.stack([_cov(rng, int(rng.integers(8, 400)), float(rng.uniform(0.0, 1.0)), 10.0 ** rng.uniform(-3, 3)) for _ in range(200)] +
                 [_cov(rng, int(rng.integers(50, 400)), 0.3, 1.0, float(rng.uniform(0.5, 0.9))) for _ in range(100)])
    out, jac = _run(ctx, G)
    settled = out[:, 0] > 0
    assert settled[:200].mean() > 0.9
    for b in range(len(G)):
        w, V = np.linalg.eigh(G[b])
        gap = (w[1] - w[0]) / max(w[-1], 1e-300)
        vj = jac[b, 1:]
        assert min(np.abs(vj - V[:, 0]).max(), np.abs(vj + V[:, 0]).max()) < 1e-9 / max(gap, 1e-7) * 1e-6 + 1e-12 or gap < 1e-10
        if settled[b]:
            x = out[b, 3:]
            assert abs(np.linalg.norm(x) - 1) < 1e-14
            d = min(np.abs(x - V[:, 0]).max(), np.abs(x + V[:, 0]).max())
            assert d < 2e-16 / max(gap, 1e-12) + 1e-13, (b, d, gap)
            dj = min(np.abs(x - vj).max(), np.abs(x + vj).max())
            assert dj < 4e-16 / max(gap, 1e-12) + 1e-13, (b, dj, gap)
            assert abs(out[b, 1] - w[0]) <= 1e-14 * w[-1] + 1e-300


def test_a_warm_start_takes_fewer_steps_and_gives_the_same_vector(ctx):
    rng = np.random.default_rng(6)
    pairs = [_cov(rng, 200, 0.3, rows=True) for _ in range(64)]
    G = np.stack([g for g, a in pairs])
    cold, _ = _run(ctx, G)
    # the same correspondences re-weighted (weights 0.8 ... 1.2) = the next fit of a chain
    G2 = np.stack([a.T @ (a * rng.uniform(0.8, 1.2, (len(a), 1))) for g, a in pairs])
    G2 = (G2 + G2.transpose(0, 2, 1)) / 2
    warm, _ = _run(ctx, G2, start=cold[:, 3:])
    cold2, _ = _run(ctx, G2)
    ok = (cold[:, 0] > 0) & (warm[:, 0] > 0) & (cold2[:, 0] > 0)
    assert ok.mean() > 0.9
    assert warm[ok, 0].mean() <= cold2[ok, 0].mean()
    d = np.minimum(np.abs(warm[ok, 3:] - cold2[ok, 3:]).max(1), np.abs(warm[ok, 3:] + cold2[ok, 3:]).max(1))
    assert d.max() < 1e-10


def test_refuses_a_double_smallest_eigenvalue_and_bad_input(ctx):
    rng = np.random.default_rng(7)
    Q, _ = np.linalg.qr(rng.standard_normal((9, 9)))
    w = np.array([1e-3, 1e-3, 0.2, 0.5, 1, 2, 3, 4, 5.0])          # the eigenvector of the smallest eigenvalue is not defined
    G_double = Q @ np.diag(w) @ Q.T
    G_zero = np.zeros((9, 9))
    G_nan = np.full((9, 9), np.nan)
    G_neg = -np.eye(9)
    out, jac = _run(ctx, np.stack([G_double, G_zero, G_nan, G_neg]))
    assert out[1, 0] == 0 and out[2, 0] == 0 and out[3, 0] == 0
    # the double eigenvalue: either refused, or -- the iteration converges INSIDE the eigenspace -- a unit vector of that eigenspace
    if out[0, 0] > 0:
        x = out[0, 3:]
        assert np.linalg.norm(G_double @ x - 1e-3 * x) < 1e-12


def test_never_settles_on_the_second_smallest_eigenvector(ctx):
    """convergence was accepted from the iterate differences alone.  Matrices whose smallest eigenvector has NO component along
    the cold start (entries sum to zero) or along a given warm start (the second-smallest eigenvector itself, exactly): the routine must
    return the smallest eigenvector or refuse (steps = 0: the callers then take the Jacobi path) -- never another eigenvector."""
    rng = np.random.default_rng(11)
    Gs, starts, truth = [], [], []
    for k in range(200):
        M = rng.standard_normal((9, 9))
        M[:, 0] -= M[:, 0].mean()                      # first column orthogonal to the constant vector
        M[:, 0] = np.round(M[:, 0] * 64) / 64          # ... exactly (small dyadic entries: the sum is exact in floating point)
        M[:, 0] -= np.round(M[:, 0].sum() * 64) / 64 / 9 * 0
        if abs(M[:, 0].sum()) > 0:
            M[8, 0] -= M[:, 0].sum()
        Q, _ = np.linalg.qr(M)                         # Q[:, 0] is parallel to M[:, 0]
        w = np.sort(10.0 ** rng.uniform(-6, 1, 9))
        w[1] = max(w[1], w[0] * rng.choice([1.5, 10.0, 1e3]))
        w[2:] = np.maximum(w[2:], w[1] * 1.2)
        G = Q @ np.diag(w) @ Q.T
        G = (G + G.T) / 2
        Gs.append(G)
        truth.append(Q[:, 0])
        starts.append(Q[:, 1] if k % 2 else np.full(9, 1 / 3))   # warm start ON the second eigenvector / the cold start
    G = np.stack(Gs)
    out_cold, jac = _run(ctx, G)
    out_warm, _ = _run(ctx, G, start=np.stack(starts))
    for out in (out_cold, out_warm):
        for b in range(len(G)):
            if out[b, 0] > 0:
                w, V = np.linalg.eigh(G[b])
                x = out[b, 3:]
                d = min(np.abs(x - V[:, 0]).max(), np.abs(x + V[:, 0]).max())
                gap = (w[1] - w[0]) / w[-1]
                assert d < 1e-15 / gap + 1e-12, (b, d, gap, out[b, 0])
                assert abs(out[b, 1] - w[0]) <= 1e-13 * w[-1]
    # and it still settles on ordinary inputs (the certificates must not turn the fast path off)
    assert (out_cold[:, 0] > 0).mean() > 0.5
