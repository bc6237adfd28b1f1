"""Golden vectors G2 (squared L2), G4 (glibc stream + runRANSAC trace) and G5 (cheirality) of SURVEY 8(c): the oracle against
them on the CPU, the HIP path against them on the GPU.  Generator: tests/golden/make_golden.py (data only)."""
import os
import zlib

import numpy as np
import pytest

from matchinglib_poselib_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def e_dist(a, b):
    return min(np.linalg.norm(a - b), np.linalg.norm(a + b))


# ---------------------------------------------------------------- CPU: oracle vs goldens
def test_g2_oracle_l2(oracle):
    g = np.load(os.path.join(GOLD, "l2_integer_sift.npz"))
    for tag in ("sift128", "d64"):
        idx, dist = oracle.knn_l2sq(g[f"{tag}_q"], g[f"{tag}_t"])
        assert np.array_equal(idx, g[f"{tag}_idx"]) and dist.tobytes() == g[f"{tag}_d2"].tobytes()
        tf = g[f"{tag}_tie_free"]
        assert np.array_equal(idx[tf], g[f"{tag}_nms_idx"][tf])          # the reference's vendored NMSLIB agrees where no tie
        rc, m = oracle.get_matches_linear(len(idx), len(g[f"{tag}_t"]), g[f"{tag}_q"], g[f"{tag}_t"])
        assert rc == 0 and np.array_equal(m["queryIdx"], g[f"{tag}_match_q"])


def test_g4_glibc_stream_fixture(oracle):
    g = np.load(os.path.join(GOLD, "ransac_trace.npz"))
    for k, seed in enumerate(g["rand_seeds"]):
        st = oracle.rand_stream(int(seed), 100000).astype(np.int32)
        assert np.array_equal(st[:2000], g["rand_head"][k])
        assert zlib.crc32(st.tobytes()) == int(g["rand_crc32_100000"][k])


@pytest.mark.parametrize("tag", ["ref", "full"])
def test_g4_oracle_ransac_trace(oracle, tag):
    g = np.load(os.path.join(GOLD, "ransac_trace.npz"))
    n, conf, iters, seed, th = g[f"{tag}_params"]
    p1, p2, R, t, mask, th_s = synth.pose_scene(int(n), seed=20260103)
    assert th_s == th
    o = oracle.ransac_essential(p1, p2, th, confidence=conf, max_iters=int(iters), lesqu=False, seed=int(seed), trace=True)
    k = int(g[f"{tag}_iters"])
    assert o["iters"] == k and o["n_inliers"] == int(g[f"{tag}_n_inliers"])
    assert np.array_equal(o["E"], g[f"{tag}_E"]) and np.array_equal(np.packbits(o["mask"]), g[f"{tag}_mask"])
    tr = o["trace"]
    assert np.array_equal(np.array([list(tr[i].idx) for i in range(k)]), g[f"{tag}_idx"])
    assert np.array_equal(np.array([tr[i].nmodels for i in range(k)]), g[f"{tag}_nmodels"])
    assert np.array_equal(np.array([list(tr[i].good) for i in range(k)]), g[f"{tag}_good"])
    assert np.array_equal(np.array([list(tr[i].err_sum) for i in range(k)]), g[f"{tag}_err_sum"])
    assert np.array_equal(np.array([tr[i].niters_after for i in range(k)]), g[f"{tag}_niters_after"])
    assert np.array_equal(np.array([tr[i].best_taken for i in range(k)]), g[f"{tag}_best_taken"])


def test_g5_oracle_cheirality_and_ground_truth(oracle):
    g = np.load(os.path.join(GOLD, "cheirality.npz"))
    for tag in ("a", "b"):
        p1, p2 = g[f"{tag}_p1"], g[f"{tag}_p2"]
        for variant in ("plain", "neg_scaled_masked"):
            key = f"{tag}_{variant}"
            mk = g[f"{tag}_mask_in"] if variant != "plain" else None
            good, R, t, Q, mo = oracle.recover_pose(g[f"{key}_E"], p1, p2, 50.0, mk)
            assert good == int(g[f"{key}_good"]) and np.array_equal(R, g[f"{key}_R"]) and np.array_equal(t, g[f"{key}_t"])
            assert np.array_equal(Q, g[f"{key}_Q"], equal_nan=True)
            # analytic truth: the known pose, and exactly the points in front of both cameras and nearer than dist pass
            assert np.abs(R - g[f"{tag}_R_true"]).max() < 1e-9 and np.abs(t - g[f"{tag}_t_true"]).max() < 1e-9
            ok = g[f"{tag}_truth_ok"].copy()
            X = g[f"{tag}_X_true"]
            sure = np.abs(X[:, 2] - 50.0) > 1e-6
            sure[60:80] = False   # the deliberately inconsistent pairs: no analytic answer, the golden outputs pin them
            if mk is not None:
                assert np.array_equal((mo != 0)[sure], (ok & (mk != 0))[sure])
                assert good == int((mo != 0).sum())
            else:
                assert abs(good - int(ok.sum())) <= int((~sure).sum())
            fin = ok & sure
            assert np.allclose(Q[fin], X[fin], rtol=1e-7, atol=1e-7)     # triangulated points = the scene's 3-D points


# ---------------------------------------------------------------- GPU: HIP path vs goldens
@pytest.mark.gpu
def test_g2_gpu_l2(ctx):
    import matchinglib_poselib_amd as mpa
    g = np.load(os.path.join(GOLD, "l2_integer_sift.npz"))
    for mode in (0, 1):   # automatic (fp16 MFMA path on this integer-valued data) and the exact fp32 kernel
        ctx.lib.mlpl_set_l2_path(ctx.handle, mode)
        try:
            for tag in ("sift128", "d64"):
                idx, dist = mpa.knn_l2sq(g[f"{tag}_q"], g[f"{tag}_t"], ctx=ctx)
                assert np.array_equal(idx, g[f"{tag}_idx"]) and dist.tobytes() == g[f"{tag}_d2"].tobytes(), (mode, tag)
                err, m = mpa.getMatches([None] * len(idx), [None] * len(g[f"{tag}_t"]), g[f"{tag}_q"], g[f"{tag}_t"],
                                        matcher_name="LINEAR", ctx=ctx)
                assert err == 0 and np.array_equal(m["queryIdx"], g[f"{tag}_match_q"])
        finally:
            ctx.lib.mlpl_set_l2_path(ctx.handle, 0)


@pytest.mark.gpu
@pytest.mark.parametrize("tag", ["ref", "full"])
def test_g4_gpu_ransac_trace(ctx, tag):
    """Final state and, per iteration of the golden trace, the solver's solution count and the scored (count, error sum) sets."""
    from matchinglib_poselib_amd import pose
    g = np.load(os.path.join(GOLD, "ransac_trace.npz"))
    n, conf, iters, seed, th = g[f"{tag}_params"]
    p1, p2, R, t, mask, th_s = synth.pose_scene(int(n), seed=20260103)
    r = pose.ransac_essential(p1, p2, th, confidence=conf, max_iters=int(iters), refit=False, seed=int(seed), ctx=ctx)
    assert r["iters"] == int(g[f"{tag}_iters"]) and r["n_inliers"] == int(g[f"{tag}_n_inliers"])
    assert np.array_equal(np.packbits(r["mask"]), g[f"{tag}_mask"]) and e_dist(r["E"], g[f"{tag}_E"]) < 1e-8
    rf = pose.ransac_essential(p1, p2, th, confidence=conf, max_iters=int(iters), refit=True, seed=int(seed), ctx=ctx)
    assert abs(rf["n_inliers"] - int(g[f"{tag}_refit_n_inliers"])) <= 2 and e_dist(rf["E"], g[f"{tag}_refit_E"]) < 1e-7
    assert (np.packbits(rf["mask"]) != g[f"{tag}_refit_mask"]).sum() <= 2
    samples = g[f"{tag}_idx"]
    E, nm = pose.solve_5pt(p1, p2, samples, ctx=ctx)
    assert np.array_equal(nm, g[f"{tag}_nmodels"])
    flat = np.concatenate([E[s, :nm[s]] for s in range(len(samples))])
    good, esum = pose.score_models(p1, p2, flat, th, ctx=ctx)
    pos, loose = 0, 0
    for s in range(len(samples)):
        k = int(nm[s])
        gg = np.sort(good[pos:pos + k])
        assert np.array_equal(gg, np.sort(g[f"{tag}_good"][s, :k])), s
        # the error sums depend on E to the last bit: the models agree to ~1e-12, so the sums agree to ~1e-9 relative -- except where
        # one correspondence has a vanishing Sampson denominator for a (bad) model and dominates its sum
        a, b = np.sort(esum[pos:pos + k]), np.sort(g[f"{tag}_err_sum"][s, :k])
        loose += not np.allclose(a, b, rtol=1e-6)
        pos += k
    assert loose <= max(2, len(samples) // 50), loose


@pytest.mark.gpu
def test_g5_gpu_cheirality(ctx):
    from matchinglib_poselib_amd import pose
    g = np.load(os.path.join(GOLD, "cheirality.npz"))
    for tag in ("a", "b"):
        p1, p2 = g[f"{tag}_p1"], g[f"{tag}_p2"]
        for variant in ("plain", "neg_scaled_masked"):
            key = f"{tag}_{variant}"
            mk = g[f"{tag}_mask_in"] if variant != "plain" else None
            good, R, t, Q, mo = pose.getPoseTriangPts(g[f"{key}_E"], p1, p2, mk, 50.0, ctx=ctx)
            assert good == int(g[f"{key}_good"])
            assert np.abs(R - g[f"{key}_R"]).max() < 1e-12 and np.abs(t.ravel() - g[f"{key}_t"]).max() < 1e-12
            assert np.abs(R - g[f"{tag}_R_true"]).max() < 1e-9
            if mk is not None:
                assert np.array_equal(mo, g[f"{key}_mask_out"])
            Qo = g[f"{key}_Q"]
            fin = np.isfinite(Qo).all(axis=1) & (np.abs(Qo).max(axis=1) < 1e6)
            assert np.allclose(Q[fin], Qo[fin], rtol=1e-9, atol=1e-9)
