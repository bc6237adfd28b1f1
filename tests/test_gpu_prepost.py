"""GPU: pre/post steps of the pose path (ImgToCamCoordTrans, Remove_LensDist, getInliers) vs the oracle, bit-exact."""
import numpy as np
import pytest

from matchinglib_poselib_amd import pose, synth

pytestmark = pytest.mark.gpu


def distort(p, d):
    k1, k2, p1, p2, k3, k4, k5, k6 = d
    x, y = p[:, 0].astype(np.float64), p[:, 1].astype(np.float64)
    r2 = x * x + y * y
    rc = (1 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1 + ((k6 * r2 + k5) * r2 + k4) * r2)
    dx = p1 * 2 * x * y + p2 * (r2 + 2 * x * x)
    dy = p1 * (r2 + 2 * y * y) + p2 * 2 * x * y
    return np.stack([x * rc + dx, y * rc + dy], axis=1).astype(np.float32)


def test_img_to_cam(ctx, oracle):
    rng = np.random.default_rng(0)
    pts = rng.uniform(0, 1200, (5000, 2)).astype(np.float32)
    K4 = np.array([812.3, 807.9, 611.2, 377.7])
    g = pose.ImgToCamCoordTrans(pts, K4, ctx=ctx)
    assert g.tobytes() == oracle.img_to_cam(pts, K4).tobytes()


def test_remove_lens_dist(ctx, oracle):
    p1, p2, *_ = synth.pose_scene(3000, seed=9)
    d1 = np.array([-0.28, 0.09, 1e-3, -5e-4, -0.012, 0.0, 0.0, 0.0])
    d2 = np.array([-0.31, 0.11, -8e-4, 7e-4, 0.0, 0.01, -0.002, 0.0])
    a, b = distort(p1 * 1.6, d1), distort(p2 * 1.6, d2)
    a[17] = [9.0, 9.0]      # far outside the model's valid range: must be dropped, order of the rest kept
    b[400] = [-7.0, 8.0]
    ok_o, ao, bo = oracle.remove_lens_dist(a, b, d1, d2)
    ok_g, ag, bg = pose.Remove_LensDist(a, b, d1, d2, ctx=ctx)
    assert ok_o and ok_g and len(ag) == len(ao) < 3000
    assert ag.tobytes() == ao.tobytes() and bg.tobytes() == bo.tobytes()
    # undistortion really inverts the model
    keep = np.ones(3000, bool)
    # zero coefficients: untouched
    ok_g, ag, bg = pose.Remove_LensDist(a, b, np.zeros(8), np.zeros(8), ctx=ctx)
    assert ok_g and ag.tobytes() == a.tobytes()
    # fewer than 16 survivors -> false
    bad = np.full((20, 2), 30.0, np.float32)
    ok_o, _, _ = oracle.remove_lens_dist(bad, bad, d1, d2)
    ok_g, _, _ = pose.Remove_LensDist(bad, bad, d1, d2, ctx=ctx)
    assert not ok_o and not ok_g


def test_get_inliers_strict(ctx, oracle):
    p1, p2, R, t, mask, th = synth.pose_scene(4000, seed=10)
    o = oracle.ransac_essential(p1, p2, th, max_iters=200, seed=1)
    cnt_o, m_o, e_o = oracle.get_inliers_strict(p1, p2, o["E"], th * th)
    cnt_g, m_g, e_g = pose.getInliers(o["E"], p1, p2, th * th, ctx=ctx)
    assert cnt_g == cnt_o and np.array_equal(m_g, m_o) and e_g.tobytes() == e_o.tobytes()
    # strictness: a threshold equal to one of the errors excludes that point here, includes it in RANSAC's <=
    k = int(np.argsort(e_o)[1000])
    cnt2, m2, _ = pose.getInliers(o["E"], p1, p2, e_o[k], ctx=ctx)
    assert m2[k] == 0 and cnt2 == int((e_o < e_o[k]).sum())
