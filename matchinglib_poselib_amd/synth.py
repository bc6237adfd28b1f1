"""Seeded synthetic workloads for the BASELINE.json configs; numpy only."""
from __future__ import annotations

import numpy as np


def orb_pair(nq: int, nt: int, nbytes: int = 32, seed: int = 20260102, match_frac: float = 0.5, flip_p: float = 0.08):
    """Binary descriptor sets: `match_frac` of the queries are a train row with Binomial(bits, flip_p) bits
    flipped (true matches), the rest are i.i.d. uniform bytes.  C1: (2048,2048,seed 20260101); C2: (8192,8192,
    seed 20260102)."""
    rng = np.random.default_rng(seed)
    t = rng.integers(0, 256, size=(nt, nbytes), dtype=np.uint8)
    q = rng.integers(0, 256, size=(nq, nbytes), dtype=np.uint8)
    n_match = int(nq * match_frac)
    which = rng.permutation(nq)[:n_match]
    src = rng.integers(0, nt, size=n_match)
    flips = rng.random((n_match, nbytes * 8)) < flip_p
    flip_bytes = np.packbits(flips, axis=1, bitorder="little")
    q[which] = t[src] ^ flip_bytes
    return q, t


def sift_pair(nq: int, nt: int, dim: int = 128, seed: int = 20260104, match_frac: float = 0.5):
    """Integer-valued 0..255 float32 descriptors in OpenCV-SIFT layout (L2-normalised to 512, clipped at 255,
    rounded); half of the queries are perturbed train rows.  C4: (4096,4096,128)."""
    rng = np.random.default_rng(seed)

    def make(n):
        x = rng.gamma(0.6, 1.0, size=(n, dim))
        x = x / np.linalg.norm(x, axis=1, keepdims=True) * 512.0
        return np.clip(np.rint(x), 0, 255)

    t = make(nt)
    q = make(nq)
    n_match = int(nq * match_frac)
    which = rng.permutation(nq)[:n_match]
    src = rng.integers(0, nt, size=n_match)
    q[which] = np.clip(np.rint(t[src] + rng.normal(0, 6.0, size=(n_match, dim))), 0, 255)
    return q.astype(np.float32), t.astype(np.float32)


def _rot(axis, deg):
    axis = np.asarray(axis, float)
    axis = axis / np.linalg.norm(axis)
    a = np.deg2rad(deg)
    K = np.array([[0, -axis[2], axis[1]], [axis[2], 0, -axis[0]], [-axis[1], axis[0], 0]])
    return np.eye(3) + np.sin(a) * K + (1 - np.cos(a)) * (K @ K)


PIX_TO_CAM = 4.0 / (np.sqrt(2.0) * 3200.0)  # f = 800 on both cameras (stereo_pose_refinement.h:280-286)


def pose_scene(n: int = 5000, inlier_frac: float = 0.5, seed: int = 20260103, noise_px: float = 0.3, rot_deg: float = 5.0,
               t_len: float = 1.0):
    """C3 scene: camera-normalised correspondences with known (R, t).

    3-D points uniform in x,y in [-2,2], z in [4,12]; R = rot((0.2,0.9,0.1), 5 deg), t = (1,0.05,-0.02)/|.|;
    Gaussian noise sigma = noise_px * PIX_TO_CAM on both views; outliers = uniform points in the same image box
    paired at random; order shuffled.  Returns p1, p2 (n x 2 float64), R, t, inlier_mask, thresh (0.8 px).
    `rot_deg` / `t_len` (baseline length; the scene is 4..12 deep) make the degenerate motions USAC tests for: t_len = 0 is a pure
    rotation, rot_deg = 0 and t_len = 0 no motion at all."""
    rng = np.random.default_rng(seed)
    R = _rot((0.2, 0.9, 0.1), rot_deg)
    t = np.array([1.0, 0.05, -0.02])
    t = t / np.linalg.norm(t)
    n_in = int(round(n * inlier_frac))
    X = np.stack([rng.uniform(-2, 2, n_in), rng.uniform(-2, 2, n_in), rng.uniform(4, 12, n_in)], axis=1)
    x1 = X[:, :2] / X[:, 2:3]
    X2 = X @ R.T + t_len * t
    x2 = X2[:, :2] / X2[:, 2:3]
    sigma = noise_px * PIX_TO_CAM
    x1 = x1 + rng.normal(0, sigma, x1.shape)
    x2 = x2 + rng.normal(0, sigma, x2.shape)
    n_out = n - n_in
    lo1, hi1 = x1.min(0), x1.max(0)
    lo2, hi2 = x2.min(0), x2.max(0)
    o1 = rng.uniform(lo1, hi1, (n_out, 2))
    o2 = rng.uniform(lo2, hi2, (n_out, 2))
    p1 = np.concatenate([x1, o1])
    p2 = np.concatenate([x2, o2])
    mask = np.concatenate([np.ones(n_in, bool), np.zeros(n_out, bool)])
    perm = rng.permutation(n)
    return (np.ascontiguousarray(p1[perm]), np.ascontiguousarray(p2[perm]), R, t, mask[perm], 0.8 * PIX_TO_CAM)


def stereo_pair(n: int = 2000, seed: int = 20260200, inlier_frac: float = 0.5, f: float = 800.0, cx: float = 320.0,
                cy: float = 240.0, nbytes: int = 32, flip_p: float = 0.04, unmatched_frac: float = 0.0):
    """One synthetic image pair for the whole pipeline (C5 unit): keypoints in pixels (float32), binary descriptors whose
    nearest neighbours are the true correspondences (train side shuffled), and the ground-truth pose.
    Returns dict(desc1, desc2, kp1, kp2, K, R, t, train_of_query)."""
    p1, p2, R, t, mask, th = pose_scene(n, inlier_frac=inlier_frac, seed=seed)
    rng = np.random.default_rng(seed + 7)
    perm = rng.permutation(n)                      # train row perm[i] belongs to query i
    kp1 = (p1 * f + np.array([cx, cy])).astype(np.float32)
    kp2_q = (p2 * f + np.array([cx, cy])).astype(np.float32)
    kp2 = np.empty_like(kp2_q)
    kp2[perm] = kp2_q
    d2 = rng.integers(0, 256, size=(n, nbytes), dtype=np.uint8)
    flips = np.packbits(rng.random((n, nbytes * 8)) < flip_p, axis=1, bitorder="little")
    d1 = d2[perm] ^ flips
    if unmatched_frac > 0:  # queries without a true neighbour: they fail the ratio test, so the match count varies with the seed
        lost = rng.random(n) < unmatched_frac
        d1[lost] = rng.integers(0, 256, size=(int(lost.sum()), nbytes), dtype=np.uint8)
    K = np.array([f, f, cx, cy], np.float64)
    return dict(desc1=d1, desc2=d2, kp1=kp1, kp2=kp2, K=K, R=R, t=t, train_of_query=perm.astype(np.int32))
