"""StereoRefine over sequences of stereo frames: the C++ drop-in (GPU estimators, tests/cpp/stereo_refine_driver.cpp) against the CPU
restatement of the reference's state machine (tests/stereo_refine_oracle.py over the CPU oracle).  The decisions (return codes, pool size,
counters, stability flags) must be identical frame by frame and the poses equal within 1e-6."""
import os
import subprocess

import numpy as np
import pytest

from stereo_refine_oracle import Cfg, StereoRefineOracle

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "stereo_refine_driver")
K = np.array([800.0, 800.0, 640.0, 480.0])   # principal point such that every keypoint of the scenes lies inside IMG_SIZE
IMG_SIZE = (1408, 1056)                      # cfg_usac.imgSize: StereoRefine::checkPoolSize indexes an image-sized table with the pool's left keypoints


def rot(rx, ry, rz):
    cx, sx, cy, sy, cz, sz = np.cos(rx), np.sin(rx), np.cos(ry), np.sin(ry), np.cos(rz), np.sin(rz)
    return (np.array([[cz, -sz, 0], [sz, cz, 0], [0, 0, 1]]) @ np.array([[cy, 0, sy], [0, 1, 0], [-sy, 0, cy]]) @
            np.array([[1, 0, 0], [0, cx, -sx], [0, sx, cx]]))


POSE_A = (rot(0.02, -0.05, 0.01), np.array([-1.0, 0.03, 0.05]))
POSE_B = (rot(-0.06, 0.10, -0.04), np.array([-0.9, -0.25, 0.30]))


def frame(rng, n, pose, outlier_frac, noise_px=0.15, dist=None):
    """n matches of one image pair in pixels: (kp1 n x 3, kp2 n x 3, descriptor distances n), float32."""
    R, t = pose
    X = np.stack([rng.uniform(-2.2, 2.2, n), rng.uniform(-1.6, 1.6, n), rng.uniform(4.0, 12.0, n)], 1)
    Y = X @ R.T + t
    x1, x2 = X[:, :2] / X[:, 2:], Y[:, :2] / Y[:, 2:]
    if dist is not None:  # forward radial distortion, first two coefficients only
        for x, d in ((x1, dist[0]), (x2, dist[1])):
            r2 = (x * x).sum(1, keepdims=True)
            x *= 1.0 + d[0] * r2 + d[1] * r2 * r2
    p1 = x1 * K[:2] + K[2:] + rng.normal(0, noise_px, (n, 2))
    p2 = x2 * K[:2] + K[2:] + rng.normal(0, noise_px, (n, 2))
    bad = rng.random(n) < outlier_frac
    p2[bad] = np.stack([rng.uniform(320, 960, bad.sum()), rng.uniform(240, 720, bad.sum())], 1)
    dd = np.where(bad, rng.uniform(50, 100, n), rng.uniform(15, 70, n))
    kp1 = np.concatenate([p1, rng.uniform(0.001, 0.1, (n, 1))], 1).astype(np.float32)
    kp2 = np.concatenate([p2, rng.uniform(0.001, 0.1, (n, 1))], 1).astype(np.float32)
    return kp1, kp2, dd.astype(np.float32)


def drifted(scale):
    return (POSE_A[0] @ rot(0.004 * scale, 0.003 * scale, 0), POSE_A[1] + scale * np.array([0, 0.02, 0.01]))


# name -> (rng seed, configuration, frames as (count, pose, outlier fraction)); `want` = branches of the state machine the sequence must reach
SEQUENCES = {
    "steady": dict(seed=5, cfg={}, frames=[(700, "A", 0.25)] * 10, want=["init", "pool"]),
    "stable": dict(seed=5, cfg=dict(absThRankingStable=0.4, minNormDistStable=0.1), frames=[(700, "A", 0.25)] * 10, want=["pool"]),
    "bad_pairs_restore_and_skip": dict(seed=5, cfg=dict(maxSkipPairs=1), frames=[(600, "A", 0.2)] * 3 + [(600, "B", 0.75)] * 3 + [(600, "A", 0.2)] * 2,
                                       want=["restore_last", "restore_last+skip_reinit"]),
    "uncertain_pairs": dict(seed=12, cfg={}, frames=[(600, "A", 0.2)] * 3 + [(600, "B", 0.5)] * 2 + [(600, "A", 0.2)] * 2, want=["pool_only"]),
    "pose_change": dict(seed=7, cfg={}, frames=[(600, "A", 0.2)] * 4 + [(600, "B", 0.2)] * 4, want=["pose_changed"]),
    "mediocre_pair": dict(seed=8, cfg={}, frames=[(600, "A", 0.15)] * 3 + [(600, "A", 0.6)] + [(600, "A", 0.15)] * 2, want=[]),
    "drift_loses_the_pair": dict(seed=5, cfg=dict(relInlRatThLast=0.6), frames=[(600, "A", 0.2)] * 2 + [(600, 0.6, 0.2)] * 4, want=["pool+pair_lost"]),
    "small_pool": dict(seed=6, cfg=dict(maxPoolCorrespondences=1300, raiseSkipCnt=(1 | (1 << 4)), absThRankingStable=0.05, minNormDistStable=0.6),
                       frames=[(650, "A", 0.2)] * 12, want=["pool+shrink"]),
    "lmeds": dict(seed=9, cfg=dict(RobMethod="LMEDS"), frames=[(500, "A", 0.2)] * 5, want=["pool"]),
    "arrsac": dict(seed=13, cfg=dict(RobMethod="ARRSAC"), frames=[(500, "A", 0.2)] * 4 + [(500, "B", 0.3)] * 3, want=["pool"]),
    "arrsac_refined": dict(seed=14, cfg=dict(RobMethod="ARRSAC", refineRTold=True), frames=[(450, "A", 0.25)] * 5, want=["pool"]),
    "pool_refined": dict(seed=15, cfg=dict(checkPoolPoseRobust=3, refineRTold_CorrPool=True), frames=[(400, "A", 0.2)] * 12,
                         want=["pool+refined", "pool+robust"]),
    "pool_refined_small_pool": dict(seed=16, cfg=dict(checkPoolPoseRobust=2, refineRTold_CorrPool=True, maxPoolCorrespondences=1500),
                                    frames=[(500, "A", 0.25)] * 14, want=["pool+refined"]),
    "auto_threshold": dict(seed=17, cfg=dict(autoTH=True, RobMethod="ARRSAC", th_pix_user=0.5), frames=[(500, "A", 0.25)] * 5, want=["pool"]),
    "auto_threshold_few_matches": dict(seed=18, cfg=dict(autoTH=True, RobMethod="ARRSAC", useRANSAC_fewMatches=True, minStartAggInlRat=0.5),
                                       frames=[(90, "A", 0.7)] + [(90, "A", 0.2)] * 2 + [(400, "A", 0.2)] * 3, want=["init"]),
    "lens_distortion_refit": dict(seed=10, cfg=dict(refineRTold=True, th_pix_user=1.0), frames=[(600, "A", 0.2)] * 6, want=["pool"],
                                  dist=(np.array([-0.12, 0.03, 0, 0, 0, 0, 0, 0.0]), np.array([-0.10, 0.02, 0, 0, 0, 0, 0, 0.0]))),
    "weak_start_few_matches": dict(seed=11, cfg=dict(useRANSAC_fewMatches=True, RobMethod="LMEDS", minStartAggInlRat=0.5),
                                   frames=[(90, "A", 0.7)] + [(90, "A", 0.2)] * 3 + [(400, "A", 0.2)] * 2, want=["init"]),
}


def sequence(name):
    spec = SEQUENCES[name]
    rng = np.random.default_rng(spec["seed"])
    cfg = Cfg(**spec["cfg"])
    dist = spec.get("dist")
    pose = lambda p: POSE_A if p == "A" else POSE_B if p == "B" else drifted(p)
    frames = [frame(rng, n, pose(p), out, dist=dist) for n, p, out in spec["frames"]]
    return cfg, cfg.RobMethod, dist, frames


def run_oracle(oracle, name, seed=777):
    cfg, method, dist, frames = sequence(name)
    d0, d1 = (np.zeros(8), np.zeros(8)) if dist is None else dist
    sr = StereoRefineOracle(oracle, cfg, K, K, d0, d1, seed)
    sr.img_size = IMG_SIZE
    out = []
    run_oracle.last = sr
    for kp1, kp2, dd in frames:
        rc = sr.add(kp1, kp2, dd)
        out.append(dict(rc=rc, inl=sr.nr_inliers, corrs=sr.nr_corrs, pool=len(sr.pool), est=sr.nr_est, skip=sr.skip,
                        stable=int(sr.stable), ml=int(sr.ml_stable), hist=len(sr.poses), branch=sr.branch,
                        E=None if sr.E is None else sr.E.copy(), R=None if sr.R is None else sr.R.copy(),
                        t=None if sr.t is None else sr.t.copy(), Eml=None if sr.E_ml is None else sr.E_ml.copy()))
    return out


def run_gpu(name, tmp_path, seed=777, options=None):
    cfg, method, dist, frames = sequence(name)
    d0, d1 = (np.zeros(8), np.zeros(8)) if dist is None else dist
    fin, fout = tmp_path / f"{name}.in", tmp_path / f"{name}.out"
    with open(fin, "wb") as f:
        np.array([len(frames), {"RANSAC": 0, "LMEDS": 1, "ARRSAC": 2}[method]], np.int32).tofile(f)
        np.array([seed], np.uint32).tofile(f)
        K.tofile(f)
        K.tofile(f)
        np.asarray(d0, np.float64).tofile(f)
        np.asarray(d1, np.float64).tofile(f)
        cfg.as_doubles().tofile(f)
        for kp1, kp2, dd in frames:
            np.array([len(dd)], np.int32).tofile(f)
            kp1.tofile(f)
            kp2.tofile(f)
            dd.tofile(f)
    env = dict(os.environ)
    if options:
        env["MLPL_OPTIONS"] = options
    subprocess.run([EXE, str(fin), str(fout)], check=True, timeout=300, stdout=subprocess.DEVNULL, env=env)
    raw = np.fromfile(fout, np.uint8)
    rec = np.dtype([("st", np.int32, 9), ("E", np.float64, 9), ("R", np.float64, 9), ("t", np.float64, 3), ("Eml", np.float64, 9)])
    return np.frombuffer(raw.tobytes(), rec)


def same_up_to_sign(a, b, tol):
    return min(np.abs(a - b).max(), np.abs(a + b).max()) < tol


@pytest.mark.parametrize("name", list(SEQUENCES))
def test_oracle_state_machine_runs_the_expected_branches(oracle, name):
    """CPU only: the restated state machine reaches the branches each sequence is built for."""
    out = run_oracle(oracle, name)
    seen = [o["branch"] for o in out]
    for b in SEQUENCES[name]["want"]:
        assert b in seen, (b, seen)
    if name == "steady":
        assert [o["rc"] for o in out] == [0] * 10 and [o["est"] for o in out] == list(range(1, 11))
        assert all(out[i]["pool"] > out[i - 1]["pool"] for i in range(1, 10))
    elif name == "stable":
        assert sum(o["stable"] for o in out) >= 3
    elif name == "bad_pairs_restore_and_skip":
        assert out[3]["skip"] == 1 and out[3]["est"] == 3 and out[4]["est"] == 1
    elif name == "pose_change":
        assert out[4]["est"] == 1 and out[4]["hist"] == 1 and out[7]["est"] == 4
    elif name == "drift_loses_the_pair":
        assert -3 in [o["rc"] for o in out] and any(o["pool"] == 0 for o in out)
    elif name == "small_pool":
        assert max(o["pool"] for o in out) <= 1300 and out[-1]["stable"] == 1  # stable through the error-range overlap fallback
        log = run_oracle.last.shrink_log      # checkPoolSize ran its density-image thinning: dilate / erode rounds and the weight-ordered rest
        assert len(log) >= 5 and all(e["density_rounds"] >= 1 for e in log) and any(e["by_weight"] for e in log), log
        assert any(e["density_rounds"] >= 2 for e in log) or any(e["shared_pixels"] for e in log), log
    elif name == "weak_start_few_matches":
        assert out[0]["est"] == 0 and out[0]["pool"] == 0 and out[1]["est"] == 1


def constraint_residual(E):
    """max |2 E E^T E - tr(E E^T) E|, |det E| of a unit-norm model: zero for an exact essential matrix."""
    E = E / np.linalg.norm(E)
    return max(np.abs(2 * E @ E.T @ E - np.trace(E @ E.T) * E).max(), abs(np.linalg.det(E)))


@pytest.mark.gpu
@pytest.mark.parametrize("polish", [0, 1])
@pytest.mark.parametrize("name", list(SEQUENCES))
def test_stereo_refine_sequence_matches_the_cpu_state_machine(oracle, tmp_path, name, polish):
    """Frame by frame: identical decisions, identical counts, poses within 1e-6.  One documented exception: when the CPU solver's
    winning model is itself off the essential-matrix constraints (residual > 1e-9; its elimination is ill conditioned for that sample,
    tests/test_gpu_baseline_configs.py), the GPU model is the accurate one (residual ~1e-16 with and without the polish step) and a
    correspondence at the threshold can change sides -- `pose_change` frame 2 is such a case (tools/debug_sr.py).  From there on the
    counts may differ by a few; the decisions still may not."""
    assert os.path.exists(EXE), "run __graft_entry__.build()"
    want = run_oracle(oracle, name)
    got = run_gpu(name, tmp_path, options=f"solver_polish={polish}")
    assert len(got) == len(want)
    drift, cpu_off = 0, False
    for i, (g, w) in enumerate(zip(got, want)):
        st = dict(zip(["rc", "inl", "corrs", "pool", "est", "skip", "stable", "ml", "hist"], g["st"].tolist()))
        arrsac = name.startswith("arrsac") or name.startswith("auto_threshold")   # its winner may be an 8-point fit or a refined matrix: no exact essential matrix to expect
        if w["E"] is not None and constraint_residual(w["E"]) > 1e-9 and not arrsac:
            cpu_off = True
        for k in ("rc", "est", "skip", "stable", "ml", "hist"):
            assert st[k] == w[k], (name, i, k, st, {k: w[k] for k in st})
        # ARRSAC draws its hypotheses from the pool's FIRST 100 correspondences, which do not change from frame to frame: successive frames
        # often re-find the same 5-point model, the pool's error history then holds two roundings of the same number, and
        # compareCorrespondences' "last error > previous error" (:2491-2495) is decided by the last bit (tools/sr_compare.py <seq> hybrid)
        for k in ("inl", "corrs", "pool"):
            assert abs(st[k] - w[k]) <= (5 if (cpu_off or arrsac) else 0), (name, i, k, st, {k: w[k] for k in st})
            drift = max(drift, abs(st[k] - w[k]))
        tol = 1e-6 if not cpu_off else 1e-2
        if w["E"] is not None:
            unit = lambda a: a / np.linalg.norm(a)  # noqa: E731
            assert same_up_to_sign(unit(g["E"].reshape(3, 3)), unit(w["E"]), tol), (name, i)
            assert np.abs(g["R"].reshape(3, 3) - w["R"]).max() < tol and np.abs(g["t"] - w["t"]).max() < tol, (name, i)
            # (a matrix that went through robustEssentialRefine has rank 2 but two unequal singular values: getClosestE keeps both)
            refined = SEQUENCES[name]["cfg"].get("refineRTold", False)
            assert polish == 0 or arrsac or refined or constraint_residual(g["E"].reshape(3, 3)) < 1e-12
        if w["Eml"] is not None:
            unit = lambda a: a / np.linalg.norm(a)  # noqa: E731
            assert same_up_to_sign(unit(g["Eml"].reshape(3, 3)), unit(w["Eml"]), tol), (name, i)
    print(f"{name}: largest count difference {drift}, cpu model off the constraints: {cpu_off}")
    if name in ("steady", "stable", "bad_pairs_restore_and_skip", "uncertain_pairs", "small_pool", "drift_loses_the_pair"):
        assert drift == 0
