"""USAC's degeneracy handling on the MI355X (mlpl_usac_params.check_degeneracy; csrc/usac_impl.h test_degeneracy / upgrade_model,
usac_degen_rows_kernel) against traces of the REFERENCE's USAC.h driven through the reference's OpenGV and PoseTools
(oracle/_ref/usac_ref with check_degeneracy -> tests/golden/usac_degen_trace.npz; generator tools/usac_degen_cases.py).

What "equal" can mean here.  On general motion nothing degenerate is found and the runs are identical event by event.  On degenerate
motion (pure rotation, no motion, a baseline of 0.02 .. 0.05 against 4 .. 12 of depth) the reference's own computation is
ill-conditioned in three places, none of them part of the degeneracy code under test: (a) the eigensolver of the R -> R + t upgrade
iterates on a forward-difference Jacobian whose step is 1.5e-8 |x| (tests/test_usac_degen_math.py), (b) the 8-point refit of the local
optimisation has a null space of more than one dimension when all correspondences satisfy a rotation, (c) the 5-point solver on a
sample without parallax.  So on those scenes the comparison is: everything up to and including the first degeneracy test (ten
two-point rotations, their evaluation on all correspondences, the n-point refits, the no-motion test) identical; the no-motion -> t
upgrade (closed form) identical candidate by candidate; the degenerate models and the decision "degenerate" equal at the end."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
import usac_degen_checks as checks  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def runs(ctx):
    from matchinglib_poselib_amd import pose

    return checks.collect(lambda p1, p2, th, seed, si, chk: pose.usac_essential(p1, p2, th, seed, sorted_idx=si, event_cap=200000,
                                                                               check_degeneracy=chk, ctx=ctx))


def test_general_motion_nothing_found_and_identical(runs):
    checks.check_general_motion_nothing_found_and_identical(runs)


def test_first_degeneracy_test_is_identical(runs):
    checks.check_first_degeneracy_test_is_identical(runs)


def test_no_motion_upgrade_is_identical_candidate_by_candidate(runs):
    checks.check_no_motion_upgrade_is_identical_candidate_by_candidate(runs)


def test_rotation_upgrade_follows_until_the_eigensolver_noise_decides(runs):
    checks.check_rotation_upgrade_follows_until_the_eigensolver_noise_decides(runs)


def test_degenerate_models_and_decision_at_the_end(runs):
    checks.check_degenerate_models_and_decision_at_the_end(runs)


def test_parameters_and_results_entry(ctx):
    import ctypes as C
    from matchinglib_poselib_amd import _lib, pose, synth
    from matchinglib_poselib_amd._lib import MlplError

    p1, p2, R, t, truth, th = synth.pose_scene(200, 0.7, seed=3)
    for bad in (dict(check_degeneracy=2), dict(check_degeneracy=4), dict(check_degeneracy=1, focal_length=0.0),
                dict(check_degeneracy=1, th_pixels=-1.0)):
        with pytest.raises(MlplError) as e:
            pose.usac_essential(p1, p2, th, 1, ctx=ctx, **bad)
        assert e.value.code == _lib.MLPL_E_BAD_INPUT
    r = pose.usac_essential(p1, p2, th, 1, ctx=ctx)                       # no tests: nothing to report, masks refused
    info = np.zeros(16)
    assert ctx.lib.mlpl_usac_last_degeneracy(ctx.handle, info.ctypes.data, None, None, 0) == 0 and info[0] == 0
    m = np.zeros(200, np.uint8)
    assert ctx.lib.mlpl_usac_last_degeneracy(ctx.handle, info.ctypes.data, m.ctypes.data, None, 200) == _lib.MLPL_E_BAD_INPUT
    r = pose.usac_essential(p1, p2, th, 1, check_degeneracy=1, ctx=ctx)
    assert r["degen"][0] == 1 and r["stats"][5] >= 1
    assert ctx.lib.mlpl_usac_last_degeneracy(ctx.handle, info.ctypes.data, m.ctypes.data, None, 199) == _lib.MLPL_E_BAD_INPUT


@pytest.mark.parametrize("scene,kw,expect", [("general", {}, 0), ("rotation", dict(t_len=0.0), 1), ("nomotion", dict(t_len=0.0, rot_deg=0.0), 1)])
def test_cpp_facade_decision_and_stereo_refine(tmp_path, scene, kw, expect):
    """poselib::estimateEssentialOrPoseUSAC with ConfigUSAC's default DEGEN_USAC_INTERNAL through the C++ drop-in: the verdict, the
    degenerate rotation (the scene's own, or the identity for "no motion"), its inlier mask; StereoRefine gives a degenerate pair up
    as the reference does (robustPoseEstimation returns -2, stereo_pose_refinement.cpp:1402-1411; addNewCorrespondences turns every
    failure of it into -1, :974-977)."""
    import subprocess
    from matchinglib_poselib_amd import synth

    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = os.path.join(root, "tests", "cpp", "usac_degen_facade")
    assert os.path.exists(exe), "run __graft_entry__.build()"
    n = 900
    p1, p2, R, t, truth, th = synth.pose_scene(n, 0.6, seed=77, **kw)
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        np.array([n], np.int32).tofile(f)
        p1.tofile(f)
        p2.tofile(f)
        np.array([th], np.float64).tofile(f)
        np.array([991], np.uint32).tofile(f)
        np.array([0.85], np.float64).tofile(f)
    subprocess.run([exe, str(fin), str(fout)], check=True, timeout=120)
    raw = open(fout, "rb").read()
    off = 0

    def take(dtype, count):
        nonlocal off
        a = np.frombuffer(raw, dtype=dtype, count=count, offset=off)
        off += a.nbytes
        return a

    rc, deg = take(np.int32, 2)
    E = take(np.float64, 9)
    mask = take(np.uint8, n)
    have_R = take(np.int32, 1)[0]
    Rd = take(np.float64, 9).reshape(3, 3)
    have_mask = take(np.int32, 1)[0]
    mask_R = take(np.uint8, n)
    sr_rc = take(np.int32, 1)[0]
    assert rc == 0 and deg == expect and np.isfinite(E).all()
    assert int((mask.astype(bool) & truth).sum()) > 0.9 * truth.sum()
    if expect:
        assert have_R == 1 and have_mask == 1
        want = R if scene == "rotation" else np.eye(3)
        assert np.abs(Rd - want).max() < 2e-4, np.abs(Rd - want).max()       # image-1 bearings -> image-2 bearings: the scene's rotation
        assert int((mask_R.astype(bool) & truth).sum()) > 0.9 * truth.sum() and int((mask_R.astype(bool) & ~truth).sum()) <= 0.02 * n
        assert sr_rc == -1            # robustPoseEstimation's -2 reaches the caller as -1 (stereo_pose_refinement.cpp:974-977)
    else:
        assert have_R == 0 and have_mask == 0 and sr_rc == 0


def test_edge_cases_terminate_with_defined_results(ctx):
    """Few correspondences, repeated correspondences, correspondences that all coincide, almost no inliers, the full 8192: every call
    returns, masks are consistent with their counts, the rotation is a rotation whenever one was stored."""
    from matchinglib_poselib_amd import pose, synth

    rng = np.random.default_rng(8)
    cases = []
    for n in (5, 6, 9, 20, 37):
        p1, p2, R, t, truth, th = synth.pose_scene(n, 1.0, seed=200 + n, t_len=0.0)
        cases.append((f"rotation_n{n}", p1, p2, th))
    p1, p2, R, t, truth, th = synth.pose_scene(300, 0.7, seed=211, t_len=0.0, rot_deg=0.0)
    cases.append(("nomotion_repeated", np.repeat(p1[:60], 5, axis=0), np.repeat(p2[:60], 5, axis=0), th))
    cases.append(("one_point", np.tile(p1[:1], (64, 1)), np.tile(p1[:1], (64, 1)), th))
    cases.append(("identical_views", p1.copy(), p1.copy(), th))
    p1, p2, R, t, truth, th = synth.pose_scene(1000, 0.08, seed=212, t_len=0.0)
    cases.append(("rotation_8_percent", p1, p2, th))
    p1, p2, R, t, truth, th = synth.pose_scene(8192, 0.5, seed=213, t_len=0.0)
    cases.append(("rotation_8192", p1, p2, th))
    p1, p2, R, t, truth, th = synth.pose_scene(8192, 0.5, seed=214, t_len=0.0, rot_deg=0.0)
    cases.append(("nomotion_8192", p1, p2, th))
    for name, a, b, th in cases:
        for chk in (1, 3):
            d = pose.usac_essential(a, b, th, 5, check_degeneracy=chk, max_hyp=3000, ctx=ctx)
            assert d["ok"], name
            assert int(d["flags_rot"].sum()) == int(d["degen"][1]) and int(d["flags_nomot"].sum()) == int(d["degen"][2]), name
            assert int(d["flags"].sum()) <= len(a) and d["final"][5] <= len(a), name
            if d["degen"][1] > 2:
                Rd = d["R_degen"].reshape(3, 3)
                assert np.isfinite(Rd).all() and np.abs(Rd @ Rd.T - np.eye(3)).max() < 1e-9 and abs(np.linalg.det(Rd) - 1) < 1e-9, name
        if name in ("rotation_8192", "nomotion_8192", "identical_views"):
            assert d["degen"][1] > 0.4 * len(a) * (0.5 if "8192" in name else 0.7), (name, d["degen"])


def test_device_equals_oracle_on_more_scenes(ctx, oracle):
    """Scenes outside the fixture, device against the CPU oracle: identical through the first degeneracy test and through no-motion
    upgrades; the same rotation-only model and verdict at the end."""
    from matchinglib_poselib_amd import pose, synth

    sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
    import usac_compare

    seen_nomot = 0
    for n, frac, seed, kw in ((900, 0.55, 301, dict(t_len=0.0)), (3000, 0.4, 302, dict(t_len=0.0, rot_deg=2.0)), (700, 0.7, 303, dict(t_len=0.0, rot_deg=0.0)),
                              (2500, 0.45, 304, dict(t_len=0.0, rot_deg=0.0)), (1200, 0.6, 305, {}), (5000, 0.5, 306, dict(t_len=0.0))):
        p1, p2, R, t, truth, th = synth.pose_scene(n, frac, seed=seed, **kw)
        for chk in (1, 3):
            o = oracle.usac_essential_degen(p1, p2, th, 77, check_degeneracy=chk, event_cap=300000)
            d = pose.usac_essential(p1, p2, th, 77, check_degeneracy=chk, event_cap=300000, ctx=ctx)
            eo, ed = o["events"], d["events"]
            i7 = checks.first_of(eo, 7)
            first, _ = usac_compare.compare(eo[:i7 + 1], ed[:i7 + 1])
            if first is not None:      # a sample without parallax before any test
                assert kw and int(eo[first][0]) in (2, 5) and first < 80, (seed, first)
                continue
            i9 = checks.first_of(eo, 9)
            if i9 is not None and eo[i9][2] == 1:
                first, _ = usac_compare.compare(eo[:i9 + 1], ed[:i9 + 1])
                assert first is None, (seed, first)
                seen_nomot += 1
            if not kw:
                first, _ = usac_compare.compare(eo, ed)
                assert first is None and np.array_equal(o["flags"], d["flags"]) and d["degen"][1] == 0
                continue
            assert abs(o["degen"][1] - d["degen"][1]) <= max(3, 0.02 * o["degen"][1]) and np.abs(o["R_degen"] - d["R_degen"]).max() < 2e-4, \
                (seed, o["degen"], d["degen"])
            for th_dec in (0.85, 1.65):
                assert checks.degenerate_decision(n, o["final"][5], o["degen"][1:3], th_dec) == \
                    checks.degenerate_decision(n, d["final"][5], d["degen"][1:3], th_dec)
    assert seen_nomot >= 2
