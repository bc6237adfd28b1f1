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

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import usac_compare  # noqa: E402
import usac_degen_cases  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def runs(ctx):
    from matchinglib_poselib_amd import pose

    g = np.load(usac_degen_cases.FIXTURE)
    out = []
    for key, name, p1, p2, th, order, truth, usac_seed, prosac, chk in usac_degen_cases.cases():
        d = pose.usac_essential(p1, p2, th, usac_seed, sorted_idx=order if prosac else None, event_cap=200000, check_degeneracy=chk, ctx=ctx)
        out.append((key, name, g, d, len(p1)))
    return out


def first_of(ev, ty, cond=None):
    for i, e in enumerate(ev):
        if int(e[0]) == ty and (cond is None or cond(e)):
            return i
    return None


def test_general_motion_nothing_found_and_identical(runs):
    seen = 0
    for key, name, g, d, n in runs:
        if name != "general":
            continue
        ev = g[key + "_events"]
        first, diffs = usac_compare.compare(ev, d["events"][:len(ev)])
        assert first is None and int(g[key + "_meta"][4]) == d["n_events"], (key, first)
        assert np.array_equal(g[key + "_final"][:8], d["final"][:8]) and np.array_equal(g[key + "_flags"], d["flags"])
        assert d["degen"][1] == 0 and d["degen"][2] == 0 and not d["flags_rot"].any() and not d["flags_nomot"].any()
        assert (ev[:, 0] == 7).sum() >= 1 and (ev[:, 0] == 9).sum() == 0        # tested, never upgraded
        seen += 1
    assert seen == 6


def test_first_degeneracy_test_is_identical(runs):
    """Ten two-point rotations, their inlier counts over all correspondences, the refits (type 8) and the verdict (type 7)."""
    parted_before = 0
    for key, name, g, d, n in runs:
        ev, dv = g[key + "_events"], d["events"]
        i7 = first_of(ev, 7)
        assert i7 is not None
        first, _ = usac_compare.compare(ev[:i7 + 1], dv[:i7 + 1])
        if first is not None:
            # (c) of the module text: a minimal sample without parallax, whose 5-point solutions are ill-conditioned -- the inlier
            # count of such a model differs between any two solvers
            assert name != "general" and int(ev[first][0]) in (2, 5) and first < 80, (key, name, first, ev[first][:9], dv[first][:9])
            parted_before += 1
            continue
        if name != "general":
            assert ev[i7][2] == 1 and ev[i7][3] == 1 and ev[i7][5] > 0.1 * n, (key, ev[i7][:8])   # degenerate, upgrade asked for
    assert parted_before <= 3


def test_no_motion_upgrade_is_identical_candidate_by_candidate(runs):
    seen = 0
    for key, name, g, d, n in runs:
        ev, dv = g[key + "_events"], d["events"]
        i9 = first_of(ev, 9)
        if i9 is None or ev[i9][2] != 1:
            continue
        first, _ = usac_compare.compare(ev[:i9 + 1], dv[:i9 + 1])
        assert first is None, (key, name, first)
        a, b = ev[:i9 + 1], dv[:i9 + 1]
        ta, tb = a[(a[:, 0] == 10)][:, 4:7], b[(b[:, 0] == 10)][:, 4:7]
        assert len(ta) > 20 and np.array_equal(ta, tb)                           # the two-point translations, to the bit
        seen += 1
    assert seen >= 4


def test_rotation_upgrade_follows_until_the_eigensolver_noise_decides(runs):
    """R -> R + t: candidates are the same correspondences (same stream), the models agree where the eigensolver converges, and the
    run stays identical at least up to the upgrade's first candidate."""
    seen, agree = 0, []
    for key, name, g, d, n in runs:
        ev, dv = g[key + "_events"], d["events"]
        i10 = first_of(ev, 10, lambda e: e[2] == 2)
        if i10 is None:
            continue
        first, _ = usac_compare.compare(ev[:i10 + 1], dv[:i10 + 1])
        if first is not None:                        # parted earlier, at (b) an 8-point refit or (c) a sample without parallax
            assert name != "general" and int(ev[first][0]) in (2, 3, 5), (key, name, first, ev[first][:9])
            continue
        m = min(len(ev), len(dv))
        first, _ = usac_compare.compare(ev[:m], dv[:m])
        stop = m if first is None else first
        a, b = ev[:stop], dv[:stop]
        ea, eb = a[(a[:, 0] == 10) & (a[:, 2] == 2)][:, 4:13], b[(b[:, 0] == 10) & (b[:, 2] == 2)][:, 4:13]
        if len(ea):
            agree.append(np.median(np.abs(ea - eb).max(1)))
        seen += 1
    assert seen >= 8 and np.median(agree) < 1e-3, (seen, agree)


def degenerate_decision(n, n_inliers, degen, th=0.85):
    """estimateEssentialOrPoseUSAC's decision (pose_estim.cpp:2101-2133): fraction of rotation / no-motion inliers among the inliers
    of E against degenDecisionTh times the inlier ratio."""
    frac_inl = n_inliers / n
    f_rot = degen[0] / n_inliers if degen[0] > 2 and n_inliers > 0 else 0.0
    f_nomot = degen[1] / n_inliers if degen[1] > 1 and n_inliers > 0 else 0.0
    return (th * frac_inl < f_rot) or (th * frac_inl < f_nomot)


def test_degenerate_models_and_decision_at_the_end(runs):
    for key, name, g, d, n in runs:
        ref_deg, dev_deg = g[key + "_degen"][:2], d["degen"][1:3]
        for th in (0.85, 1.65):                      # ConfigUSAC's default and the harness's (--USACdegenTh)
            ref_dec = degenerate_decision(n, g[key + "_final"][5], ref_deg, th)
            dev_dec = degenerate_decision(n, d["final"][5], dev_deg, th)
            assert ref_dec == dev_dec, (key, name, th, ref_deg, dev_deg)
        if name != "shortbase":                      # a short baseline is the case in between: either verdict, the same on both sides
            assert degenerate_decision(n, d["final"][5], dev_deg) == (name != "general"), (key, name, dev_deg)
        if name == "general":
            continue
        # the best rotation-only model: inlier count within 2 %, the same rotation, the same inlier set up to a few correspondences
        assert abs(ref_deg[0] - dev_deg[0]) <= max(3, 0.02 * ref_deg[0]), (key, ref_deg, dev_deg)
        Rr, Rd = g[key + "_R"].reshape(3, 3), d["R_degen"].reshape(3, 3)
        assert np.abs(Rr - Rd).max() < 2e-4 and abs(np.linalg.det(Rd) - 1) < 1e-12, (key, np.abs(Rr - Rd).max())
        assert (g[key + "_flags_rot"] != d["flags_rot"]).sum() <= max(4, 0.03 * ref_deg[0]), key
        assert int(d["flags_rot"].sum()) == int(dev_deg[0]) and int(d["flags_nomot"].sum()) == int(dev_deg[1])
        if name == "nomotion":
            assert abs(ref_deg[1] - dev_deg[1]) <= max(3, 0.02 * ref_deg[1])


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
