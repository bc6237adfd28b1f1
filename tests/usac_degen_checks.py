"""Assertions shared by the CPU test of the oracle (tests/test_oracle_usac_degeneracy.py) and the GPU test of the device path
(tests/test_gpu_usac_degeneracy.py) against the reference-built traces of USAC's degeneracy handling (tests/golden/usac_degen_trace.npz).
`runs` = list of (key, scene name, fixture, result dict of the implementation under test, n)."""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import usac_compare  # noqa: E402
import usac_degen_cases  # noqa: E402


def collect(run):
    """run(p1, p2, th, usac_seed, sorted_idx or None, check) -> result dict with events, final, flags, degen, R_degen, flags_rot, flags_nomot."""
    g = np.load(usac_degen_cases.FIXTURE)
    out = []
    for key, name, p1, p2, th, order, truth, usac_seed, prosac, chk in usac_degen_cases.cases():
        out.append((key, name, g, run(p1, p2, th, usac_seed, order if prosac else None, chk), len(p1)))
    return out


def first_of(ev, ty, cond=None):
    for i, e in enumerate(ev):
        if int(e[0]) == ty and (cond is None or cond(e)):
            return i
    return None


def check_general_motion_nothing_found_and_identical(runs):
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


def check_first_degeneracy_test_is_identical(runs):
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


def check_no_motion_upgrade_is_identical_candidate_by_candidate(runs):
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


def check_rotation_upgrade_follows_until_the_eigensolver_noise_decides(runs, agree_tol=1e-3, agree_share=0.5):
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
    # per run: the median difference of the candidates' models up to the point where the runs part; a run counts as agreeing below
    # agree_tol (a few runs part after a handful of candidates on which the eigensolver's iteration or the order of Eigen's
    # eigenvalues was decided by rounding noise)
    assert seen >= 8 and np.mean(np.array(agree) < agree_tol) >= agree_share, (seen, agree)


def degenerate_decision(n, n_inliers, degen, th=0.85):
    """estimateEssentialOrPoseUSAC's decision (pose_estim.cpp:2101-2133): fraction of rotation / no-motion inliers among the inliers
    of E against degenDecisionTh times the inlier ratio."""
    frac_inl = n_inliers / n
    f_rot = degen[0] / n_inliers if degen[0] > 2 and n_inliers > 0 else 0.0
    f_nomot = degen[1] / n_inliers if degen[1] > 1 and n_inliers > 0 else 0.0
    return (th * frac_inl < f_rot) or (th * frac_inl < f_nomot)


def check_degenerate_models_and_decision_at_the_end(runs):
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
        # the best rotation-only model: inlier count within 2 %, the same rotation, the same inlier set up to a few correspondences.
        # With a short baseline "the" rotation-only model is not sharply defined (which sample's rotation collects most depends on the
        # run after the upgrade): 10 % there.
        loose = name == "shortbase"
        assert abs(ref_deg[0] - dev_deg[0]) <= max(3, (0.10 if loose else 0.02) * ref_deg[0]), (key, ref_deg, dev_deg)
        Rr, Rd = g[key + "_R"].reshape(3, 3), d["R_degen"].reshape(3, 3)
        assert np.abs(Rr - Rd).max() < (2e-3 if loose else 2e-4) and abs(np.linalg.det(Rd) - 1) < 1e-12, (key, np.abs(Rr - Rd).max())
        if not loose:
            assert (g[key + "_flags_rot"] != d["flags_rot"]).sum() <= max(4, 0.03 * ref_deg[0]), key
        assert int(d["flags_rot"].sum()) == int(dev_deg[0]) and int(d["flags_nomot"].sum()) == int(dev_deg[1])
        if name == "nomotion":
            assert abs(ref_deg[1] - dev_deg[1]) <= max(3, 0.02 * ref_deg[1])
