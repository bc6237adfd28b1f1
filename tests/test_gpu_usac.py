"""USAC on the MI355X (mlpl_usac_essential, csrc/usac_impl.h) against the CPU oracle and against the reference-built decision traces
(tests/golden/usac_trace.npz): every sample, evaluation, refit and stored model, uniform and PROSAC sampling, with the local
optimisation as one launch and through its resume path."""
import ctypes as C
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import make_golden  # noqa: E402
import usac_compare  # noqa: E402
from test_oracle_usac import check_against_fixture, fixture_cases, refine_fixture_cases, stewenius_run  # noqa: E402

pytestmark = pytest.mark.gpu


def dev_run(ctx):
    from matchinglib_poselib_amd import pose

    return lambda p1, p2, th, s, si, cap: pose.usac_essential(p1, p2, th, s, sorted_idx=si, event_cap=cap, ctx=ctx)


@pytest.mark.parametrize("stepwise", [0, 1])
def test_device_follows_the_reference_built_traces(ctx, stepwise):
    """The fixture of the REFERENCE's USAC.h compiled in place, held directly against the device path (32 runs, 8 scenes x uniform / PROSAC
    x 2 seeds); stepwise = 1 takes every local-optimisation step through the resume path."""
    ctx.set_option("usac_lo_stepwise", stepwise)
    try:
        for g, k, n, frac, seed, prosac, usac_seed, agree in fixture_cases():
            if agree:
                check_against_fixture(dev_run(ctx), g, k, n, frac, seed, prosac, usac_seed, e5_max=1e-4)
    finally:
        ctx.set_option("usac_lo_stepwise", 0)


def test_device_follows_the_reference_running_its_default_stewenius_solver(ctx):
    """tests/golden/usac_stewenius_trace.npz (`usac_ref --stewenius`: the reference-built USAC.h with ConfigUSAC's default estimator,
    OpenGV's fivept_stewenius -- the reference's own solver).  The device serves POSE_STEWENIUS with its five-point kernels
    (estimator = 2): same decisions event by event on the runs the oracle follows too."""
    from matchinglib_poselib_amd import pose

    run = stewenius_run(lambda *a, **k: pose.usac_essential(*a, estimator=2, ctx=ctx, **k))
    checked = 0
    for g, k, n, frac, seed, prosac, usac_seed, agree in fixture_cases("usac_stewenius_trace.npz"):
        if agree:
            check_against_fixture(run, g, k, n, frac, seed, prosac, usac_seed, e5_max=5e-3, kept=make_golden.USAC_STEWENIUS_EVENTS_KEPT)
            checked += 1
    assert checked >= 21


def test_sequential_test_finished_on_bounds_takes_the_same_decisions(ctx):
    """Option usac_sprt_fast (default 1): a sequential test that survives its first 128 steps is finished word by word on upper bounds of the
    likelihood ratio instead of step by step.  Same events (start position, inliers seen, points tested, verdict, SPRT parameters) and the
    same result with the option off, on scenes from 64 to 8192 correspondences, uniform and PROSAC, with the degeneracy tests."""
    from matchinglib_poselib_amd import pose

    for sc in usac_compare.scenes():
        for prosac, chk in ((False, 0), (True, 0), (False, 3)):
            si = sc["order"] if prosac else None
            runs = []
            for fast in (1, 0):
                ctx.set_option("usac_sprt_fast", fast)
                try:
                    runs.append(pose.usac_essential(sc["p1"], sc["p2"], sc["th"], 31337, sorted_idx=si, event_cap=120000, max_hyp=4000,
                                                    check_degeneracy=chk, ctx=ctx))
                finally:
                    ctx.set_option("usac_sprt_fast", 1)
            a, b = runs
            assert a["n_events"] == b["n_events"] and np.array_equal(a["events"], b["events"]), sc["name"]
            assert np.array_equal(a["final"], b["final"]) and np.array_equal(a["flags"], b["flags"]) and np.array_equal(a["E"], b["E"])


def test_prosac_table_kept_by_the_context_gives_the_same_run(ctx):
    """init_prosac's non-randomness table depends on (subset size, beta, confidence) only; the context keeps the last one.  A run that copies
    it (same beta, fewer correspondences: a prefix of the kept table) equals the run of a fresh context event by event, and a run with
    another beta -- which recomputes -- does too."""
    import matchinglib_poselib_amd as mpa
    from matchinglib_poselib_amd import pose

    p1, p2, th, truth, order = make_golden.usac_scene(5000, 0.5, 20260103)
    q1, q2, qth, qtruth, qorder = make_golden.usac_scene(300, 0.5, 13)
    pose.usac_essential(p1, p2, th, 5, sorted_idx=order, ctx=ctx)                       # fills the table (1001 sizes, beta 0.09)
    for beta in (0.09, 0.05, 0.09):
        kept = pose.usac_essential(q1, q2, qth, 77, sorted_idx=qorder, prosac_beta=beta, event_cap=20000, ctx=ctx)
        fresh_ctx = mpa.Context(0)
        try:
            fresh = pose.usac_essential(q1, q2, qth, 77, sorted_idx=qorder, prosac_beta=beta, event_cap=20000, ctx=fresh_ctx)
        finally:
            fresh_ctx.close()
        assert kept["n_events"] == fresh["n_events"] and np.array_equal(kept["events"], fresh["events"]), beta
        assert np.array_equal(kept["final"], fresh["final"]) and np.array_equal(kept["flags"], fresh["flags"])


def test_device_equals_oracle_turn_by_turn(ctx, oracle):
    """Ten more scenes incl. C3 (5000 correspondences, 50 % inliers) and 8192 correspondences at 25 %: identical decisions, models to 1e-8.
    A run may part from the oracle only at a sample whose solution COUNT differs (a double root on the 1e-10 imaginary-part line).
    Minimal models: 98 % within 1e-8; the rest are the ill-conditioned samples on which the CPU root path itself is inaccurate (the
    device polishes every solution on the cubic constraints) -- with identical decisions all the same."""
    from matchinglib_poselib_amd import pose

    parted = 0
    for sc in usac_compare.scenes():
        for prosac in (False, True):
            si = sc["order"] if prosac else None
            o = oracle.usac_essential(sc["p1"], sc["p2"], sc["th"], 4242, sorted_idx=si, event_cap=120000, max_hyp=6000)
            d = pose.usac_essential(sc["p1"], sc["p2"], sc["th"], 4242, sorted_idx=si, event_cap=120000, max_hyp=6000, ctx=ctx)
            first, diffs = usac_compare.compare(o["events"], d["events"])
            if first is not None:
                a, b = o["events"][first], d["events"][first]
                assert int(a[0]) == 1 and int(b[0]) == 1 and np.array_equal(a[1:7], b[1:7]) and a[7] != b[7], (sc["name"], first, a[:9], b[:9])
                parted += 1
                continue
            assert diffs["sprt"] < 1e-12 and diffs.get("E3", 0) < 1e-8 and diffs.get("E5_q98", 0) < 1e-8 and diffs.get("E5", 0) < 1e-2, \
                (sc["name"], diffs)
            assert np.array_equal(o["final"][:8], d["final"][:8]) and np.abs(o["final"][8:] - d["final"][8:]).max() < 1e-12
            assert np.array_equal(o["flags"], d["flags"])
            Eo, Ed = o["E"] / np.linalg.norm(o["E"]), d["E"] / np.linalg.norm(d["E"])
            assert min(np.abs(Eo - Ed).max(), np.abs(Eo + Ed).max()) < 1e-8
            assert d["stats"][2] == o["final"][1] - o["final"][3]          # samples consumed = hypotheses - pre-validation rejections
    assert parted <= 2


def test_device_pointer_entry_and_mask(ctx, oracle):
    import torch
    from matchinglib_poselib_amd import pose, synth

    p1, p2, R, t, truth, th = synth.pose_scene(3000, 0.6, seed=5)
    o = oracle.usac_essential(p1, p2, th, 99)
    P = pose.UsacParams()
    ctx.lib.mlpl_usac_default_params(C.addressof(P), float(th))
    P.seed = 99
    d1, d2 = torch.from_numpy(p1).cuda(), torch.from_numpy(p2).cuda()
    mask = torch.zeros(len(p1), dtype=torch.uint8, device="cuda")
    E, res = np.zeros(9), np.zeros(12)
    torch.cuda.synchronize()
    rc = ctx.lib.mlpl_usac_essential_dev(ctx.handle, d1.data_ptr(), d2.data_ptr(), len(p1), C.addressof(P), E.ctypes.data, mask.data_ptr(),
                                         res.ctypes.data, None)
    assert rc == 0
    assert np.array_equal(mask.cpu().numpy(), o["flags"]) and np.array_equal(res[:8], o["final"][:8])
    assert min(np.abs(E - o["E"]).max(), np.abs(E + o["E"]).max()) < 1e-8
    # the returned model explains the true inliers
    assert (o["flags"].astype(bool) & truth).sum() > 0.97 * truth.sum()


def test_argument_checks_and_refusals(ctx):
    from matchinglib_poselib_amd import _lib, pose, synth

    p1, p2, R, t, truth, th = synth.pose_scene(200, 0.6, seed=6)
    assert not pose.usac_essential(p1[:4], p2[:4], th, 1, ctx=ctx)["ok"]                                   # solve() refuses: < 5
    assert not pose.usac_essential(p1[:12], p2[:12], th, 1, sorted_idx=np.arange(12), ctx=ctx)["ok"]      # PROSAC: < 20
    assert pose.usac_essential(p1, p2, th, 1, estimator=2, ctx=ctx)["ok"]                                  # POSE_STEWENIUS: same solver
    for refine in (1, 2, 3, 8):                                                                            # REF_8PT_PSEUDOHUBER, REF_EIG_KNEIP(_WEIGHTS): not built
        with pytest.raises(_lib.MlplError) as e:
            pose.usac_essential(p1, p2, th, 1, refine=refine, ctx=ctx)
        assert e.value.code == _lib.MLPL_E_UNSUPPORTED
    assert pose.usac_essential(p1, p2, th, 1, refine=5, estimator=2, ctx=ctx)["ok"]                        # ConfigUSAC's defaults
    with pytest.raises(_lib.MlplError):
        pose.usac_essential(p1, p2, th, 1, refine=5, check_degeneracy=3, ctx=ctx)                          # tests after LO: 8-point refinements only
    with pytest.raises(_lib.MlplError):
        pose.usac_essential(p1, p2, th, 1, estimator=1, ctx=ctx)                                           # Kneip's eigensolver: not built
    with pytest.raises(_lib.MlplError):
        pose.usac_essential(p1, p2, th, 1, sorted_idx=np.full(200, 200), ctx=ctx)                          # index out of range


def test_threshold_relaxation_after_half_the_budget(ctx, oracle):
    """No model within max_hyp / 2 hypotheses: the inlier threshold grows by 1.33 (USAC.h:361-368) and the speculation must not run
    across that hypothesis.  Pure outliers, tiny budget."""
    from matchinglib_poselib_amd import pose

    rng = np.random.default_rng(8)
    p1, p2 = rng.uniform(-0.4, 0.4, (400, 2)), rng.uniform(-0.4, 0.4, (400, 2))
    o = oracle.usac_essential(p1, p2, 1e-5, 3, max_hyp=300, event_cap=60000)
    d = pose.usac_essential(p1, p2, 1e-5, 3, max_hyp=300, event_cap=60000, ctx=ctx)
    first, diffs = usac_compare.compare(o["events"], d["events"])
    assert first is None and np.array_equal(o["final"][:8], d["final"][:8]) and np.array_equal(o["flags"], d["flags"])
    thr = np.unique(o["events"][o["events"][:, 0] == 2][:, 10])
    assert len(thr) >= 2                                                                                   # the relaxed threshold was in force


@pytest.mark.parametrize("refine,stepwise", [(5, 0), (5, 1), (4, 0), (7, 0), (7, 1), (6, 0)])
def test_device_follows_the_reference_with_the_five_point_refinements(ctx, refine, stepwise):
    """tests/golden/usac_refine_trace.npz against the device path (usac5_* chains): REF_STEWENIUS_WEIGHTS (ConfigUSAC's default) and
    REF_STEWENIUS on the reference-built USAC.h + OpenGV Stewenius traces, REF_NISTER(_WEIGHTS) on the control-flow traces; same
    decisions event by event, refined models to 1e-8.  stepwise = 1: every step of a local-optimisation chain through the resume path."""
    from matchinglib_poselib_amd import pose

    ctx.set_option("usac_lo_stepwise", stepwise)
    try:
        run = stewenius_run(lambda *a, **k: pose.usac_essential(*a, estimator=2 if refine in (4, 5) else 0, refine=refine, ctx=ctx, **k))
        checked = 0
        for g, k, n, frac, seed, prosac, usac_seed, agree, rf in refine_fixture_cases():
            if rf == refine and agree:
                check_against_fixture(run, g, k, n, frac, seed, prosac, usac_seed, e5_max=5e-3, kept=make_golden.USAC_REFINE_EVENTS_KEPT)
                checked += 1
        assert checked >= 12
    finally:
        ctx.set_option("usac_lo_stepwise", 0)


@pytest.mark.parametrize("refine", [5, 7, 4, 6])
def test_device_equals_oracle_with_the_five_point_refinements(ctx, oracle, refine):
    """Ten more scenes, uniform and PROSAC, with the degeneracy tests of DEGEN_USAC_INTERNAL on for half of them (check_degeneracy = 1:
    what estimateEssentialMatUsac configures for these refinements): identical decisions, refined models to 1e-8, same masks."""
    from matchinglib_poselib_amd import pose

    parted = 0
    for si_, sc in enumerate(usac_compare.scenes()):
        for prosac in (False, True):
            si = sc["order"] if prosac else None
            chk = 1 if (si_ + prosac) % 2 else 0
            kw = dict(sorted_idx=si, event_cap=120000, max_hyp=6000, refine=refine, sprt_ms=6.0, sprt_tm=2736.0)
            if chk:
                o = oracle.usac_essential_degen(sc["p1"], sc["p2"], sc["th"], 4242, check_degeneracy=1, **kw)
            else:
                o = oracle.usac_essential(sc["p1"], sc["p2"], sc["th"], 4242, **kw)
            d = pose.usac_essential(sc["p1"], sc["p2"], sc["th"], 4242, check_degeneracy=chk, ctx=ctx, **kw)
            first, diffs = usac_compare.compare(o["events"], d["events"])
            if first is not None:   # only at a sample whose solution COUNT differs (a double root)
                a, b = o["events"][first], d["events"][first]
                assert int(a[0]) == 1 and int(b[0]) == 1 and np.array_equal(a[1:7], b[1:7]) and a[7] != b[7], (sc["name"], first, a[:9], b[:9])
                parted += 1
                continue
            assert diffs["sprt"] < 1e-12 and diffs.get("E3", 0) < 1e-8 and diffs.get("E5_q98", 0) < 1e-8, (sc["name"], diffs)
            assert np.array_equal(o["final"][:8], d["final"][:8]) and np.array_equal(o["flags"], d["flags"])
            Eo, Ed = o["E"] / np.linalg.norm(o["E"]), d["E"] / np.linalg.norm(d["E"])
            assert min(np.abs(Eo - Ed).max(), np.abs(Eo + Ed).max()) < 1e-8
    assert parted <= 2
