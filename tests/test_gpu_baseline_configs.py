"""GPU parity at the BASELINE.json sizes: C3 exactly as bench.py times it, the RANSAC driver's internal size thresholds, the C5
unit at 8192 keypoints, and a classified comparison of the 5-point solver's E-sets with the oracle's."""
import numpy as np
import pytest

from matchinglib_poselib_amd import pose, synth

pytestmark = pytest.mark.gpu


def e_dist(a, b):
    return min(np.linalg.norm(a - b), np.linalg.norm(a + b))


def assert_same_run(g, o, p1, p2, oracle, e_tol=1e-8):
    assert g["ok"] and o["ok"]
    assert g["iters"] == o["iters"], (g["iters"], o["iters"])
    assert g["n_inliers"] == o["n_inliers"], (g["n_inliers"], o["n_inliers"])
    assert np.array_equal(g["mask"], o["mask"])
    assert e_dist(g["E"], o["E"]) < e_tol
    go, Ro, to, Qo, mo = oracle.recover_pose(o["E"], p1, p2, 50.0, o["mask"])
    gg, Rg, tg, Qg, mg = oracle.recover_pose(g["E"], p1, p2, 50.0, g["mask"])
    assert gg == go
    assert np.abs(Rg - Ro).max() < 1e-6 and np.abs(tg - to).max() < 1e-6   # north_star bar for a fixed seed


def test_c3_exactly_as_benched(ctx, oracle):
    """BASELINE config 3 as bench_extras.py runs it: pose_scene(5000, seed 20260103), 20 000 iterations, confidence 1.0 (no
    early exit), srand(12345) -- every iteration, against the oracle's serial loop (~6 s on one core)."""
    p1, p2, R, t, mask, th = synth.pose_scene(5000, seed=20260103)
    g = pose.ransac_essential(p1, p2, th, confidence=1.0, max_iters=20000, refit=False, seed=12345, ctx=ctx)
    o = oracle.ransac_essential(p1, p2, th, confidence=1.0, max_iters=20000, lesqu=False, seed=12345)
    assert g["iters"] == o["iters"] == 20000
    assert_same_run(g, o, p1, p2, oracle)
    # and the device-pointer entry bench.py actually calls
    import torch
    d1, d2 = torch.from_numpy(p1).cuda(), torch.from_numpy(p2).cuda()
    dm = torch.empty(5000, dtype=torch.uint8, device="cuda")
    r = pose.ransac_essential_device(d1, d2, th, confidence=1.0, max_iters=20000, refit=False, seed=12345, ctx=ctx, mask_out=dm)
    torch.cuda.synchronize()
    assert r["iters"] == 20000 and r["n_inliers"] == o["n_inliers"] and e_dist(r["E"], o["E"]) < 1e-8
    assert np.array_equal(dm.cpu().numpy(), o["mask"])


def test_c3_with_refit_and_reference_confidence(ctx, oracle):
    """Same scene, the reference's confidence with C3's iteration cap, refit on: the adaptive stop inside a 20 000-slot pass."""
    p1, p2, R, t, mask, th = synth.pose_scene(5000, seed=20260103)
    g = pose.ransac_essential(p1, p2, th, confidence=0.999, max_iters=20000, refit=True, seed=12345, ctx=ctx)
    o = oracle.ransac_essential(p1, p2, th, confidence=0.999, max_iters=20000, lesqu=True, seed=12345)
    assert g["iters"] == o["iters"] and g["n_inliers"] == o["n_inliers"]
    assert e_dist(g["E"], o["E"]) < 1e-7
    # The refit is a 9 x 9 eigen-problem on ~2500 inliers: device and oracle agree on its model to ~1e-8, not bit for bit, and the final mask is
    # err(E_refit) <= th^2.  A correspondence may therefore differ ONLY if the oracle's own float error of it sits on the threshold: within
    # the relative change a 1e-7 model difference can cause (VERDICT r4 #8a: assert THAT, not a flip count).
    flips = np.nonzero(g["mask"] != o["mask"])[0]
    if len(flips):
        err = oracle.sampson_err(p1, p2, o["E"]).astype(np.float64)
        t2 = th * th
        assert len(flips) <= 4 and (np.abs(err[flips] - t2) <= 2e-5 * t2).all(), (flips, err[flips], t2)


@pytest.mark.parametrize("iters", [2047, 2048, 2049, 4095, 4096, 4097, 8191])
def test_ransac_pass_size_thresholds(ctx, oracle, iters):
    """The driver changes shape with the pass size: per-hypothesis maxima in a grid-wide kernel above 2048 hypotheses, a second
    solver slice that overlaps the host's sample drawing above 4096.  No early exit, so every hypothesis of the pass counts."""
    p1, p2, R, t, mask, th = synth.pose_scene(600, seed=777 + iters)
    g = pose.ransac_essential(p1, p2, th, confidence=1.0, max_iters=iters, refit=False, seed=iters, ctx=ctx)
    o = oracle.ransac_essential(p1, p2, th, confidence=1.0, max_iters=iters, lesqu=False, seed=iters)
    assert g["iters"] == o["iters"] == iters
    assert_same_run(g, o, p1, p2, oracle)


@pytest.mark.parametrize("n,iters", [(300, 33000), (64, 70000)])
def test_ransac_more_iterations_than_one_default_pass(ctx, oracle, n, iters):
    """max_iters above the default 32 768-hypothesis pass: the replay state crosses a pass boundary at the DEFAULT pass size
    (small n keeps the oracle's serial loop short)."""
    p1, p2, R, t, mask, th = synth.pose_scene(n, inlier_frac=0.4, seed=4000 + n)
    g = pose.ransac_essential(p1, p2, th, confidence=1.0, max_iters=iters, refit=False, seed=n, ctx=ctx)
    o = oracle.ransac_essential(p1, p2, th, confidence=1.0, max_iters=iters, lesqu=False, seed=n)
    # (confidence 1.0 still adapts when n is tiny: log(DBL_MIN) / log(1 - w^5) drops below max_iters once w^5 > ~0.01)
    assert g["iters"] == o["iters"] and g["iters"] > 32768
    assert_same_run(g, o, p1, p2, oracle, e_tol=1e-7)


@pytest.mark.parametrize("cap", [1, 3])
@pytest.mark.parametrize("conf,iters,host_table", [(0.999, 1000, 0), (1.0, 6000, 0), (0.999, 3000, 1)])
def test_ransac_record_list_overflow_takes_the_serial_path(ctx, oracle, cap, conf, iters, host_table):
    """The candidate / replay kernels note the record-breaking hypotheses of a pass in a 1024-entry list and expand them in parallel; with
    more records than entries they fall back to expanding in place (candidates) and to a serial replay.  A list of 1 or 3 entries forces
    those paths on ordinary data: the run must not change."""
    p1, p2, R, t, mask, th = synth.pose_scene(2500, inlier_frac=0.45, seed=77)
    want = oracle.ransac_essential(p1, p2, th, confidence=conf, max_iters=iters, lesqu=False, seed=9)
    ctx.set_option("ransac_event_cap", cap)
    ctx.set_option("ransac_host_table", host_table)
    try:
        g = pose.ransac_essential(p1, p2, th, confidence=conf, max_iters=iters, refit=False, seed=9, ctx=ctx)
    finally:
        ctx.set_option("ransac_event_cap", 0)
        ctx.set_option("ransac_host_table", 0)
    assert_same_run(g, want, p1, p2, oracle)


@pytest.mark.parametrize("cache_max", [0, 40, 1000])
def test_rand_stream_cache_across_calls_seeds_and_its_cap(ctx, oracle, cache_max):
    """The context keeps the raw rand() stream of the last seed.  Calls with the same seed and different n, a changed seed, LMedS in
    between, and a cache far too small for a call (a private generator continues behind it) must all see the stream srand(seed) gives."""
    ctx.set_option("rand_cache_max", cache_max)
    try:
        for n, seed, iters in [(900, 5, 800), (1300, 5, 1500), (700, 6, 600), (900, 5, 800), (2000, 5, 5000)]:
            p1, p2, R, t, mask, th = synth.pose_scene(n, inlier_frac=0.5, seed=n)
            g = pose.ransac_essential(p1, p2, th, confidence=0.999, max_iters=iters, refit=False, seed=seed, ctx=ctx)
            o = oracle.ransac_essential(p1, p2, th, confidence=0.999, max_iters=iters, lesqu=False, seed=seed)
            assert_same_run(g, o, p1, p2, oracle)
            gl = pose.lmeds_essential(p1, p2, seed=seed + 1, ctx=ctx)
            ol = oracle.lmeds_essential(p1, p2, seed=seed + 1)
            assert gl["ok"] == ol["ok"] and gl["n_inliers"] == ol["n_inliers"]
    finally:
        ctx.set_option("rand_cache_max", 0)


def test_ransac_tiny_inlier_fraction_uses_every_iteration(ctx, oracle):
    """20 % inliers at the reference's settings: the adaptive bound stays above max_iters for a long time."""
    p1, p2, R, t, mask, th = synth.pose_scene(3000, inlier_frac=0.2, seed=99)
    g = pose.ransac_essential(p1, p2, th, confidence=0.999, max_iters=5000, refit=False, seed=5, ctx=ctx)
    o = oracle.ransac_essential(p1, p2, th, confidence=0.999, max_iters=5000, lesqu=False, seed=5)
    assert_same_run(g, o, p1, p2, oracle)


def cubic_residual(E):
    """max |2 E E^T E - tr(E E^T) E| and |det E| of a Frobenius-normalised model: zero for an exact essential matrix."""
    return max(np.abs(2 * E @ E.T @ E - np.trace(E @ E.T) * E).max(), abs(np.linalg.det(E)))


def compare_solver_with_oracle(ctx, oracle, p1, p2, samples, tol=1e-8):
    """GPU E-sets vs the oracle's per sample.  Returns (count_mismatches, unexplained, worst_gpu_residual, n_differ).
    A differing oracle model is `explained` when the ORACLE's model violates the essential-matrix constraints by more than 1e-11:
    the elimination + root path is ill conditioned for the CPU path's null-space basis on that sample and the CPU result itself is
    off (the GPU polishes every solution on the constraints, so its own residual is at rounding level for every model)."""
    E, nm = pose.solve_5pt(p1, p2, samples, ctx=ctx)
    count_mismatch, unexplained, worst, differ = [], [], 0.0, 0
    for s in range(len(samples)):
        q1, q2 = p1[samples[s]], p2[samples[s]]
        Eo = oracle.run5point(q1, q2)
        Eg = E[s, :nm[s]]
        for e in Eg:
            worst = max(worst, cubic_residual(e))
        if len(Eo) != len(Eg):
            _, c, roots, z = oracle.run5point_dbg(q1, q2)
            count_mismatch.append((s, len(Eg), len(Eo), np.abs(roots.imag).tolist(), z.tolist()))
            continue
        bad_here = False
        for e in Eo:
            d = min(e_dist(e, x) for x in Eg) if len(Eg) else np.inf
            if d > tol:
                bad_here = True
                if cubic_residual(e) < 1e-11:
                    unexplained.append((s, d, cubic_residual(e)))
        differ += bad_here
    return count_mismatch, unexplained, worst, differ


def test_solve_5pt_agrees_wherever_the_cpu_path_is_accurate(ctx, oracle):
    """10 000 samples of the C3 scene.  Solution counts must be equal; every oracle model is reproduced to 1e-8 unless the oracle's
    own model violates the essential-matrix constraints (residual > 1e-11); every GPU model satisfies them to 1e-12."""
    p1, p2, R, t, mask, th = synth.pose_scene(5000, seed=20260103)
    samples = oracle.sample_table(12345, p1, p2, 10000)
    cm, unexplained, worst, differ = compare_solver_with_oracle(ctx, oracle, p1, p2, samples)
    print(f"\nsolver: {differ} of {len(samples)} samples differ by > 1e-8 (all with an inaccurate CPU model), "
          f"{len(cm)} count mismatches, worst GPU constraint residual {worst:.2e}")
    for row in cm:
        print("  count mismatch", row)
    assert not unexplained, unexplained[:5]
    assert worst < 1e-12
    # a count mismatch needs a root whose |imag| (or solveZ z) sits within rounding of the 1e-10 acceptance thresholds
    for s, ng, no, im, z in cm:
        near = any(1e-12 < v < 1e-8 for v in im) or any(abs(v) < 1e-8 for v in z if v == v)
        assert near, (s, ng, no)
    assert len(cm) <= 2


def test_solver_polish_off_is_the_plain_root_path(ctx, oracle):
    """A/B: without the polish the same solution counts, the same models to 1e-4, and the known ~0.5 % of ill-conditioned samples."""
    p1, p2, R, t, mask, th = synth.pose_scene(5000, seed=20260103)
    samples = oracle.sample_table(12345, p1, p2, 2000)
    E1, n1 = pose.solve_5pt(p1, p2, samples, ctx=ctx)
    ctx.set_option("solver_polish", 0)
    try:
        E0, n0 = pose.solve_5pt(p1, p2, samples, ctx=ctx)
    finally:
        ctx.set_option("solver_polish", 1)
    assert np.array_equal(n0, n1)
    d = np.abs(E0 - E1).reshape(len(samples), -1).max(axis=1)
    assert d.max() < 1e-3 and (d > 1e-8).sum() < 40 and np.median(d) < 1e-13


def test_the_safeguard_is_what_keeps_the_device_on_the_cpu_path(ctx, oracle):
    """Which setting of solver_polish is closer to the CPU path, per minimal model (tools/polish_default_ab.py on 3 000 samples)?  With the
    safeguard (default) every model that differs from the oracle's by more than 1e-8 is one where the ORACLE's own model violates the
    essential-matrix constraints (its residual > 100 x the device's) and they are under 0.6 % of the models; without it the device's own
    ill-conditioned samples come on top (models whose DEVICE residual is the large one) and more models differ."""
    p1, p2, R, t, mask, th = synth.pose_scene(5000, seed=20260103)
    samples = oracle.sample_table(12345, p1, p2, 3000)
    Eo_all = [oracle.run5point(p1[s], p2[s]) for s in samples]

    def tally():
        E, nm = pose.solve_5pt(p1, p2, samples, ctx=ctx)
        models = differ = oracle_off = device_off = 0
        for s in range(len(samples)):
            Eo, Eg = Eo_all[s], E[s, :nm[s]]
            assert len(Eo) == len(Eg), s
            for e in Eo:
                j = int(np.argmin([e_dist(e, x) for x in Eg]))
                models += 1
                if e_dist(e, Eg[j]) > 1e-8:
                    differ += 1
                    ro, rg = cubic_residual(e), cubic_residual(Eg[j])
                    oracle_off += ro > 100 * rg
                    device_off += rg > 100 * ro
        return models, differ, oracle_off, device_off

    assert ctx.get_option("solver_polish") == 1
    m1, d1, o1, g1 = tally()
    ctx.set_option("solver_polish", 0)
    try:
        m0, d0, o0, g0 = tally()
    finally:
        ctx.set_option("solver_polish", 1)
    print(f"\nsafeguard on: {d1} of {m1} models differ by > 1e-8 (oracle's model the inaccurate one: {o1}, the device's: {g1}); "
          f"off: {d0} of {m0} (oracle's {o0}, device's {g0})")
    assert d1 == o1 and g1 == 0 and d1 <= 0.006 * m1
    assert g0 > 0 and d0 > d1


def test_c5_unit_at_8192_keypoints(ctx, oracle):
    """The C5 unit at its real size: 8192 ORB keypoints per image through mlpl_pair_pose_dev vs the oracle pipeline."""
    import torch
    from matchinglib_poselib_amd import batch
    from test_gpu_batch import oracle_pipeline

    for seed, unmatched in ((20260200, 0.30), (20260201, 0.0)):
        sp = synth.stereo_pair(8192, seed=seed, unmatched_frac=unmatched)
        dev = torch.device("cuda", 0)
        args = [torch.from_numpy(sp[k]).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")]
        rec = batch.process_pair_on_device(ctx, *args, sp["K"], sp["K"], seed=100, pair_id=3)[0]
        nm, r, R, t = oracle_pipeline(oracle, sp, 100)
        assert rec["status"] == 0 and rec["n_matches"] == nm and rec["n_inliers"] == r["n_inliers"], (rec, nm, r["n_inliers"])
        E = rec["E"].reshape(3, 3)
        assert e_dist(E, r["E"]) < 1e-8
        assert np.abs(rec["R"].reshape(3, 3) - R).max() < 1e-6 and np.abs(rec["t"] - t).max() < 1e-6
