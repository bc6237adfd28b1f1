"""The BATCHED entries held directly against the CPU oracle -- not against the single-problem HIP entries, which
tests/test_gpu_usac_batch.py and tests/test_gpu_arrsac_batch.py do:
  * mlpl_usac_essential_batch_dev:   16 mixed problems, the decision trace of every problem against oracle/usac_oracle.cpp event by event;
  * mlpl_arrsac_essential_batch_dev: 16 mixed problems against oracle/arrsac_oracle.cpp (turn statistics, stream positions, inliers, mask, model);
  * mlpl_pair_pose_batch_usac_dev / _arrsac_dev: image pairs of 8192 keypoints (BASELINE config 5's unit) against the oracle PIPELINE
    (LINEAR matching -> ImgToCamCoordTrans -> USAC / ARRSAC oracle -> recoverPose), as test_c5_unit_at_8192_keypoints does for RANSAC;
  * the pair entries as the FIRST call of a fresh context (the nested matcher must not invalidate the entry's pinned block).
Reference: P/include/usac/estimators/USAC.h, EssentialMatEstimator.h; P/source/five-point-nister/modelest.cpp:197-341; the harness loop
T/poselib-test/main.cpp:1440-2072."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tools"))
import make_golden  # noqa: E402
import usac_compare  # noqa: E402

pytestmark = pytest.mark.gpu


def e_dist(a, b):
    a, b = np.asarray(a).ravel(), np.asarray(b).ravel()
    a, b = a / np.linalg.norm(a), b / np.linalg.norm(b)
    return min(np.abs(a - b).max(), np.abs(a + b).max())


def _mixed_problems(B, seed):
    rng = np.random.default_rng(seed)
    sizes = [int(v) for v in rng.choice([64, 150, 300, 800, 1200, 2000, 3000, 5000], B)]
    fr = rng.uniform(0.3, 0.9, B)
    scenes = [make_golden.usac_scene(sizes[b], float(fr[b]), 4100 + 17 * b + seed) for b in range(B)]
    stride = max(sizes)
    p1, p2 = np.zeros((B, stride, 2)), np.zeros((B, stride, 2))
    for b, (a, c, t, truth, order) in enumerate(scenes):
        p1[b, :sizes[b]], p2[b, :sizes[b]] = a, c
    return sizes, scenes, p1, p2


@pytest.mark.parametrize("refine,estimator", [(0, 0), (5, 2)])
def test_usac_batch_against_the_oracle_event_by_event(ctx, oracle, refine, estimator):
    """16 problems of 64 ... 5000 correspondences, 30-90 % inliers, uniform and PROSAC sampling mixed, own seeds, in ONE call of
    mlpl_usac_essential_batch_dev: every problem's decision trace equals the oracle's event by event (a run may part only at a sample whose
    solution COUNT differs -- the double-root category, at most 2 of 16), result block, inlier flags and model as in the single-problem test."""
    import torch
    from matchinglib_poselib_amd import pose

    B = 16
    sizes, scenes, p1, p2 = _mixed_problems(B, 1 + refine)
    th = scenes[0][2]
    orders = [scenes[b][4] if b % 3 == 1 else None for b in range(B)]
    seeds = [7000 + 29 * b for b in range(B)]
    dev = torch.device("cuda:0")
    d1, d2 = torch.from_numpy(p1).to(dev), torch.from_numpy(p2).to(dev)
    masks = torch.zeros((B, p1.shape[1]), dtype=torch.uint8, device=dev)
    cap = 60000
    got = pose.usac_essential_batch(d1, d2, sizes, th, seeds, sorted_idx=orders, event_cap=cap, masks_out=masks, max_hyp=4000, refine=refine,
                                    estimator=estimator, ctx=ctx)
    mh = masks.cpu().numpy()
    parted = 0
    for b in range(B):
        n = sizes[b]
        o = oracle.usac_essential(p1[b, :n], p2[b, :n], th, seeds[b], refine=refine, sorted_idx=orders[b], event_cap=cap, max_hyp=4000)
        g = got[b]
        assert g["ok"] == o["ok"], b
        assert g["n_events"] <= cap and o["n_events"] <= cap
        first, diffs = usac_compare.compare(o["events"], g["events"])
        if first is not None:
            a, c = o["events"][first], g["events"][first]
            assert int(a[0]) == 1 and int(c[0]) == 1 and np.array_equal(a[1:7], c[1:7]) and a[7] != c[7], (b, first, a[:9], c[:9])
            parted += 1
            continue
        assert diffs["sprt"] < 1e-12 and diffs.get("E3", 0) < 1e-7 and diffs.get("E5_q98", 0) < 1e-7, (b, diffs)
        assert np.array_equal(o["final"][:8], g["final"][:8]) and np.abs(o["final"][8:] - g["final"][8:]).max() < 1e-12, b
        assert np.array_equal(mh[b, :n], o["flags"]), b
        assert e_dist(o["E"], g["E"]) < 1e-7, b
    assert parted <= 2


@pytest.mark.parametrize("refine", [True, False])
def test_arrsac_batch_against_the_oracle(ctx, oracle, refine):
    """16 problems in ONE call of mlpl_arrsac_essential_batch_dev against the sequential oracle: stream positions after the call, inlier
    count, mask and model per problem (the single-entry test's criteria; its turn statistics are not exported by the batch entry -- the
    sampler streams stand where the oracle's stand only if the control flow took the same turns)."""
    import torch
    from matchinglib_poselib_amd import pose

    B = 16
    sizes, scenes, p1, p2 = _mixed_problems(B, 9)
    th = scenes[0][2]
    dev = torch.device("cuda:0")
    d1, d2 = torch.from_numpy(p1).to(dev), torch.from_numpy(p2).to(dev)
    masks = torch.zeros((B, p1.shape[1]), dtype=torch.uint8, device=dev)
    states = np.array([[0xFFFFFFFF + 131 * b, 0xFFFFFFFF + 7 * b * b] for b in range(B)], np.uint64)
    st_batch = states.copy()
    got = pose.arrsac_essential_batch(d1, d2, sizes, th, refine=refine, rng_states=st_batch, masks_out=masks, ctx=ctx)
    mh = masks.cpu().numpy()
    differing = 0
    for b in range(B):
        n = sizes[b]
        st_o = states[b].copy()
        o = oracle.arrsac_essential(p1[b, :n], p2[b, :n], th, refine=refine, rng_state=st_o)
        g = got[b]
        if not np.array_equal(st_o, st_batch[b]):     # a first-stage hypothesis more or less (double root): DESIGN section 8, at most 1 in 16 here
            differing += 1
            continue
        assert g["ok"] == o["ok"], b
        assert g["n_inliers"] == o["n_inliers"], b
        if o["n_inliers"]:
            assert np.array_equal(mh[b, :n], o["mask"]), b
        if o["ok"]:
            assert e_dist(g["E"], o["E"]) < 1e-7, (b, e_dist(g["E"], o["E"]))   # default options = the reference's arithmetic
    assert differing <= 1


def _cam(p, K):
    return np.stack([((p[:, 0].astype(np.float64) - K[2]) / K[0]).astype(np.float32),
                     ((p[:, 1].astype(np.float64) - K[3]) / K[1]).astype(np.float32)], axis=1).astype(np.float64)


@pytest.mark.parametrize("name,prosac,refine", [("uniform", False, 0), ("prosac", True, 0), ("default_refinement", True, 5)])
def test_c5_unit_with_usac_at_8192_keypoints_against_the_oracle_pipeline(ctx, oracle, name, prosac, refine):
    """mlpl_pair_pose_batch_usac_dev on 4 image pairs of 8192 ORB keypoints (the harness' cfgUSAC: POSE_STEWENIUS, SPRT 6.0 / 2736; uniform,
    PROSAC in the order of the matching costs, and ConfigUSAC's default refinement) against the oracle pipeline pair by pair: match count,
    hypotheses, inliers exact; E to 1e-7, R, t to 1e-6.  (Until round 5 this check lived only in bench_extras.py.)"""
    import torch
    from matchinglib_poselib_amd import batch, synth

    dev = torch.device("cuda:0")
    B, nk = 4, 8192
    sps = [synth.stereo_pair(nk, seed=20260200 + i, unmatched_frac=0.30 + 0.02 * (i % 8)) for i in range(B)]
    K = sps[0]["K"]
    stk = [torch.from_numpy(np.stack([sp[k] for sp in sps])).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")]
    seeds = [100 + i for i in range(B)]
    rec, raw = batch.process_pairs_batched_usac(ctx, *stk, K, K, seeds, prosac=prosac, refine=refine)
    th = 0.8 * 4.0 / (np.sqrt(2.0) * (2 * K[0] + 2 * K[1]))
    for i in range(B):
        sp = sps[i]
        rc, mm = oracle.get_matches_linear(nk, nk, sp["desc1"], sp["desc2"])
        assert rc == 0 and raw["status"][i] == 0 and len(mm) == raw["n_matches"][i], i
        p1, p2 = _cam(sp["kp1"][mm["queryIdx"]], K), _cam(sp["kp2"][mm["trainIdx"]], K)
        order = None
        if prosac:   # poselib::getSortedMatchIdx (pose_helper.cpp:2896-2923): std::sort of the matching costs; stable order of numpy differs on ties,
            order = np.zeros(len(mm), np.uint32)   # so the host helper the library exports for exactly this (no GPU work) gives the order
            mmc = np.ascontiguousarray(mm)
            assert ctx.lib.mlpl_sorted_match_idx(mmc.ctypes.data, len(mm), order.ctypes.data) == 0
            assert (np.diff(mm["distance"][order]) >= 0).all() and sorted(order.tolist()) == list(range(len(mm)))
        o = oracle.usac_essential(p1, p2, th, seeds[i], refine=refine, sorted_idx=order, prosac_beta=0.05, sprt_ms=6.0, sprt_tm=2736.0)
        assert o["ok"], i
        assert int(o["final"][1]) == raw["iters"][i] and int(o["final"][5]) == raw["n_inliers"][i], (name, i, o["final"][:8], raw[i])
        assert e_dist(o["E"], raw["E"][i]) < 1e-7, (name, i)
        good, R, t, Q, mk = oracle.recover_pose(o["E"], p1, p2, 50.0, o["flags"])
        assert good == raw["n_good"][i], (name, i)
        assert np.abs(raw["R"][i].reshape(3, 3) - R).max() < 1e-6 and np.abs(raw["t"][i] - np.asarray(t).ravel()).max() < 1e-6, (name, i)


def test_c5_unit_with_arrsac_at_8192_keypoints_against_the_oracle_pipeline(ctx, oracle):
    """mlpl_pair_pose_batch_arrsac_dev (estimateEssentialMat's default method + robustEssentialRefine) on 4 image pairs of 8192 keypoints against
    the oracle pipeline; a pair whose sampler streams end elsewhere (one first-stage hypothesis more or less) is skipped, at most one."""
    import torch
    from matchinglib_poselib_amd import batch, pose, synth

    dev = torch.device("cuda:0")
    B, nk = 4, 8192
    sps = [synth.stereo_pair(nk, seed=20260300 + i, unmatched_frac=0.30 + 0.02 * (i % 8)) for i in range(B)]
    K = sps[0]["K"]
    stk = [torch.from_numpy(np.stack([sp[k] for sp in sps])).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")]
    states = np.tile(np.array(pose.ARRSAC_RNG_FRESH, np.uint64), (B, 1))
    rec, raw = batch.process_pairs_batched_arrsac(ctx, *stk, K, K, refine=True, rng_states=states)
    th = 0.8 * 4.0 / (np.sqrt(2.0) * (2 * K[0] + 2 * K[1]))
    skipped = 0
    for i in range(B):
        sp = sps[i]
        rc, mm = oracle.get_matches_linear(nk, nk, sp["desc1"], sp["desc2"])
        assert rc == 0 and len(mm) == raw["n_matches"][i], i
        p1, p2 = _cam(sp["kp1"][mm["queryIdx"]], K), _cam(sp["kp2"][mm["trainIdx"]], K)
        o = oracle.arrsac_essential(p1, p2, th, refine=True)
        if not np.array_equal(o["rng_state"], states[i]):
            skipped += 1
            continue
        assert o["ok"] and raw["status"][i] == 0 and o["n_inliers"] == raw["n_inliers"][i], (i, o["n_inliers"], raw[i])
        assert e_dist(o["E"], raw["E"][i]) < 1e-7, i
        good, R, t, Q, mk = oracle.recover_pose(o["E"], p1, p2, 50.0, o["mask"])
        assert good == raw["n_good"][i], i
        assert np.abs(raw["R"][i].reshape(3, 3) - R).max() < 1e-6 and np.abs(raw["t"][i] - np.asarray(t).ravel()).max() < 1e-6, i   # north_star's bar
    assert skipped <= 1


@pytest.mark.parametrize("B", [1, 2, 8])
def test_pair_entries_as_the_first_call_of_a_fresh_context(B):
    """the USAC / ARRSAC pair entries took the context's pinned block BEFORE the nested matching call, which reallocated it
    when it built its split table (B * 4096-keypoint pairs hit exactly that path on 256 CUs) -- the entry then wrote through dangling
    pointers.  Each entry as the very first call on a fresh context, against the same entry on a warmed-up context, byte for byte."""
    import torch
    import matchinglib_poselib_amd as mpa
    from matchinglib_poselib_amd import batch, synth

    dev = torch.device("cuda:0")
    nk = 4096
    sps = [synth.stereo_pair(nk, seed=990 + i, unmatched_frac=0.3) for i in range(B)]
    K = sps[0]["K"]
    stk = [torch.from_numpy(np.stack([sp[k] for sp in sps])).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")]
    seeds = [5 + i for i in range(B)]
    warm = mpa.Context(0)
    try:
        for _ in range(2):
            ref_u = batch.process_pairs_batched_usac(warm, *stk, K, K, seeds, prosac=False)[1].copy()
            ref_p = batch.process_pairs_batched_usac(warm, *stk, K, K, seeds, prosac=True)[1].copy()
            ref_a = batch.process_pairs_batched_arrsac(warm, *stk, K, K, refine=True)[1].copy()
    finally:
        warm.close()
    for which, ref in (("usac", ref_u), ("usac_prosac", ref_p), ("arrsac", ref_a)):
        fresh = mpa.Context(0)
        try:
            if which == "arrsac":
                got = batch.process_pairs_batched_arrsac(fresh, *stk, K, K, refine=True)[1]
            else:
                got = batch.process_pairs_batched_usac(fresh, *stk, K, K, seeds, prosac=(which == "usac_prosac"))[1]
            assert (got["status"] == 0).all() and got.tobytes() == ref.tobytes(), (which, B)
        finally:
            fresh.close()
