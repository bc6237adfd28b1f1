"""mlpl_usac_essential_batch_dev (csrc/batch_hub.h: every problem's sequential program on its own stack -- a fiber on a worker thread --,
the launches of all runs merged into one launch per kernel, up to four cohorts of runs in flight) against mlpl_usac_essential problem by problem: results, masks and decision traces identical."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden  # noqa: E402

pytestmark = pytest.mark.gpu


def _problems(B, rng):
    sizes = [int(v) for v in rng.choice([64, 150, 300, 800, 1200, 2000, 3000, 5000], B)]
    fr = rng.uniform(0.25, 0.9, B)
    scenes = [make_golden.usac_scene(sizes[b], float(fr[b]), 500 + b) for b in range(B)]
    return sizes, scenes


@pytest.mark.parametrize("refine,estimator,degen", [(0, 0, 0), (5, 2, 1), (0, 0, 3), (7, 0, 0)])
def test_batch_equals_the_single_problem_entry_on_64_problems(ctx, refine, estimator, degen):
    """64 problems of 64 ... 5000 correspondences, 25-90 % inliers, uniform and PROSAC sampling mixed, each with its own seed; REF_WEIGHTS and
    ConfigUSAC's defaults (POSE_STEWENIUS + REF_STEWENIUS_WEIGHTS + the degeneracy tests): per problem the same results[12], model, inlier
    mask, degeneracy verdict and the same decision trace event by event as mlpl_usac_essential."""
    import torch
    from matchinglib_poselib_amd import pose

    rng = np.random.default_rng(3 + refine)
    B = 64
    sizes, scenes = _problems(B, rng)
    stride = max(sizes)
    p1, p2 = np.zeros((B, stride, 2)), np.zeros((B, stride, 2))
    th = scenes[0][2]
    orders = []
    for b, (a, c, t, truth, order) in enumerate(scenes):
        p1[b, :sizes[b]], p2[b, :sizes[b]] = a, c
        orders.append(order if b % 3 == 1 else None)
    sizes[5] = 4            # solve() refuses
    seeds = [9000 + 13 * b for b in range(B)]
    dev = torch.device("cuda:0")
    d1, d2 = torch.from_numpy(p1).to(dev), torch.from_numpy(p2).to(dev)
    masks = torch.zeros((B, stride), dtype=torch.uint8, device=dev)
    kw = dict(refine=refine, estimator=estimator, check_degeneracy=degen, max_hyp=3000, sprt_ms=6.0, sprt_tm=2736.0)
    cap = 40000
    got = pose.usac_essential_batch(d1, d2, sizes, th, seeds, sorted_idx=orders, event_cap=cap, masks_out=masks, ctx=ctx, **kw)
    mh = masks.cpu().numpy()
    for b in range(B):
        n = sizes[b]
        one = pose.usac_essential(p1[b, :n], p2[b, :n], th, seeds[b], sorted_idx=orders[b], event_cap=cap, ctx=ctx, **kw)
        g = got[b]
        assert g["ok"] == one["ok"], b
        assert np.array_equal(g["final"], one["final"]), (b, g["final"], one["final"])
        if not one["ok"]:
            continue
        assert np.array_equal(g["E"].view(np.uint64), one["E"].view(np.uint64)), b
        assert np.array_equal(mh[b, :n], one["flags"]), b
        assert g["n_events"] == one["n_events"] and g["n_events"] <= cap and np.array_equal(g["events"], one["events"]), b
        if degen:
            assert np.array_equal(g["degen"], one["degen"]) and np.array_equal(g["R_degen"], one["R_degen"]), b
    assert got[0]["stats"][0] > 0 and got[0]["stats"][1] > 0      # rounds, merged launches


def test_batch_of_one_and_of_more_than_an_internal_batch(ctx):
    """B = 1, 7 (one cohort), 33 (four ragged cohorts), 150, and 600 (five cohorts of 128 on four lanes: a lane serves two): same as
    the single entry."""
    import torch
    from matchinglib_poselib_amd import pose

    a, c, th, truth, order = make_golden.usac_scene(600, 0.6, 77)
    dev = torch.device("cuda:0")
    for B in (1, 7, 33, 150, 600):
        d1 = torch.from_numpy(np.repeat(a[None], B, 0).copy()).to(dev)
        d2 = torch.from_numpy(np.repeat(c[None], B, 0).copy()).to(dev)
        seeds = [40 + (b % 7) for b in range(B)]
        got = pose.usac_essential_batch(d1, d2, [600] * B, th, seeds, refine=5, estimator=2, ctx=ctx)
        ref = {s: pose.usac_essential(a, c, th, s, refine=5, estimator=2, ctx=ctx) for s in sorted(set(seeds))}
        for b in range(B):
            assert got[b]["ok"] and np.array_equal(got[b]["final"], ref[seeds[b]]["final"]) and np.array_equal(got[b]["E"], ref[seeds[b]]["E"]), (B, b)


@pytest.mark.parametrize("prosac,refine", [(False, 0), (True, 0), (False, 5)])
def test_batched_image_pairs_with_usac_equal_the_single_problem_entries(ctx, prosac, refine):
    """mlpl_pair_pose_batch_usac_dev (match -> gather -> USAC -> cheirality for a batch of image pairs) against the chain of single-problem
    entries per pair: match counts, USAC inliers and hypotheses, E, R, t bit for bit; a pair with too few matches -> status -1."""
    import ctypes as C
    import torch
    from matchinglib_poselib_amd import batch, pose, synth
    from matchinglib_poselib_amd.matching import match_hamming_device

    dev = torch.device("cuda:0")
    B, nk = 20, 1024
    sps = [synth.stereo_pair(nk, seed=700 + i, unmatched_frac=0.3 + 0.03 * (i % 5)) for i in range(B)]
    rng = np.random.default_rng(1)
    sps[3]["desc2"] = rng.integers(0, 256, sps[3]["desc2"].shape, dtype=np.uint8)     # nothing matches: status -1
    K = sps[0]["K"]
    stk = [torch.from_numpy(np.stack([sp[k] for sp in sps])).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")]
    seeds = [300 + 7 * i for i in range(B)]
    kw = dict(estimator=2, refine=refine, sprt_ms=6.0, sprt_tm=2736.0)
    rec, raw = batch.process_pairs_batched_usac(ctx, *stk, K, K, seeds, prosac=prosac, **kw)
    th = 0.8 * 4.0 / (np.sqrt(2.0) * (2 * K[0] + 2 * K[1]))
    k4 = (C.c_double * 4)(*K)
    for i in range(B):
        m = match_hamming_device(stk[0][i], stk[1][i], ctx=ctx)
        cnt = int(m["count"][0].item())
        assert raw["n_matches"][i] == cnt
        if cnt < 16:
            assert raw["status"][i] == -1
            continue
        mm = m["matches"][0, :cnt].contiguous()
        d1 = torch.empty((cnt, 2), dtype=torch.float64, device=dev)
        d2 = torch.empty((cnt, 2), dtype=torch.float64, device=dev)
        st = torch.cuda.current_stream(dev).cuda_stream
        assert ctx.lib.mlpl_gather_match_points_dev(ctx.handle, mm.data_ptr(), cnt, stk[2][i].data_ptr(), stk[3][i].data_ptr(), k4, k4, d1.data_ptr(),
                                                    d2.data_ptr(), st) == 0
        torch.cuda.synchronize()
        order = None
        if prosac:   # poselib::getSortedMatchIdx: std::sort by the matching cost (integer Hamming distances: many ties)
            mh = np.ascontiguousarray(mm.cpu().numpy())
            order = np.zeros(cnt, np.uint32)
            assert ctx.lib.mlpl_sorted_match_idx(mh.ctypes.data, cnt, order.ctypes.data) == 0
            dist = mh[:, 3].view(np.float32)
            assert sorted(order.tolist()) == list(range(cnt)) and (np.diff(dist[order]) >= 0).all() and len(np.unique(dist)) < cnt
        one = pose.usac_essential(d1.cpu().numpy(), d2.cpu().numpy(), th, seeds[i], sorted_idx=order, prosac_beta=0.05, th_pixels=0.8,
                                  focal_length=float((2 * K[0] + 2 * K[1]) / 4.0), ctx=ctx, **kw)
        assert raw["status"][i] == 0 and one["ok"], i
        assert raw["iters"][i] == int(one["final"][1]) and raw["n_inliers"][i] == int(one["final"][5]), (i, raw[i], one["final"])
        assert np.array_equal(raw["E"][i].view(np.uint64), one["E"].view(np.uint64)), i
        ng, R, t = pose.getPoseTriangPts_device(one["E"].reshape(3, 3), d1, d2, mask=torch.from_numpy(one["flags"]).to(dev), ctx=ctx)
        assert raw["n_good"][i] == ng and np.array_equal(raw["R"][i].view(np.uint64), R.ravel().view(np.uint64)), i
        assert np.array_equal(raw["t"][i].view(np.uint64), t.ravel().view(np.uint64)), i


@pytest.mark.parametrize("opts", [dict(hub_lanes=1), dict(hub_lanes=2, hub_workers=3), dict(hub_lanes=4, hub_cohort=8, hub_workers=1),
                                  dict(hub_lanes=3, hub_cohort=16, hub_workers=64, hub_blocking_sync=0),
                                  dict(hub_lanes=8, hub_cohort=8), dict(hub_lanes=6, usac_lo5_fused_fit=0), dict(usac_lo5_fused_fit=0)])
def test_results_do_not_depend_on_lanes_cohorts_or_workers(opts):
    """How the runs are dealt to cohorts, lanes and worker threads is scheduling only: 40 problems under several settings of the options,
    among them one worker for all runs of a cohort (every wait is a fiber switch on one thread), more workers than runs, eight cohorts in
    flight (round 5), and the fit of the 5-point refinement chains as three launches instead of one (usac_lo5_fused_fit = 0; the
    default context runs the fused launch): same models, same decision traces."""
    import torch
    import matchinglib_poselib_amd as mpa
    from matchinglib_poselib_amd import pose

    rng = np.random.default_rng(11)
    B = 40
    sizes, scenes = _problems(B, rng)
    stride = max(sizes)
    p1, p2 = np.zeros((B, stride, 2)), np.zeros((B, stride, 2))
    th = scenes[0][2]
    for b, (a, c, t, truth, order) in enumerate(scenes):
        p1[b, :sizes[b]], p2[b, :sizes[b]] = a, c
    seeds = [77 + b for b in range(B)]
    dev = torch.device("cuda:0")
    d1, d2 = torch.from_numpy(p1).to(dev), torch.from_numpy(p2).to(dev)
    kw = dict(refine=5, estimator=2, max_hyp=2000, sprt_ms=6.0, sprt_tm=2736.0)
    base = mpa.Context(0)
    want = pose.usac_essential_batch(d1, d2, sizes, th, seeds, event_cap=20000, ctx=base, **kw)
    base.close()
    c2 = mpa.Context(0)
    for k, v in opts.items():
        c2.set_option(k, v)
    for _ in range(2):   # the second call reuses the kept fibers, workers and item tables
        got = pose.usac_essential_batch(d1, d2, sizes, th, seeds, event_cap=20000, ctx=c2, **kw)
        for b in range(B):
            assert got[b]["ok"] == want[b]["ok"] and np.array_equal(got[b]["final"], want[b]["final"]), (opts, b)
            assert np.array_equal(got[b]["E"].view(np.uint64), want[b]["E"].view(np.uint64)), (opts, b)
            assert got[b]["n_events"] == want[b]["n_events"] and np.array_equal(got[b]["events"], want[b]["events"]), (opts, b)
    c2.close()


def test_debug_account_of_a_batch_prints_and_changes_nothing():
    """MLPL_USAC_PROF=1 (diagnostics: the runs' host time by section, their statistics, every cohort's times) prints one line per batched call
    to stderr and leaves the results alone."""
    import subprocess
    code = r'''
import os, sys, hashlib
import numpy as np, torch
sys.path.insert(0, os.path.join(os.getcwd(), "tests", "golden"))
import make_golden
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import pose
B = 12
scenes = [make_golden.usac_scene(400, 0.6, 900 + b) for b in range(B)]
p1 = np.stack([s[0] for s in scenes]); p2 = np.stack([s[1] for s in scenes])
dev = torch.device("cuda:0")
ctx = mpa.Context(0)
got = pose.usac_essential_batch(torch.from_numpy(p1).to(dev), torch.from_numpy(p2).to(dev), [400] * B, scenes[0][2], [5 + b for b in range(B)],
                                refine=5, estimator=2, max_hyp=2000, ctx=ctx)
print(hashlib.sha1(b"".join(g["E"].tobytes() + g["final"].tobytes() for g in got)).hexdigest())
'''
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    outs = []
    for prof in ("0", "1"):
        env = dict(os.environ, MLPL_USAC_PROF=prof)
        r = subprocess.run([sys.executable, "-c", code], cwd=root, env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0, r.stderr[-2000:]
        outs.append((r.stdout.strip().splitlines()[-1], r.stderr))
    assert outs[0][0] == outs[1][0]
    assert "[mlpl usac prof]" in outs[1][1] and "[mlpl usac prof]" not in outs[0][1]
    assert "samples solved" in outs[1][1] and "cohorts" in outs[1][1]
