"""GPU: the whole per-pair pipeline on device-resident inputs (C5 unit) against the oracle pipeline."""
import numpy as np
import pytest
import torch

from matchinglib_poselib_amd import batch, synth

pytestmark = pytest.mark.gpu


def oracle_pipeline(oracle, sp, seed, max_iters=1000):
    n = len(sp["desc1"])
    rc, m = oracle.get_matches_linear(n, n, sp["desc1"], sp["desc2"])
    assert rc == 0
    K = sp["K"]
    a = sp["kp1"][m["queryIdx"]]
    b = sp["kp2"][m["trainIdx"]]
    cam = lambda p: np.stack([((p[:, 0].astype(np.float64) - K[2]) / K[0]).astype(np.float32),  # noqa: E731
                              ((p[:, 1].astype(np.float64) - K[3]) / K[1]).astype(np.float32)], axis=1).astype(np.float64)
    p1, p2 = cam(a), cam(b)
    th = 0.8 * 4.0 / (np.sqrt(2.0) * (2 * K[0] + 2 * K[1]))
    r = oracle.ransac_essential(p1, p2, th, confidence=0.999, max_iters=max_iters, lesqu=False, seed=seed)
    good, R, t, Q, mk = oracle.recover_pose(r["E"], p1, p2, 50.0, r["mask"])
    return len(m), r, R, t


@pytest.mark.parametrize("seed", [11, 12])
def test_pair_pipeline_vs_oracle(ctx, oracle, seed):
    sp = synth.stereo_pair(2500, seed=20260200 + seed)
    dev = torch.device("cuda", 0)
    dq = torch.from_numpy(sp["desc1"]).to(dev)
    dt = torch.from_numpy(sp["desc2"]).to(dev)
    k1 = torch.from_numpy(sp["kp1"]).to(dev)
    k2 = torch.from_numpy(sp["kp2"]).to(dev)
    rec = batch.process_pair_on_device(ctx, dq, dt, k1, k2, sp["K"], sp["K"], seed=seed, pair_id=5)[0]
    nm, r, R, t = oracle_pipeline(oracle, sp, seed)
    assert rec["pair_id"] == 5 and rec["status"] == 0
    assert rec["n_matches"] == nm and rec["n_inliers"] == r["n_inliers"]
    E = rec["E"].reshape(3, 3)
    assert min(np.abs(E - r["E"]).max(), np.abs(E + r["E"]).max()) < 1e-8
    assert np.abs(rec["R"].reshape(3, 3) - R).max() < 1e-6 and np.abs(rec["t"] - t).max() < 1e-6
    # and the pose is the scene's pose
    assert np.abs(rec["R"].reshape(3, 3) - sp["R"]).max() < 2e-2


def test_single_rank_gather_is_identity(ctx):
    local = np.zeros(3, batch.RECORD_DTYPE)
    local["pair_id"] = [0, 1, 2]
    local["n_matches"] = [5, 6, 7]
    rec = batch.gather_records(local, 3, 0, 1, device=torch.device("cuda", 0))
    assert rec.tobytes() == local.tobytes()


def test_concurrent_pair_workers_equal_sequential(ctx):
    """Several pairs in flight on one GPU (independent contexts / streams / threads) give the records of the sequential loop."""
    import torch

    dev = torch.device("cuda:0")
    sps = [synth.stereo_pair(2048, seed=900 + i) for i in range(6)]
    pairs = [tuple(torch.from_numpy(sp[k]).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")) for sp in sps]
    K = sps[0]["K"]
    seq = np.concatenate([batch.process_pair_on_device(ctx, *pairs[i], K, K, seed=40 + i, pair_id=i) for i in range(6)])
    pw = batch.PairWorkers(0, workers=3)
    try:
        con = pw.process(pairs, K, K, seeds=[40 + i for i in range(6)])
    finally:
        pw.close()
    assert con.tobytes() == seq.tobytes()
    assert (seq["status"] == 0).all() and (seq["n_inliers"] > 100).all()


def _stack(sps, dev):
    return tuple(torch.from_numpy(np.stack([sp[k] for sp in sps])).to(dev) for k in ("desc1", "desc2", "kp1", "kp2"))


def test_batched_pairs_equal_the_single_pair_pipeline(ctx):
    """mlpl_pair_pose_batch_dev: 70 pairs of varied inlier ratio and match count -- pairs that stop in the
    first pass of 324 iterations and pairs that need the second -- give byte-identical records to mlpl_pair_pose_dev pair by pair."""
    dev = torch.device("cuda:0")
    fracs = [0.5, 0.3, 0.7, 0.2, 0.45, 0.9, 0.25]
    sps = [synth.stereo_pair(1536, seed=700 + i, inlier_frac=fracs[i % 7], unmatched_frac=0.1 * (i % 4)) for i in range(70)]
    seeds = [9000 + 13 * i for i in range(70)]
    K = sps[0]["K"]
    one = np.concatenate([batch.process_pair_on_device(ctx, *(torch.from_numpy(sp[k]).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")), K, K,
                                                       seed=seeds[i], pair_id=i) for i, sp in enumerate(sps)])
    dq, dt, k1, k2 = _stack(sps, dev)
    bat = batch.process_pairs_batched(ctx, dq, dt, k1, k2, K, K, seeds)
    stats = np.zeros(8, np.int64)
    ctx.lib.mlpl_pair_batch_last_stats(ctx.handle, stats.ctypes.data)
    for i in range(70):
        assert bat[i].tobytes() == one[i].tobytes(), (i, bat[i], one[i])
    assert (one["status"] == 0).all()
    assert stats[0] == 2 and 70 < stats[1] < 140, stats      # a second pass ran, and only for part of the pairs
    assert stats[6] >= 70 * 20 and stats[4] > stats[6] and stats[5] > 100 * stats[4], stats   # iterations, matrices scored, evaluations
    # smaller internal batches (16 + 16 + 16 + 16 + 6) give the same records
    ctx.set_option("pair_batch", 16)
    try:
        assert batch.process_pairs_batched(ctx, dq, dt, k1, k2, K, K, seeds).tobytes() == bat.tobytes()
    finally:
        ctx.set_option("pair_batch", 0)


def test_batched_pairs_edge_cases(ctx, oracle):
    """A pair with too few matches (status -1), a pair of pure outliers (status -2 or a model with few inliers, as the single-pair
    pipeline), long runs (max_iters 3000: three passes), and the refit variant (served pair by pair)."""
    dev = torch.device("cuda:0")
    sps = [synth.stereo_pair(1024, seed=800 + i, inlier_frac=0.5) for i in range(5)]
    rng = np.random.default_rng(3)
    sps[1]["desc1"] = rng.integers(0, 256, sps[1]["desc1"].shape, dtype=np.uint8)            # nothing passes the ratio test
    sps[3]["kp2"] = rng.uniform(0, 640, sps[3]["kp2"].shape).astype(np.float32)              # matches without geometry
    K = sps[0]["K"]
    seeds = [5, 6, 7, 8, 9]
    dq, dt, k1, k2 = _stack(sps, dev)
    for kw in (dict(), dict(max_iters=3000, confidence=0.9999999), dict(refit=True)):
        one = np.concatenate([batch.process_pair_on_device(ctx, *(torch.from_numpy(sp[k]).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")), K, K,
                                                           seed=seeds[i], pair_id=i, **kw) for i, sp in enumerate(sps)])
        bat = batch.process_pairs_batched(ctx, dq, dt, k1, k2, K, K, seeds, **kw)
        assert bat.tobytes() == one.tobytes(), (kw, bat, one)
        assert one["status"][1] == -1 and one["status"][0] == 0


def test_batched_pairs_few_matches_and_short_raw_streams(ctx):
    """The samples of a batch are drawn on the device from the pairs' raw rand() streams (one wave per pair, speculating "no repeated
    index" 64 samples at a time).  Pairs with 16-60 matches repeat an index in up to half of their samples: the draw-by-draw path of the
    kernel; a raw stream cut short (test option) runs out in the first or in the second pass: those pairs are redone by the single-pair
    entry.  Records byte-identical to mlpl_pair_pose_dev every time."""
    dev = torch.device("cuda:0")
    sizes = [40, 48, 64, 96, 128, 256, 40, 56, 1024, 72, 1536, 88]
    sps = [synth.stereo_pair(sizes[i], seed=900 + i, inlier_frac=0.3 if i in (8, 10) else 0.55 + 0.03 * (i % 5), unmatched_frac=0.05 * (i % 3))
           for i in range(12)]   # (pairs 8 and 10 need every one of the 1000 iterations: three passes)
    nmax = max(sizes)
    for sp in sps:   # common shape: pad the descriptor / keypoint blocks with rows that match nothing well
        k = nmax - len(sp["desc1"])
        if k:
            rng = np.random.default_rng(len(sp["desc1"]))
            for d in ("desc1", "desc2"):
                sp[d] = np.concatenate([sp[d], rng.integers(0, 256, (k, sp[d].shape[1]), dtype=np.uint8)])
            for kp in ("kp1", "kp2"):
                sp[kp] = np.concatenate([sp[kp], rng.uniform(0, 600, (k, 2)).astype(np.float32)])
    seeds = [4000 + 7 * i for i in range(12)]
    K = sps[0]["K"]
    one = np.concatenate([batch.process_pair_on_device(ctx, *(torch.from_numpy(sp[k]).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")), K, K,
                                                       seed=seeds[i], pair_id=i) for i, sp in enumerate(sps)])
    assert 16 <= one["n_matches"].min() < 64 and (one["status"] == 0).sum() >= 10, one["n_matches"]
    dq, dt, k1, k2 = _stack(sps, dev)
    stats = np.zeros(8, np.int64)
    for cap, redone in ((0, False), (1500, True), (2200, True), (400, True)):
        ctx.set_option("pair_batch_raw_cap", cap)
        try:
            bat = batch.process_pairs_batched(ctx, dq, dt, k1, k2, K, K, seeds)
        finally:
            ctx.set_option("pair_batch_raw_cap", 0)
        ctx.lib.mlpl_pair_batch_last_stats(ctx.handle, stats.ctypes.data)
        for i in range(12):
            assert bat[i].tobytes() == one[i].tobytes(), (cap, i, bat[i], one[i])
        assert (stats[2] > 0) == redone, (cap, stats)


@pytest.mark.parametrize("recover", [True, False])
def test_batched_correspondence_sets_equal_the_single_problem_entries(ctx, recover):
    """mlpl_ransac_essential_batch_dev: 40 correspondence sets of 5...3000 points, 20-90 % inliers, each with its own seed, against
    mlpl_ransac_essential_dev (refit = 0) + mlpl_recover_pose_dev problem by problem: iteration counts, inlier counts, E, R, t and the masks
    bit for bit; fewer than 6 correspondences -> status -1; pure noise -> whatever the single entry says."""
    from matchinglib_poselib_amd import pose

    dev = torch.device("cuda:0")
    rng = np.random.default_rng(11)
    sizes = [5, 6, 9, 16, 40, 100, 333, 1000, 3000, 64] * 4
    stride = max(sizes)
    p1 = np.zeros((40, stride, 2))
    p2 = np.zeros((40, stride, 2))
    ths = None
    for i, n in enumerate(sizes):
        a, b, R, t, mask, th = synth.pose_scene(max(n, 8), inlier_frac=float(rng.choice([0.2, 0.5, 0.7, 0.9])), seed=1200 + i)
        if i == 17:
            b = rng.uniform(-0.4, 0.4, b.shape)       # no geometry at all
        p1[i, :n], p2[i, :n] = a[:n], b[:n]
        ths = th
    seeds = [700 + 3 * i for i in range(40)]
    d1, d2 = torch.from_numpy(p1).to(dev), torch.from_numpy(p2).to(dev)
    masks = torch.zeros((40, stride), dtype=torch.uint8, device=dev)
    got = batch.ransac_pose_batched(ctx, d1, d2, sizes, seeds, ths, recover_pose=recover, masks_out=masks)
    mh = masks.cpu().numpy()
    # the same batch with the raw rand() streams cut short: the small problems (many redraws) and the ones that need a second pass run out
    # of stream and are redone through the single-problem entry -- several of ONE internal batch, one after the other (the nested calls
    # reuse the context's pinned block: ADVICE r3), with the same records and masks
    masks2 = torch.zeros((40, stride), dtype=torch.uint8, device=dev)
    stats = np.zeros(8, np.int64)
    ctx.set_option("pair_batch_raw_cap", 1700)
    try:
        got2 = batch.ransac_pose_batched(ctx, d1, d2, sizes, seeds, ths, recover_pose=recover, masks_out=masks2)
    finally:
        ctx.set_option("pair_batch_raw_cap", 0)
    ctx.lib.mlpl_pair_batch_last_stats(ctx.handle, stats.ctypes.data)
    assert stats[2] >= 2, stats
    for g, h in zip(got, got2):
        assert all(np.array_equal(np.asarray(g[k]), np.asarray(h[k])) for k in g), (g, h)
    mh2 = masks2.cpu().numpy()
    for i, n in enumerate(sizes):
        if n >= 6 and got[i]["status"] == 0:
            assert np.array_equal(mh2[i, :n], mh[i, :n]), i
    for i, n in enumerate(sizes):
        g = got[i]
        if n < 6:
            assert g["status"] == -1 and g["n"] == n
            continue
        a, b = d1[i, :n].contiguous(), d2[i, :n].contiguous()
        r = pose.ransac_essential_device(a, b, ths, confidence=0.999, max_iters=1000, refit=False, seed=seeds[i], ctx=ctx)
        if not r["ok"]:
            assert g["status"] == -2, i
            continue
        assert g["status"] == 0 and g["iters"] == r["iters"] and g["n_inliers"] == r["n_inliers"], (i, g, r)
        assert np.array_equal(g["E"].view(np.uint64), np.asarray(r["E"]).view(np.uint64)), i
        m = r["mask"]
        if recover:
            ng, R, t = pose.getPoseTriangPts_device(r["E"], a, b, mask=m, ctx=ctx)
            assert g["n_good"] == ng and np.array_equal(g["R"].view(np.uint64), R.view(np.uint64)) and np.array_equal(g["t"].view(np.uint64), t.ravel().view(np.uint64)), i
        else:
            assert g["n_good"] == 0 and not g["R"].any()
        assert np.array_equal(mh[i, :n], m.cpu().numpy()), i


def test_two_batched_calls_in_flight_give_the_same_records(ctx):
    """batch.BatchLanes: the batch split over two library contexts / streams / host threads -- byte-identical records and match lists."""
    import torch
    from matchinglib_poselib_amd import batch, synth

    dev = torch.device("cuda", 0)
    sps = [synth.stereo_pair(1024, seed=900 + i, unmatched_frac=0.3) for i in range(5)]
    K = sps[0]["K"]
    B = 13
    stk = [torch.from_numpy(np.stack([sps[i % 5][k] for i in range(B)])).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")]
    seeds = [7 + i for i in range(B)]
    m1 = torch.zeros((B, 1024, 4), dtype=torch.int32, device=dev)
    m2 = torch.zeros((B, 1024, 4), dtype=torch.int32, device=dev)
    one = batch.process_pairs_batched(ctx, *stk, K, K, seeds, matches_out=m1)
    lanes = batch.BatchLanes(0, lanes=2, first_ctx=ctx)
    m3 = torch.zeros((B, 1024, 4), dtype=torch.int32, device=dev)
    try:
        two = lanes.process(*stk, K, K, seeds, matches_out=m2)                    # lanes inside the library (mlpl_pair_pose_batch_lanes_dev)
        two_py = lanes.process(*stk, K, K, seeds, matches_out=m3, native=False)   # lanes as Python threads
        l3 = batch.BatchLanes(0, lanes=3)
        try:
            three = l3.process(*stk, K, K, seeds)
            one_pair = l3.process(*[t[:1] for t in stk], K, K, seeds[:1])         # fewer pairs than lanes
        finally:
            l3.close()
    finally:
        lanes.close()
    assert one.tobytes() == two.tobytes() == three.tobytes() == two_py.tobytes() and one_pair.tobytes() == one[:1].tobytes()
    assert all(b >= a for a, b in lanes.last_lane_span)
    for b in range(B):
        k = int(one["n_matches"][b])
        assert torch.equal(m1[b, :k], m2[b, :k]) and torch.equal(m1[b, :k], m3[b, :k])


def test_lanes_entry_refuses_a_context_shared_by_two_lanes(ctx):
    """mlpl_pair_pose_batch_lanes_dev: a context serves one call at a time (its workspaces and pinned blocks belong to the call), so two lanes
    on ONE context are refused with MLPL_E_BAD_INPUT and a message instead of racing."""
    import ctypes as C

    import torch
    from matchinglib_poselib_amd import batch, synth

    dev = torch.device("cuda", 0)
    sp = synth.stereo_pair(512, seed=31, unmatched_frac=0.3)
    B = 4
    stk = [torch.from_numpy(np.stack([sp[k]] * B)).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")]
    K = sp["K"]
    k0 = (C.c_double * 4)(*K)
    ctxs = (C.c_void_p * 2)(ctx.handle, ctx.handle)
    s0, s1 = torch.cuda.Stream(dev), torch.cuda.Stream(dev)
    strs = (C.c_void_p * 2)(s0.cuda_stream, s1.cuda_stream)
    sd = np.arange(B, dtype=np.uint32)
    res = (batch._PairResult * B)()
    th = 0.8 * 4.0 / (np.sqrt(2.0) * 2 * (K[0] + K[1]))
    torch.cuda.synchronize()
    rc = ctx.lib.mlpl_pair_pose_batch_lanes_dev(ctxs, strs, 2, B, stk[0].data_ptr(), 512, stk[1].data_ptr(), 512, 32, stk[2].data_ptr(), stk[3].data_ptr(),
                                                k0, k0, float(th), 1000, 0.999, sd.ctypes.data, 50.0, C.addressof(res), None, None)
    assert rc == -1 and b"share a context" in ctx.lib.mlpl_last_error()   # MLPL_E_BAD_INPUT
