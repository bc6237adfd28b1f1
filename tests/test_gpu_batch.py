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
