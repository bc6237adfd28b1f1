"""mlpl_arrsac_essential_batch_dev (every problem's ARRSAC on its own host thread, launches merged over the problems: csrc/batch_hub.h) against
mlpl_arrsac_essential problem by problem: model, inlier count, mask and the two sampler stream states after the call."""
import os
import sys

import numpy as np
import pytest

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden"))
import make_golden  # noqa: E402

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("refine", [True, False])
def test_batch_equals_the_single_problem_entry(ctx, refine):
    import torch
    from matchinglib_poselib_amd import pose

    rng = np.random.default_rng(17)
    B = 48
    sizes = [int(v) for v in rng.choice([80, 150, 300, 800, 1200, 2500, 5000], B)]
    fr = rng.uniform(0.3, 0.95, B)
    scenes = [make_golden.usac_scene(sizes[b], float(fr[b]), 900 + b) for b in range(B)]
    stride = max(sizes)
    p1, p2 = np.zeros((B, stride, 2)), np.zeros((B, stride, 2))
    th = scenes[0][2]
    for b, (a, c, t, truth, order) in enumerate(scenes):
        p1[b, :sizes[b]], p2[b, :sizes[b]] = a, c
    dev = torch.device("cuda:0")
    d1, d2 = torch.from_numpy(p1).to(dev), torch.from_numpy(p2).to(dev)
    masks = torch.zeros((B, stride), dtype=torch.uint8, device=dev)
    states = np.array([[0xFFFFFFFF + 977 * b, 0xFFFFFFFF + 31 * b * b] for b in range(B)], np.uint64)
    st_batch = states.copy()
    got = pose.arrsac_essential_batch(d1, d2, sizes, th, refine=refine, rng_states=st_batch, masks_out=masks, ctx=ctx)
    mh = masks.cpu().numpy()
    oks = 0
    for b in range(B):
        n = sizes[b]
        st = states[b].copy()
        one = pose.arrsac_essential(p1[b, :n], p2[b, :n], th, refine=refine, rng_state=st, ctx=ctx)
        g = got[b]
        assert g["ok"] == one["ok"], b
        assert np.array_equal(st, st_batch[b]), (b, st, st_batch[b])          # the samplers' streams stand where the single call left them
        if not one["ok"]:
            continue
        oks += 1
        assert g["n_inliers"] == one["n_inliers"], b
        assert np.array_equal(g["E"].view(np.uint64), one["E"].ravel().view(np.uint64)), b
        assert np.array_equal(mh[b, :n], one["mask"]), b
    assert oks >= B - 6


def test_batched_image_pairs_with_arrsac_equal_the_single_problem_entries(ctx):
    """mlpl_pair_pose_batch_arrsac_dev against match -> gather -> mlpl_arrsac_essential -> getPoseTriangPts per pair."""
    import ctypes as C
    import torch
    from matchinglib_poselib_amd import batch, pose, synth
    from matchinglib_poselib_amd.matching import match_hamming_device

    dev = torch.device("cuda:0")
    B, nk = 16, 1024
    sps = [synth.stereo_pair(nk, seed=800 + i, unmatched_frac=0.3 + 0.03 * (i % 5)) for i in range(B)]
    rng = np.random.default_rng(2)
    sps[2]["desc2"] = rng.integers(0, 256, sps[2]["desc2"].shape, dtype=np.uint8)     # nothing matches: status -1
    K = sps[0]["K"]
    stk = [torch.from_numpy(np.stack([sp[k] for sp in sps])).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")]
    states = np.array([[0xFFFFFFFF + 11 * b, 0xFFFFFFFF + 5 * b] for b in range(B)], np.uint64)
    st_batch = states.copy()
    rec, raw = batch.process_pairs_batched_arrsac(ctx, *stk, K, K, refine=True, rng_states=st_batch)
    th = 0.8 * 4.0 / (np.sqrt(2.0) * (2 * K[0] + 2 * K[1]))
    k4 = (C.c_double * 4)(*K)
    for i in range(B):
        m = match_hamming_device(stk[0][i], stk[1][i], ctx=ctx)
        cnt = int(m["count"][0].item())
        assert raw["n_matches"][i] == cnt
        if cnt < 16:
            assert raw["status"][i] == -1 and np.array_equal(st_batch[i], states[i])
            continue
        mm = m["matches"][0, :cnt].contiguous()
        d1 = torch.empty((cnt, 2), dtype=torch.float64, device=dev)
        d2 = torch.empty((cnt, 2), dtype=torch.float64, device=dev)
        assert ctx.lib.mlpl_gather_match_points_dev(ctx.handle, mm.data_ptr(), cnt, stk[2][i].data_ptr(), stk[3][i].data_ptr(), k4, k4, d1.data_ptr(),
                                                    d2.data_ptr(), torch.cuda.current_stream(dev).cuda_stream) == 0
        torch.cuda.synchronize()
        st = states[i].copy()
        one = pose.arrsac_essential(d1.cpu().numpy(), d2.cpu().numpy(), th, refine=True, rng_state=st, ctx=ctx)
        assert np.array_equal(st, st_batch[i]), i
        if not one["ok"]:
            assert raw["status"][i] == -2
            continue
        assert raw["status"][i] == 0 and raw["n_inliers"][i] == one["n_inliers"], (i, raw[i])
        assert np.array_equal(raw["E"][i].view(np.uint64), one["E"].ravel().view(np.uint64)), i
        ng, R, t = pose.getPoseTriangPts_device(one["E"], d1, d2, mask=torch.from_numpy(one["mask"]).to(dev), ctx=ctx)
        assert raw["n_good"][i] == ng and np.array_equal(raw["R"][i].view(np.uint64), R.ravel().view(np.uint64)), i
