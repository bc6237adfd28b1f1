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
