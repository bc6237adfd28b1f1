"""Soak of the launch hub: random batches of USAC and ARRSAC problems (random sizes, inlier ratios, seeds, refinements, PROSAC on some)
under random settings of hub_lanes / hub_cohort / hub_workers, every problem's results against the single-problem entry's (bit for bit).
usage: python tools/stress_batch.py [seconds=120]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import pose
import make_golden

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 120.0
rng = np.random.default_rng(20261005)
dev = torch.device("cuda:0")
single = mpa.Context(0)
scenes = {}
def scene(n, fr, sd):
    k = (n, round(fr, 2), sd)
    if k not in scenes:
        scenes[k] = make_golden.usac_scene(n, fr, sd)
    return scenes[k]
t0 = time.time(); rounds = probs = bad = 0
while time.time() - t0 < budget:
    B = int(rng.choice([1, 3, 9, 17, 40, 64, 130, 300]))
    sizes = [int(v) for v in rng.choice([40, 120, 300, 800, 2000], B)]
    frs = rng.choice([0.3, 0.5, 0.8], B)
    sds = rng.integers(0, 6, B)
    sc = [scene(sizes[b], float(frs[b]), int(sds[b])) for b in range(B)]
    stride = max(sizes)
    p1, p2 = np.zeros((B, stride, 2)), np.zeros((B, stride, 2))
    for b in range(B):
        p1[b, :sizes[b]], p2[b, :sizes[b]] = sc[b][0], sc[b][1]
    th = sc[0][2]
    d1, d2 = torch.from_numpy(p1).to(dev), torch.from_numpy(p2).to(dev)
    ctx = mpa.Context(0)
    opts = dict(hub_lanes=int(rng.integers(0, 9)), hub_cohort=int(rng.choice([8, 16, 64, 128])), hub_workers=int(rng.choice([1, 2, 7, 16])),
                hub_blocking_sync=int(rng.integers(0, 2)), usac_lo5_fused_fit=int(rng.integers(0, 2)))   # (round 5: up to 8 lanes, 0 = the estimator's default; the fused fit launch on / off)
    for k, v in opts.items():
        ctx.set_option(k, v)
    seeds = [int(v) for v in rng.integers(1, 1 << 30, B)]
    if rng.random() < 0.6:
        refine, est = [(0, 0), (5, 2), (7, 0)][int(rng.integers(0, 3))]
        orders = [sc[b][4] if rng.random() < 0.3 else None for b in range(B)]
        kw = dict(refine=refine, estimator=est, max_hyp=1500, sprt_ms=6.0, sprt_tm=2736.0, check_degeneracy=int(rng.choice([0, 0, 1])) if refine == 0 else 0)
        masks = torch.zeros((B, stride), dtype=torch.uint8, device=dev)
        got = pose.usac_essential_batch(d1, d2, sizes, th, seeds, sorted_idx=orders, masks_out=masks, ctx=ctx, **kw)
        mh = masks.cpu().numpy()
        for b in range(B):
            one = pose.usac_essential(p1[b, :sizes[b]], p2[b, :sizes[b]], th, seeds[b], sorted_idx=orders[b], ctx=single, **kw)
            ok = got[b]["ok"] == one["ok"] and np.array_equal(got[b]["final"], one["final"])
            if ok and one["ok"]:
                ok = np.array_equal(got[b]["E"].view(np.uint64), one["E"].view(np.uint64)) and np.array_equal(mh[b, :sizes[b]], one["flags"])
            if not ok:
                bad += 1
                print("USAC MISMATCH", opts, kw, b, sizes[b], seeds[b], got[b]["final"], one["final"], flush=True)
    else:
        st = np.tile(np.array(pose.ARRSAC_RNG_FRESH, np.uint64), (B, 1))
        refine = bool(rng.integers(0, 2))
        masks = torch.zeros((B, stride), dtype=torch.uint8, device=dev)
        got = pose.arrsac_essential_batch(d1, d2, sizes, th, refine=refine, rng_states=st, masks_out=masks, ctx=ctx)
        mh = masks.cpu().numpy()
        for b in range(B):
            s1 = np.array(pose.ARRSAC_RNG_FRESH, np.uint64)
            one = pose.arrsac_essential(p1[b, :sizes[b]], p2[b, :sizes[b]], th, refine=refine, rng_state=s1, ctx=single)
            ok = got[b]["ok"] == one["ok"] and np.array_equal(st[b], s1)
            if ok and one["ok"]:
                ok = np.array_equal(got[b]["E"].ravel().view(np.uint64), one["E"].ravel().view(np.uint64)) and np.array_equal(mh[b, :sizes[b]], one["mask"]) and got[b]["n_inliers"] == one["n_inliers"]
            if not ok:
                bad += 1
                print("ARRSAC MISMATCH", opts, refine, b, sizes[b], flush=True)
    ctx.close()
    rounds += 1; probs += B
print(f"{rounds} batches, {probs} problems in {time.time() - t0:.0f} s: {bad} mismatches against the single-problem entries")
