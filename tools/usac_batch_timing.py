"""mlpl_usac_essential_batch_dev: wall time per call for B problems of the C3 shape (5000 correspondences, 50 % inliers, distinct scenes and
seeds), per inner refinement and sampling, beside B single-problem calls.  usage: python tools/usac_batch_timing.py [B=512] [reps=3]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np, torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import pose
import make_golden

B = int(sys.argv[1]) if len(sys.argv) > 1 else 512
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
OPTS = sys.argv[3:]   # name=value library options (hub_lanes=2 hub_cohort=128 ...)
n = 5000
ctx = mpa.Context(0)
for o in OPTS:
    ctx.set_option(o.split("=")[0], int(o.split("=")[1]))
distinct = 16
sc = [make_golden.usac_scene(n, 0.5, 20260103 + i) for i in range(distinct)]
th = sc[0][2]
p1 = np.stack([sc[b % distinct][0] for b in range(B)])
p2 = np.stack([sc[b % distinct][1] for b in range(B)])
orders = [sc[b % distinct][4] for b in range(B)]
dev = torch.device("cuda:0")
d1, d2 = torch.from_numpy(p1).to(dev), torch.from_numpy(p2).to(dev)
masks = torch.zeros((B, n), dtype=torch.uint8, device=dev)
seeds = [1000 + b for b in range(B)]
ONLY = os.environ.get("UBT_ONLY")   # "refine,prosac": one configuration only (for a kernel trace of it alone)
for refine, est in ((0, 0), (5, 2)):
    for prosac in (0, 1):
        if ONLY and ONLY != f"{refine},{prosac}":
            continue
        kw = dict(refine=refine, estimator=est, sprt_ms=6.0, sprt_tm=2736.0, ctx=ctx)
        si = orders if prosac else None
        got = pose.usac_essential_batch(d1, d2, [n] * B, th, seeds, sorted_idx=si, masks_out=masks, **kw)
        ts = []
        cpu0, wall0 = time.process_time(), time.perf_counter()
        for _ in range(reps):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            got = pose.usac_essential_batch(d1, d2, [n] * B, th, seeds, sorted_idx=si, masks_out=masks, **kw)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        cores = (time.process_time() - cpu0) / (time.perf_counter() - wall0)   # host cores the batched calls kept busy (all threads)
        k = min(B, 32)
        t0 = time.perf_counter()
        for b in range(k):
            pose.usac_essential(p1[b], p2[b], th, seeds[b], sorted_idx=orders[b] if prosac else None, **kw)
        single = (time.perf_counter() - t0) * 1e3 / k
        inl = np.mean([g["final"][5] for g in got])
        hyp = np.mean([g["final"][1] for g in got])
        print(f"B {B} refine {refine} prosac {prosac}: batch {min(ts):.2f} ms (runs {[round(t, 1) for t in ts]}, {cores:.1f} host cores busy) = {min(ts) / B * 1e3:.1f} us per problem; one at a time "
              f"{single:.3f} ms per problem ({single * B:.0f} ms for B); mean hyps {hyp:.0f} inliers {inl:.0f}; rounds {got[0]['stats'][0]} merged launches {got[0]['stats'][1]} hub waiting for host us {got[0]['stats'][2]} device us {got[0]['stats'][3]} thread spawn us {got[0]['stats'][4]} runs total us {got[0]['stats'][5]}", flush=True)
