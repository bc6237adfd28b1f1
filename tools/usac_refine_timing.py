"""Wall time of one USAC call on the C3 scene per inner refinement (poselib::RefineAlg 0 = REF_WEIGHTS, 5 = REF_STEWENIUS_WEIGHTS --
ConfigUSAC's default --, 4, 7, 6), uniform and PROSAC sampling, with the library's statistics of the last call:
[batches, samples solved, samples consumed, LO launches, LO resumes, choice rechecks, chains re-run after a recheck, Jacobi sweeps]."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))
import numpy as np
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import pose
import make_golden

ctx = mpa.Context(0)
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 20
if len(sys.argv) > 2:   # A/B of the warm-started refits (option usac_lo_warm_start)
    ctx.set_option("usac_lo_warm_start", int(sys.argv[2]))
for (n, frac, seed) in ((5000, 0.5, 20260103), (2000, 0.7, 12), (8192, 0.25, 14)):
    p1, p2, th, truth, order = make_golden.usac_scene(n, frac, seed)
    for refine in (0, 5, 4, 7, 6):
        for prosac in (0, 1):
            kw = dict(sorted_idx=order if prosac else None, refine=refine, estimator=2 if refine in (4, 5) else 0, sprt_ms=6.0, sprt_tm=2736.0, ctx=ctx)
            r = pose.usac_essential(p1, p2, th, 12345, **kw)
            ts = []
            for k in range(reps):
                t0 = time.perf_counter()
                r = pose.usac_essential(p1, p2, th, 12345 + k, **kw)
                ts.append((time.perf_counter() - t0) * 1e3)
            print(f"n {n} inl {frac} refine {refine} prosac {prosac}: median {np.median(ts):.3f} min {min(ts):.3f} max {max(ts):.3f} ms; last: hyps {int(r['final'][1])} "
                  f"inliers {int(r['final'][5])} LOs {int(r['final'][7])} stats {r['stats'].tolist()}", flush=True)
