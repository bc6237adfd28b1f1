"""Round-2 diagnostics (GPU box): (1) the 70 000-iteration n=64 run that stopped early, (2) solver disagreements in detail."""
import sys, os
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import oracle_lib
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import pose, synth

ora = oracle_lib.load()
ctx = mpa.Context(0)

def e_dist(a, b):
    return min(np.linalg.norm(a - b), np.linalg.norm(a + b))

if "ransac" in sys.argv:
    n, iters = 64, 70000
    p1, p2, R, t, mask, th = synth.pose_scene(n, inlier_frac=0.4, seed=4000 + n)
    o = ora.ransac_essential(p1, p2, th, confidence=1.0, max_iters=iters, lesqu=False, seed=n, trace=True)
    print("oracle", o["iters"], o["n_inliers"])
    tr = o["trace"]
    best = 0
    for i in range(o["iters"]):
        if tr[i].best_taken >= 0:
            print("  oracle record at iter", i, "good", list(tr[i].good)[:tr[i].nmodels], "taken", tr[i].best_taken, "niters_after", tr[i].niters_after)
    for chunk in (0, 8192, 70000, 1000):
        ctx.set_option("ransac_chunk", chunk)
        for lazy in (1, 0):
            ctx.set_option("ransac_lazy_sums", lazy)
            g = pose.ransac_essential(p1, p2, th, confidence=1.0, max_iters=iters, refit=False, seed=n, ctx=ctx)
            print("gpu chunk", chunk, "lazy", lazy, g["iters"], g["n_inliers"], e_dist(g["E"], o["E"]))
    ctx.set_option("ransac_chunk", 0); ctx.set_option("ransac_lazy_sums", 1)
    # which hypothesis gives the GPU its count?  score every model of the GPU solver and of the oracle near the stop iteration
    samples = ora.sample_table(n, p1, p2, iters)
    E, nm = pose.solve_5pt(p1, p2, samples, ctx=ctx)
    flat = np.concatenate([E[s, :nm[s]] for s in range(iters)])
    owner = np.concatenate([np.full(nm[s], s) for s in range(iters)])
    good, esum = pose.score_models(p1, p2, flat, th, ctx=ctx)
    top = np.argsort(-good, kind="stable")[:10]
    print("gpu top models:", [(int(owner[k]), int(good[k])) for k in top])
    for k in top[:5]:
        s = int(owner[k])
        Eo = ora.run5point(p1[samples[s]], p2[samples[s]])
        og = [ora.find_inliers(p1, p2, e, th)[0] for e in Eo]
        gg = [int(x) for x in good[owner == s]]
        print("  iter", s, "gpu goods", gg, "oracle goods", og, "nm", nm[s], len(Eo))

if "solver" in sys.argv:
    p1, p2, R, t, mask, th = synth.pose_scene(5000, seed=20260103)
    samples = ora.sample_table(12345, p1, p2, 2000)
    E, nm = pose.solve_5pt(p1, p2, samples, ctx=ctx)
    np.set_printoptions(precision=4, linewidth=200)
    for s in (40, 95, 299, 485, 506, 707, 986, 1007):
        q1, q2 = p1[samples[s]], p2[samples[s]]
        Eo, c, roots, z = ora.run5point_dbg(q1, q2)
        Eg = E[s, :nm[s]]
        x1 = np.c_[q1, np.ones(5)]; x2 = np.c_[q2, np.ones(5)]
        print("sample", s, "roots", roots)
        for e in Eo:
            d = [e_dist(e, x) for x in Eg]
            j = int(np.argmin(d))
            res_o = np.abs(np.einsum("ij,jk,ik->i", x2, e, x1)).max()
            res_g = np.abs(np.einsum("ij,jk,ik->i", x2, Eg[j], x1)).max()
            cub = lambda M: np.abs(2 * M @ M.T @ M - np.trace(M @ M.T) * M).max()
            print(f"   dist {d[j]:.3e}  epi oracle {res_o:.2e} gpu {res_g:.2e}  cubic oracle {cub(e):.2e} gpu {cub(Eg[j]):.2e} det o {np.linalg.det(e):.1e} g {np.linalg.det(Eg[j]):.1e}")
        # numpy roots of the oracle polynomial for comparison
        print("   np.roots real:", np.sort(np.roots(c[::-1])[np.abs(np.roots(c[::-1]).imag) < 1e-8].real))
        # conditioning of the 5x9 system: singular values
        Q = np.c_[x1[:, 0] * x2[:, 0], x1[:, 1] * x2[:, 0], x2[:, 0], x1[:, 0] * x2[:, 1], x1[:, 1] * x2[:, 1], x2[:, 1], x1[:, 0], x1[:, 1], np.ones(5)]
        print("   sv(Q):", np.linalg.svd(Q, compute_uv=False))
ctx.close()
