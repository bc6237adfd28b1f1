"""Random scenes: mlpl_arrsac_essential against the sequential oracle (statistics, stream positions, masks, E).  Prints every mismatch."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import pose, synth
import oracle_lib
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
polish = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ora = oracle_lib.load(); ctx = mpa.Context(0); ctx.set_option("solver_polish", polish)
rng = np.random.default_rng(99)
bad = 0; kinds = {}
st_g = np.array(pose.ARRSAC_RNG_FRESH, np.uint64); st_o = st_g.copy()      # ONE stream pair across all scenes, like one process
t0 = time.time()
for it in range(N):
    n = int(rng.choice([60, 99, 100, 101, 150, 250, 600, 1500, 4000]))
    frac = float(rng.choice([0.2, 0.35, 0.5, 0.7, 0.85, 0.95, 1.0]))
    noise = float(rng.choice([0.0, 0.3, 1.0]))
    p1, p2, R, t, truth, th = synth.pose_scene(n, frac, seed=1000 + it, noise_px=noise)
    refine = bool(it & 1)
    g = pose.arrsac_essential(p1, p2, th, refine=refine, rng_state=st_g, ctx=ctx)
    o = ora.arrsac_essential(p1, p2, th, refine=refine, rng_state=st_o)
    same = g["ok"] == o["ok"] and g["stats"][:8].tolist() == o["stats"].tolist() and np.array_equal(st_g, st_o) and np.array_equal(g["mask"], o["mask"])
    dE = 0.0
    if same and g["ok"]:
        a, b = g["E"] / np.linalg.norm(g["E"]), o["E"] / np.linalg.norm(o["E"])
        dE = min(np.abs(a - b).max(), np.abs(a + b).max())
        same = dE < (2e-5 if polish else 1e-7)
    key = (o["stats"][3] > 0, o["stats"][5] > 0, bool(o["ok"]))
    kinds[key] = kinds.get(key, 0) + 1
    if not same:
        bad += 1
        print(f"MISMATCH scene {it}: n={n} frac={frac} noise={noise} refine={refine} ok {g['ok']}/{o['ok']} stats {g['stats'][:8].tolist()} vs {o['stats'].tolist()} "
              f"mask diff {(g['mask'] != o['mask']).sum()} dE {dE:.2e}")
        st_g[:] = st_o                                                       # resynchronise the streams
print(f"{N} scenes, {bad} mismatches, {time.time() - t0:.1f} s; (inner RANSAC used, generation branch used, ok) histogram: {kinds}")
