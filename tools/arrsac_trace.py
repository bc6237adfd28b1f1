"""Turn-by-turn comparison of ARRSAC's first stage: device batches + host control flow vs the sequential oracle."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import pose, synth
import oracle_lib
n, frac, seed, polish = int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4])
ora = oracle_lib.load(); ctx = mpa.Context(0); ctx.set_option("solver_polish", polish)
p1, p2, R, t, truth, th = synth.pose_scene(n, frac, seed=seed)
bg = np.zeros(20 * 4000, np.int32); bo = np.zeros(20 * 4000, np.int32)
ctx.lib.mlpl_debug_arrsac_trace(ctx.handle, bg.ctypes.data, len(bg))
ora.lib.oracle_arrsac_trace.argtypes = [C.c_void_p, C.c_int]
ora.lib.oracle_arrsac_trace(bo.ctypes.data, len(bo))
g = pose.arrsac_essential(p1, p2, th, refine=False, rng_state=np.array(pose.ARRSAC_RNG_FRESH, np.uint64), ctx=ctx)
o = ora.arrsac_essential(p1, p2, th, refine=False)
lg = ctx.lib.mlpl_debug_arrsac_trace(ctx.handle, None, 0); lo = ora.lib.oracle_arrsac_trace(None, 0)
tg = bg[:lg].reshape(-1, 20); to = bo[:lo].reshape(-1, 20)
print("turns", len(tg), len(to), "stats", g["stats"].tolist(), o["stats"].tolist())
for i in range(min(len(tg), len(to))):
    if not np.array_equal(tg[i], to[i]):
        print("first difference at turn", i)
        for j in range(max(0, i - 2), min(i + 3, len(tg), len(to))):
            print(" gpu", tg[j].tolist()); print(" cpu", to[j].tolist())
        idx = to[i, 3:3 + min(5, to[i, 2])]
        if to[i, 2] == 5:
            Eo = np.zeros((10, 9)); 
            nm = ora.run5point(p1[idx], p2[idx])
            print(" oracle 5pt models:", len(nm) if hasattr(nm, '__len__') else nm)
        break
else:
    print("traces agree on the common prefix")
