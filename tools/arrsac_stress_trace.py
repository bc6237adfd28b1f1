"""First differing turn of scene K of tools/stress_arrsac.py (same scene generator, same stream pair: the oracle replays scenes 0..K-1)."""
import ctypes as C, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests"))
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import pose, synth
import oracle_lib
K = int(sys.argv[1]); polish = int(sys.argv[2]) if len(sys.argv) > 2 else 0
ora = oracle_lib.load(); ctx = mpa.Context(0); ctx.set_option("solver_polish", polish)
rng = np.random.default_rng(99)
st = np.array(pose.ARRSAC_RNG_FRESH, np.uint64)
for it in range(K + 1):
    n = int(rng.choice([60, 99, 100, 101, 150, 250, 600, 1500, 4000]))
    frac = float(rng.choice([0.2, 0.35, 0.5, 0.7, 0.85, 0.95, 1.0]))
    noise = float(rng.choice([0.0, 0.3, 1.0]))
    p1, p2, R, t, truth, th = synth.pose_scene(n, frac, seed=1000 + it, noise_px=noise)
    refine = bool(it & 1)
    if it < K:
        ora.arrsac_essential(p1, p2, th, refine=refine, rng_state=st)
bg = np.zeros(20 * 4000, np.int32); bo = np.zeros(20 * 4000, np.int32)
ctx.lib.mlpl_debug_arrsac_trace(ctx.handle, bg.ctypes.data, len(bg))
ora.lib.oracle_arrsac_trace.argtypes = [C.c_void_p, C.c_int]
ora.lib.oracle_arrsac_trace(bo.ctypes.data, len(bo))
g = pose.arrsac_essential(p1, p2, th, refine=False, rng_state=st.copy(), ctx=ctx)
o = ora.arrsac_essential(p1, p2, th, refine=False, rng_state=st.copy())
lg = ctx.lib.mlpl_debug_arrsac_trace(ctx.handle, None, 0); lo = ora.lib.oracle_arrsac_trace(None, 0)
tg = bg[:lg].reshape(-1, 20); to = bo[:lo].reshape(-1, 20)
print("turns", len(tg), len(to), g["stats"][:8].tolist(), o["stats"].tolist())
for i in range(min(len(tg), len(to))):
    if not np.array_equal(tg[i], to[i]):
        print("first difference at turn", i); print(" gpu", tg[i].tolist()); print(" cpu", to[i].tolist())
        m = to[i, 2]
        if m in (6, 7):
            idx = list(to[i, 3:8]) + [int(to[i, 19]) % 100] + ([int(to[i, 19]) // 100] if m == 7 else [])
            idx = np.array(idx)
            Eo = ora.run5point(p1[idx], p2[idx])
            A = np.array([[a[0] * b[0], a[1] * b[0], b[0], a[0] * b[1], a[1] * b[1], b[1], a[0], a[1], 1.0] for a, b in zip(p1[idx], p2[idx])])
            print(" sample", idx.tolist(), "singular values of the", A.shape, "system:", np.linalg.svd(A, compute_uv=False))
            w, V = np.linalg.eigh(A.T @ A)
            print(" eigenvalues of the Gram matrix:", w)
            for e in Eo:
                e = np.asarray(e).reshape(3, 3)
                err = ora.sampson_err(p1[:100], p2[:100], e) if hasattr(ora, "sampson_err") else None
                print("   oracle model inliers on the first 100:", None if err is None else int((err < th * th).sum()), "valid", ora.valid_model(p1[idx], p2[idx], e), ora.valid_model(p1[idx], p2[idx], -e))
        if m == 5:
            idx = to[i, 3:8]
            Eo = ora.run5point(p1[idx], p2[idx])
            Eg, nm = pose.solve_5pt(p1, p2, idx[None, :].astype(np.int32), ctx=ctx)
            print(" oracle models", len(Eo), "gpu models", int(nm[0]))
            for e in Eo:
                e = np.asarray(e).reshape(3, 3)
                d = min(min(np.abs(e - x).max(), np.abs(e + x).max()) for x in Eg[0, :nm[0]]) if nm[0] else None
                sv = np.linalg.svd(e, compute_uv=False)
                print("   oracle model: nearest gpu", d, "sv ratio", sv[0] / sv[1], "sv2/sv1", sv[2] / sv[1], "valid", ora.valid_model(p1[idx], p2[idx], e), ora.valid_model(p1[idx], p2[idx], -e))
        break
