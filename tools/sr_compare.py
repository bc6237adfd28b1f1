import sys, os, pathlib
sys.path.insert(0,'/root/repo'); sys.path.insert(0,'/root/repo/tests')
import numpy as np, oracle_lib
import test_gpu_stereo_refine as T
ora = oracle_lib.load()
name = sys.argv[1]
want = T.run_oracle(ora, name)
got = T.run_gpu(name, pathlib.Path(os.environ.get('TMPDIR', '/tmp')), options="solver_polish=0")
for i,(g,w) in enumerate(zip(got,want)):
    st = dict(zip(["rc", "inl", "corrs", "pool", "est", "skip", "stable", "ml", "hist"], g["st"].tolist()))
    d = None if w["E"] is None else min(np.abs(g["E"].reshape(3,3)/np.linalg.norm(g["E"])-w["E"]/np.linalg.norm(w["E"])).max(), np.abs(g["E"].reshape(3,3)/np.linalg.norm(g["E"])+w["E"]/np.linalg.norm(w["E"])).max())
    print(i, st, {k:w[k] for k in st}, w["branch"], 'dE', d)

# hybrid: the CPU state machine with the GPU's ARRSAC as its estimator -- does the difference come from the estimator's output?
if len(sys.argv) > 2 and sys.argv[2] == "hybrid":
    from stereo_refine_oracle import StereoRefineOracle
    from matchinglib_poselib_amd import pose
    import matchinglib_poselib_amd as mpa
    ctx = mpa.Context(0); ctx.set_option("solver_polish", 0)
    cfg, method, dist, frames = T.sequence(name)
    sr = StereoRefineOracle(ora, cfg, T.K, T.K, np.zeros(8), np.zeros(8), 777)
    st = np.array([0xFFFFFFFF, 0xFFFFFFFF], np.uint64)
    real = ora.arrsac_essential
    st_o = np.array([0xFFFFFFFF, 0xFFFFFFFF], np.uint64)
    log = []
    def fake(a, b, th, refine=True, rng_state=None):
        g = pose.arrsac_essential(a, b, th, refine=refine, rng_state=st, ctx=ctx)
        o = real(a, b, th, refine=refine, rng_state=st_o)
        d = min(np.abs(g["E"] - o["E"]).max(), np.abs(g["E"] + o["E"]).max())
        log.append((len(a), g["n_inliers"], o["n_inliers"], d, np.linalg.norm(o["E"]), (g["mask"] != o["mask"]).sum(), float((g["E"] * o["E"]).sum())))
        return g
    ora.arrsac_essential = fake
    for i, (kp1, kp2, dd) in enumerate(frames):
        rc = sr.add(kp1, kp2, dd)
        print("hybrid", i, rc, sr.nr_inliers, sr.nr_corrs, len(sr.pool), log[-1] if log else None)
