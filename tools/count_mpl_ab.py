"""A/B of the packed-fp32 counting kernel with one and two models per lane (option ransac_count_mpl): C3 call and the C5 batch, same process,
per-kernel times by the library's HIP events.  Also checks that the results are identical."""
import ctypes as C, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import batch, pose, synth

ctx = mpa.Context(0)
dev = torch.device("cuda:0")
def prof(kid):
    ms, cnt = C.c_double(0), C.c_int(0)
    ctx.lib.mlpl_profile_read(ctx.handle, kid, C.byref(ms), C.byref(cnt))
    return ms.value, cnt.value
n, iters = 5000, 20000
p1, p2, R, t, mask, th = synth.pose_scene(n, seed=20260103)
d1, d2 = torch.from_numpy(p1).to(dev), torch.from_numpy(p2).to(dev)
dm = torch.empty(n, dtype=torch.uint8, device=dev)
total = 512
sps = [synth.stereo_pair(8192, seed=20260200 + i, unmatched_frac=0.30 + 0.02 * (i % 8)) for i in range(8)]
K = sps[0]["K"]
stk = [torch.from_numpy(np.stack([sps[i % 8][k] for i in range(total)])).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")]
seeds = [100 + i for i in range(total)]
ref = {}
for rnd in range(2):
    for mpl, defer in ((2, 0), (2, 1), (1, 0)):
        ctx.set_option("ransac_count_mpl", mpl)
        ctx.set_option("ransac_count_defer", defer)
        call = lambda: pose.ransac_essential_device(d1, d2, th, confidence=1.0, max_iters=iters, refit=False, seed=12345, ctx=ctx, mask_out=dm)
        r = call(); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(20): r = call()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 20
        ctx.lib.mlpl_profile_reset(ctx.handle); ctx.lib.mlpl_profile_enable(ctx.handle, 1)
        for _ in range(5): call()
        torch.cuda.synchronize(); ctx.lib.mlpl_profile_enable(ctx.handle, 0)
        sc, scn = prof(3)
        key = (r["iters"], r["n_inliers"], r["E"].tobytes(), dm.cpu().numpy().tobytes())
        assert ref.setdefault("c3", key) == key
        rec = batch.process_pairs_batched(ctx, *stk, K, K, seeds); torch.cuda.synchronize()
        ts = []
        for _ in range(5):
            t0 = time.perf_counter(); rec = batch.process_pairs_batched(ctx, *stk, K, K, seeds); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
        ctx.lib.mlpl_profile_reset(ctx.handle); ctx.lib.mlpl_profile_enable(ctx.handle, 1)
        batch.process_pairs_batched(ctx, *stk, K, K, seeds); torch.cuda.synchronize(); ctx.lib.mlpl_profile_enable(ctx.handle, 0)
        s5, s5n = prof(3)
        assert ref.setdefault("c5", rec.tobytes()) == rec.tobytes()
        print(f"round {rnd} mpl {mpl} defer {defer}: C3 call {dt * 1e3:.3f} ms, counting pass {sc / 5:.3f} ms per call; C5 512 pairs {min(ts) * 1e3:.2f} ms (median {np.median(ts) * 1e3:.2f}), counting {s5:.3f} ms per step ({s5n} launches)", flush=True)
