"""One-off soak: the division-free inlier predicate (count-only kernels, both shapes) against the dividing kernels' counts on random
models, points and thresholds, including thresholds placed on actual error values.  Usage: python tools/stress_inlier_count.py [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import matchinglib_poselib_amd as mpa  # noqa: E402
from matchinglib_poselib_amd import pose, synth  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
ctx = mpa.Context(0)
rng = np.random.default_rng(777)
t0 = time.time()
cases = 0
evals = 0
while time.time() - t0 < budget:
    n = int(rng.integers(6, 6000))
    p1, p2, R, t, mask, th = synth.pose_scene(n, seed=int(rng.integers(1, 1 << 30)))
    nm = int(rng.integers(1, 200))
    E = rng.normal(size=(nm, 3, 3))
    if rng.random() < 0.5:  # some models near the true one: many errors near the threshold scale
        tx = np.array([[0, -t[2, 0], t[1, 0]], [t[2, 0], 0, -t[0, 0]], [-t[1, 0], t[0, 0], 0]]) if t.ndim == 2 else np.array(
            [[0, -t[2], t[1]], [t[2], 0, -t[0]], [-t[1], t[0], 0]])
        E0 = tx @ R
        E[: nm // 2] = E0 / np.linalg.norm(E0) + 1e-4 * rng.normal(size=(nm // 2, 3, 3))
    t2 = float(th * th * 10 ** rng.uniform(-2, 2))
    good, esum = pose.score_models(p1, p2, E, np.sqrt(t2), ctx=ctx)
    t2 = float(np.sqrt(t2)) ** 2        # score_models squares its argument: use exactly that value
    ctx.set_option("ransac_count_defer", int(rng.integers(0, 2)))   # the deferred queue of the counting kernel and its inline form
    ctx.set_option("ransac_count_threads", int(rng.choice([256, 512])))   # 4-wave (default since round 6) and 8-wave workgroups
    for shape in (1, 2):
        c = pose.count_models(p1, p2, E, t2, shape=shape, ctx=ctx)
        if not np.array_equal(c, good):
            print("MISMATCH", n, nm, t2, shape, np.nonzero(c != good)[0][:5], flush=True)
            sys.exit(1)
    cases += 1
    evals += n * nm
print(f"{cases} random cases ({evals / 1e6:.0f} M error evaluations) in {time.time() - t0:.0f} s: division-free counts == dividing kernels")
