"""Sweep counts of the root iteration (roots_kernel_t) over many 5-point samples: mean, maximum, histogram, and the slowest sample.
    python tools/dk_stats.py [n_samples]"""
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import matchinglib_poselib_amd as mpa  # noqa: E402
from matchinglib_poselib_amd import pose, synth  # noqa: E402

ns = int(sys.argv[1]) if len(sys.argv) > 1 else 200000
ctx = mpa.Context(0)
edges = ["<=8", "<=12", "<=16", "<=24", "<=32", "<=64", "<=128", "<=256", "<400", "=400"]
for scene_seed, n, frac in ((20260103, 5000, 0.5), (7, 5000, 0.2), (11, 2000, 0.8)):
    p1, p2, R, t, mask, th = synth.pose_scene(n, inlier_frac=frac, seed=scene_seed)
    rng = np.random.default_rng(scene_seed)
    samples = np.stack([rng.choice(n, 5, replace=False) for _ in range(ns)]).astype(np.int32)
    ctx.lib.mlpl_debug_dk_stats(ctx.handle, 1, None)
    E, nm = pose.solve_5pt(p1, p2, samples, ctx=ctx)
    st = (C.c_int * 16)()
    ctx.lib.mlpl_debug_dk_stats(ctx.handle, 0, st)
    print(f"scene {scene_seed}: sweeps mean {st[0] / max(st[1], 1):.2f} over {st[1]} solves, max {st[2]} (sample {st[4]}: {samples[st[4]].tolist()}), "
          f"models {nm.sum()}")
    print("   histogram:", {e: st[5 + i] for i, e in enumerate(edges)})
