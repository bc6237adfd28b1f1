import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import pose, synth
ctx = mpa.Context(0)
p1, p2, R, t, mask, th = synth.pose_scene(5000, seed=20260103)
ctx.lib.mlpl_debug_dk_stats(ctx.handle, 1, None)
r = pose.ransac_essential(p1, p2, th, confidence=1.0, max_iters=20000, refit=False, seed=12345, ctx=ctx)
st = (C.c_int * 3)()
ctx.lib.mlpl_debug_dk_stats(ctx.handle, 0, st)
print("DK sweeps: mean %.1f over %d solves, max %d" % (st[0] / max(st[1], 1), st[1], st[2]))
