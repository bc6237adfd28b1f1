"""Device USAC vs the CPU oracle, decision by decision, on the scenes of tools/usac_compare.py (GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT); sys.path.insert(0, os.path.join(ROOT, "tests")); sys.path.insert(0, os.path.join(ROOT, "tools"))
import numpy as np
import oracle_lib, usac_compare as uc
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import pose

ora = oracle_lib.load()
ctx = mpa.Context(0)
np.set_printoptions(linewidth=220, precision=6, suppress=True)
bad = 0
quick = len(sys.argv) > 1 and sys.argv[1] == "quick"
for sc in uc.scenes():
    if quick and len(sc["p1"]) > 5000: continue
    for prosac in (False, True):
        for seed in (12345, 7):
            for stepwise in (0, 1):
                ctx.set_option("usac_lo_stepwise", stepwise)
                si = sc["order"] if prosac else None
                o = ora.usac_essential(sc["p1"], sc["p2"], sc["th"], seed, sorted_idx=si, event_cap=200000, max_hyp=20000)
                t0 = time.perf_counter()
                d = pose.usac_essential(sc["p1"], sc["p2"], sc["th"], seed, sorted_idx=si, event_cap=200000, max_hyp=20000, ctx=ctx)
                dt = time.perf_counter() - t0
                first, diffs = uc.compare(o["events"], d["events"])
                same = np.array_equal(o["final"][:8], d["final"][:8]) and np.array_equal(o["flags"], d["flags"])
                print(f"{sc['name']:14s} prosac {int(prosac)} seed {seed:5d} step {stepwise}: hyps {int(o['final'][1]):5d} inl {int(o['final'][5]):5d} LO {int(o['final'][7])} "
                      f"first_diff {first} final_equal {same} {diffs} dev {dt*1e3:.2f} ms stats {d['stats'][:5]}", flush=True)
                if first is not None or not same:
                    bad += 1
                    if first is not None:
                        print(o["events"][max(0, first - 1):first + 2, :14]); print(d["events"][max(0, first - 1):first + 2, :14])
print("differences:", bad)
