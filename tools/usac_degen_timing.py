"""Wall time of mlpl_usac_essential with and without the degeneracy handling on the scenes of tools/usac_degen_cases.py (GPU box).
    python tools/usac_degen_timing.py"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import usac_degen_cases
    from matchinglib_poselib_amd import pose, synth

    seen = set()
    for name, n, frac, seed, kw in usac_degen_cases.SCENES + [("general", 5000, 0.5, 20260103, {}), ("rotation", 5000, 0.5, 41, dict(t_len=0.0))]:
        if (name, n) in seen:
            continue
        seen.add((name, n))
        p1, p2, R, t, truth, th = synth.pose_scene(n, frac, seed=seed, **kw)
        row = []
        for chk in (0, 1, 3):
            ts, launches, cands = [], 0, 0
            for rep in range(7):
                t0 = time.perf_counter()
                d = pose.usac_essential(p1, p2, th, 100 + rep, check_degeneracy=chk, event_cap=400000 if rep == 0 else 0)
                ts.append(time.perf_counter() - t0)
                if rep == 0:
                    launches = int(d["stats"][5])
                    cands = int((d["events"][:, 0] == 10).sum())
            row.append(f"check {chk}: {1e3 * np.median(ts[1:]):7.2f} ms (first run: {launches:3d} test launches, {cands:5d} upgrade candidates)")
        print(f"{name:15s} n {n:5d}: " + " | ".join(row), flush=True)


if __name__ == "__main__":
    main()
