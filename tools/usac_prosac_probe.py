"""Wall time of a USAC call on the C3 scene with uniform sampling, with a good PROSAC ordering and with an uninformative one (GPU box).
Diagnostic: the run that located the host-side costs described in DESIGN 4.7."""
from __future__ import annotations

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests", "golden"))


def main():
    import make_golden
    from matchinglib_poselib_amd import pose

    p1, p2, th, truth, order = make_golden.usac_scene(5000, 0.5, 20260103)
    for name, si in (("uniform", None), ("prosac", order), ("prosac, identity order", np.arange(5000, dtype=np.uint32))):
        pose.usac_essential(p1, p2, th, 1, sorted_idx=si)
        ts = []
        for c in range(30):
            t0 = time.perf_counter()
            d = pose.usac_essential(p1, p2, th, 100 + c, sorted_idx=si)
            ts.append(time.perf_counter() - t0)
        print(f"{name}: median {1e3 * np.median(ts):.3f} ms, min {1e3 * min(ts):.3f}; hypotheses {int(d['final'][1])}, stats {d['stats'][:6]}", flush=True)


if __name__ == "__main__":
    main()
