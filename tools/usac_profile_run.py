"""A fixed USAC workload for rocprofv3 (kernel trace): `calls` estimations on the C3 scene (general motion, 5000 correspondences) and on a
pure-rotation scene of the same size, degeneracy handling on.  Prints wall time per call and the library's own counters.
    rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/r03/usac_trace -- python tools/usac_profile_run.py"""
from __future__ import annotations

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    from matchinglib_poselib_amd import pose, synth

    calls = int(sys.argv[1]) if len(sys.argv) > 1 else 20
    for name, kw, uk in (("general", {}, dict(check_degeneracy=3)), ("rotation", dict(t_len=0.0), dict(check_degeneracy=3)),
                         ("general, ConfigUSAC's defaults (POSE_STEWENIUS, REF_STEWENIUS_WEIGHTS, DEGEN_USAC_INTERNAL)", {},
                          dict(check_degeneracy=1, refine=5, estimator=2, sprt_ms=6.0, sprt_tm=2736.0))):
        p1, p2, R, t, truth, th = synth.pose_scene(5000, 0.5, seed=20260103, **kw)
        pose.usac_essential(p1, p2, th, 1, **uk)   # warm-up: workspaces
        ts, st = [], np.zeros(8)
        for c in range(calls):
            t0 = time.perf_counter()
            d = pose.usac_essential(p1, p2, th, 100 + c, **uk)
            ts.append(time.perf_counter() - t0)
            st += d["stats"]
        st /= calls
        print(f"{name}: {1e3 * np.median(ts):.2f} ms per call (median of {calls}); per call: {st[0]:.1f} solver batches, {st[1]:.0f} samples solved, "
              f"{st[2]:.0f} consumed, {st[3]:.1f} LO launches ({st[4]:.1f} resumes), {st[5]:.1f} degeneracy-test launches, {st[7]:.0f} Jacobi sweeps in LO fits", flush=True)


if __name__ == "__main__":
    main()
