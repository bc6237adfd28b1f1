"""Kernel timeline of ARRSAC calls on the C3 scene (or `n inlier_fraction seed`).  Run under
`rocprofv3 --kernel-trace --output-format csv -d <dir> -- python tools/arrsac_timeline.py [n frac seed]`, then summarise with
`python tools/arrsac_timeline.py summarise <dir>`."""
import glob, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
if len(sys.argv) > 2 and sys.argv[1] == "summarise":
    import csv
    f = glob.glob(os.path.join(sys.argv[2], "**", "*kernel_trace.csv"), recursive=True)[0]
    rows = sorted(csv.DictReader(open(f)), key=lambda r: int(r["Start_Timestamp"]))
    names = [r["Kernel_Name"] for r in rows]
    last = max(i for i, nm in enumerate(names) if "pack_points" in nm)      # the last call
    t0 = int(rows[last]["Start_Timestamp"])
    for r in rows[last:]:
        nm = r["Kernel_Name"].split("(")[0].replace("void mlpl::(anonymous namespace)::", "")
        print(f"{(int(r['Start_Timestamp']) - t0) / 1e3:9.1f} us  +{(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3:7.1f}  {nm}  grid {r['Grid_Size_X']}")
    sys.exit(0)
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import pose, synth
ctx = mpa.Context(0)
if os.environ.get("POLISH") is not None:
    ctx.set_option("solver_polish", int(os.environ["POLISH"]))
_n, _f, _s = (int(sys.argv[1]), float(sys.argv[2]), int(sys.argv[3])) if len(sys.argv) > 3 else (5000, 0.5, 20260103)
p1, p2, R, t, truth, th = synth.pose_scene(_n, _f, seed=_s)
for _ in range(5):
    g = pose.arrsac_essential(p1, p2, th, refine=True, rng_state=np.array(pose.ARRSAC_RNG_FRESH, np.uint64), ctx=ctx)
t0 = time.perf_counter()
g = pose.arrsac_essential(p1, p2, th, refine=True, rng_state=np.array(pose.ARRSAC_RNG_FRESH, np.uint64), ctx=ctx)
print("ms", (time.perf_counter() - t0) * 1e3, g["stats"].tolist())
