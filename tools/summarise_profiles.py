"""Turns gpurun_out/<round>/ (tools/collect_profiles.sh) into the committed summaries under profiles/.
Usage: python tools/summarise_profiles.py r01"""
import collections
import csv
import glob
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
R = sys.argv[1] if len(sys.argv) > 1 else "r01"
SRC = os.path.join(ROOT, "gpurun_out", R)
DST = os.environ.get("MLPL_PROFILE_DST") or os.path.join(ROOT, "profiles")  # collect_profiles.sh summarises on the GPU box into gpurun_out/<round>_summary
os.makedirs(DST, exist_ok=True)


def short(name):
    return name.replace("mlpl::(anonymous namespace)::", "").replace("void ", "")[:70]


def one(pattern):
    f = sorted(glob.glob(os.path.join(SRC, pattern)), key=os.path.getmtime)  # gpurun merges runs into one tree: newest wins
    return f[-1] if f else None


# kernel stats of the default bench run
ks = one("trace/*/*kernel_stats.csv")
if ks:
    shutil.copy(ks, os.path.join(DST, f"{R}_bench_default_kernel_stats.csv"))
for name in ("bench_plain.json", "bench_under_rocprof.json"):
    p = os.path.join(SRC, name)
    if os.path.exists(p):
        text = open(p).read().splitlines()
        line = [x for x in text if x.startswith("{")]
        if line:   # the final line = what the driver parses
            json.dump(json.loads(line[-1]), open(os.path.join(DST, f"{R}_{name}"), "w"), indent=1)
        for tag in ("bench_detail ", "bench_secondary "):   # round 6 on: the two lines in front of it
            more = [x for x in text if x.startswith(tag)]
            if more:
                json.dump(json.loads(more[-1][len(tag):]), open(os.path.join(DST, f"{R}_{name}".replace(".json", "_" + tag.strip()[6:] + ".json")), "w"), indent=1)

summary = {}
for tag in ("pmc_fetch", "pmc_write", "pmc_sq", "pmc_grbm"):
    f = one(f"{tag}/*/*counter_collection.csv")
    if not f:
        continue
    agg = collections.defaultdict(lambda: collections.defaultdict(list))
    dur = collections.defaultdict(list)
    rows = list(csv.DictReader(open(f)))
    # The Hamming kernel is launched with ONE grid for different work: 64 pairs x 1 train split (the headline) and 8 pairs x 8 splits
    # (the 8-pair continuity figure) are both 512 workgroups.  They differ eightfold in work: launches shorter than half the longest of
    # their (kernel, grid) are kept apart as "(short)".
    longest = collections.defaultdict(int)
    for r in rows:
        k = short(r["Kernel_Name"])
        if k.startswith("knn_hamming_mfma_lds_kernel<4, 0"):
            g = k + " @grid " + r.get("Grid_Size", r.get("Grid_Size_X", "?"))
            longest[g] = max(longest[g], int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for r in rows:
        k = short(r["Kernel_Name"])
        if k.startswith(("knn_hamming_mfma_lds_kernel<4, 0", "solve5pt", "roots_kernel_t<true>")):  # launched at several batch sizes (headline, 8-pair continuity, C5 extras): keep apart
            k += " @grid " + r.get("Grid_Size", r.get("Grid_Size_X", "?"))
            if k in longest and 2 * (int(r["End_Timestamp"]) - int(r["Start_Timestamp"])) < longest[k]:
                k += " (short)"
        agg[k][r["Counter_Name"]].append(float(r["Counter_Value"]))
        dur[k].append(int(r["End_Timestamp"]) - int(r["Start_Timestamp"]))
    for k, v in agg.items():
        e = summary.setdefault(k, {})
        for c, vals in v.items():
            e[c] = {"mean": sum(vals) / len(vals), "dispatches": len(vals)}
        e.setdefault("duration_ns_under_pmc", {})[tag] = sum(dur[k]) / len(dur[k])
json.dump(summary, open(os.path.join(DST, f"{R}_pmc_summary.json"), "w"), indent=1)

# roofline.traffic of bench.py: HBM-side bytes per launch of the dominant kernel (FETCH_SIZE doubled on gfx950 for 16-B/lane streams,
# MI355X_MICROARCH.md HBM section)
PAIRS = 64
bp = os.path.join(DST, f"{R}_bench_plain.json")
if os.path.exists(bp):
    PAIRS = int(json.load(open(bp)).get("config", {}).get("pairs_per_gpu", PAIRS))
HEADLINE_GRID = str(4096 * PAIRS)  # 8 workgroups of 512 threads per image pair: one per 256 queries, ONE train split (round 5; with two splits 8192 * PAIRS)
traffic = {"round": R, "pairs_per_launch": PAIRS,
           "how": "rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes of `bench.py --steps 20 --warmup 2 "
                  f"--no-cpu-baseline` (dispatches of {PAIRS} image pairs per launch, grid {HEADLINE_GRID}); (2*FETCH_SIZE + WRITE_SIZE)*1024 B, FETCH_SIZE doubled per "
                  "MI355X_MICROARCH.md (gfx950 tallies 128-B requests at 64 B for 16-B/lane streams)"}
for k, e in summary.items():
    if "FETCH_SIZE" in e and "WRITE_SIZE" in e:
        b = (2 * e["FETCH_SIZE"]["mean"] + e["WRITE_SIZE"]["mean"]) * 1024
        # the headline launch (8 pairs per launch): round 1 knn_hamming_mfma_kernel<4, 4>, since round 2 the LDS-ring kernel <4, 0>
        # (<1, 0> / <4, 1> are the single-pair extras launches)
        if k.startswith("knn_hamming_mfma_lds_kernel<4, 0") and k.endswith("@grid " + HEADLINE_GRID):
            traffic["knn_hamming_mfma_bytes_per_launch"] = b
            traffic["mfma_fetch_KiB_raw"] = e["FETCH_SIZE"]["mean"]
            traffic["mfma_write_KiB"] = e["WRITE_SIZE"]["mean"]
        if k.startswith("knn_hamming_partial_kernel"):
            traffic["knn_hamming_partial_bytes_per_launch"] = b
        if k.startswith("hamming_expand_kernel"):
            traffic["hamming_expand_bytes_per_launch"] = b
# matrix-core busy fraction of the headline launch: SQ_VALU_MFMA_BUSY_CYCLES over the SIMD-cycles of the launch (GRBM_GUI_ACTIVE sums the
# eight XCDs' clocks; 1024 SIMDs), and the kernel's duration under `--kernel-trace` for the same grid
for k, e in summary.items():
    if k.startswith("knn_hamming_mfma_lds_kernel<4, 0") and k.endswith("@grid " + HEADLINE_GRID):
        if "SQ_VALU_MFMA_BUSY_CYCLES" in e and "GRBM_GUI_ACTIVE" in e:
            cyc = e["GRBM_GUI_ACTIVE"]["mean"] / 8.0
            traffic["mfma_busy_frac"] = e["SQ_VALU_MFMA_BUSY_CYCLES"]["mean"] / (1024.0 * cyc)
            traffic["mfma_busy_how"] = "SQ_VALU_MFMA_BUSY_CYCLES / (1024 SIMDs x GRBM_GUI_ACTIVE / 8 XCDs), separate --pmc passes"
            traffic["shader_clock_GHz_under_pmc"] = cyc / e["duration_ns_under_pmc"]["pmc_grbm"]
kt = one("trace/*/*kernel_trace.csv")
if kt:
    d = [int(r["End_Timestamp"]) - int(r["Start_Timestamp"]) for r in csv.DictReader(open(kt))
         if r["Kernel_Name"].find("knn_hamming_mfma_lds_kernel<4, 0") >= 0 and r.get("Grid_Size_X", r.get("Grid_Size", "")) == HEADLINE_GRID]
    d = [x for x in d if 2 * x >= max(d)]  # (without the 8-pair launches of the same grid, see above)
    if d:
        traffic["kernel_us_rocprof_trace"] = sum(d) / len(d) / 1e3
        traffic["kernel_launches_rocprof_trace"] = len(d)
old = os.path.join(DST, "pmc_traffic.json")
prev_file = os.path.join(ROOT, "profiles", "pmc_traffic.json")
if os.path.exists(prev_file):
    prev = json.load(open(prev_file))
    for k in ("knn_hamming_partial_bytes_per_launch",):
        if k not in traffic and k in prev:
            traffic[k] = prev[k]
            traffic[k + "_note"] = "VALU kernel (hamming_variant 0), measured earlier in round 1 before the matrix-core kernel became the default"
json.dump(traffic, open(old, "w"), indent=1)
print(json.dumps(traffic, indent=1))
for k, e in summary.items():
    print(k, {c: round(v["mean"], 1) for c, v in e.items() if isinstance(v, dict) and "mean" in v})

# round 4 on: the C5 batch, the sequential estimators and C5 with USAC under the kernel trace
for sub, name in (("c5_trace", "c5_kernel_stats.csv"), ("usac_trace", "usac_kernel_stats.csv"), ("c5_usac_trace", "c5_usac_kernel_stats.csv")):
    f = one(f"{sub}/*/*kernel_stats.csv")
    if f:
        shutil.copy(f, os.path.join(DST, f"{R}_{name}"))
p = os.path.join(SRC, "usac_profile_run.txt")
if os.path.exists(p):
    open(os.path.join(DST, f"{R}_usac_profile_run.txt"), "w").write("".join(l for l in open(p) if not l.startswith(("W2", "E2", "/opt"))))
