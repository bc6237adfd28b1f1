"""Per-kernel (name, grid) launch statistics from a rocprofv3 results .db (rocpd) or kernel_trace.csv.
    python tools/rocpd_kernels.py <results.db | kernel_trace.csv> [name-substring]"""
import collections
import csv
import sqlite3
import sys


def rows_of(path):
    if path.endswith(".db"):
        cur = sqlite3.connect(path).cursor()
        return [(n, g, s, e) for n, g, s, e in cur.execute("select name, grid_x, start, end from kernels order by start")]
    out = []
    for r in csv.DictReader(open(path)):
        out.append((r["Kernel_Name"], int(r["Grid_Size_X"]) if "Grid_Size_X" in r else int(r.get("Grid_Size", 0)), int(r["Start_Timestamp"]), int(r["End_Timestamp"])))
    return out


def main():
    rows = rows_of(sys.argv[1])
    sub = sys.argv[2] if len(sys.argv) > 2 else ""
    agg = collections.defaultdict(list)
    for n, g, s, e in rows:
        if sub in n:
            agg[(n.replace("mlpl::(anonymous namespace)::", "").replace("void ", "")[:56], g)].append((e - s) / 1e3)
    for k, v in sorted(agg.items(), key=lambda kv: -sum(kv[1])):
        print(f"{k[0]:58s} grid {k[1]:9d} calls {len(v):5d} avg {sum(v) / len(v):9.2f} us  min {min(v):9.2f}  total {sum(v) / 1e3:9.3f} ms")


if __name__ == "__main__":
    main()
