"""Timeline of the kernels in a rocprofv3 results .db: start (ms from the first kernel), gap before, duration, name, grid.
    python tools/rocpd_timeline.py <results.db> [from_ms] [to_ms]"""
import sqlite3
import sys

cur = sqlite3.connect(sys.argv[1]).cursor()
rows = list(cur.execute("select name, grid_x, start, end from kernels order by start"))
lo = float(sys.argv[2]) if len(sys.argv) > 2 else 0.0
hi = float(sys.argv[3]) if len(sys.argv) > 3 else 1e30
t0 = rows[0][2]
prev_end = None
busy = idle = 0.0
for n, g, s, e in rows:
    ts = (s - t0) / 1e6
    if lo <= ts <= hi:
        gap = (s - prev_end) / 1e3 if prev_end is not None else 0.0
        nm = n.replace("mlpl::(anonymous namespace)::", "").replace("void ", "")[:44]
        print(f"{ts:10.3f} ms  gap {gap:8.1f} us  dur {(e - s) / 1e3:8.1f} us  {nm:44s} grid {g}")
        busy += (e - s) / 1e3
        idle += max(gap, 0.0)
    prev_end = e if prev_end is None else max(prev_end, e)
print(f"busy {busy / 1e3:.3f} ms, gaps {idle / 1e3:.3f} ms")
