"""Prints the kernel timeline of ONE call out of a `rocprofv3 --kernel-trace --output-format csv` capture: start offset, duration and the gap to
the previous kernel's end, for the kernels between two host-side gaps of more than `gap_us`.  usage: python tools/kernel_timeline.py DIR [call_index] [gap_us]"""
import csv, glob, sys
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
which = int(sys.argv[2]) if len(sys.argv) > 2 else -2
gap = float(sys.argv[3]) * 1e3 if len(sys.argv) > 3 else 100e3
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name']) for r in csv.DictReader(open(f)))
calls, cur = [], [rows[0]]
for r in rows[1:]:
    if r[0] - max(x[1] for x in cur) > gap:
        calls.append(cur); cur = [r]
    else:
        cur.append(r)
calls.append(cur)
c = calls[which]
t0 = c[0][0]; prev = t0
print(f"{len(calls)} calls; call {which}: {len(c)} kernels, span {(max(x[1] for x in c) - t0) / 1e3:.1f} us")
for s_, e_, n in c:
    n = n.replace('mlpl::(anonymous namespace)::', '').replace('void ', '')
    print(f"  +{(s_ - t0) / 1e3:8.1f} us  {(e_ - s_) / 1e3:7.1f} us  gap {(s_ - prev) / 1e3:6.1f}  {n[:70]}")
    prev = max(prev, e_)
