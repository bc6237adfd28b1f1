"""Where one image pair's ~22 us go: per-kernel durations and the gaps between them over the steady part of a `rocprofv3 --kernel-trace
--output-format csv` capture of tools/single_pair_opt_ab.py (back-to-back single-pair steps: expand -> ring kernel -> merge -> ratio_write).
usage: python tools/single_pair_timeline.py DIR"""
import csv, glob, sys
from collections import defaultdict
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), r['Kernel_Name'].replace('mlpl::(anonymous namespace)::', '').replace('void ', '').split('(')[0]) for r in csv.DictReader(open(f)))
def kind(n):
    return 'expand' if 'expand' in n else 'ring' if 'mfma_lds' in n else 'merge' if 'merge' in n else 'ratio_write' if 'ratio_write' in n else None
steps, cur = [], []
for r in rows:
    k = kind(r[2])
    if k is None:
        continue
    if k == 'expand' and cur:
        steps.append(cur); cur = []
    cur.append((k,) + r)
steps = [s for s in steps if [x[0] for x in s] == ['expand', 'ring', 'merge', 'ratio_write']]
steps = steps[len(steps) // 4:]          # the steady part
dur, gap, names = defaultdict(list), defaultdict(list), {}
period = []
for a, b in zip(steps, steps[1:]):
    if b[0][1] - a[0][1] < 60_000:       # consecutive steps of one timed loop (not across an option change)
        period.append(b[0][1] - a[0][1])
    prev_end = None
    for k, s_, e_, n in a:
        dur[k].append(e_ - s_); names[k] = n
        if prev_end is not None:
            gap[k].append(s_ - prev_end)
        prev_end = e_
    gap['expand(next step)'].append(b[0][1] - a[-1][2])
med = lambda v: sorted(v)[len(v) // 2] / 1e3
print(f"{len(steps)} steady single-pair steps; step period (start of expand to start of the next expand), median: {med(period):.2f} us")
tot = 0.0
for k in ('expand', 'ring', 'merge', 'ratio_write'):
    g = med(gap[k]) if gap[k] else 0.0
    print(f"  {k:12s} kernel {med(dur[k]):6.2f} us   idle before it {g:5.2f} us   {names[k][:60]}")
    tot += med(dur[k]) + g
print(f"  idle before the next step's expand {med(gap['expand(next step)']):5.2f} us;  kernels + gaps = {tot + med(gap['expand(next step)']):.2f} us")
