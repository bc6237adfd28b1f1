"""The device side of ONE lane (stream) of a batched USAC call with ConfigUSAC's default refinement, out of a `rocprofv3 --kernel-trace
--output-format csv` capture of `tools/c5_opt_ab.py usac_lo5_fused_fit 1 usac_default_refine 2`: per kernel kind the launches, the mean
duration and the mean idle time in front of it on that stream, the share of the lane's span spent in kernels, and 36 consecutive launches.
usage: python tools/usac_lane_timeline.py DIR"""
import csv, glob, sys
from collections import defaultdict, Counter
f = glob.glob(sys.argv[1] + '/*/*kernel_trace.csv')[0]
R = list(csv.DictReader(open(f)))
sk = 'Stream_Id' if 'Stream_Id' in R[0] else 'Queue_Id'
def short(n):
    n = n.replace('mlpl::(anonymous namespace)::', '').replace('void ', '')
    if 'hub_batch_kernel' in n or 'hub_single_kernel' in n:
        a = n.find('Args')
        b = n.rfind('::', 0, a) if a > 0 else -1
        return n[b + 2:a + 4] if a > 0 else n[:40]
    return n.split('(')[0].split('<')[0]
rows = sorted((int(r['Start_Timestamp']), int(r['End_Timestamp']), short(r['Kernel_Name']), r[sk]) for r in R)
# the last call of the capture: kernels behind the last host gap of more than 3 ms
cut = 0
for i in range(1, len(rows)):
    if rows[i][0] - rows[i - 1][1] > 3_000_000:
        cut = i
call = rows[cut:]
streams = Counter(r[3] for r in call if 'Usac5' in r[2])
lane = streams.most_common(1)[0][0]
L = [r for r in call if r[3] == lane]
span = L[-1][1] - L[0][0]
busy = sum(e - s for s, e, _, _ in L)
print(f"last call: {len(call)} kernels on {len(set(r[3] for r in call))} streams, span {(call[-1][1] - call[0][0]) / 1e6:.2f} ms; lane {sk} {lane}: {len(L)} launches, span {span / 1e6:.2f} ms, "
      f"in kernels {busy / 1e6:.2f} ms = {busy / span:.2f} of its span")
dur, gap = defaultdict(list), defaultdict(list)
prev = None
for s, e, n, _ in L:
    dur[n].append(e - s)
    if prev is not None:
        gap[n].append(max(0, s - prev))
    prev = e
for n in sorted(dur, key=lambda k: -sum(dur[k])):
    print(f"  {n:24s} {len(dur[n]):5d} launches  mean {sum(dur[n]) / len(dur[n]) / 1e3:7.1f} us  sum {sum(dur[n]) / 1e6:6.2f} ms   idle in front of it (this stream), mean {sum(gap[n]) / max(1, len(gap[n])) / 1e3:6.1f} us")
i0 = next(i for i, r in enumerate(L) if 'Usac5' in r[2]) + 40
prev = L[i0][0]
print("  36 consecutive launches of the lane (offset us, duration us, idle before us, kernel):")
for s, e, n, _ in L[i0:i0 + 36]:
    print(f"    +{(s - L[i0][0]) / 1e3:8.1f}  {(e - s) / 1e3:7.1f}  {(s - prev) / 1e3:6.1f}  {n}")
    prev = e
