"""The tail of the two-lane C5 mode (one step in ~20 takes 14-15 ms instead of 9.3): N steps of batch.BatchLanes in one process,
every step's wall time, each lane's span and -- from the library's hop timeline (mlpl_debug_hop_trace) -- when each host hop of each lane's
call returned; the slow steps are printed in full beside a typical one, and the attribution (which lane, which hop) is summarised.
usage: python tools/lanes_tail_probe.py [steps=120] [pairs=512]"""
import ctypes as C, json, os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np, torch
import matchinglib_poselib_amd as mpa
from matchinglib_poselib_amd import batch, synth

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 120
total = int(sys.argv[2]) if len(sys.argv) > 2 else 512
native = (sys.argv[3] != "python") if len(sys.argv) > 3 else True   # lanes inside the library (default) or as Python threads
n = 8192
dev = torch.device("cuda:0")
ctx = mpa.Context(0)
sps = [synth.stereo_pair(n, seed=20260200 + i, unmatched_frac=0.30 + 0.02 * (i % 8)) for i in range(8)]
K = sps[0]["K"]
st = {k: torch.from_numpy(np.stack([sps[i % 8][k] for i in range(total)])).to(dev) for k in ("desc1", "desc2", "kp1", "kp2")}
seeds = [100 + i for i in range(total)]
d_matches = torch.zeros((total, n, 4), dtype=torch.int32, device=dev)
lanes = batch.BatchLanes(0, lanes=2, first_ctx=ctx)


def hops(c):
    us, codes, g = np.zeros(48, np.float32), np.zeros(48, np.int32), C.c_longlong(0)
    m = c.lib.mlpl_debug_hop_trace(c.handle, us.ctypes.data, codes.ctypes.data, 48, C.byref(g))
    return [(int(codes[i]), round(float(us[i]) / 1e3, 2)) for i in range(m)], g.value


recs = []
for k in range(steps + 3):
    t0 = time.perf_counter()
    rec = lanes.process(st["desc1"], st["desc2"], st["kp1"], st["kp2"], K, K, seeds, matches_out=d_matches, native=native)
    t1 = time.perf_counter()
    torch.cuda.synchronize()
    t2 = time.perf_counter()
    if k < 3:
        continue
    recs.append({"step": k - 3, "ms": (t2 - t0) * 1e3, "process_ms": (t1 - t0) * 1e3, "lane_spans_ms": [[round((a - t0) * 1e3, 2), round((b - t0) * 1e3, 2)] for a, b in lanes.last_lane_span],
                 "hops": [hops(c) for c in lanes.ctxs]})
ms = np.array([r["ms"] for r in recs])
med = float(np.median(ms))
slow = [r for r in recs if r["ms"] > 1.25 * med]
print(json.dumps({"lanes": "native" if native else "python", "steps": steps, "median_ms": med, "p90_ms": float(np.percentile(ms, 90)), "max_ms": float(ms.max()), "slow_steps": len(slow),
                  "all_ms": [round(float(x), 2) for x in ms]}))
typ = min(recs, key=lambda r: abs(r["ms"] - med))
print("typical:", json.dumps(typ))
for r in slow[:6]:
    print("slow:   ", json.dumps(r))
    # attribution: the largest gap between consecutive hops of either lane compared with the typical step's gap at the same position
    for w in range(2):
        h, ht = r["hops"][w][0], typ["hops"][w][0]
        gaps = [(h[i][0], round(h[i][1] - (h[i - 1][1] if i else 0.0), 2), round((ht[i][1] - (ht[i - 1][1] if i else 0.0)) if i < len(ht) else float("nan"), 2)) for i in range(len(h))]
        worst = max(gaps, key=lambda g: g[1] - (g[2] if g[2] == g[2] else 0))
        print(f"         lane {w}: largest excess at hop code {worst[0]}: {worst[1]} ms (typical {worst[2]} ms); gaps (code, ms, typical ms): {gaps}")
lanes.close()
