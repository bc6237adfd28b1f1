"""Compares the event traces of two USAC runs (reference-built usac_ref, the CPU oracle, the device path).
    python tools/usac_compare.py            # oracle vs usac_ref --solver-oracle on a set of scenes (build container only)
Decision columns must be identical; float columns are reported as maximum differences."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

DEC = {1: [1, 2, 3, 4, 5, 6, 7], 2: [1, 2, 3, 4, 5, 6], 3: [1, 2, 3, 4], 4: [1, 2, 3], 5: [1, 2], 6: [1, 2],
       7: [1, 2, 3, 4, 5, 6, 7], 8: [1, 2, 3, 4, 5], 9: [1, 2, 3, 4], 10: [1, 2, 3]}


def compare(a, b):
    """Returns (index of the first differing decision or None, dict of max float differences)."""
    m = min(len(a), len(b))
    for i in range(m):
        ty = int(a[i, 0])
        if int(b[i, 0]) != ty or not np.array_equal(a[i, DEC[ty]], b[i, DEC[ty]]):
            return i, {}
    if len(a) != len(b):
        return m, {}
    d = {}
    sa, sb = a[a[:, 0] == 2], b[b[:, 0] == 2]
    d["sprt"] = float(np.abs(sa[:, 7:11] - sb[:, 7:11]).max()) if len(sa) else 0.0
    ca, cb = a[a[:, 0] == 10], b[b[:, 0] == 10]
    for br, width, name in ((1, 3, "t10"), (2, 9, "E10")):  # upgrade candidates: translations (unit) / essential matrices
        va, vb = ca[ca[:, 2] == br][:, 4:4 + width], cb[cb[:, 2] == br][:, 4:4 + width]
        keep = (np.linalg.norm(va, axis=1) > 0) & np.isfinite(va).all(1) & np.isfinite(vb).all(1)
        va, vb = va[keep], vb[keep]
        if len(va):
            dd = np.abs(va - vb).max(1)
            d[name] = float(dd.max())
            d[name + "_q98"] = float(np.quantile(dd, 0.98))
    for ty, lo in ((3, 5), (5, 3)):
        ea, eb = a[a[:, 0] == ty][:, lo:lo + 9], b[b[:, 0] == ty][:, lo:lo + 9]
        keep = np.linalg.norm(ea, axis=1) > 0
        ea, eb = ea[keep], eb[keep]
        if len(ea):
            ea = ea / np.linalg.norm(ea, axis=1, keepdims=True)
            eb = eb / np.linalg.norm(eb, axis=1, keepdims=True)
            dd = np.minimum(np.abs(ea - eb).max(1), np.abs(ea + eb).max(1))
            d["E%d" % ty] = float(dd.max())
            d["E%d_q98" % ty] = float(np.quantile(dd, 0.98))
    return None, d


def scenes():
    from matchinglib_poselib_amd import synth

    out = []
    for n, frac, seed in ((5000, 0.5, 20260103), (800, 0.3, 11), (2000, 0.7, 12), (300, 0.5, 13), (8192, 0.25, 14), (1200, 0.9, 15),
                          (150, 0.6, 16), (3000, 0.15, 17), (64, 0.8, 18), (4000, 0.4, 19)):
        p1, p2, R, t, mask, th = synth.pose_scene(n, frac, seed=seed)
        rng = np.random.default_rng(seed)
        # PROSAC order: a noisy quality score (inliers tend to come first)
        score = rng.random(n) + 0.6 * (~mask)
        out.append(dict(p1=p1, p2=p2, th=th, mask=mask, order=np.argsort(score, kind="stable").astype(np.uint32), name=f"n{n}_f{frac}"))
    return out


def main():
    import oracle_lib
    import usac_ref_tool as u

    ora = oracle_lib.load()
    np.set_printoptions(linewidth=220, precision=6, suppress=True)
    bad = 0
    for sc in scenes():
        for refine in (0, 6):
            for prosac in (False, True):
                for seed in (12345, 7):
                    si = sc["order"] if prosac else None
                    r = u.run(sc["p1"], sc["p2"], sc["th"], seed, refine=refine, sorted_idx=si, solver_oracle=True)
                    o = ora.usac_essential(sc["p1"], sc["p2"], sc["th"], seed, refine=refine, sorted_idx=si, event_cap=200000)
                    first, d = compare(r["events"], o["events"])
                    same_fin = np.array_equal(r["final"][:8], o["final"][:8]) and np.array_equal(r["flags"], o["flags"])
                    print(f"{sc['name']:14s} refine {refine} prosac {int(prosac)} seed {seed:5d}: events {len(r['events']):6d} hyps {int(r['final'][1]):5d} "
                          f"inl {int(r['final'][5]):5d} LO {int(r['final'][7])} first_diff {first} final_equal {same_fin} {d}")
                    if first is not None or not same_fin:
                        bad += 1
                        if first is not None:
                            print(r["events"][max(0, first - 1):first + 2, :14])
                            print(o["events"][max(0, first - 1):first + 2, :14])
    print("scenes with differences:", bad)


if __name__ == "__main__":
    main()
