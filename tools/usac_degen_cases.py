"""Scenes and reference-built traces for USAC's degeneracy handling (rotation only, no motion, short baseline, general motion).
    python tools/usac_degen_cases.py make      # build container: runs oracle/_ref/usac_ref, writes tests/golden/usac_degen_trace.npz
    python tools/usac_degen_cases.py check     # GPU box: the device path against that fixture, run by run
Test infrastructure (the fixture's generator and a diagnostic runner); tests/test_gpu_usac_degeneracy.py holds the assertions."""
from __future__ import annotations

import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))
FIXTURE = os.path.join(ROOT, "tests", "golden", "usac_degen_trace.npz")
EVENTS_KEPT = 2500

# (name, n, inlier fraction, scene seed, pose_scene keywords)
SCENES = [
    ("general", 600, 0.6, 31, {}),
    ("general", 2000, 0.5, 32, {}),
    ("rotation", 600, 0.6, 31, dict(t_len=0.0)),
    ("rotation", 2000, 0.5, 33, dict(t_len=0.0)),
    ("rotation_clean", 400, 0.8, 34, dict(t_len=0.0, noise_px=0.05)),
    ("nomotion", 600, 0.6, 31, dict(t_len=0.0, rot_deg=0.0)),
    ("nomotion", 1500, 0.5, 35, dict(t_len=0.0, rot_deg=0.0)),
    ("shortbase", 600, 0.6, 31, dict(t_len=0.02)),
    ("shortbase", 2000, 0.5, 36, dict(t_len=0.05)),
]


def cases():
    """(key, p1, p2, th, order, truth, usac_seed, prosac, check) of every fixture run."""
    from matchinglib_poselib_amd import synth

    k = 0
    for name, n, frac, seed, kw in SCENES:
        p1, p2, R, t, truth, th = synth.pose_scene(n, frac, seed=seed, **kw)
        score = np.random.default_rng(seed).random(n) + 0.6 * (~truth)
        order = np.argsort(score, kind="stable").astype(np.uint32)
        for usac_seed, prosac, check in ((12345, False, 1), (7, True, 3), (99, False, 3)):
            yield f"k{k}", name, p1, p2, th, order, truth, usac_seed, prosac, check
            k += 1


def make():
    import usac_ref_tool as u

    out = {}
    for key, name, p1, p2, th, order, truth, usac_seed, prosac, check in cases():
        r = u.run(p1, p2, th, usac_seed, refine=0, sorted_idx=order if prosac else None, solver_oracle=True, check_degeneracy=check,
                  eigvec_smallest=False)
        ev = r["events"]
        out[key + "_meta"] = np.array([len(p1), usac_seed, int(prosac), check, len(ev)], np.float64)
        out[key + "_events"] = ev[:EVENTS_KEPT]
        out[key + "_final"] = r["final"]
        out[key + "_E"] = r["E"]
        out[key + "_flags"] = r["flags"]
        out[key + "_degen"] = r["degen"]
        out[key + "_R"] = r["R_degen"]
        out[key + "_flags_rot"] = r["flags_rot"]
        out[key + "_flags_nomot"] = r["flags_nomot"]
        cnt = {int(t): int((ev[:, 0] == t).sum()) for t in (7, 8, 9, 10)}
        print(f"{key} {name:15s} n {len(p1):5d} seed {usac_seed:5d} prosac {int(prosac)} check {check}: events {len(ev):6d} {cnt} "
              f"hyps {int(r['final'][1])} best {int(r['final'][5])} degen {r['degen'][:3]}")
    np.savez_compressed(FIXTURE, **out)
    print("written", FIXTURE, os.path.getsize(FIXTURE), "bytes")


def check():
    import usac_compare
    from matchinglib_poselib_amd import pose

    g = np.load(FIXTURE)
    np.set_printoptions(linewidth=220, precision=6, suppress=True)
    bad = 0
    for key, name, p1, p2, th, order, truth, usac_seed, prosac, chk in cases():
        ev = g[key + "_events"]
        d = pose.usac_essential(p1, p2, th, usac_seed, sorted_idx=order if prosac else None, event_cap=200000, check_degeneracy=chk)
        first, diffs = usac_compare.compare(ev, d["events"][:len(ev)])
        total = int(g[key + "_meta"][4])
        fin = np.array_equal(g[key + "_final"][:8], d["final"][:8]) and np.array_equal(g[key + "_flags"], d["flags"])
        dg = np.array_equal(g[key + "_degen"][:3], d["degen"][1:4][[0, 1, 2]]) if False else \
            (g[key + "_degen"][0] == d["degen"][1] and g[key + "_degen"][1] == d["degen"][2] and g[key + "_degen"][2] == d["degen"][3])
        print(f"{key} {name:15s} n {len(p1):5d}: ref events {total:6d} dev {d['n_events']:6d} first_diff {first} final {fin} degen {dg} "
              f"R {np.abs(g[key + '_R'] - d['R_degen']).max():.2e} {diffs} launches {int(d['stats'][5])}")
        if first is not None:
            bad += 1
            print(ev[max(0, first - 2):first + 2, :14])
            print(d["events"][max(0, first - 2):first + 2, :14])
    print("runs with differences:", bad)


if __name__ == "__main__":
    {"make": make, "check": check}[sys.argv[1]]()
