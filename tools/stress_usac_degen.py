"""Soak of USAC's degeneracy handling: random scenes (general motion, pure rotation, no motion, short baselines; 5 .. 4000 correspondences;
inlier ratios 0.1 .. 1; uniform and PROSAC sampling), device path against the CPU oracle.
    python tools/stress_usac_degen.py [seconds]          (GPU box)
Per run: both sides must agree event by event up to and including the first degeneracy test unless they part earlier at a model of a
sample without parallax (types 2 / 5 / 6: its solutions, their number or the oriented-constraint verdict on one of them); a no-motion upgrade must agree candidate by candidate; the verdict "degenerate" (both decision
thresholds in use) and the rotation-only model must agree at the end.  Prints the tally; exits non-zero on any violation."""
from __future__ import annotations

import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))
sys.path.insert(0, os.path.join(ROOT, "tools"))


def main():
    import warnings

    warnings.simplefilter("ignore", RuntimeWarning)
    import oracle_lib
    import usac_compare
    import usac_degen_checks as checks
    from matchinglib_poselib_amd import pose, synth

    budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
    ora = oracle_lib.load()
    rng = np.random.default_rng(int(os.environ.get("STRESS_SEED", "1")))
    t0 = time.time()
    tally = dict(runs=0, identical=0, parted_before_test=0, parted_after_test=0, nomotion_upgrades=0, verdict_true=0, violations=0)
    while time.time() - t0 < budget:
        kind = rng.integers(0, 5)
        n = int(rng.choice([5, 6, 8, 12, 20, 50, 120, 300, 700, 1500, 4000]))
        frac = float(rng.choice([0.1, 0.25, 0.5, 0.7, 0.9, 1.0]))
        kw = [{}, dict(t_len=0.0), dict(t_len=0.0, rot_deg=0.0), dict(t_len=float(rng.choice([0.01, 0.03, 0.1]))), dict(t_len=0.0, rot_deg=float(rng.uniform(0.2, 20)))][kind]
        noise = float(rng.choice([0.05, 0.3, 1.0]))
        if round(n * frac) < 2:
            continue
        p1, p2, R, t, truth, th = synth.pose_scene(n, frac, seed=int(rng.integers(1 << 30)), noise_px=noise, **kw)
        seed = int(rng.integers(1 << 30))
        chk = int(rng.choice([1, 3]))
        si = None
        if n >= 20 and rng.random() < 0.4:
            score = rng.random(n) + 0.6 * (~truth)
            si = np.argsort(score, kind="stable").astype(np.uint32)
        o = ora.usac_essential_degen(p1, p2, th, seed, check_degeneracy=chk, sorted_idx=si, event_cap=400000, max_hyp=5000)
        d = pose.usac_essential(p1, p2, th, seed, check_degeneracy=chk, sorted_idx=si, event_cap=400000, max_hyp=5000)
        tally["runs"] += 1
        tag = (kind, n, frac, noise, seed, chk, si is not None)
        if o["ok"] != d["ok"]:
            print("VIOLATION ok flag", tag, flush=True)
            tally["violations"] += 1
            continue
        if not o["ok"]:
            continue
        eo, ed = o["events"], d["events"]
        m = min(len(eo), len(ed))
        first, _ = usac_compare.compare(eo[:m], ed[:m])
        if first is None and len(eo) == len(ed):
            tally["identical"] += 1
        i7 = checks.first_of(eo, 7)
        if i7 is None:
            if first is not None and int(eo[first][0]) not in (1, 2, 3, 5):
                print("VIOLATION no test, parted at", tag, first, eo[first][:6], ed[first][:6], flush=True)
                tally["violations"] += 1
            continue
        if first is not None and first <= i7:
            if int(eo[first][0]) in (1, 2, 3, 5, 6) and kind != 0:   # 6 / 2: the oriented-constraint verdict of such a model differs (either side)
                tally["parted_before_test"] += 1
                continue
            if kind == 0 and int(eo[first][0]) in (1, 5):   # a double root: the known category of the plain run
                tally["parted_before_test"] += 1
                continue
            print("VIOLATION before / in the first test", tag, first, eo[first][:8], ed[first][:8], flush=True)
            tally["violations"] += 1
            continue
        if first is not None:
            tally["parted_after_test"] += 1
        i9 = checks.first_of(eo, 9)
        if i9 is not None and eo[i9][2] == 1 and (first is None or first > i9):
            tally["nomotion_upgrades"] += 1
        elif i9 is not None and eo[i9][2] == 1 and first is not None and first <= i9 and int(eo[first][0]) not in (1, 2, 3, 5):
            print("VIOLATION in a no-motion upgrade", tag, first, eo[first][:8], ed[first][:8], flush=True)
            tally["violations"] += 1
            continue
        for th_dec in (0.85, 1.65):
            a = checks.degenerate_decision(n, o["final"][5], o["degen"][1:3], th_dec)
            b = checks.degenerate_decision(n, d["final"][5], d["degen"][1:3], th_dec)
            # a verdict on the knife edge (the two sides' E-inlier counts differ by a few after the runs parted) is not a disagreement
            edge = any(abs(th_dec * r["final"][5] / n - r["degen"][k] / max(r["final"][5], 1)) < 0.02 for r in (o, d) for k in (1, 2))
            # (with 1 px of noise against the 0.8 px threshold the degenerate model explains a quarter of the true correspondences only:
            #  which sample's rotation collects most depends on the run, so the verdict is compared where the motion is well defined)
            if a != b and not edge and noise > 0.3:
                tally["verdict_differs_at_high_noise"] = tally.get("verdict_differs_at_high_noise", 0) + 1
            elif a != b and not edge and kind in (0, 1, 2) and n >= 50 and frac >= 0.25:
                print("VIOLATION verdict", tag, th_dec, o["degen"], d["degen"], o["final"][5], d["final"][5], flush=True)
                tally["violations"] += 1
        tally["verdict_true"] += int(checks.degenerate_decision(n, d["final"][5], d["degen"][1:3]))
    print(f"{time.time() - t0:.0f} s:", tally, flush=True)
    sys.exit(1 if tally["violations"] else 0)


if __name__ == "__main__":
    main()
