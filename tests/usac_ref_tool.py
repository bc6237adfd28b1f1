"""Runs oracle/_ref/usac_ref (the reference's USAC.h compiled in place, build container only) and parses its event trace.
Test infrastructure: used by tests/golden/make_golden.py to generate tests/golden/usac_trace.npz and by the `ref`-marked tests."""
from __future__ import annotations

import os
import struct
import subprocess
import tempfile

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
TOOL = os.path.join(_HERE, "..", "oracle", "_ref", "usac_ref")


def available() -> bool:
    return os.path.exists(TOOL)


def run(p1, p2, th, seed, refine=0, sorted_idx=None, max_hyp=50000, prosac_beta=0.09, sprt_delta=0.05, sprt_epsilon=0.15,
        sprt_ms=8.5, sprt_tm=2314.0, conf=0.99, native_order=False, solver_oracle=False, check_degeneracy=False, th_pixels=0.8,
        focal=800.0, eigvec_smallest=False, stewenius=False):
    """Returns dict(events=(k,16) float64, final=(12,), E=(9,), flags=(n,) uint8, pool=(n,) int32); with check_degeneracy also
    degen=(4,) [inliers of the rotation, of "no motion", degeneracy type, 0], R_degen=(9,), flags_rot, flags_nomot=(n,) uint8."""
    n = len(p1)
    pts = np.concatenate([np.asarray(p1, np.float64), np.asarray(p2, np.float64)], axis=1)
    with tempfile.TemporaryDirectory() as d:
        fi, fo = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fi, "wb") as f:
            f.write(struct.pack("8i", n, int(seed), int(refine), 0 if sorted_idx is None else 1, int(max_hyp), int(check_degeneracy), 0, 0))
            f.write(struct.pack("8d", th, prosac_beta, sprt_delta, sprt_epsilon, sprt_ms, sprt_tm, conf, float(th_pixels) / float(focal)))
            f.write(np.ascontiguousarray(pts).tobytes())
            if sorted_idx is not None:
                f.write(np.ascontiguousarray(sorted_idx, np.uint32).tobytes())
        subprocess.run([TOOL, fi, fo] + (["--native-order"] if native_order else []) + (["--solver-oracle"] if solver_oracle else [])
                       + (["--eigvec-smallest"] if eigvec_smallest else []) + (["--stewenius"] if stewenius else []),
                       check=True)
        raw = open(fo, "rb").read()
    ne = struct.unpack("i", raw[:4])[0]
    off = 4
    ev = np.frombuffer(raw, np.float64, ne * 16, off).reshape(ne, 16).copy()
    off += ne * 128
    fin = np.frombuffer(raw, np.float64, 12, off).copy()
    off += 96
    E = np.frombuffer(raw, np.float64, 9, off).copy()
    off += 72
    flags = np.frombuffer(raw, np.float64, n, off).astype(np.uint8)
    off += 8 * n
    pool = np.frombuffer(raw, np.float64, n, off).astype(np.int32)
    out = dict(events=ev, final=fin, E=E, flags=flags, pool=pool)
    if check_degeneracy:
        off += 8 * n
        out["degen"] = np.frombuffer(raw, np.float64, 4, off).copy()
        out["R_degen"] = np.frombuffer(raw, np.float64, 9, off + 32).copy()
        out["flags_rot"] = np.frombuffer(raw, np.float64, n, off + 104).astype(np.uint8)
        out["flags_nomot"] = np.frombuffer(raw, np.float64, n, off + 104 + 8 * n).astype(np.uint8)
    return out
