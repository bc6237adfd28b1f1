"""One-off soak: many random shapes, the matrix-core Hamming kernel against the integer VALU kernel (both exact by their own
tests against the oracle).  Usage: python tools/stress_hamming.py [seconds]"""
import os
import sys
import time

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np  # noqa: E402

import matchinglib_poselib_amd as mpa  # noqa: E402

budget = float(sys.argv[1]) if len(sys.argv) > 1 else 60.0
ctx = mpa.Context(0)
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 12345)
t0 = time.time()
cases = batched = 0


def batched_case():
    """Round 5: a FULL chip -- many pairs per call, so that a train set is one or two long splits (up to the 8192-row bound of the row
    fraction) -- matrix-core against VALU kernel through the batched device entry, both split caps drawn at random."""
    import torch
    from matchinglib_poselib_amd.matching import match_hamming_device
    B = int(rng.choice([8, 16, 33, 64]))
    nq = int(rng.choice([2048, 4096, 5000, 8192]))
    nt = int(rng.choice([4096, 4097, 6000, 8191, 8192, 8193, 12000, 16384]))
    alphabet = int(rng.choice([1, 3, 40, 100000]))
    if alphabet >= 100000:
        q = rng.integers(0, 256, (B, nq, 32), dtype=np.uint8)
        t = rng.integers(0, 256, (B, nt, 32), dtype=np.uint8)
    else:
        base = rng.integers(0, 256, (alphabet, 32), dtype=np.uint8)
        q = base[rng.integers(0, alphabet, (B, nq))]
        t = base[rng.integers(0, alphabet, (B, nt))]
        flip = rng.random(t.shape) < 0.01
        t = np.where(flip, rng.integers(0, 256, t.shape, dtype=np.uint8), t)
    dq, dt = torch.from_numpy(np.ascontiguousarray(q)).cuda(), torch.from_numpy(np.ascontiguousarray(t)).cuda()
    ctx.set_option("hamming_variant", 0)
    o0 = match_hamming_device(dq, dt, ctx=ctx)
    r0 = [o0[k].cpu().numpy().copy() for k in ("idx", "dist", "count")]
    ctx.set_option("hamming_variant", 3)
    ctx.set_option("hamming_mfma_qt", 0)
    ctx.set_option("hamming_split_rows", int(rng.choice([0, 4096])))
    ctx.set_option("hamming_train01", int(rng.choice([0, 1])))
    o3 = match_hamming_device(dq, dt, ctx=ctx)
    r3 = [o3[k].cpu().numpy().copy() for k in ("idx", "dist", "count")]
    ctx.set_option("hamming_split_rows", 0)
    if not all(np.array_equal(a, b) for a, b in zip(r0, r3)):
        print("MISMATCH (batched)", B, nq, nt, alphabet, flush=True)
        sys.exit(1)


while time.time() - t0 < budget:
    if rng.random() < 0.02:
        batched_case()
        batched += 1
        continue
    nq = int(rng.integers(1, 3000))
    nt = int(rng.integers(2, 20000))
    nbytes = int(rng.choice([1, 2, 3, 4, 7, 8, 9, 15, 16, 17, 24, 31, 32, 33, 40, 48, 61, 63, 64]))
    k = int(rng.integers(1, 3))
    alphabet = int(rng.choice([1, 2, 5, 50, 100000]))
    if alphabet >= 100000:
        q = rng.integers(0, 256, (nq, nbytes), dtype=np.uint8)
        t = rng.integers(0, 256, (nt, nbytes), dtype=np.uint8)
    else:
        base = rng.integers(0, 256, (alphabet, nbytes), dtype=np.uint8)
        t = base[rng.integers(0, alphabet, nt)]
        q = base[rng.integers(0, alphabet, nq)]
        if rng.random() < 0.5:
            flip = rng.random(q.shape) < 0.03
            q = np.where(flip, rng.integers(0, 256, q.shape, dtype=np.uint8), q)
    ctx.set_option("hamming_variant", 0)
    i0, d0 = mpa.knn_hamming(q, t, k=k, ctx=ctx)
    ctx.set_option("hamming_variant", 3)
    ctx.set_option("hamming_mfma_qt", int(rng.choice([0, 1, 2, 4])))
    ctx.set_option("hamming_train01", int(rng.choice([0, 1])))   # both encodings of the train operand
    i3, d3 = mpa.knn_hamming(q, t, k=k, ctx=ctx)
    if not (np.array_equal(i0, i3) and np.array_equal(d0, d3)):
        print("MISMATCH", nq, nt, nbytes, k, alphabet, flush=True)
        sys.exit(1)
    cases += 1
print(f"{cases} random cases + {batched} full-chip batches in {time.time() - t0:.0f} s: matrix-core == VALU kernel on all of them")
