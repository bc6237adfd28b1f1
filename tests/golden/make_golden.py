"""Generates the golden vectors under tests/golden/ (run in the build container, where /root/reference exists):

    make -C oracle ref && python tests/golden/make_golden.py

Matching vectors: inputs from matchinglib_poselib_amd.synth, expected (idx, dist) from two independent sources:
the reference's vendored NMSLIB seq_search (oracle/_ref/nmslib_knn, built from /root/reference sources in place)
for the distances and the tie-free indices, and a numpy brute force with lexicographic (dist, idx) selection for
the cvflann tie order.  Fixtures are data only (inputs + expected outputs).
"""
import os
import subprocess
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "tests"))

import oracle_lib  # noqa: E402
from matchinglib_poselib_amd import synth  # noqa: E402


def run_nmslib(mode, q, t):
    tool = oracle_lib.ref_tool("nmslib_knn")
    assert tool, "build oracle/_ref first: make -C oracle ref"
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            np.array([q.shape[0], t.shape[0], q.shape[1]], np.int32).tofile(f)
            q.tofile(f)
            t.tofile(f)
        subprocess.run([tool, mode, fin, fout], check=True)
        n = q.shape[0] * 2
        if mode.startswith("hamming"):
            raw = np.fromfile(fout, np.int32)
            return raw[:n].reshape(-1, 2), raw[n:].reshape(-1, 2)
        idx = np.fromfile(fout, np.int32, count=n).reshape(-1, 2)
        dist = np.fromfile(fout, np.float32, offset=4 * n).reshape(-1, 2)
        return idx, dist


def hamming_case(name, q, t):
    idx, dist = oracle_lib.numpy_knn_hamming(q, t)
    nidx, ndist = run_nmslib("hamming", q, t)
    assert np.array_equal(dist, ndist), "NMSLIB distances disagree with numpy brute force"
    lut = np.array([bin(i).count("1") for i in range(256)], np.int32)
    tie_free = np.zeros(q.shape[0], bool)
    for i in range(q.shape[0]):
        d = lut[np.bitwise_xor(t, q[i][None, :])].sum(axis=1)
        tie_free[i] = (d <= dist[i, 1]).sum() == 2 and dist[i, 0] < dist[i, 1]
    assert np.array_equal(idx[tie_free], nidx[tie_free])
    keep = dist[:, 0].astype(np.float32) < np.float32(0.75) * dist[:, 1].astype(np.float32)
    np.savez_compressed(os.path.join(HERE, name + ".npz"), q=q, t=t, idx=idx, dist=dist, nms_idx=nidx,
                        nms_tie_free=tie_free, match_q=np.nonzero(keep)[0].astype(np.int32),
                        match_t=idx[keep, 0])
    print(name, q.shape, t.shape, "tie-free", int(tie_free.sum()), "matches", int(keep.sum()))


def main():
    q, t = synth.orb_pair(257, 263, seed=20260111)
    hamming_case("hamming_257x263", q, t)
    # adversarial ties: duplicated train rows, zero descriptors, exact copies as queries
    rng = np.random.default_rng(20260112)
    t = rng.integers(0, 256, (80, 32), dtype=np.uint8)
    t[7] = t[5]; t[33] = t[5]; t[60] = 0; t[61] = 0; t[79] = t[0]
    q = rng.integers(0, 256, (96, 32), dtype=np.uint8)
    q[:20] = t[:20]
    q[20] = 0
    q[21:40] = t[40:59] ^ np.uint8(1)
    hamming_case("hamming_ties_96x80", q, t)
    q, t = synth.orb_pair(2048, 2048, seed=20260101)  # BASELINE config C1
    hamming_case("hamming_c1_2048", q, t)




# ---- 5-point solver vectors: inputs + E-sets of the reference's vendored OpenGV fivept_nister ----------------
def run_opengv(pts):
    tool = oracle_lib.ref_tool("opengv_5pt")
    assert tool, "build oracle/_ref first: make -C oracle ref"
    ns, npts = pts.shape[:2]
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            np.array([ns, npts], np.int32).tofile(f)
            np.ascontiguousarray(pts, np.float64).tofile(f)
        subprocess.run([tool, fin, fout], check=True)
        raw = np.fromfile(fout, np.uint8).reshape(ns, 4 + 720)
    cnt = raw[:, :4].copy().view(np.int32)[:, 0]
    Es = raw[:, 4:].copy().view(np.float64).reshape(ns, 10, 3, 3)
    return cnt, Es


def fivept_case():
    p1, p2, R, t, mask, th = synth.pose_scene(5000, seed=20260103)
    rng = np.random.default_rng(20260113)
    ns = 120
    samples = np.stack([rng.choice(5000, 5, replace=False) for _ in range(ns)]).astype(np.int32)
    pts = np.concatenate([p1[samples], p2[samples]], axis=2)  # ns x 5 x (x1,y1,x2,y2)
    cnt, Es = run_opengv(pts)
    np.savez_compressed(os.path.join(HERE, "fivept_opengv_120.npz"), pts=pts, count=cnt, E=Es)
    print("fivept_opengv_120: mean #solutions", cnt.mean())


def nms_wrapper_case():
    """BRUTEFORCENMS: expected (idx, dist) straight from the reference's vendored NMSLIB driven with the reference
    wrapper's own packing (2 bytes per int, no length word => last 2 bytes ignored) and its 'l2' space (sqrt distances)."""
    out = {}
    for nb in (32, 31):
        q, t = synth.orb_pair(200, 260, nbytes=nb, seed=20260120 + nb)
        idx, dist = run_nmslib("hamming_wrapper", q, t)
        out[f"u8_{nb}_q"], out[f"u8_{nb}_t"], out[f"u8_{nb}_idx"], out[f"u8_{nb}_dist"] = q, t, idx, dist
    rng = np.random.default_rng(20260121)
    t = rng.normal(size=(260, 64)).astype(np.float32)
    q = rng.normal(size=(200, 64)).astype(np.float32)
    q[:60] = t[:60] + rng.normal(size=(60, 64)).astype(np.float32) * 0.05
    idx, dist = run_nmslib("l2", q, t)
    out["f32_q"], out["f32_t"], out["f32_idx"], out["f32_dist"] = q, t, idx, dist
    np.savez_compressed(os.path.join(HERE, "bruteforce_nms.npz"), **out)
    print("bruteforce_nms: ok")


if __name__ == "__main__":
    what = sys.argv[1] if len(sys.argv) > 1 else "all"
    if what in ("all", "nms"):
        nms_wrapper_case()
    if what in ("all", "matching"):
        main()
    if what in ("all", "pose"):
        fivept_case()
