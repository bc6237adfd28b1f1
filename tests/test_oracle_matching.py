"""CPU-only: the matching oracle against an independent numpy brute force, the committed golden vectors
(generated with the reference's vendored NMSLIB, tests/golden/make_golden.py) and, when present, NMSLIB itself."""
import os
import subprocess
import tempfile

import numpy as np
import pytest

import oracle_lib
from matchinglib_poselib_amd import synth

GOLD = os.path.join(os.path.dirname(__file__), "golden")


def test_hamming_oracle_vs_numpy(oracle):
    q, t = synth.orb_pair(257, 263, seed=11)
    i0, d0 = oracle.knn_hamming(q, t)
    i1, d1 = oracle_lib.numpy_knn_hamming(q, t)
    assert np.array_equal(d0, d1)
    assert np.array_equal(i0, i1)


def test_hamming_oracle_ties(oracle):
    # duplicated train rows, all-zero descriptors, exact matches: ties resolve to the smaller train index
    rng = np.random.default_rng(5)
    t = rng.integers(0, 256, (64, 32), dtype=np.uint8)
    t[10] = t[3]
    t[40] = t[3]
    t[50] = 0
    t[51] = 0
    q = np.concatenate([t[3:4], np.zeros((1, 32), np.uint8), t[20:21]])
    idx, dist = oracle.knn_hamming(q, t)
    assert idx[0].tolist() == [3, 10] and dist[0].tolist() == [0, 0]
    assert idx[1].tolist() == [50, 51] and dist[1].tolist() == [0, 0]
    assert idx[2, 0] == 20 and dist[2, 0] == 0
    i1, d1 = oracle_lib.numpy_knn_hamming(q, t)
    assert np.array_equal(idx, i1) and np.array_equal(dist, d1)


def test_ratio_filter_semantics(oracle):
    idx = np.array([[1, 2], [3, 4], [5, 6], [7, 8]], np.int32)
    dist = np.array([[3, 4], [0, 0], [74, 100], [75, 100]], np.int32)
    m = oracle.ratio_filter(idx, dist)
    # 3 < 3.0 false ; 0 < 0 false (d1 = 0 never passes) ; 74 < 75 true ; 75 < 75 false
    assert m["queryIdx"].tolist() == [2]
    assert m["trainIdx"].tolist() == [5] and m["distance"].tolist() == [74.0] and m["imgIdx"].tolist() == [-1]


def test_l2_oracle_exact_on_integer_data(oracle):
    q, t = synth.sift_pair(64, 300, seed=3)
    idx, dist = oracle.knn_l2sq(q, t)
    d = ((q[:, None, :].astype(np.int64) - t[None, :, :].astype(np.int64)) ** 2).sum(-1)
    for i in range(q.shape[0]):
        order = np.lexsort((np.arange(t.shape[0]), d[i]))[:2]
        assert idx[i].tolist() == order.tolist()
        assert dist[i].tolist() == d[i][order].astype(np.float32).tolist()


def test_get_matches_codes(oracle):
    q, t = synth.orb_pair(40, 50, seed=2)
    assert oracle.get_matches_linear(14, 50, q[:14], t)[0] == -4
    assert oracle.get_matches_linear(41, 50, q, t)[0] == -1
    rc, m = oracle.get_matches_linear(40, 50, q, t)
    assert rc == 0 and len(m) >= 2
    assert np.all(np.diff(m["queryIdx"]) > 0)
    rc, m1 = oracle.get_matches_linear(40, 50, q, t, ratio_test=False)
    assert rc == 0 and len(m1) == 40


@pytest.mark.parametrize("name", ["hamming_257x263", "hamming_ties_96x80", "hamming_c1_2048"])
def test_hamming_oracle_vs_golden(oracle, name):
    g = np.load(os.path.join(GOLD, name + ".npz"))
    idx, dist = oracle.knn_hamming(g["q"], g["t"])
    assert np.array_equal(dist, g["dist"])          # distances pinned by NMSLIB
    assert np.array_equal(idx, g["idx"])            # indices: lexicographic (dist, idx), pinned by numpy
    # where NMSLIB's answer is tie-free it must agree on the indices too
    tf = g["nms_tie_free"]
    assert np.array_equal(idx[tf], g["nms_idx"][tf])
    m = oracle.ratio_filter(idx, dist)
    assert np.array_equal(m["queryIdx"], g["match_q"]) and np.array_equal(m["trainIdx"], g["match_t"])


@pytest.mark.ref
def test_hamming_oracle_vs_nmslib_live(oracle):
    tool = oracle_lib.ref_tool("nmslib_knn")
    if tool is None:
        pytest.skip("oracle/_ref/nmslib_knn not built (needs /root/reference)")
    q, t = synth.orb_pair(300, 500, seed=77)
    with tempfile.TemporaryDirectory() as d:
        fin, fout = os.path.join(d, "in.bin"), os.path.join(d, "out.bin")
        with open(fin, "wb") as f:
            np.array([q.shape[0], t.shape[0], q.shape[1]], np.int32).tofile(f)
            q.tofile(f)
            t.tofile(f)
        subprocess.run([tool, "hamming", fin, fout], check=True)
        raw = np.fromfile(fout, np.int32)
    n = q.shape[0] * 2
    nidx, ndist = raw[:n].reshape(-1, 2), raw[n:].reshape(-1, 2)
    idx, dist = oracle.knn_hamming(q, t)
    assert np.array_equal(dist, ndist)
