"""mlpl_sorted_match_idx = poselib::getSortedMatchIdx (pose_helper.cpp:2896-2923): the indices of the matches in the order std::sort leaves
them when comparing the distances.  A host function: checked here without a GPU.  With distinct distances the order is the ascending one;
with ties (integer Hamming costs: always) it is whatever this C++ library's std::sort does with them -- the same call the reference makes --
so only the properties every std::sort has are asserted for those."""
import numpy as np

from matchinglib_poselib_amd import _lib

DMATCH = np.dtype([("queryIdx", np.int32), ("trainIdx", np.int32), ("imgIdx", np.int32), ("distance", np.float32)])


def _order(dist):
    lib = _lib.load_library()
    m = np.zeros(len(dist), DMATCH)
    m["queryIdx"] = np.arange(len(dist))
    m["distance"] = dist
    out = np.zeros(len(dist), np.uint32)
    assert lib.mlpl_sorted_match_idx(m.ctypes.data, len(dist), out.ctypes.data) == 0
    return out


def test_distinct_distances_come_out_ascending():
    rng = np.random.default_rng(1)
    for n in (1, 2, 17, 1000, 5180):
        d = rng.permutation(n).astype(np.float32) * 0.5
        assert np.array_equal(_order(d), np.argsort(d, kind="stable").astype(np.uint32))


def test_ties_give_a_permutation_sorted_by_distance_and_the_same_one_every_time():
    rng = np.random.default_rng(2)
    d = rng.integers(0, 60, 5180).astype(np.float32)     # Hamming costs: many equal
    o = _order(d)
    assert np.array_equal(np.sort(o), np.arange(len(d), dtype=np.uint32))
    assert np.all(np.diff(d[o]) >= 0)
    assert np.array_equal(o, _order(d))


def test_empty_list():
    assert len(_order(np.zeros(0, np.float32))) == 0
