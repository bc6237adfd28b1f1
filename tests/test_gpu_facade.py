"""GPU: the C++ drop-in facade (reference signatures, include/matchinglib_poselib/*.h) driven by a C++ program, checked
against the CPU oracle."""
import os
import subprocess

import numpy as np
import pytest

import oracle_lib
from matchinglib_poselib_amd import synth

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
EXE = os.path.join(ROOT, "tests", "cpp", "facade_check")


def test_cpp_facade_end_to_end(oracle, tmp_path):
    assert os.path.exists(EXE), "run __graft_entry__.build()"
    q, t = synth.orb_pair(700, 900, seed=31)
    p1, p2, R, tv, mask, th = synth.pose_scene(1200, seed=32)
    seed = 4242
    fin, fout = tmp_path / "in.bin", tmp_path / "out.bin"
    with open(fin, "wb") as f:
        np.array([q.shape[0], t.shape[0], 32], np.int32).tofile(f)
        q.tofile(f)
        t.tofile(f)
        np.array([p1.shape[0]], np.int32).tofile(f)
        p1.tofile(f)
        p2.tofile(f)
        np.array([th], np.float64).tofile(f)
        np.array([seed], np.uint32).tofile(f)
    subprocess.run([EXE, str(fin), str(fout)], check=True, timeout=120)
    raw = open(fout, "rb").read()
    off = 0

    def take(dtype, count):
        nonlocal off
        a = np.frombuffer(raw, dtype=dtype, count=count, offset=off)
        off += a.nbytes
        return a

    err, nm = take(np.int32, 2)
    matches = take(oracle_lib.DMATCH, int(nm))
    rc, om = oracle.get_matches_linear(700, 900, q, t)
    assert err == rc == 0 and matches.tobytes() == om.tobytes()

    ok = take(np.int32, 1)[0]
    E = take(np.float64, 9).reshape(3, 3)
    m = take(np.uint8, 1200)
    o = oracle.ransac_essential(p1, p2, th, confidence=0.999, max_iters=1000, lesqu=False, seed=seed)
    assert ok == 1 and o["ok"]
    assert min(np.abs(E - o["E"]).max(), np.abs(E + o["E"]).max()) < 1e-8
    assert np.array_equal(m, o["mask"])
    ng = take(np.int32, 1)[0]
    Rg = take(np.float64, 9).reshape(3, 3)
    tg = take(np.float64, 3)
    go, Ro, to, Qo, mo = oracle.recover_pose(o["E"], p1, p2, 50.0, o["mask"])
    assert ng == go and np.abs(Rg - Ro).max() < 1e-6 and np.abs(tg - to).max() < 1e-6

    sr_rc = take(np.int32, 1)[0]
    sr_E = take(np.float64, 9).reshape(3, 3)
    sr_inl = take(np.int32, 1)[0]
    assert sr_rc == 0
    # StereoRefine works on float-rounded camera coordinates (ImgToCamCoordTrans) with th = 0.8 px * pixToCamFact
    f32 = lambda a: ((a * 800 + np.array([320, 240])).astype(np.float32).astype(np.float64) - np.array([320, 240])) / 800  # noqa: E731
    q1 = f32(p1).astype(np.float32).astype(np.float64)
    q2 = f32(p2).astype(np.float32).astype(np.float64)
    o2 = oracle.ransac_essential(q1, q2, 0.8 * synth.PIX_TO_CAM, confidence=0.999, max_iters=1000, lesqu=False, seed=seed)
    assert sr_inl == o2["n_inliers"]
    assert min(np.abs(sr_E - o2["E"]).max(), np.abs(sr_E + o2["E"]).max()) < 1e-8

    okl = take(np.int32, 1)[0]
    El = take(np.float64, 9).reshape(3, 3)
    ml = take(np.uint8, 1200)
    o3 = oracle.lmeds_essential(p1, p2, confidence=0.999, max_iters=2000, seed=seed)
    assert okl == 1 and o3["ok"]
    assert min(np.abs(El - o3["E"]).max(), np.abs(El + o3["E"]).max()) < 1e-8
    assert (ml != o3["mask"]).sum() <= 2

    # the default method, ARRSAC with refinement, called twice in one process: the second call starts from the stream positions the
    # first one left (function-local static cv::RNGs in the reference)
    st = np.array([0xFFFFFFFF, 0xFFFFFFFF], np.uint64)
    for call in range(2):
        oka = take(np.int32, 1)[0]
        Ea = take(np.float64, 9).reshape(3, 3)
        sta = take(np.uint64, 2)
        o4 = oracle.arrsac_essential(p1, p2, 1.6, refine=True, rng_state=st)
        assert oka == int(o4["ok"]) and sta.tolist() == st.tolist(), (call, oka, o4["ok"], sta, st)
        if oka:
            a, b = Ea / np.linalg.norm(Ea), o4["E"] / np.linalg.norm(o4["E"])
            assert min(np.abs(a - b).max(), np.abs(a + b).max()) < 2e-5

    # AutoThEpi::estimateEVarTH: the same loop over the CPU oracle, continuing the samplers' streams
    from stereo_refine_oracle import AutoThEpiOracle
    rca = take(np.int32, 1)[0]
    tha = take(np.float64, 1)[0]
    nga = take(np.int32, 1)[0]
    Eat = take(np.float64, 9).reshape(3, 3)
    auto = AutoThEpiOracle(oracle, th / 0.8, lambda t: oracle.arrsac_essential(p1, p2, t, refine=True, rng_state=st))
    rc_o, E_o, mask_o, th_o, ng_o = auto.estimate_e_var_th(p1, p2, th)
    assert rca == rc_o == 0 and nga == ng_o and abs(tha - th_o) < 1e-9 * th_o, (rca, rc_o, nga, ng_o, tha, th_o)
    a, b = Eat / np.linalg.norm(Eat), E_o / np.linalg.norm(E_o)
    assert min(np.abs(a - b).max(), np.abs(a + b).max()) < 2e-5
