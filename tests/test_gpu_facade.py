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
    # LMedS' mask is err <= sigma^2 with sigma from the median error of the best model: device and oracle agree on that model to 1e-8, so a
    # correspondence may differ only where the oracle's own error sits on ITS boundary (between its largest inlier and smallest outlier error)
    flips = np.nonzero(ml != o3["mask"])[0]
    if len(flips):
        err = oracle.sampson_err(p1, p2, o3["E"]).astype(np.float64)
        edge_in, edge_out = err[o3["mask"] != 0].max(), err[o3["mask"] == 0].min()
        assert len(flips) <= 4 and all(min(abs(err[i] - edge_in), abs(err[i] - edge_out)) <= 2e-5 * edge_in for i in flips), (flips, err[flips], edge_in, edge_out)

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
            assert min(np.abs(a - b).max(), np.abs(a + b).max()) < 1e-7

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
    assert min(np.abs(a - b).max(), np.abs(a + b).max()) < 1e-7

    # estimateEssentialOrPoseUSAC: PROSAC in the order of the matching costs, delta from the convex hulls of the keypoints, epsilon from
    # the share of matches a flow filter kept; the second call starts from the delta / epsilon the first one handed back
    from scipy.spatial import ConvexHull
    n = len(p1)
    dist = (np.arange(n) * 7919 % n).astype(np.float32) + 0.5
    order = np.argsort(dist, kind="stable").astype(np.uint32)
    kpa = (p1 * 800 + np.array([320, 240])).astype(np.float32)
    kpb = (p2 * 800 + np.array([320, 240])).astype(np.float32)
    area = min(ConvexHull(kpa.astype(np.float64)).volume, ConvexHull(kpb.astype(np.float64)).volume)
    max_epi = np.sqrt(640.0 ** 2 + 480.0 ** 2) * 2 * 0.8
    delta0 = max(max_epi / max(area, 6 * max_epi), 0.001)
    eps0 = min(max(0.8 * (n // 2) / n, 0.1), 0.4)
    d_old, d_new, e_old, e_new, numhyps, models = 0.0, delta0, 0.0, eps0, 0, 0
    for call in range(2):
        rcu, deg = take(np.int32, 2)
        Eu = take(np.float64, 9)
        mu = take(np.uint8, n)
        if call == 0:
            delta, eps, ms = delta0, eps0, 8.5
        else:   # pose_estim.cpp:1812-1875: no valid statistic yet -> the last result if it moved by less than 33 % / 40 %, else a new estimate
            delta = d_new if abs((d_old - d_new) / d_old) < 0.33 else delta0
            eps = e_new if abs((e_old - e_new) / e_old) < 0.4 else eps0
            ms = models / numhyps
        o5 = oracle.usac_essential(p1, p2, th, seed + 1 + call, sorted_idx=order, prosac_beta=delta, sprt_delta=delta, sprt_epsilon=eps,
                                   sprt_ms=ms)
        assert rcu == 0 and deg == 0 and o5["ok"]
        assert np.array_equal(mu, o5["flags"]), (call, int((mu != o5["flags"]).sum()))
        a5, b5 = Eu / np.linalg.norm(Eu), o5["E"] / np.linalg.norm(o5["E"])
        assert min(np.abs(a5 - b5).max(), np.abs(a5 + b5).max()) < 1e-8
        numhyps, models = numhyps + int(o5["final"][1]), models + int(o5["final"][2])
        e_res = o5["final"][9] / 2 if (o5["final"][1] > 2000 and o5["final"][9] > 0.2) else o5["final"][9]
        d_old, d_new, e_old, e_new = d_new, o5["final"][8], e_new, e_res
    # StereoRefine with RobMethod = "USAC" (the harness default): first call = robust initialisation on float-rounded camera coordinates
    rc2 = take(np.int32, 1)[0]
    E2 = take(np.float64, 9)
    inl2 = take(np.int32, 1)[0]
    area2 = min(ConvexHull(kpa.astype(np.float64)).volume, ConvexHull(kpb.astype(np.float64)).volume)
    o6 = oracle.usac_essential(q1, q2, 0.8 * synth.PIX_TO_CAM, seed + 3, sorted_idx=order, prosac_beta=delta0, sprt_delta=delta0,
                               sprt_epsilon=eps0, sprt_ms=8.5)   # resetUsacHistory() also forgets the models-per-sample average
    assert rc2 == 0 and area2 == area
    assert inl2 == int(o6["flags"].sum())
    a6, b6 = E2 / np.linalg.norm(E2), o6["E"] / np.linalg.norm(o6["E"])
    assert min(np.abs(a6 - b6).max(), np.abs(a6 + b6).max()) < 1e-8
    # a default-constructed ConfigUSAC: POSE_STEWENIUS + REF_STEWENIUS_WEIGHTS + DEGEN_USAC_INTERNAL on a process' first call (mS = 6,
    # tM = 2736; the rotation-only / no-motion tests after every new best model) -- the algorithm the configuration names
    rcd, degd = take(np.int32, 2)
    Ed = take(np.float64, 9)
    md = take(np.uint8, n)
    o7 = oracle.usac_essential_degen(p1, p2, th, seed + 5, check_degeneracy=1, refine=5, sorted_idx=order, prosac_beta=delta0, sprt_delta=delta0,
                                     sprt_epsilon=eps0, sprt_ms=6.0, sprt_tm=2736.0)
    assert rcd == 0 and degd == 0 and o7["ok"]
    assert np.array_equal(md, o7["flags"]), int((md != o7["flags"]).sum())
    a7, b7 = Ed / np.linalg.norm(Ed), o7["E"] / np.linalg.norm(o7["E"])
    assert min(np.abs(a7 - b7).max(), np.abs(a7 + b7).max()) < 1e-8
    # ... and what is not built is refused, not substituted: POSE_EIG_KNEIP, REF_8PT_PSEUDOHUBER, REF_EIG_KNEIP(_WEIGHTS), DEGEN_QDEGSAC
    assert take(np.int32, 5).tolist() == [-1] * 5
