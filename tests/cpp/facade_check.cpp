// facade_check.cpp -- drives the C++ drop-in facade exactly as a user of the reference would; the pytest wrapper
// (tests/test_gpu_facade.py) compares what it writes against the CPU oracle.
//   in : int32 nq, nt, nbytes ; q bytes ; t bytes ; int32 n ; p1 (n x 2 f64) ; p2 (n x 2 f64) ; f64 thresh ; uint32 seed
//   out: int32 err ; int32 n_matches ; DMatch[n_matches] ; int32 ok ; E[9] ; mask[n] ; int32 n_good ; R[9] ; t[3] ;
//        int32 sr_rc ; sr_E[9] ; int32 sr_inliers
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "matchinglib_poselib/matchinglib_matchers.h"
#include "matchinglib_poselib/pose_estim.h"
#include "matchinglib_poselib/stereo_pose_refinement.h"

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[3];
    if (fread(hdr, 4, 3, f) != 3) return 2;
    const int nq = hdr[0], nt = hdr[1], nb = hdr[2];
    cv::Mat d1(nq, nb, CV_8U), d2(nt, nb, CV_8U);
    if (fread(d1.data, 1, (size_t)nq * nb, f) != (size_t)nq * nb) return 2;
    if (fread(d2.data, 1, (size_t)nt * nb, f) != (size_t)nt * nb) return 2;
    int32_t n;
    if (fread(&n, 4, 1, f) != 1) return 2;
    cv::Mat p1(n, 2, CV_64F), p2(n, 2, CV_64F);
    if (fread(p1.data, 8, (size_t)n * 2, f) != (size_t)n * 2) return 2;
    if (fread(p2.data, 8, (size_t)n * 2, f) != (size_t)n * 2) return 2;
    double th;
    uint32_t seed;
    if (fread(&th, 8, 1, f) != 1 || fread(&seed, 4, 1, f) != 1) return 2;
    fclose(f);

    FILE *o = fopen(argv[2], "wb");
    std::vector<cv::KeyPoint> kp1((size_t)nq), kp2((size_t)nt);
    std::vector<cv::DMatch> matches;
    int32_t err = matchinglib::getMatches(kp1, kp2, d1, d2, cv::Size(640, 480), matches, "LINEAR", false, true);
    int32_t nm = (int32_t)matches.size();
    fwrite(&err, 4, 1, o);
    fwrite(&nm, 4, 1, o);
    fwrite(matches.data(), sizeof(cv::DMatch), matches.size(), o);

    poselib::setRansacSeed(seed);
    cv::Mat E, mask;
    int32_t ok = poselib::estimateEssentialMat(E, p1, p2, "RANSAC", th, false, mask) ? 1 : 0;
    fwrite(&ok, 4, 1, o);
    fwrite(E.data, 8, 9, o);
    fwrite(mask.data, 1, (size_t)n, o);
    cv::Mat R, t, Q;
    int32_t ng = poselib::getPoseTriangPts(E, p1, p2, R, t, Q, mask, 50.0);
    fwrite(&ng, 4, 1, o);
    fwrite(R.data, 8, 9, o);
    fwrite(t.data, 8, 3, o);

    // StereoRefine on pixel keypoints: K = [[800,0,320],[0,800,240],[0,0,1]] both cameras
    cv::Mat K = cv::Mat::zeros(3, 3, CV_64F);
    K.at<double>(0, 0) = 800, K.at<double>(1, 1) = 800, K.at<double>(0, 2) = 320, K.at<double>(1, 2) = 240, K.at<double>(2, 2) = 1;
    // the configuration struct filled field by field exactly as the reference harness does (tests/poselib-test/main.cpp:1389-1432)
    cv::Mat dist0_8 = cv::Mat::zeros(1, 8, CV_64F), dist1_8 = cv::Mat::zeros(1, 8, CV_64F);
    cv::Mat K0 = K, K1 = K;
    poselib::ConfigPoseEstimation cfg;
    cfg.dist0_8 = &dist0_8;
    cfg.dist1_8 = &dist1_8;
    cfg.K0 = &K0;
    cfg.K1 = &K1;
    cfg.keypointType = "ORB";
    cfg.descriptorType = "ORB";
    cfg.th_pix_user = 0.8;
    cfg.verbose = 0;
    cfg.Halign = 0;
    cfg.autoTH = false;
    cfg.BART = 0;
    cfg.kneipInsteadBA = false;
    cfg.refineMethod = poselib::RefinePostAlg::PR_NO_REFINEMENT;
    cfg.refineRTold = false;
    cfg.RobMethod = "RANSAC";
    cfg.refineMethod_CorrPool = poselib::RefinePostAlg::PR_STEWENIUS | poselib::RefinePostAlg::PR_PSEUDOHUBER_WEIGHTS;
    cfg.refineRTold_CorrPool = false;
    cfg.kneipInsteadBA_CorrPool = false;
    cfg.BART_CorrPool = 0;
    cfg.minStartAggInlRat = 0.2;
    cfg.relInlRatThLast = 0.35;
    cfg.relInlRatThNew = 0.2;
    cfg.minInlierRatSkip = 0.38;
    cfg.relMinInlierRatSkip = 0.7;
    cfg.maxSkipPairs = 5;
    cfg.minInlierRatioReInit = 0.6;
    cfg.minPtsDistance = 3.f;
    cfg.maxPoolCorrespondences = 30000;
    cfg.minContStablePoses = 3;
    cfg.absThRankingStable = 0.075;
    cfg.useRANSAC_fewMatches = false;
    cfg.checkPoolPoseRobust = 3;
    cfg.minNormDistStable = 0.5;
    cfg.raiseSkipCnt = (1 | (2 << 4));
    cfg.maxRat3DPtsFar = 0.5;
    cfg.maxDist3DPtsZ = 50.0;
    // ConfigUSAC likewise (main.cpp:1441-1458 fills it per image pair); carried, USAC itself is not part of this library
    poselib::ConfigUSAC cfg_usac;
    cfg_usac.focalLength = 800.0;
    cfg_usac.th_pixels = 0.8;
    cfg_usac.degeneracyCheck = poselib::UsacChkDegenType::DEGEN_USAC_INTERNAL;
    cfg_usac.estimator = poselib::PoseEstimator::POSE_STEWENIUS;
    cfg_usac.refinealg = poselib::RefineAlg::REF_STEWENIUS_WEIGHTS;
    cfg_usac.prevalidateSample = false;
    cfg_usac.noAutomaticProsacParamters = false;
    cfg_usac.automaticSprtInit = poselib::SprtInit::SPRT_DELTA_AUTOM_INIT | poselib::SprtInit::SPRT_EPSILON_AUTOM_INIT;
    cfg_usac.imgSize = cv::Size(640, 480);
    cfg_usac.degenDecisionTh = 0.85;
    static_assert(PIX_MIN_GOOD_TH == 1.6, "reference pose_estim.h:59");
    poselib::StereoRefine sr(cfg);
    std::vector<cv::KeyPoint> a((size_t)n), b((size_t)n);
    std::vector<cv::DMatch> mm((size_t)n);
    for (int i = 0; i < n; ++i) {
        a[i].pt = cv::Point2f((float)(p1.at<double>(i, 0) * 800 + 320), (float)(p1.at<double>(i, 1) * 800 + 240));
        b[i].pt = cv::Point2f((float)(p2.at<double>(i, 0) * 800 + 320), (float)(p2.at<double>(i, 1) * 800 + 240));
        mm[i].queryIdx = i, mm[i].trainIdx = i;
    }
    poselib::setRansacSeed(seed);
    cfg_usac.matches = &mm, cfg_usac.keypoints1 = &a, cfg_usac.keypoints2 = &b;
    cfg_usac.nrMatchesVfcFiltered = (unsigned)mm.size();
    int32_t rc = sr.addNewCorrespondences(mm, a, b, cfg_usac);
    int32_t inl = (int32_t)sr.nrInliersNew();
    double zero[9] = {0};
    fwrite(&rc, 4, 1, o);
    fwrite(rc == 0 ? (const void *)sr.E_new.data : (const void *)zero, 8, 9, o);
    fwrite(&inl, 4, 1, o);

    // estimateEssentialMat(..., "LMEDS") (pose_estim.cpp:874-877): same seed control, no refit
    poselib::setRansacSeed(seed);
    cv::Mat El, maskl;
    int32_t okl = poselib::estimateEssentialMat(El, p1, p2, "LMEDS", th, true, maskl) ? 1 : 0;
    fwrite(&okl, 4, 1, o);
    fwrite(okl ? (const void *)El.data : (const void *)zero, 8, 9, o);
    std::vector<uint8_t> ml((size_t)n, 0);
    if (okl) std::memcpy(ml.data(), maskl.data, (size_t)n);
    fwrite(ml.data(), 1, (size_t)n, o);

    // estimateEssentialMat with its DEFAULT method (ARRSAC, pose_estim.h:207), twice: the second call continues the samplers' streams
    for (int call = 0; call < 2; ++call) {
        cv::Mat Ea, maska;
        int32_t oka = poselib::estimateEssentialMat(Ea, p1, p2) ? 1 : 0;
        fwrite(&oka, 4, 1, o);
        fwrite(oka ? (const void *)Ea.data : (const void *)zero, 8, 9, o);
        uint64_t st[2];
        poselib::getArrsacRngState(&st[0], &st[1]);
        fwrite(st, 8, 2, o);
    }
    // AutoThEpi (pose_estim.cpp:81-300): ARRSAC with the threshold estimated from its own error statistics, starting at the file's threshold
    {
        const double pixToCam = th / 0.8;   // the scene generator's threshold is 0.8 pixels in camera units
        poselib::AutoThEpi ath(pixToCam);
        double tha = th;
        int32_t ng = 0;
        cv::Mat Eat, mat;
        int32_t rca = ath.estimateEVarTH(p1, p2, Eat, mat, &tha, &ng);
        fwrite(&rca, 4, 1, o);
        fwrite(&tha, 8, 1, o);
        fwrite(&ng, 4, 1, o);
        fwrite(rca == 0 ? (const void *)Eat.data : (const void *)zero, 8, 9, o);
    }
    // estimateEssentialOrPoseUSAC (pose_estim.h:212-223), twice: the second call starts its sequential test from what the first one handed
    // back (function-local statics in the reference); then StereoRefine with the harness' default RobMethod
    {
        for (int i = 0; i < n; ++i) mm[i].distance = (float)((i * 7919) % n) + 0.5f;   // distinct matching costs: a definite PROSAC order
        poselib::resetUsacHistory();
        poselib::ConfigUSAC cu;
        cu.focalLength = 800.0, cu.th_pixels = 0.8;
        cu.degeneracyCheck = poselib::UsacChkDegenType::DEGEN_NO_CHECK;
        cu.estimator = poselib::PoseEstimator::POSE_NISTER;
        cu.refinealg = poselib::RefineAlg::REF_WEIGHTS;
        cu.automaticSprtInit = poselib::SprtInit::SPRT_DELTA_AUTOM_INIT | poselib::SprtInit::SPRT_EPSILON_AUTOM_INIT;
        cu.imgSize = cv::Size(640, 480);
        cu.matches = &mm, cu.keypoints1 = &a, cu.keypoints2 = &b;
        cu.nrMatchesVfcFiltered = (unsigned)(n / 2);
        for (int call = 0; call < 2; ++call) {
            poselib::setRansacSeed(seed + 1 + call);
            cv::Mat Eu, inl;
            bool degenerate = true;
            int32_t rcu = poselib::estimateEssentialOrPoseUSAC(p1, p2, Eu, th, cu, degenerate, inl);
            int32_t deg = degenerate ? 1 : 0;
            fwrite(&rcu, 4, 1, o);
            fwrite(&deg, 4, 1, o);
            fwrite(rcu == 0 ? (const void *)Eu.data : (const void *)zero, 8, 9, o);
            std::vector<uint8_t> mu((size_t)n, 0);
            if (rcu == 0) std::memcpy(mu.data(), inl.data, (size_t)n);
            fwrite(mu.data(), 1, (size_t)n, o);
        }
        poselib::resetUsacHistory();
        poselib::ConfigPoseEstimation cfg2 = cfg;
        cfg2.RobMethod = "USAC";
        poselib::StereoRefine sr2(cfg2);
        poselib::setRansacSeed(seed + 3);
        int32_t rc2 = sr2.addNewCorrespondences(mm, a, b, cu);
        int32_t inl2 = (int32_t)sr2.nrInliersNew();
        fwrite(&rc2, 4, 1, o);
        fwrite(rc2 == 0 ? (const void *)sr2.E_new.data : (const void *)zero, 8, 9, o);
        fwrite(&inl2, 4, 1, o);
    }
    // a default-constructed ConfigUSAC (pose_estim.h:94-132: POSE_STEWENIUS, REF_STEWENIUS_WEIGHTS, DEGEN_USAC_INTERNAL, automatic SPRT
    // start values) with the caller's matches / keypoints / image size filled in, as the harness does per image pair: it runs the
    // algorithm it names.  Then the options this library does not build: each is refused with -1 like an unsupported configuration in
    // the reference, never served by another algorithm.
    {
        poselib::resetUsacHistory();
        poselib::ConfigUSAC cd;
        cd.imgSize = cv::Size(640, 480);
        cd.matches = &mm, cd.keypoints1 = &a, cd.keypoints2 = &b;
        cd.nrMatchesVfcFiltered = (unsigned)(n / 2);
        poselib::setRansacSeed(seed + 5);
        cv::Mat Eu, inl;
        bool degenerate = true;
        int32_t rcu = poselib::estimateEssentialOrPoseUSAC(p1, p2, Eu, th, cd, degenerate, inl);
        int32_t deg = degenerate ? 1 : 0;
        fwrite(&rcu, 4, 1, o);
        fwrite(&deg, 4, 1, o);
        fwrite(rcu == 0 ? (const void *)Eu.data : (const void *)zero, 8, 9, o);
        std::vector<uint8_t> mu((size_t)n, 0);
        if (rcu == 0) std::memcpy(mu.data(), inl.data, (size_t)n);
        fwrite(mu.data(), 1, (size_t)n, o);
        int32_t refused[5];
        for (int k = 0; k < 5; ++k) {
            poselib::ConfigUSAC cr = cd;
            if (k == 0) cr.estimator = poselib::PoseEstimator::POSE_EIG_KNEIP;
            if (k == 1) cr.refinealg = poselib::RefineAlg::REF_8PT_PSEUDOHUBER;
            if (k == 2) cr.refinealg = poselib::RefineAlg::REF_EIG_KNEIP;
            if (k == 3) cr.refinealg = poselib::RefineAlg::REF_EIG_KNEIP_WEIGHTS;
            if (k == 4) cr.degeneracyCheck = poselib::UsacChkDegenType::DEGEN_QDEGSAC;
            cv::Mat Er;
            bool dg = false;
            refused[k] = poselib::estimateEssentialOrPoseUSAC(p1, p2, Er, th, cr, dg);
        }
        fwrite(refused, 4, 5, o);
    }
    fclose(o);
    return 0;
}
