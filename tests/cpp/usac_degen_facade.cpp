// usac_degen_facade.cpp -- the drop-in facade's DEGEN_USAC_INTERNAL path as a user of the reference would call it
// (poselib::estimateEssentialOrPoseUSAC, pose_estim.h:212-223; StereoRefine with RobMethod "USAC"); tests/test_gpu_usac_degeneracy.py
// checks what it writes.
//   in : int32 n ; p1 (n x 2 f64) ; p2 (n x 2 f64) ; f64 thresh ; uint32 seed ; f64 degenDecisionTh
//   out: int32 rc ; int32 degenerate ; E[9] ; mask[n] ; int32 have_R ; R_degenerate[9] ; int32 have_mask ; inliers_degenerate_R[n] ;
//        int32 stereo_refine_rc
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <vector>

#include "matchinglib_poselib/pose_estim.h"
#include "matchinglib_poselib/stereo_pose_refinement.h"

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t n;
    if (fread(&n, 4, 1, f) != 1) return 2;
    cv::Mat p1(n, 2, CV_64F), p2(n, 2, CV_64F);
    if (fread(p1.data, 8, (size_t)n * 2, f) != (size_t)n * 2 || fread(p2.data, 8, (size_t)n * 2, f) != (size_t)n * 2) return 2;
    double th, decision;
    uint32_t seed;
    if (fread(&th, 8, 1, f) != 1 || fread(&seed, 4, 1, f) != 1 || fread(&decision, 8, 1, f) != 1) return 2;
    fclose(f);
    FILE *o = fopen(argv[2], "wb");
    const double zero[9] = {0};

    poselib::resetUsacHistory();
    poselib::ConfigUSAC cu;  // defaults: DEGEN_USAC_INTERNAL, focal length 800, 0.8 pixels
    cu.estimator = poselib::PoseEstimator::POSE_NISTER;
    cu.refinealg = poselib::RefineAlg::REF_WEIGHTS;
    cu.automaticSprtInit = poselib::SprtInit::SPRT_DEFAULT_INIT;
    cu.noAutomaticProsacParamters = true;
    cu.degenDecisionTh = decision;
    poselib::setRansacSeed(seed);
    cv::Mat E, inl, Rd, inl_R;
    bool degenerate = false;
    int32_t rc = poselib::estimateEssentialOrPoseUSAC(p1, p2, E, th, cu, degenerate, inl, Rd, inl_R);
    int32_t deg = degenerate ? 1 : 0, have_R = Rd.empty() ? 0 : 1, have_mask = inl_R.empty() ? 0 : 1;
    fwrite(&rc, 4, 1, o);
    fwrite(&deg, 4, 1, o);
    fwrite(rc == 0 ? (const void *)E.data : (const void *)zero, 8, 9, o);
    std::vector<uint8_t> m((size_t)n, 0);
    if (rc == 0) std::memcpy(m.data(), inl.data, (size_t)n);
    fwrite(m.data(), 1, (size_t)n, o);
    fwrite(&have_R, 4, 1, o);
    fwrite(have_R ? (const void *)Rd.data : (const void *)zero, 8, 9, o);
    fwrite(&have_mask, 4, 1, o);
    std::fill(m.begin(), m.end(), 0);
    if (have_mask) std::memcpy(m.data(), inl_R.data, (size_t)n);
    fwrite(m.data(), 1, (size_t)n, o);

    // StereoRefine, RobMethod "USAC": a degenerate configuration ends robustPoseEstimation with -2 (stereo_pose_refinement.cpp:1402-1411),
    // which addNewCorrespondences reports as -1 (:974-977)
    cv::Mat K = cv::Mat::zeros(3, 3, CV_64F);
    K.at<double>(0, 0) = 800, K.at<double>(1, 1) = 800, K.at<double>(0, 2) = 320, K.at<double>(1, 2) = 240, K.at<double>(2, 2) = 1;
    cv::Mat dist0 = cv::Mat::zeros(1, 8, CV_64F), dist1 = cv::Mat::zeros(1, 8, CV_64F), K0 = K, K1 = K;
    poselib::ConfigPoseEstimation cfg;
    cfg.dist0_8 = &dist0, cfg.dist1_8 = &dist1, cfg.K0 = &K0, cfg.K1 = &K1;
    cfg.th_pix_user = 0.8, cfg.verbose = 0, cfg.autoTH = false, cfg.BART = 0, cfg.kneipInsteadBA = false;
    cfg.refineMethod = poselib::RefinePostAlg::PR_NO_REFINEMENT, cfg.refineRTold = false;
    cfg.RobMethod = "USAC";
    poselib::StereoRefine sr(cfg);
    std::vector<cv::KeyPoint> a((size_t)n), b((size_t)n);
    std::vector<cv::DMatch> mm((size_t)n);
    for (int i = 0; i < n; ++i) {
        a[i].pt = cv::Point2f((float)(p1.at<double>(i, 0) * 800 + 320), (float)(p1.at<double>(i, 1) * 800 + 240));
        b[i].pt = cv::Point2f((float)(p2.at<double>(i, 0) * 800 + 320), (float)(p2.at<double>(i, 1) * 800 + 240));
        mm[i].queryIdx = i, mm[i].trainIdx = i, mm[i].distance = (float)((i * 7919) % n) + 0.5f;
    }
    poselib::resetUsacHistory();
    poselib::setRansacSeed(seed + 1);
    cu.imgSize = cv::Size(640, 480);
    cu.matches = &mm, cu.keypoints1 = &a, cu.keypoints2 = &b;
    cu.nrMatchesVfcFiltered = (unsigned)mm.size();
    int32_t src = sr.addNewCorrespondences(mm, a, b, cu);
    fwrite(&src, 4, 1, o);
    fclose(o);
    return 0;
}
