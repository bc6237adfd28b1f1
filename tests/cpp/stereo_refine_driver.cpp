// stereo_refine_driver.cpp -- runs poselib::StereoRefine over a sequence of stereo frames the way the reference harness does
// (tests/poselib-test/main.cpp:1460-1530: one addNewCorrespondences() per image pair) and dumps the state after every frame;
// tests/test_gpu_stereo_refine.py compares the dump with the CPU restatement of the state machine (tests/stereo_refine_oracle.py).
//   in : int32 nframes, robMethod(0 RANSAC, 1 LMEDS, 2 ARRSAC) ; uint32 seed ; f64 K0[4], K1[4] (fx fy cx cy) ; f64 dist0[8], dist1[8] ; f64 cfg[21] ;
//        per frame: int32 n ; f32 kp1[n][3] (x, y, response) ; f32 kp2[n][3] ; f32 descrDist[n]      (match i joins kp1[i] and kp2[i])
//   out: per frame: int32 rc, nr_inliers_new, nr_corrs_new, pool, nrEstimation, skipCount, poseIsStable, mostLikelyPose_stable, history ;
//        f64 E_new[9], R_new[9], t_new[3], E_mostLikely[9]  (zeros while empty)
#include <cstdint>
#include <cstdio>
#include <vector>

#include "matchinglib_poselib/stereo_pose_refinement.h"

static void dump(FILE *o, const cv::Mat &m, int count) {
    std::vector<double> v((size_t)count, 0.0);
    if (!m.empty())
        for (int i = 0; i < count; ++i) v[(size_t)i] = m.cols == 1 ? m.at<double>(i, 0) : m.at<double>(i / m.cols, i % m.cols);
    fwrite(v.data(), 8, (size_t)count, o);
}

int main(int argc, char **argv) {
    if (argc != 3) return 2;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[2];
    uint32_t seed;
    double k0[4], k1[4], d0[8], d1[8], c[21];
    if (fread(hdr, 4, 2, f) != 2 || fread(&seed, 4, 1, f) != 1 || fread(k0, 8, 4, f) != 4 || fread(k1, 8, 4, f) != 4 ||
        fread(d0, 8, 8, f) != 8 || fread(d1, 8, 8, f) != 8 || fread(c, 8, 21, f) != 21)
        return 2;
    cv::Mat K0 = cv::Mat::zeros(3, 3, CV_64F), K1 = cv::Mat::zeros(3, 3, CV_64F), dist0(1, 8, CV_64F), dist1(1, 8, CV_64F);
    K0.at<double>(0, 0) = k0[0], K0.at<double>(1, 1) = k0[1], K0.at<double>(0, 2) = k0[2], K0.at<double>(1, 2) = k0[3], K0.at<double>(2, 2) = 1;
    K1.at<double>(0, 0) = k1[0], K1.at<double>(1, 1) = k1[1], K1.at<double>(0, 2) = k1[2], K1.at<double>(1, 2) = k1[3], K1.at<double>(2, 2) = 1;
    for (int i = 0; i < 8; ++i) dist0.at<double>(0, i) = d0[i], dist1.at<double>(0, i) = d1[i];
    poselib::ConfigPoseEstimation cfg;
    cfg.K0 = &K0, cfg.K1 = &K1, cfg.dist0_8 = &dist0, cfg.dist1_8 = &dist1;
    cfg.RobMethod = hdr[1] == 1 ? "LMEDS" : hdr[1] == 2 ? "ARRSAC" : "RANSAC";
    cfg.refineMethod_CorrPool = poselib::RefinePostAlg::PR_STEWENIUS | poselib::RefinePostAlg::PR_PSEUDOHUBER_WEIGHTS;
    cfg.th_pix_user = c[0];
    cfg.minStartAggInlRat = c[1];
    cfg.relInlRatThLast = c[2];
    cfg.relInlRatThNew = c[3];
    cfg.minInlierRatSkip = c[4];
    cfg.relMinInlierRatSkip = c[5];
    cfg.maxSkipPairs = (size_t)c[6];
    cfg.minInlierRatioReInit = c[7];
    cfg.minPtsDistance = (float)c[8];
    cfg.maxPoolCorrespondences = (size_t)c[9];
    cfg.minContStablePoses = (size_t)c[10];
    cfg.absThRankingStable = c[11];
    cfg.useRANSAC_fewMatches = c[12] != 0;
    cfg.minNormDistStable = c[13];
    cfg.raiseSkipCnt = (int)c[14];
    cfg.maxRat3DPtsFar = c[15];
    cfg.maxDist3DPtsZ = c[16];
    cfg.refineRTold = c[17] != 0;
    cfg.checkPoolPoseRobust = (size_t)c[18];
    cfg.refineRTold_CorrPool = c[19] != 0;
    cfg.autoTH = c[20] != 0;
    poselib::ConfigUSAC cfg_usac;
    cfg_usac.imgSize = cv::Size(1408, 1056);  // every keypoint of the test scenes lies inside (StereoRefine::checkPoolSize indexes an image-sized table)
    poselib::setRansacSeed(seed);
    poselib::StereoRefine sr(cfg);
    FILE *o = fopen(argv[2], "wb");
    for (int fr = 0; fr < hdr[0]; ++fr) {
        int32_t n;
        if (fread(&n, 4, 1, f) != 1) return 2;
        std::vector<float> a((size_t)n * 3), b((size_t)n * 3), dd((size_t)n);
        if (fread(a.data(), 4, a.size(), f) != a.size() || fread(b.data(), 4, b.size(), f) != b.size() ||
            fread(dd.data(), 4, dd.size(), f) != dd.size())
            return 2;
        std::vector<cv::KeyPoint> kp1((size_t)n), kp2((size_t)n);
        std::vector<cv::DMatch> m((size_t)n);
        for (int i = 0; i < n; ++i) {
            kp1[i].pt = cv::Point2f(a[3 * i], a[3 * i + 1]), kp1[i].response = a[3 * i + 2];
            kp2[i].pt = cv::Point2f(b[3 * i], b[3 * i + 1]), kp2[i].response = b[3 * i + 2];
            m[i].queryIdx = i, m[i].trainIdx = i, m[i].distance = dd[i];
        }
        cfg_usac.matches = &m, cfg_usac.keypoints1 = &kp1, cfg_usac.keypoints2 = &kp2;
        const int32_t rc = sr.addNewCorrespondences(m, kp1, kp2, cfg_usac);
        const int32_t st[9] = {rc, (int32_t)sr.nrInliersNew(), (int32_t)sr.nrCorrsNew(), (int32_t)sr.getCorrespondencePoolSize(),
                               (int32_t)sr.nrEstimations(), (int32_t)sr.skipCounter(), sr.poseIsStable ? 1 : 0,
                               sr.mostLikelyPose_stable ? 1 : 0, (int32_t)sr.poseHistorySize()};
        fwrite(st, 4, 9, o);
        dump(o, sr.E_new, 9);
        dump(o, sr.R_new, 9);
        dump(o, sr.t_new, 3);
        dump(o, sr.E_mostLikely, 9);
    }
    fclose(o);
    fclose(f);
    return 0;
}
