// stereo_pose_refinement.h -- the hot-path slice of the reference's poselib::StereoRefine
// (poselib/include/poselib/stereo_pose_refinement.h:100-313, poselib/source/stereo_pose_refinement.cpp:416-478,
// 1272-1579).  What is kept: construction from ConfigPoseEstimation (K0, K1 mandatory), the pixel -> camera threshold
// (th = th_pix_user * 4/(sqrt(2)(fx0+fy0+fx1+fy1))), addNewCorrespondences() = gather matched keypoints ->
// ImgToCamCoordTrans + Remove_LensDist (pose_helper.cpp:1100-1109, 1169-1279; both on the GPU) -> robust estimation with the
// non-USAC branch (estimateEssentialMat(RobMethod, th, refineRTold) + getPoseTriangPts(maxDist3DPtsZ)) on the GPU.
// What is NOT built ("next"): the correspondence pool, pose history and
// stability logic, refinement/BA; every call is a fresh robust estimation (the reference's first-call path).
#pragma once
#include <string>
#include <vector>

#include "matchinglib_poselib/cv_compat.h"
#include "matchinglib_poselib/pose_estim.h"

namespace poselib {

// reference stereo_pose_refinement.h:100-176: every field, the reference's defaults (incl. RobMethod = "USAC", which this library
// does not build: StereoRefine reports -1 for it, see addNewCorrespondences).
struct ConfigPoseEstimation {
    ConfigPoseEstimation()
        : dist0_8(nullptr),
          dist1_8(nullptr),
          K0(nullptr),
          K1(nullptr),
          keypointType("FAST"),
          descriptorType("FREAK"),
          th_pix_user(0.8),
          autoTH(false),
          Halign(0),
          RobMethod("USAC"),
          refineMethod(poselib::RefinePostAlg::PR_NO_REFINEMENT),
          refineRTold(false),
          kneipInsteadBA(false),
          BART(0),
          refineMethod_CorrPool(poselib::RefinePostAlg::PR_STEWENIUS | poselib::RefinePostAlg::PR_PSEUDOHUBER_WEIGHTS),
          refineRTold_CorrPool(false),
          kneipInsteadBA_CorrPool(false),
          BART_CorrPool(0),
          verbose(7),
          minStartAggInlRat(0.2),
          relInlRatThLast(0.35),
          relInlRatThNew(0.20),
          minInlierRatSkip(0.38),
          relMinInlierRatSkip(0.7),
          maxSkipPairs(5),
          minInlierRatioReInit(0.6),
          minPtsDistance(3.f),
          maxPoolCorrespondences(30000),
          minContStablePoses(3),
          absThRankingStable(0.075),
          useRANSAC_fewMatches(false),
          checkPoolPoseRobust(3),
          minNormDistStable(0.5),
          raiseSkipCnt(0),
          maxRat3DPtsFar(0.5),
          maxDist3DPtsZ(50.0) {}

    cv::Mat *dist0_8;  // 8 OpenCV-ordered distortion coefficients (CV_64F) of the first / second camera
    cv::Mat *dist1_8;
    cv::Mat *K0;  // 3x3 CV_64F camera matrices
    cv::Mat *K1;
    std::string keypointType;
    std::string descriptorType;
    double th_pix_user;
    bool autoTH;
    int Halign;
    std::string RobMethod;  // USAC, RANSAC, ARRSAC, LMEDS
    int refineMethod;       // enum RefinePostAlg
    bool refineRTold;
    bool kneipInsteadBA;
    int BART;
    int refineMethod_CorrPool;
    bool refineRTold_CorrPool;
    bool kneipInsteadBA_CorrPool;
    int BART_CorrPool;
    int verbose;
    double minStartAggInlRat;
    double relInlRatThLast;
    double relInlRatThNew;
    double minInlierRatSkip;
    double relMinInlierRatSkip;
    size_t maxSkipPairs;
    double minInlierRatioReInit;
    float minPtsDistance;
    size_t maxPoolCorrespondences;
    size_t minContStablePoses;
    double absThRankingStable;
    bool useRANSAC_fewMatches;
    size_t checkPoolPoseRobust;
    double minNormDistStable;
    int raiseSkipCnt;
    double maxRat3DPtsFar;
    double maxDist3DPtsZ;
};

class StereoRefine {
   public:
    cv::Mat E_new, Q, R_new, t_new;
    cv::Mat E_mostLikely, R_mostLikely, t_mostLikely;  // == the newest pose (no history in this slice)
    bool poseIsStable = false, mostLikelyPose_stable = false;
    cv::Mat mask_E_new, mask_Q_new;
    size_t nr_inliers_new = 0, nr_corrs_new = 0;

    explicit StereoRefine(ConfigPoseEstimation cfg_pose_, bool verbose_ = false);
    void setNewParameters(ConfigPoseEstimation cfg_pose_);
    // 0 ok, -1 too few correspondences / bad configuration, -2 robust estimation failed (reference codes:
    // stereo_pose_refinement.cpp:411-414).
    int addNewCorrespondences(std::vector<cv::DMatch> matches, std::vector<cv::KeyPoint> kp1, std::vector<cv::KeyPoint> kp2,
                              const poselib::ConfigUSAC &cfg);
    size_t getCorrespondencePoolSize() { return 0; }
    double inlierThreshold() const { return th; }

   private:
    ConfigPoseEstimation cfg_pose;
    double pixToCamFact = 0, th = 0;
    bool verbose = false;
    void init();
};

}  // namespace poselib
