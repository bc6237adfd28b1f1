// stereo_pose_refinement.h -- drop-in for the reference's poselib::StereoRefine
// (poselib/include/poselib/stereo_pose_refinement.h:100-334, poselib/source/stereo_pose_refinement.cpp).
//
// Built here: the whole multi-frame state machine of addNewCorrespondences() as the reference runs it for the estimators of the hot
// path -- first-call robust initialisation; on later frames the strict getInliers() test against the last pose (GPU), the decision tree
// on the inlier ratios (re-estimate / restore the last pose / re-initialise, :486-560), the correspondence pool with its minPtsDistance
// filter and quality comparison (:2107-2316, :2450-2548), robust re-estimation on the pool (:1075-1128), pose history, the
// near-to-mean pose rating and the stability flags (:2817-3298), maxSkipPairs handling (:3300-3317).
// Every robust estimation and every error evaluation runs on the MI355X (estimateEssentialMat / getPoseTriangPts / mlpl_get_inliers_strict).
//
// Not built (outside the hot path): homography alignment (Halign), the linear refinement solvers
// (refineMethod / refineMethod_CorrPool: Nister/Stewenius/Kneip/8pt with weights) and bundle adjustment (BART).  refineRTold IS built: the
// estimator's own refinement step plus poselib::robustEssentialRefine on the inliers (:1460-1474), on the device.  Consequences, all
// reported once on std::cout:
//   * RobMethod is "USAC" (the reference's default; estimateEssentialOrPoseUSAC with the ConfigUSAC handed to addNewCorrespondences, a
//     degenerate pair -> -2, :1355-1413), "RANSAC", "LMEDS" or "ARRSAC"; Halign makes addNewCorrespondences() return -1; autoTH (ARRSAC
//     with poselib::AutoThEpi's threshold estimation, :1330-1342) is built;
//   * refinement / BA options are ignored;
//   * between robust estimations the pool pose is REFINED (checkPoolPoseRobust != 1, the reference's schedule :680-716) only with
//     refineRTold_CorrPool, the refinement that is built (refinePoseFromPool :1767-2084 with robustEssentialRefine on the device);
//     with the linear solvers selected the pool is re-estimated robustly on every frame (the reference's checkPoolPoseRobust = 1);
//   * thinning an over-full pool follows checkPoolSize (:2550-2816): a density image of the pool's left keypoints, elliptic dilate /
//     erode with the reference's border handling, then the weights;
//   * the radius search over the pool returns neighbours by ascending squared distance, as the reference's
//     keyPointTreeInterface::radiusSearch does (nanoflannInterface.cpp:269-300); equal distances come by index.
#pragma once
#include <memory>
#include <string>
#include <vector>

#include "matchinglib_poselib/cv_compat.h"
#include "matchinglib_poselib/pose_estim.h"

namespace poselib {

// reference stereo_pose_refinement.h:100-176: every field, the reference's defaults (incl. RobMethod = "USAC").
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
    // reference stereo_pose_refinement.h:283-291
    cv::Mat E_new;         // newest essential matrix
    cv::Mat Q;             // 3-D points of the latest estimation
    cv::Mat R_new;         // newest rotation
    cv::Mat t_new;         // newest translation (unit norm)
    cv::Mat E_mostLikely;  // pose of the history that is nearest to the centre of gravity of all stored poses
    cv::Mat R_mostLikely;
    cv::Mat t_mostLikely;
    bool poseIsStable = false;
    bool mostLikelyPose_stable = false;

    explicit StereoRefine(ConfigPoseEstimation cfg_pose_, bool verbose_ = false);
    ~StereoRefine();
    void setNewParameters(ConfigPoseEstimation cfg_pose_);
    // 0 ok; -1 robust estimation failed / too few matches / unsupported configuration; -2 the pool had to be re-initialised;
    // -3 too low an inlier ratio after the estimation on the pool (reference codes: stereo_pose_refinement.cpp:411-414).
    int addNewCorrespondences(std::vector<cv::DMatch> matches, std::vector<cv::KeyPoint> kp1, std::vector<cv::KeyPoint> kp2,
                              const poselib::ConfigUSAC &cfg);
    size_t getCorrespondencePoolSize();

    // Read-only views of the private state the reference keeps (same names), for tests and diagnostics.
    double inlierThreshold() const;
    size_t nrInliersNew() const;
    size_t nrCorrsNew() const;
    size_t nrEstimations() const;
    size_t skipCounter() const;
    size_t poseHistorySize() const;
    const cv::Mat &maskENew() const;

   private:
    struct Impl;
    std::unique_ptr<Impl> d;
};

}  // namespace poselib
