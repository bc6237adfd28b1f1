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

struct ConfigPoseEstimation {
    cv::Mat *dist0_8 = nullptr;  // 8 OpenCV-ordered distortion coefficients (CV_64F), null/empty = none
    cv::Mat *dist1_8 = nullptr;
    cv::Mat *K0 = nullptr;  // 3x3 CV_64F camera matrices
    cv::Mat *K1 = nullptr;
    double th_pix_user = 0.8;          // reference default, stereo_pose_refinement.h:108
    std::string RobMethod = "RANSAC";  // the reference default is "USAC" (not built); only "RANSAC" is accepted
    bool refineRTold = false;          // passed as `refine` to estimateEssentialMat (stereo_pose_refinement.cpp:1416)
    double maxDist3DPtsZ = 50.0;
    int verbose = 0;
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
