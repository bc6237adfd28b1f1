// pose_estim.h -- drop-in for the hot-path part of the reference's poselib/include/poselib/pose_estim.h:192-210.
#pragma once
#include <string>

#include "matchinglib_poselib/cv_compat.h"

#define PIX_MIN_GOOD_TH 0.8  // reference pose_estim.h:56

namespace poselib {

// Placeholder for the reference's ConfigUSAC (pose_estim.h:94-132); USAC is outside the hot path built here.
struct ConfigUSAC {};

// RANSAC seed control.  The reference seeds std::srand(std::time(nullptr)) in the estimator constructor
// (five-point-nister/modelest.cpp:58) and its setSeed() is never called on this path, so its results are time-seeded.
// Default here: the same (time-seeded).  setRansacSeed(s) fixes the glibc rand() stream for reproducible runs;
// clearRansacSeed() returns to time seeding.  Thread-local.
void setRansacSeed(unsigned seed);
void clearRansacSeed();

// poselib::estimateEssentialMat (pose_estim.h:204-210, pose_estim.cpp:857-890).  p1, p2: n x 2 CV_64F camera
// coordinates.  method "RANSAC" runs on the GPU (1000 iterations, confidence 0.999, `refine` = least-squares refit on
// the inliers); "LMEDS" runs on the GPU as well (2000 iterations, no refit; `threshold` unused, as in the reference).
// "USAC" / unknown method: prints the reference's message and calls exit(1), like the reference.
// "ARRSAC" (the reference's default argument) is not part of this library: it also exits(1) with a message.
bool estimateEssentialMat(cv::OutputArray E, cv::InputArray p1, cv::InputArray p2, const std::string &method = "ARRSAC",
                          double threshold = PIX_MIN_GOOD_TH, bool refine = true, cv::OutputArray mask = cv::noArray());

// poselib::getPoseTriangPts (pose_estim.h:192-200, pose_estim.cpp:913-946).  Returns the number of valid 3-D points,
// or -1 when R, t or Q is cv::noArray().  translatE = true: E is a translational essential matrix, R = I.
int getPoseTriangPts(cv::InputArray E, cv::InputArray p1, cv::InputArray p2, cv::OutputArray R, cv::OutputArray t,
                     cv::OutputArray Q, cv::InputOutputArray mask = cv::noArray(), const double dist = 50.0,
                     bool translatE = false);

// Convenience named in BASELINE.json: estimateEssentialMat(..., "RANSAC", ...) followed by getPoseTriangPts, the
// sequence the reference's README prescribes (README.md:497-521).
bool estimateRelativePose(cv::InputArray p1, cv::InputArray p2, cv::OutputArray E, cv::OutputArray R, cv::OutputArray t,
                          cv::OutputArray Q, cv::OutputArray mask, double threshold = PIX_MIN_GOOD_TH, bool refine = true,
                          double dist = 50.0);

}  // namespace poselib
