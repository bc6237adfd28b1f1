// matchinglib_matchers.h -- drop-in for the reference's matchinglib/include/matchinglib/matchinglib_matchers.h:61-64.
// Same name, arguments, defaults and return codes; the LINEAR (brute-force) matcher runs on the MI355X through
// libmlpl_hip.so (include/mlpl_c.h).  Every other matcher name returns -2 ("Matcher not supported"): those matchers are
// outside the hot path this library accelerates.
#pragma once
#include <string>
#include <vector>

#include "matchinglib_poselib/cv_compat.h"

namespace matchinglib {

// Return value: 0 ok, -1 wrong input data, -2 matcher not supported, -3 matching failed (< 2 matches left),
// -4 too few keypoints (< 15).  Throws cv::Exception when descriptors1.type() != descriptors2.type() (CV_Assert,
// reference matchers.cpp:119).
int getMatches(const std::vector<cv::KeyPoint> &keypoints1, const std::vector<cv::KeyPoint> &keypoints2,
               cv::Mat const &descriptors1, cv::Mat const &descriptors2, cv::Size imgSi,
               std::vector<cv::DMatch> &finalMatches, std::string const &matcher_name = "GMBSOF", bool VFCrefine = false,
               bool ratioTest = true, std::string const &descriptor_name = "", std::string idxPars_NMSLIB = "",
               std::string queryPars_NMSLIB = "", const size_t nr_threads = 0);

}  // namespace matchinglib
