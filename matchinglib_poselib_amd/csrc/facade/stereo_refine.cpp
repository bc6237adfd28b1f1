// stereo_refine.cpp -- poselib::StereoRefine: the multi-frame state machine of the reference
// (poselib/source/stereo_pose_refinement.cpp) as host C++ over the GPU entry points of libmlpl_hip.so.  See the header for what is
// built and what is not.  Citations below are lines of the reference's stereo_pose_refinement.cpp unless another file is named.
#include <algorithm>
#include <array>
#include <climits>
#include <cmath>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <list>
#include <unordered_map>
#include <unordered_set>

#include "facade_internal.h"
#include "matchinglib_poselib/stereo_pose_refinement.h"
#include "mlpl_c.h"

namespace poselib {

namespace {

inline bool nearZero(double d) { return d < 1e-3 && d > -1e-3; }  // pose_helper.h:82-87

struct StatVals {
    double medErr = 0, arithErr = 0, arithStd = 0, medStd = 0;
};

// getStatsfromVec(vals, &stats, false, false) (pose_helper.cpp:358-413)
StatVals statsFromVec(const std::vector<double> &vals) {
    StatVals st;
    if (vals.empty()) return st;
    const int n = (int)vals.size();
    std::vector<double> v(vals);
    std::sort(v.begin(), v.end());
    st.medErr = (n % 2) ? v[(n - 1) / 2] : (v[n / 2] + v[n / 2 - 1]) / 2.0;
    double sum = 0, sum2 = 0;
    std::vector<double> mad;
    mad.reserve(n);
    for (int i = 0; i < n; ++i) {
        sum += v[i];
        sum2 += v[i] * v[i];
        mad.push_back(std::abs(v[i] - st.medErr));
    }
    st.arithErr = sum / (double)n;
    std::sort(mad.begin(), mad.end());
    st.medStd = (n % 2) ? 1.4826 * mad[(n - 1) / 2] : 1.4826 * (mad[n / 2] + mad[n / 2 - 1]) / 2.0;
    const double hlp = sum2 - (double)n * st.arithErr * st.arithErr;
    st.arithStd = std::sqrt(hlp / ((double)n - 1.0));
    return st;
}

// getSampsonL2Error (pose_helper.cpp:3011-3020) for one correspondence, E row-major
double sampsonL2(const double *E, double x1, double y1, double x2, double y2) {
    const double x2E0 = x2 * E[0] + y2 * E[3] + E[6], x2E1 = x2 * E[1] + y2 * E[4] + E[7], x2E2 = x2 * E[2] + y2 * E[5] + E[8];
    const double r = x2E0 * x1 + x2E1 * y1 + x2E2;
    const double rx = E[0] * x1 + E[1] * y1 + E[2], ry = E[3] * x1 + E[4] * y1 + E[5];
    return r * r / (x2E0 * x2E0 + x2E1 * x2E1 + rx * rx + ry * ry);
}

inline double weightInv(double value, double max_value) { return 1.0 - value / max_value; }  // getWeightingValuesInv (min = 0)
inline double weight(double value, double max_value) { return value / max_value; }           // getWeightingValues

struct CoordinateProps {  // poselib/include/poselib/stereo_pose_types.h
    cv::Point2f pt1, pt2;
    float descrDist = 0;
    float keyPResponses[2] = {0, 0};
    std::vector<double> SampsonErrors;
    double meanSampsonError = 0;
    double Q[3] = {0, 0, 0};
    bool Q_tooFar = false;
    size_t age = 0, nrFound = 0, ptIdx = 0, poolIdx = 0;
};

struct CoordinatePropsNew {
    cv::Point2f pt1, pt2;
    float descrDist;
    float keyPResponses[2];
    double sampsonError;
};

struct PoseHist {
    double E[9], R[9], t[3];
};

void copy9(const cv::Mat &m, double *o) {
    for (int i = 0; i < 9; ++i) o[i] = m.at<double>(i / 3, i % 3);
}
cv::Mat mat33(const double *v) {
    cv::Mat m(3, 3, CV_64F);
    for (int i = 0; i < 9; ++i) m.at<double>(i / 3, i % 3) = v[i];
    return m;
}
cv::Mat mat31(const double *v) {
    cv::Mat m(3, 1, CV_64F);
    for (int i = 0; i < 3; ++i) m.at<double>(i, 0) = v[i];
    return m;
}

}  // namespace

struct StereoRefine::Impl {
    StereoRefine *self = nullptr;
    ConfigPoseEstimation cfg_pose;
    ConfigUSAC cfg_usac;
    double pixToCamFact = 0;
    size_t nrEstimation = 0, skipCount = 0;
    std::vector<cv::Point2f> points1new, points2new;  // camera coordinates (float), newest pair
    std::vector<double> p1new, p2new;                 // points1newMat / points2newMat: n x 2 doubles
    std::vector<double> p1new_tmp, p2new_tmp;         // ..._tmp: also the correspondences filtered out later
    std::vector<double> p1Cam, p2Cam;                 // pool coordinates (points1Cam / points2Cam)
    size_t newAddedPoolCorrs = 0;
    double th = 0, th2 = 0;
    float descrDist_max = 0, keyPRespons_max = 0;
    std::vector<uint8_t> mask_E_new, mask_Q_new;
    cv::Mat mask_E_new_mat;
    bool have_Q = false;
    std::vector<double> Qv;  // n x 3
    size_t nr_inliers_new = 0, nr_corrs_new = 0;
    bool maxPoolSizeReached = false;
    size_t checkPoolPoseRobust_tmp = 0, initNumberInliers = 0, nrConsecStablePoses = 0, maxSkipPairsNew = 0;
    std::list<CoordinateProps> correspondencePool;
    std::unordered_map<size_t, std::list<CoordinateProps>::iterator> correspondencePoolIdx;
    size_t corrIdx = 0;
    bool tree = false;  // kdTreeLeft exists
    bool verbose = false;
    double E_[9] = {0}, R_[9] = {0}, t_[3] = {0};
    bool have_pose = false;
    std::vector<PoseHist> pose_history;
    std::vector<double> pose_history_rating;
    std::vector<size_t> mostLikelyPoseIdxs;
    std::vector<double> inlier_ratio_history;
    std::vector<StatVals> errorStatistic_history;
    size_t nr_Q_tooFar = 0, nr_Qs = 0;
    // function-local statics of the reference (addNewCorrespondences: failed_refinements, nr_since_robust; checkPoseStability: nr_tries)
    size_t nr_since_robust = 0, nr_tries = 0;
    bool noticed = false;

    // ---- uniform grid over pt1 of the pool: the radius search of filterNewCorrespondences ----
    std::unordered_map<long long, std::vector<size_t>> grid;
    float cell = 4.f;
    long long cellKey(float x, float y) const {
        return ((long long)std::floor(x / cell) << 32) ^ ((long long)std::floor(y / cell) & 0xFFFFFFFFll);
    }
    void gridAdd(size_t idx, const cv::Point2f &p) { grid[cellKey(p.x, p.y)].push_back(idx); }
    void gridRemove(size_t idx, const cv::Point2f &p) {
        auto it = grid.find(cellKey(p.x, p.y));
        if (it == grid.end()) return;
        auto &v = it->second;
        v.erase(std::remove(v.begin(), v.end(), idx), v.end());
        if (v.empty()) grid.erase(it);
    }
    // keyPointTreeInterface::radiusSearch (nanoflannInterface.cpp:269-300): squared distances, here sorted ascending
    size_t radiusSearch(const cv::Point2f &q, float radius, std::vector<std::pair<size_t, float>> &result) {
        result.clear();
        const float r2 = radius * radius;
        const int reach = (int)std::ceil(radius / cell);
        const long long cx = (long long)std::floor(q.x / cell), cy = (long long)std::floor(q.y / cell);
        for (long long ix = cx - reach; ix <= cx + reach; ++ix)
            for (long long iy = cy - reach; iy <= cy + reach; ++iy) {
                auto it = grid.find((ix << 32) ^ (iy & 0xFFFFFFFFll));
                if (it == grid.end()) continue;
                for (size_t idx : it->second) {
                    const CoordinateProps &c = *correspondencePoolIdx[idx];
                    const float dx = c.pt1.x - q.x, dy = c.pt1.y - q.y;
                    const float d2 = dx * dx + dy * dy;
                    if (d2 < r2) result.emplace_back(idx, d2);
                }
            }
        std::sort(result.begin(), result.end(), [](const std::pair<size_t, float> &a, const std::pair<size_t, float> &b) {
            return a.second < b.second || (a.second == b.second && a.first < b.first);
        });
        return result.size();
    }

    void notice(const char *msg) {
        if (!noticed) std::cout << "StereoRefine (MI355X hot-path library): " << msg << std::endl;
    }

    void publish() {
        if (have_pose) {
            self->E_new = mat33(E_);
            self->R_new = mat33(R_);
            self->t_new = mat31(t_);
        }
        if (have_Q) {
            const int n = (int)(Qv.size() / 3);
            self->Q = cv::Mat(n, 3, CV_64F);
            for (int i = 0; i < n; ++i)
                for (int c = 0; c < 3; ++c) self->Q.at<double>(i, c) = Qv[(size_t)i * 3 + c];
        } else {
            self->Q = cv::Mat();
        }
        mask_E_new_mat = cv::Mat(1, (int)mask_E_new.size(), CV_8U);
        if (!mask_E_new.empty()) std::memcpy(mask_E_new_mat.ptr<uint8_t>(0), mask_E_new.data(), mask_E_new.size());
    }

    void init();
    void checkInputParamters();
    void clearHistoryAndPool();
    size_t getInliers(const double *E, const std::vector<double> &a, const std::vector<double> &b, std::vector<uint8_t> &mask,
                      std::vector<double> &error);
    void reprojErrors(const std::vector<double> &a, const std::vector<double> &b, const double *E, std::vector<double> &error);
    int robustPoseEstimation();
    int robustInitialization(double &inlier_ratio, std::vector<cv::DMatch> &matches, std::vector<cv::KeyPoint> &kp1,
                             std::vector<cv::KeyPoint> &kp2);
    bool initDataAfterReinitialization(double &inlier_ratio, std::vector<cv::DMatch> &matches, std::vector<cv::KeyPoint> &kp1,
                                       std::vector<cv::KeyPoint> &kp2);
    bool reinitializeSystem(double &inlier_ratio, std::vector<cv::DMatch> &matches, std::vector<cv::KeyPoint> &kp1,
                            std::vector<cv::KeyPoint> &kp2);
    int robustEstimationOnPool();
    int refinePoseFromPool();
    size_t failed_refinements = 0;  // function-local static in the reference (:678): process-wide there, per object here
    int addCorrespondencesToPool(const std::vector<cv::DMatch> &matches, const std::vector<cv::KeyPoint> &kp1,
                                 const std::vector<cv::KeyPoint> &kp2);
    int filterNewCorrespondences(std::vector<cv::DMatch> &matches, const std::vector<cv::KeyPoint> &kp1,
                                 const std::vector<cv::KeyPoint> &kp2, const std::vector<double> &error);
    bool compareCorrespondences(const CoordinatePropsNew &n, const CoordinateProps &o);
    int poolCorrespondenceDelete(std::vector<size_t> delete_list);
    int checkPoolSize(long long maxPoolSize);
    double computeCorrespondenceWeight(double error, double descrDist, double resp1, double resp2, bool z3DtooFar = false,
                                       double zValue3D = 0);
    int getNearToMeanPose();
    int checkPoseStability();
    void updateMaxSkipPairs();
    int addNewCorrespondences(std::vector<cv::DMatch> &matches, std::vector<cv::KeyPoint> &kp1, std::vector<cv::KeyPoint> &kp2,
                              const ConfigUSAC &cfg);
};

// ---- construction, parameters (:87-400) ---------------------------------------------------------------------------------------------

void StereoRefine::Impl::init() {
    CV_Assert(cfg_pose.K0 != nullptr && cfg_pose.K1 != nullptr && cfg_pose.dist0_8 != nullptr && cfg_pose.dist1_8 != nullptr);
    pixToCamFact = 4.0 / (std::sqrt(2.0) * (cfg_pose.K0->at<double>(0, 0) + cfg_pose.K0->at<double>(1, 1) +
                                            cfg_pose.K1->at<double>(0, 0) + cfg_pose.K1->at<double>(1, 1)));
    th = cfg_pose.th_pix_user * pixToCamFact;
    th2 = th * th;
    checkInputParamters();
    maxSkipPairsNew = cfg_pose.maxSkipPairs;
}

void StereoRefine::Impl::checkInputParamters() {  // :187-400 (values only; the reference also prints an explanation per clamp)
    ConfigPoseEstimation &c = cfg_pose;
    if (((c.refineMethod_CorrPool & 0xF) == PR_NO_REFINEMENT) && !c.refineRTold)
        c.refineMethod_CorrPool = PR_STEWENIUS | PR_PSEUDOHUBER_WEIGHTS;
    if (c.kneipInsteadBA && ((c.refineMethod & 0xF) != PR_KNEIP)) c.refineMethod = (c.refineMethod & 0xF0) | PR_KNEIP;
    if (c.kneipInsteadBA_CorrPool && ((c.refineMethod_CorrPool & 0xF) != PR_KNEIP))
        c.refineMethod_CorrPool = (c.refineMethod_CorrPool & 0xF0) | PR_KNEIP;
    if (c.minStartAggInlRat < 0.075) c.minStartAggInlRat = 0.1;
    else if (c.minStartAggInlRat > 0.75) c.minStartAggInlRat = 0.75;
    if (c.relInlRatThLast > 0.75) c.relInlRatThLast = 0.6;
    else if (c.relInlRatThLast < 0.01) c.relInlRatThLast = 0.1;
    if (c.relInlRatThNew < 0.04) c.relInlRatThNew = 0.04;
    else if (c.relInlRatThNew > 0.55) c.relInlRatThNew = 0.35;
    if (c.minInlierRatSkip > 0.95) c.minInlierRatSkip = 0.95;
    else if (c.minInlierRatSkip < 0.01) c.minInlierRatSkip = 0.1;
    if (c.relMinInlierRatSkip < 0.01) c.relMinInlierRatSkip = 0.1;
    else if (c.relMinInlierRatSkip > 1.0) c.relMinInlierRatSkip = 1.0;
    if (c.maxSkipPairs == 0) c.maxSkipPairs = 1;
    else if (c.maxSkipPairs > 200) c.maxSkipPairs = 200;
    if (c.minInlierRatioReInit <= c.minInlierRatSkip) c.minInlierRatioReInit = c.minInlierRatSkip + 0.05;
    if (c.minInlierRatioReInit > 0.8) c.minInlierRatioReInit = 0.8;
    else if (c.minInlierRatioReInit < 0.15) c.minInlierRatioReInit = 0.15;
    if (c.minPtsDistance < 1.5f) c.minPtsDistance = 1.5f;
    if (c.maxPoolCorrespondences > (size_t)INT_MAX) c.maxPoolCorrespondences = (size_t)INT_MAX;
    if (c.minContStablePoses <= 2) c.minContStablePoses = 3;
    if (c.absThRankingStable < 0.01) c.absThRankingStable = 0.01;
    else if (c.absThRankingStable > 0.9) c.absThRankingStable = 0.6;
    // The pool is REFINED between robust estimations (checkPoolPoseRobust != 1) only through refineRTold_CorrPool, the refinement that is
    // built (robustEssentialRefine on the device); with the linear solvers (refineMethod_CorrPool) selected instead, the pool is
    // re-estimated robustly on every frame (the reference's checkPoolPoseRobust = 1).
    if (c.checkPoolPoseRobust != 1 && !c.refineRTold_CorrPool) {
        notice("the linear refinement solvers of the correspondence pool are not built; the pool is re-estimated robustly on every frame "
               "(the reference's checkPoolPoseRobust = 1); set refineRTold_CorrPool for the refinement that is built");
        c.checkPoolPoseRobust = 1;
    }
    if ((c.refineMethod & 0xF) != PR_NO_REFINEMENT || c.refineRTold || c.kneipInsteadBA || c.BART || c.BART_CorrPool)
        notice("refinement after the robust estimation (refineMethod, refineRTold beyond the estimator's own refit, kneipInsteadBA, "
               "BART) is not built and is skipped");
    noticed = true;
}

void StereoRefine::Impl::clearHistoryAndPool() {  // :1038-1064
    p1Cam.clear();
    p2Cam.clear();
    correspondencePool.clear();
    correspondencePoolIdx.clear();
    grid.clear();
    tree = false;
    corrIdx = 0;
    nrEstimation = 0;
    skipCount = 0;
    pose_history.clear();
    pose_history_rating.clear();
    inlier_ratio_history.clear();
    errorStatistic_history.clear();
    mostLikelyPoseIdxs.clear();
    maxPoolSizeReached = false;
    self->poseIsStable = false;
    self->mostLikelyPose_stable = false;
    nrConsecStablePoses = 0;
    maxSkipPairsNew = cfg_pose.maxSkipPairs;
    nr_Q_tooFar = 0;
    nr_Qs = 0;
}

// ---- GPU-backed primitives ---------------------------------------------------------------------------------------------------------

// computeReprojError2 (pose_helper.cpp:639-664) + getInlierMask (:3030-3045): double errors, STRICT err < th^2
size_t StereoRefine::Impl::getInliers(const double *E, const std::vector<double> &a, const std::vector<double> &b,
                                      std::vector<uint8_t> &mask, std::vector<double> &error) {
    const int n = (int)(a.size() / 2);
    error.assign((size_t)n, 0.0);
    mask.assign((size_t)n, 0);
    if (n == 0) return 0;
    const int cnt = mlpl_get_inliers_strict(mlpl_facade_default_ctx(), a.data(), b.data(), n, E, th2, error.data(), mask.data());
    if (cnt < 0) throw cv::Exception(std::string("mlpl_get_inliers_strict: ") + mlpl_last_error());
    return (size_t)cnt;
}

void StereoRefine::Impl::reprojErrors(const std::vector<double> &a, const std::vector<double> &b, const double *E,
                                      std::vector<double> &error) {
    std::vector<uint8_t> m;
    getInliers(E, a, b, m, error);
}

// robustPoseEstimation (:1272-1760), the branches of the estimators built here: optional switch to RANSAC for < 100 matches,
// estimateEssentialMat(RobMethod, th, refineRTold), getPoseTriangPts(maxDist3DPtsZ), t normalised.
int StereoRefine::Impl::robustPoseEstimation() {
    have_Q = false;
    Qv.clear();
    std::string method = cfg_pose.RobMethod;
    if (cfg_pose.Halign) {
        std::cout << "StereoRefine (MI355X hot-path library): Halign is not built." << std::endl;
        return -1;
    }
    const int n = (int)(p1new.size() / 2);
    bool autoTH = cfg_pose.autoTH;
    // :1295-1323: below 100 matches RANSAC replaces the configured estimator (and the automatic threshold) for this one estimation
    if (cfg_pose.useRANSAC_fewMatches && n < 100 && (method != "RANSAC" || autoTH)) method = "RANSAC", autoTH = false;
    cv::Mat P1(n, 2, CV_64F, p1new.data()), P2(n, 2, CV_64F, p2new.data());
    cv::Mat E, mask;
    if (autoTH) {
        // :1330-1342: ARRSAC with the threshold estimated from its own error statistics; `th` keeps the estimate for the frames to come
        // (th2, the squared threshold of the strict inlier test, is NOT updated -- as in the reference, :159-160)
        int inlierPoints = 0;
        AutoThEpi Eautoth(pixToCamFact);
        if (Eautoth.estimateEVarTH(P1, P2, E, mask, &th, &inlierPoints) != 0) {
            std::cout << "Estimation of essential matrix using automatic threshold estimation and ARRSAC failed!" << std::endl;
            return -1;
        }
        std::cout << "Estimated threshold: " << th / pixToCamFact << " pixels" << std::endl;
    } else if (method == "USAC") {
        // :1355-1413: the harness default.  cfg_usac.matches / keypoints describe the correspondences the estimation runs on (the new
        // pair, or the pool: robustEstimationOnPool)
        bool isDegenerate = false;
        cv::Mat R_degenerate, inliers_degenerate_R;
        if (estimateEssentialOrPoseUSAC(P1, P2, E, th, cfg_usac, isDegenerate, mask, R_degenerate, inliers_degenerate_R, cv::noArray(),
                                        cv::noArray(), verbose) != 0) {
            std::cout << "Estimation of essential matrix using USAC failed!" << std::endl;
            return -1;
        }
        if (isDegenerate) return -2;  // :1400-1411: "Camera configuration is degenerate and, thus, rotation only. Skipping further calculations!"
    } else if (!estimateEssentialMat(E, P1, P2, method, th, cfg_pose.refineRTold, mask)) {
        std::cout << "Estimation of essential matrix using " << method << " failed!" << std::endl;
        return -1;
    }
    mask_E_new.assign(mask.ptr<uint8_t>(0), mask.ptr<uint8_t>(0) + n);
    nr_inliers_new = 0;
    for (uint8_t v : mask_E_new) nr_inliers_new += v != 0;
    if (cfg_pose.refineRTold)  // :1460-1474: the "old" robust refinement on the inliers, threshold th / 10; the mask stays
        robustEssentialRefine(P1, P2, E, E, th / 10.0, 0, true, nullptr, nullptr, cv::noArray(), mask, 0);
    cv::Mat R, t, Q3;
    if (getPoseTriangPts(E, P1, P2, R, t, Q3, mask, cfg_pose.maxDist3DPtsZ) <= 0) {  // :1557
        std::cout << "Unable to triangulate 3D points" << std::endl;
        return -1;
    }
    copy9(E, E_);
    copy9(R, R_);
    double nrm = 0;
    for (int i = 0; i < 3; ++i) nrm += t.at<double>(i, 0) * t.at<double>(i, 0);
    nrm = std::sqrt(nrm);
    for (int i = 0; i < 3; ++i) t_[i] = t.at<double>(i, 0) / nrm;  // :1727-1728
    have_pose = true;
    mask_Q_new.assign(mask.ptr<uint8_t>(0), mask.ptr<uint8_t>(0) + n);
    Qv.resize((size_t)n * 3);
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) Qv[(size_t)i * 3 + c] = Q3.at<double>(i, c);
    have_Q = true;
    return 0;
}

int StereoRefine::Impl::robustInitialization(double &inlier_ratio, std::vector<cv::DMatch> &matches, std::vector<cv::KeyPoint> &kp1,
                                             std::vector<cv::KeyPoint> &kp2) {  // :968-989
    checkPoolPoseRobust_tmp = cfg_pose.checkPoolPoseRobust;
    if (robustPoseEstimation()) return -1;
    initNumberInliers = nr_inliers_new;
    inlier_ratio = (double)nr_inliers_new / (double)nr_corrs_new;
    if (inlier_ratio < cfg_pose.minStartAggInlRat) {
        std::cout << "Inlier ratio too small! Skipping aggregation of correspondences! "
                     "The output pose of this and the next iteration will be like in the mono camera case!"
                  << std::endl;
        return -3;
    }
    if (!initDataAfterReinitialization(inlier_ratio, matches, kp1, kp2)) return -2;
    return 0;
}

bool StereoRefine::Impl::initDataAfterReinitialization(double &inlier_ratio, std::vector<cv::DMatch> &matches,
                                                       std::vector<cv::KeyPoint> &kp1, std::vector<cv::KeyPoint> &kp2) {  // :1001-1012
    if (addCorrespondencesToPool(matches, kp1, kp2)) return false;
    PoseHist ph;
    std::memcpy(ph.E, E_, 72);
    std::memcpy(ph.R, R_, 72);
    std::memcpy(ph.t, t_, 24);
    pose_history.push_back(ph);
    inlier_ratio_history.push_back(inlier_ratio);
    nrEstimation++;
    return true;
}

bool StereoRefine::Impl::reinitializeSystem(double &inlier_ratio, std::vector<cv::DMatch> &matches, std::vector<cv::KeyPoint> &kp1,
                                            std::vector<cv::KeyPoint> &kp2) {  // :1025-1032
    clearHistoryAndPool();
    return initDataAfterReinitialization(inlier_ratio, matches, kp1, kp2);
}

int StereoRefine::Impl::robustEstimationOnPool() {  // :1075-1128: the robust estimator runs on the pool coordinates
    // USAC's PROSAC order and SPRT start values come from matches / keypoints: rebuilt from the pool (:1080-1102), restored afterwards
    const ConfigUSAC saved = cfg_usac;
    const size_t ps = correspondencePool.size();
    std::vector<cv::DMatch> pool_matches(ps);
    std::vector<cv::KeyPoint> kp1_tmp(ps), kp2_tmp(ps);
    if (cfg_pose.RobMethod == "USAC") {
        size_t i = 0;
        for (const CoordinateProps &c : correspondencePool) {
            pool_matches[i].queryIdx = pool_matches[i].trainIdx = (int)i, pool_matches[i].imgIdx = -1, pool_matches[i].distance = c.descrDist;
            kp1_tmp[i].pt = c.pt1, kp1_tmp[i].size = 10.f, kp1_tmp[i].angle = -1.f, kp1_tmp[i].response = c.keyPResponses[0];
            kp2_tmp[i].pt = c.pt2, kp2_tmp[i].size = 10.f, kp2_tmp[i].angle = -1.f, kp2_tmp[i].response = c.keyPResponses[1];
            ++i;
        }
        cfg_usac.matches = &pool_matches, cfg_usac.keypoints1 = &kp1_tmp, cfg_usac.keypoints2 = &kp2_tmp;
        cfg_usac.nrMatchesVfcFiltered = ps > UINT_MAX ? UINT_MAX : (unsigned int)ps;
    }
    std::swap(p1Cam, p1new);
    std::swap(p2Cam, p2new);
    const int rc = robustPoseEstimation();
    std::swap(p1Cam, p1new);
    std::swap(p2Cam, p2new);
    cfg_usac = saved;
    return rc ? -1 : 0;
}

// refinePoseFromPool (:1767-2084) for refineRTold_CorrPool: robustEssentialRefine of the current E on ALL pool correspondences (th / 10),
// R, t and the 3-D points from the refined matrix, E rebuilt from them (getEfromRT, pose_helper.cpp:785-788).  No correspondence is
// marked as an outlier on this path.
int StereoRefine::Impl::refinePoseFromPool() {
    const int n = (int)(p1Cam.size() / 2);
    nr_inliers_new = (size_t)n;
    have_Q = false;
    Qv.clear();
    cv::Mat P1(n, 2, CV_64F, p1Cam.data()), P2(n, 2, CV_64F, p2Cam.data());
    cv::Mat E(3, 3, CV_64F), mask(1, n, CV_8U);
    for (int i = 0; i < 9; ++i) E.at<double>(i / 3, i % 3) = E_[i];
    std::memset(mask.ptr<uint8_t>(0), 1, (size_t)n);
    robustEssentialRefine(P1, P2, E, E, th / 10.0, 0, true, nullptr, nullptr, cv::noArray(), mask, 0);
    mask_E_new.assign((size_t)n, 1);
    cv::Mat R, t, Q3;
    if (getPoseTriangPts(E, P1, P2, R, t, Q3, mask, cfg_pose.maxDist3DPtsZ) <= 0) {
        std::cout << "No 3D points left after triangulation!" << std::endl;
        return -1;
    }
    double nrm = 0;
    for (int i = 0; i < 3; ++i) nrm += t.at<double>(i, 0) * t.at<double>(i, 0);
    nrm = std::sqrt(nrm);
    double tn[3];
    for (int i = 0; i < 3; ++i) tn[i] = t.at<double>(i, 0) / nrm;
    const double S[9] = {0, -tn[2], tn[1], tn[2], 0, -tn[0], -tn[1], tn[0], 0};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double v = 0;
            for (int k = 0; k < 3; ++k) v += S[r * 3 + k] * R.at<double>(k, c);
            E_[r * 3 + c] = v;
        }
    copy9(R, R_);
    for (int i = 0; i < 3; ++i) t_[i] = tn[i];
    mask_Q_new.assign(mask.ptr<uint8_t>(0), mask.ptr<uint8_t>(0) + n);
    Qv.resize((size_t)n * 3);
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < 3; ++c) Qv[(size_t)i * 3 + c] = Q3.at<double>(i, c);
    have_Q = true;
    return 0;
}

// ---- the pool (:1143-1266, :2107-2548) ------------------------------------------------------------------------------------------------

int StereoRefine::Impl::addCorrespondencesToPool(const std::vector<cv::DMatch> &matches, const std::vector<cv::KeyPoint> &kp1,
                                                 const std::vector<cv::KeyPoint> &kp2) {
    std::vector<double> errors;
    size_t nrEntries = 0;
    bool isInitMat = true;
    newAddedPoolCorrs = 0;
    if (!p1Cam.empty()) {
        for (auto &it : correspondencePool) it.age++;
        isInitMat = false;
        nrEntries = p1Cam.size() / 2;
    }
    tree = true;
    for (size_t i = 0, count = nrEntries; i < nr_corrs_new; i++) {
        if (!mask_E_new[i]) continue;
        CoordinateProps tmp;
        p1Cam.push_back(p1new[2 * i]);
        p1Cam.push_back(p1new[2 * i + 1]);
        p2Cam.push_back(p2new[2 * i]);
        p2Cam.push_back(p2new[2 * i + 1]);
        tmp.age = 1;
        tmp.descrDist = matches[i].distance;
        if (descrDist_max < tmp.descrDist) descrDist_max = tmp.descrDist;
        tmp.keyPResponses[0] = kp1[(size_t)matches[i].queryIdx].response;
        if (keyPRespons_max < tmp.keyPResponses[0]) keyPRespons_max = tmp.keyPResponses[0];
        tmp.keyPResponses[1] = kp2[(size_t)matches[i].trainIdx].response;
        if (keyPRespons_max < tmp.keyPResponses[1]) keyPRespons_max = tmp.keyPResponses[1];
        if (isInitMat) {
            tmp.meanSampsonError = sampsonL2(E_, p1new[2 * i], p1new[2 * i + 1], p2new[2 * i], p2new[2 * i + 1]);
            tmp.SampsonErrors.push_back(tmp.meanSampsonError);
            errors.push_back(tmp.meanSampsonError);
            if (have_Q) {
                for (int c = 0; c < 3; ++c) tmp.Q[c] = Qv[i * 3 + c];
                tmp.Q_tooFar = !mask_Q_new[i];
                nr_Qs++;
                if (tmp.Q_tooFar) nr_Q_tooFar++;
            }
        }
        tmp.nrFound = 1;
        tmp.pt1 = kp1[(size_t)matches[i].queryIdx].pt;
        tmp.pt2 = kp2[(size_t)matches[i].trainIdx].pt;
        tmp.ptIdx = count;
        tmp.poolIdx = corrIdx;
        correspondencePool.push_back(tmp);
        correspondencePoolIdx.insert({corrIdx, --correspondencePool.end()});
        gridAdd(corrIdx, tmp.pt1);
        newAddedPoolCorrs++;
        corrIdx++;
        count++;
    }
    if (isInitMat) errorStatistic_history.push_back(statsFromVec(errors));
    if ((double)correspondencePool.size() / (double)corrIdx < 0.5) {  // :1233-1259: re-number the pool when half of the indices are dead
        corrIdx = 0;
        correspondencePoolIdx.clear();
        grid.clear();
        for (auto it = correspondencePool.begin(); it != correspondencePool.end(); ++it) {
            it->poolIdx = corrIdx;
            correspondencePoolIdx.insert({corrIdx, it});
            gridAdd(corrIdx, it->pt1);
            corrIdx++;
        }
    }
    return 0;
}

double StereoRefine::Impl::computeCorrespondenceWeight(double error, double descrDist, double resp1, double resp2, bool z3DtooFar,
                                                       double zValue3D) {  // :2514-2542
    const double weighting_terms[3] = {0.3, 0.5, 0.2};
    const double weight_error = weightInv(error, th2);
    const double weight_descrDist = weightInv(descrDist, (double)descrDist_max);
    const double weight_response = (weight(resp1, (double)keyPRespons_max) + weight(resp2, (double)keyPRespons_max)) / 2.0;
    double overall = weighting_terms[0] * weight_error + weighting_terms[1] * weight_descrDist + weighting_terms[2] * weight_response;
    if (z3DtooFar) {
        double z_weight = 1.0;
        if (zValue3D > 0) z_weight = 0.5 + 0.9 * cfg_pose.maxDist3DPtsZ / (2.0 * zValue3D);
        else if (zValue3D < 0) z_weight = 0.25;
        overall *= z_weight;
    }
    return overall;
}

bool StereoRefine::Impl::compareCorrespondences(const CoordinatePropsNew &n, const CoordinateProps &o) {  // :2450-2497
    const double weight_th = 0.2;
    const size_t max_age = 15;
    const double w0 = computeCorrespondenceWeight(n.sampsonError, (double)n.descrDist, (double)n.keyPResponses[0], (double)n.keyPResponses[1]);
    const double w1 = computeCorrespondenceWeight(o.SampsonErrors.back(), (double)o.descrDist, (double)o.keyPResponses[0],
                                                  (double)o.keyPResponses[1]);
    if (!(w0 > w1)) {
        const double rel_diff = (w1 - w0) / w1;
        if (rel_diff < 0.05 || rel_diff > weight_th) return false;
    } else {
        const double rel_diff = (w0 - w1) / w0;
        if (rel_diff < 0.05) return false;
        if (rel_diff > weight_th) return true;
    }
    if (o.age > max_age) return true;
    if (o.SampsonErrors.size() > 1 && o.SampsonErrors.back() > o.SampsonErrors[o.SampsonErrors.size() - 2]) return true;
    return false;
}

int StereoRefine::Impl::filterNewCorrespondences(std::vector<cv::DMatch> &matches, const std::vector<cv::KeyPoint> &kp1,
                                                 const std::vector<cv::KeyPoint> &kp2, const std::vector<double> &error) {  // :2107-2316
    std::vector<CoordinatePropsNew> corrProbsNew;
    std::vector<double> a, b;
    std::vector<cv::Point2f> pa, pb;
    std::vector<cv::DMatch> mnew;
    for (size_t i = 0; i < nr_corrs_new; i++) {
        if (!mask_E_new[i]) continue;
        const int q = matches[i].queryIdx, t = matches[i].trainIdx;
        CoordinatePropsNew c;
        c.pt1 = kp1[(size_t)q].pt;
        c.pt2 = kp2[(size_t)t].pt;
        c.descrDist = matches[i].distance;
        c.keyPResponses[0] = kp1[(size_t)q].response;
        c.keyPResponses[1] = kp2[(size_t)t].response;
        c.sampsonError = error[i];
        corrProbsNew.push_back(c);
        a.push_back(p1new[2 * i]);
        a.push_back(p1new[2 * i + 1]);
        b.push_back(p2new[2 * i]);
        b.push_back(p2new[2 * i + 1]);
        pa.push_back(points1new[i]);
        pb.push_back(points2new[i]);
        mnew.push_back(matches[i]);
    }
    p1new = a;
    p2new = b;
    points1new = pa;
    points2new = pb;
    matches = mnew;
    if (!tree) return -1;
    std::vector<size_t> delete_list_new, delete_list_old;
    if (!correspondencePool.empty()) {
        for (size_t i = 0; i < nr_inliers_new; i++) {
            std::vector<std::pair<size_t, float>> result;
            const size_t nr_found = radiusSearch(corrProbsNew[i].pt1, cfg_pose.minPtsDistance, result);
            if (!nr_found) continue;
            bool deletionMarked = false;
            size_t j = 0;
            for (; j < nr_found; j++) {
                const CoordinateProps &corr_tmp = *correspondencePoolIdx[result[j].first];
                if (result[j].second < 2.0f) {
                    const float dx = corr_tmp.pt2.x - corrProbsNew[i].pt2.x, dy = corr_tmp.pt2.y - corrProbsNew[i].pt2.y;
                    const double diff_dist = (double)dx * (double)dx + (double)dy * (double)dy;
                    if (diff_dist < 2.0) {
                        if (diff_dist < 0.01 && result[j].second < 0.01f) {
                            delete_list_new.push_back(i);
                            deletionMarked = true;
                            correspondencePoolIdx[result[j].first]->nrFound++;
                            break;
                        }
                        if (compareCorrespondences(corrProbsNew[i], corr_tmp)) {
                            delete_list_old.push_back(result[j].first);
                        } else {
                            delete_list_new.push_back(i);
                            deletionMarked = true;
                            correspondencePoolIdx[result[j].first]->nrFound++;
                            break;
                        }
                    }
                } else {
                    break;
                }
            }
            if (!deletionMarked && j == 0) {
                for (; j < nr_found; j++)
                    if (!compareCorrespondences(corrProbsNew[i], *correspondencePoolIdx[result[j].first])) {
                        delete_list_new.push_back(i);
                        break;
                    }
                if (j >= nr_found)
                    for (j = 0; j < nr_found; j++) delete_list_old.push_back(result[j].first);
            }
        }
    }
    if (!delete_list_new.empty()) {
        const size_t n_del = delete_list_new.size();
        if (n_del == nr_inliers_new) {
            p1new.clear();
            p2new.clear();
            points1new.clear();
            points2new.clear();
            matches.clear();
            mask_E_new.clear();
            mask_Q_new.clear();
            nr_inliers_new = 0;
            nr_corrs_new = 0;
        } else {
            std::vector<uint8_t> drop(nr_inliers_new, 0);
            for (size_t i : delete_list_new) drop[i] = 1;
            a.clear();
            b.clear();
            pa.clear();
            pb.clear();
            mnew.clear();
            for (size_t i = 0; i < nr_inliers_new; ++i) {
                if (drop[i]) continue;
                a.push_back(p1new[2 * i]);
                a.push_back(p1new[2 * i + 1]);
                b.push_back(p2new[2 * i]);
                b.push_back(p2new[2 * i + 1]);
                pa.push_back(points1new[i]);
                pb.push_back(points2new[i]);
                mnew.push_back(matches[i]);
            }
            p1new = a;
            p2new = b;
            points1new = pa;
            points2new = pb;
            matches = mnew;
            const size_t n_new = nr_inliers_new - n_del;
            mask_E_new.assign(n_new, 1);
            mask_Q_new.clear();
            nr_inliers_new = n_new;
            nr_corrs_new = n_new;
        }
    } else {
        nr_corrs_new = matches.size();
        nr_inliers_new = nr_corrs_new;
        mask_E_new.assign(nr_corrs_new, 1);
        mask_Q_new.clear();
    }
    have_Q = false;
    Qv.clear();
    if (!delete_list_old.empty()) {
        std::sort(delete_list_old.begin(), delete_list_old.end());
        delete_list_old.erase(std::unique(delete_list_old.begin(), delete_list_old.end()), delete_list_old.end());
        if (poolCorrespondenceDelete(delete_list_old)) return -1;
    }
    return 0;
}

int StereoRefine::Impl::poolCorrespondenceDelete(std::vector<size_t> delete_list) {  // :2318-2436
    const size_t nrToDel = delete_list.size(), poolSize = correspondencePool.size();
    if (nrToDel == 0) return 0;
    if (nrToDel == poolSize) {
        tree = false;
        grid.clear();
        p1Cam.clear();
        p2Cam.clear();
        correspondencePool.clear();
        correspondencePoolIdx.clear();
        corrIdx = 0;
        nr_Q_tooFar = 0;
        nr_Qs = 0;
        return 0;
    }
    std::vector<uint8_t> dead(poolSize, 0);
    for (size_t i = 0; i < nrToDel; i++) {
        auto it = correspondencePoolIdx.find(delete_list[i]);
        if (it == correspondencePoolIdx.end() || it->second == correspondencePool.end()) return -1;  // "Invalid pool iterator"
        dead[it->second->ptIdx] = 1;
    }
    std::vector<double> a, b;
    a.reserve(p1Cam.size());
    b.reserve(p2Cam.size());
    std::vector<size_t> shift(poolSize, 0);
    size_t removed = 0;
    for (size_t i = 0; i < poolSize; ++i) {
        shift[i] = removed;
        if (dead[i]) {
            removed++;
            continue;
        }
        a.push_back(p1Cam[2 * i]);
        a.push_back(p1Cam[2 * i + 1]);
        b.push_back(p2Cam[2 * i]);
        b.push_back(p2Cam[2 * i + 1]);
    }
    p1Cam.swap(a);
    p2Cam.swap(b);
    for (size_t i = 0; i < nrToDel; i++) {
        auto it = correspondencePoolIdx.find(delete_list[i]);
        gridRemove(delete_list[i], it->second->pt1);
        if (!nearZero(100.0 * (it->second->Q[0] + it->second->Q[1] + it->second->Q[2]))) {
            if (nr_Qs) {
                nr_Qs--;
                if (it->second->Q_tooFar && nr_Q_tooFar) nr_Q_tooFar--;
            }
        }
        correspondencePool.erase(it->second);
        it->second = correspondencePool.end();
    }
    for (auto &c : correspondencePool) c.ptIdx -= shift[c.ptIdx];
    return 0;
}

// checkPoolSize (:2550-2816): too many correspondences in the pool -> thin it where the image is densely covered.  First the
// correspondences that share a LEFT pixel (rounded position) with another one go, the best weight of a pixel staying; then a density
// image of the occupied pixels is dilated with an elliptic element of minPtsDistance and eroded with the next larger one (what survives
// are pixels inside densely covered regions), the survivors are deleted -- all of them while fewer than the quota, with a growing element,
// otherwise the lowest weights among them.  cv::getStructuringElement / dilate / erode / findNonZero (OpenCV imgproc, not in the
// reference's tree) are restated from their published definitions: ellipse rows of half-width round(c * sqrt(1 - dy^2 / r^2)), anchor at
// the element's centre (size / 2), constant border 0 (the reference passes cv::Scalar(0) for BOTH operations, so the erosion also eats
// the image border), row-major scan order of findNonZero.
namespace {
struct Elem {
    int w = 0, h = 0;
    std::vector<uint8_t> k;
};
Elem ellipseElement(int size) {  // cv::getStructuringElement(MORPH_ELLIPSE, Size(size, size))
    Elem e;
    e.w = e.h = size;
    e.k.assign((size_t)size * size, 0);
    const int r = size / 2, c = size / 2;
    const double inv_r2 = r ? 1. / ((double)r * r) : 0;
    for (int i = 0; i < size; i++) {
        int j1 = 0, j2 = 0;
        const int dy = i - r;
        if (std::abs(dy) <= r) {
            const int dx = (int)std::nearbyint(c * std::sqrt((r * r - dy * dy) * inv_r2));  // saturate_cast<int>(double) = cvRound
            j1 = std::max(c - dx, 0);
            j2 = std::min(c + dx + 1, size);
        }
        for (int j = j1; j < j2; j++) e.k[(size_t)i * size + j] = 1;
    }
    return e;
}
// dst(y, x) = max (dilate) / min (erode) of src(y + i - ay, x + j - ax) over the element's non-zero cells, 0 outside the image
void morph(const std::vector<uint8_t> &src, std::vector<uint8_t> &dst, int W, int H, const Elem &e, bool dilate) {
    const int ax = e.w / 2, ay = e.h / 2;
    std::vector<std::pair<int, int>> offs;
    for (int i = 0; i < e.h; ++i)
        for (int j = 0; j < e.w; ++j)
            if (e.k[(size_t)i * e.w + j]) offs.emplace_back(i - ay, j - ax);
    dst.assign((size_t)W * H, dilate ? 0 : 255);
    if (dilate) {  // scatter from the (few) set pixels
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x)
                if (src[(size_t)y * W + x])
                    for (const auto &o : offs) {
                        const int yy = y - o.first, xx = x - o.second;
                        if (yy >= 0 && yy < H && xx >= 0 && xx < W) dst[(size_t)yy * W + xx] = 255;
                    }
    } else {
        for (int y = 0; y < H; ++y)
            for (int x = 0; x < W; ++x) {
                uint8_t v = 255;
                for (const auto &o : offs) {
                    const int yy = y + o.first, xx = x + o.second;
                    if (yy < 0 || yy >= H || xx < 0 || xx >= W || !src[(size_t)yy * W + xx]) {
                        v = 0;
                        break;
                    }
                }
                dst[(size_t)y * W + x] = v;
            }
    }
}
}  // namespace

int StereoRefine::Impl::checkPoolSize(long long maxPoolSize) {
    size_t n_del = 0;
    const size_t pool_Size = correspondencePool.size();
    if (maxPoolSize < 0 && pool_Size > 20) n_del = pool_Size / 2;
    else if (maxPoolSize < 0) return 0;
    else if (pool_Size <= (size_t)maxPoolSize) return 0;
    else n_del = pool_Size - (size_t)maxPoolSize;
    if (pool_Size - n_del < 10) {
        if (pool_Size > 20) n_del = pool_Size / 2;
        else return 0;
    }
    std::vector<size_t> delIdx(n_del);
    size_t delIdxIdx = 0;
    const int W = cfg_usac.imgSize.width, H = cfg_usac.imgSize.height;
    auto weightOf = [&](size_t poolIdx) {
        const CoordinateProps &c = *correspondencePoolIdx[poolIdx];
        return computeCorrespondenceWeight(c.SampsonErrors.back(), c.descrDist, c.keyPResponses[0], c.keyPResponses[1], c.Q_tooFar, c.Q[2]);
    };
    // pool indices per (rounded) left pixel, in pool order; pixels holding more than one in the order they got their second
    std::unordered_map<long long, std::vector<size_t>> idxPos;
    std::vector<std::pair<int, int>> posMultCorrs;  // (x, y)
    auto keyOf = [&](int x, int y) { return (long long)y * (long long)W + x; };
    for (auto &c : correspondencePool) {
        const int y = (int)std::round(c.pt1.y), x = (int)std::round(c.pt1.x);
        if (x < 0 || y < 0 || x >= W || y >= H)
            throw cv::Exception("StereoRefine::checkPoolSize: a pool correspondence lies outside cfg_usac.imgSize (the reference indexes an image-sized table with it)");
        std::vector<size_t> &v = idxPos[keyOf(x, y)];
        v.push_back(c.poolIdx);
        if (v.size() == 2) posMultCorrs.emplace_back(x, y);
    }
    if (!posMultCorrs.empty()) {
        std::vector<std::pair<size_t, size_t>> idx1;
        std::vector<size_t> nr_entries(posMultCorrs.size());
        for (size_t i = 0; i < posMultCorrs.size(); i++) {
            nr_entries[i] = idxPos[keyOf(posMultCorrs[i].first, posMultCorrs[i].second)].size();
            for (size_t j = 0; j < nr_entries[i]; j++) idx1.emplace_back(i, j);
        }
        const size_t n_multi = idx1.size() - posMultCorrs.size();
        if (n_multi <= n_del) {
            for (auto &pm : posMultCorrs) {
                std::vector<size_t> &cell = idxPos[keyOf(pm.first, pm.second)];
                std::vector<std::pair<double, size_t>> w;
                for (size_t j = 0; j < cell.size(); j++) w.emplace_back(weightOf(cell[j]), j);
                std::sort(w.begin(), w.end(), [](const std::pair<double, size_t> &f, const std::pair<double, size_t> &g) { return f.first > g.first; });
                for (size_t j = 1; j < w.size(); j++) delIdx[delIdxIdx++] = cell[w[j].second];
                const size_t keep = cell[w[0].second];
                cell.clear();
                cell.push_back(keep);
            }
            n_del -= n_multi;
        } else {
            // (the reference sizes this vector with idx1.size() value-initialised entries and then APPENDS the real ones: the first
            // idx1.size() entries are (0.0, 0) and sort to the front -- they point at entry 0 of idx1 again and again, whose pixel loses
            // one correspondence per visit while it has more than one.  Kept as it is: it decides which correspondences go.)
            std::vector<std::pair<double, size_t>> w(idx1.size());
            for (size_t i = 0; i < idx1.size(); i++) {
                const auto &pm = posMultCorrs[idx1[i].first];
                w.emplace_back(weightOf(idxPos[keyOf(pm.first, pm.second)][idx1[i].second]), i);
            }
            std::sort(w.begin(), w.end(), [](const std::pair<double, size_t> &f, const std::pair<double, size_t> &g) { return f.first < g.first; });
            for (size_t i = 0, count = 0; i < w.size(); i++) {
                const auto &ent = idx1[w[i].second];
                if (nr_entries[ent.first] > 1) {
                    const auto &pm = posMultCorrs[ent.first];
                    delIdx[delIdxIdx++] = idxPos[keyOf(pm.first, pm.second)][ent.second];
                    nr_entries[ent.first]--;
                    count++;
                }
                if (count >= n_del) break;
            }
            n_del = 0;
        }
    }
    if (n_del) {
        std::vector<uint8_t> density((size_t)W * H, 0), init, tmp;
        for (auto &c : correspondencePool) density[(size_t)((int)std::round(c.pt1.y)) * W + (int)std::round(c.pt1.x)] = 255;
        init = density;
        const double mpd = (double)cfg_pose.minPtsDistance;
        const double frac = mpd - std::ceil(mpd);
        int erosion_size = (frac < 1e-3 && frac > -1e-3) ? (int)std::ceil(mpd) + 1 : (int)std::ceil(mpd);  // nearZero (pose_helper.h:82)
        int nr_erosions = 0;
        do {
            morph(density, tmp, W, H, ellipseElement(erosion_size), true);
            morph(tmp, density, W, H, ellipseElement(erosion_size + 1), false);
            for (size_t i = 0; i < density.size(); ++i) density[i] &= init[i];
            std::vector<std::pair<int, int>> locations;  // (x, y), row-major scan (cv::findNonZero)
            for (int y = 0; y < H; ++y)
                for (int x = 0; x < W; ++x)
                    if (density[(size_t)y * W + x]) locations.emplace_back(x, y);
            size_t n_loc = locations.size();
            if (n_loc <= n_del && nr_erosions < 100) {
                for (auto &l : locations) delIdx[delIdxIdx++] = idxPos[keyOf(l.first, l.second)][0];
                n_del -= n_loc;
                if (n_del) {
                    for (size_t i = 0; i < density.size(); ++i) density[i] = (uint8_t)(~density[i]) & init[i];
                    init = density;
                    erosion_size++;
                    nr_erosions++;
                }
            } else {
                if (nr_erosions >= 100) {
                    locations.clear();
                    for (int y = 0; y < H; ++y)
                        for (int x = 0; x < W; ++x)
                            if (init[(size_t)y * W + x]) locations.emplace_back(x, y);
                    n_loc = locations.size();
                }
                std::vector<std::pair<double, size_t>> w(n_loc);
                for (size_t i = 0; i < n_loc; ++i) w[i] = std::make_pair(weightOf(idxPos[keyOf(locations[i].first, locations[i].second)][0]), i);
                std::sort(w.begin(), w.end(), [](const std::pair<double, size_t> &f, const std::pair<double, size_t> &g) { return f.first < g.first; });
                for (size_t i = 0; i < n_del; i++) {
                    const auto &l = locations[w[i].second];
                    delIdx[delIdxIdx++] = idxPos[keyOf(l.first, l.second)][0];
                }
                n_del = 0;
            }
        } while (n_del);
    }
    if (poolCorrespondenceDelete(delIdx)) return -1;
    maxPoolSizeReached = true;
    return 0;
}

// ---- pose history: rating and stability (:2817-3317) --------------------------------------------------------------------------------------

int StereoRefine::Impl::getNearToMeanPose() {
    const size_t n_p = pose_history.size();
    if (n_p < 5) return -1;
    const double point[3] = {0.5, 0.5, 0.5};
    std::vector<std::array<double, 3>> res(n_p);
    std::vector<std::pair<double, size_t>> xyz[3];
    for (int a = 0; a < 3; ++a) xyz[a].resize(n_p);
    for (size_t i = 0; i < n_p; i++) {
        const PoseHist &p = pose_history[i];
        for (int r = 0; r < 3; ++r) {
            res[i][r] = p.R[r * 3] * point[0] + p.R[r * 3 + 1] * point[1] + p.R[r * 3 + 2] * point[2] + p.t[r];
            xyz[r][i] = std::make_pair(res[i][r], i);
        }
    }
    for (int a = 0; a < 3; ++a)
        std::stable_sort(xyz[a].begin(), xyz[a].end(),
                         [](const std::pair<double, size_t> &f, const std::pair<double, size_t> &s) { return f.first < s.first; });
    const double range_th = 0.05, medArithDiffAbsRel[2] = {0.02, 1.33}, stdDevMult = 3.0;
    size_t q_idx[2];
    q_idx[0] = (size_t)std::floor((double)n_p * 0.25 + 0.5);
    q_idx[1] = n_p - q_idx[0];
    bool overRangeTh = false;
    double medianXYZ[3], arith[3] = {0, 0, 0}, c2[3] = {0, 0, 0}, arith_u[3] = {0, 0, 0}, c2_u[3] = {0, 0, 0}, arith_o[3] = {0, 0, 0},
                         c2_o[3] = {0, 0, 0}, stdv[3], thXYZ[3][2];
    for (int a = 0; a < 3; ++a) {
        if (std::abs(xyz[a][n_p - 1].first - xyz[a][0].first) > range_th) overRangeTh = true;
        medianXYZ[a] = (n_p % 2) ? xyz[a][(n_p - 1) / 2].first : (xyz[a][n_p / 2].first + xyz[a][n_p / 2 - 1].first) / 2.0;
    }
    const size_t nq = n_p - 2 * q_idx[0];
    for (int a = 0; a < 3; ++a) {
        for (size_t i = 0; i < q_idx[0]; i++) arith_u[a] += xyz[a][i].first, c2_u[a] += xyz[a][i].first * xyz[a][i].first;
        for (size_t i = q_idx[0]; i < q_idx[1]; i++) arith[a] += xyz[a][i].first, c2[a] += xyz[a][i].first * xyz[a][i].first;
        for (size_t i = q_idx[1]; i < n_p; i++) arith_o[a] += xyz[a][i].first, c2_o[a] += xyz[a][i].first * xyz[a][i].first;
        arith_o[a] += arith_u[a] + arith[a];
        arith_o[a] /= (double)n_p;
        arith[a] /= (double)nq;
        if (overRangeTh) {
            stdv[a] = std::sqrt((c2[a] - (double)nq * arith[a] * arith[a]) / ((double)nq - 1.0));
            thXYZ[a][0] = arith[a] - stdDevMult * stdv[a];
            thXYZ[a][1] = arith[a] + stdDevMult * stdv[a];
        }
    }
    if (!overRangeTh)
        for (int a = 0; a < 3; ++a) {
            c2[a] += c2_u[a] + c2_o[a];
            stdv[a] = std::sqrt((c2[a] - (double)n_p * arith_o[a] * arith_o[a]) / ((double)n_p - 1.0));
            thXYZ[a][0] = arith_o[a] - stdDevMult * stdv[a];
            thXYZ[a][1] = arith_o[a] + stdDevMult * stdv[a];
        }
    bool statFilterPossible[3] = {true, true, true};
    for (int i = 0; i < 3; i++) {
        if ((arith_o[i] > 0 && medianXYZ[i] > 0) || (arith_o[i] < 0 && medianXYZ[i] < 0)) {
            if ((arith_o[i] / medianXYZ[i] > medArithDiffAbsRel[1]) || (medianXYZ[i] / arith_o[i] > medArithDiffAbsRel[1]) ||
                (std::abs(arith_o[i] - medianXYZ[i]) > medArithDiffAbsRel[0]))
                statFilterPossible[i] = false;
        } else if (nearZero(arith_o[i]) || nearZero(medianXYZ[i])) {
            if (std::abs(arith_o[i] - medianXYZ[i]) > medArithDiffAbsRel[0]) statFilterPossible[i] = false;
        } else {
            statFilterPossible[i] = false;
        }
    }
    std::vector<size_t> valid_idx;
    const auto &x = xyz[0], &y = xyz[1], &z = xyz[2];
    if (!statFilterPossible[0] && !statFilterPossible[1] && !statFilterPossible[2]) {
        for (size_t i = q_idx[0]; i < q_idx[1]; i++)
            for (size_t j = q_idx[0]; j < q_idx[1]; j++)
                if (x[i].second == y[j].second) {
                    for (size_t k = q_idx[0]; k < q_idx[1]; k++)
                        if (x[i].second == z[k].second) {
                            valid_idx.push_back(x[i].second);
                            break;
                        }
                    break;
                }
    } else {
        std::vector<size_t> idx[3];
        for (int a = 0; a < 3; ++a) {
            if (statFilterPossible[a]) {
                for (size_t i = 0; i < n_p; i++)
                    if (xyz[a][i].first > thXYZ[a][0] && xyz[a][i].first < thXYZ[a][1]) idx[a].push_back(xyz[a][i].second);
            } else {
                for (size_t i = q_idx[0]; i < q_idx[1]; i++) idx[a].push_back(xyz[a][i].second);
            }
        }
        for (size_t i = 0; i < idx[0].size(); i++)
            for (size_t j = 0; j < idx[1].size(); j++)
                if (idx[0][i] == idx[1][j]) {
                    for (size_t k = 0; k < idx[2].size(); k++)
                        if (idx[0][i] == idx[2][k]) {
                            valid_idx.push_back(x[i].second);  // (sic) the reference pushes x[i].second, not x_idx[i] (:3037)
                            break;
                        }
                    break;
                }
    }
    if (valid_idx.size() < 3) return -2;
    double cg[3] = {0, 0, 0};
    for (size_t v : valid_idx)
        for (int a = 0; a < 3; ++a) cg[a] += res[v][a];
    for (int a = 0; a < 3; ++a) cg[a] /= (double)valid_idx.size();
    const double point_norm = std::sqrt(cg[0] * cg[0] + cg[1] * cg[1] + cg[2] * cg[2]);
    std::vector<double> dist2(n_p);
    size_t imin = 0, imax = 0;
    for (size_t i = 0; i < n_p; i++) {
        const double dx = res[i][0] - cg[0], dy = res[i][1] - cg[1], dz = res[i][2] - cg[2];
        dist2[i] = std::sqrt(dx * dx + dy * dy + dz * dz);
        if (dist2[i] < dist2[imin]) imin = i;     // std::minmax_element: first smallest,
        if (!(dist2[i] < dist2[imax])) imax = i;  // last largest
    }
    const PoseHist &best = pose_history[imin];
    self->R_mostLikely = mat33(best.R);
    self->t_mostLikely = mat31(best.t);
    self->E_mostLikely = mat33(best.E);
    mostLikelyPoseIdxs.push_back(imin);
    pose_history_rating.assign(n_p, 0.0);
    const double max_dist = dist2[imax] + point_norm * 0.0075;
    for (size_t i = 0; i < n_p; i++) pose_history_rating[i] = 1.0 - dist2[i] / max_dist;
    return 0;
}

void StereoRefine::Impl::updateMaxSkipPairs() {  // :3300-3317
    if ((cfg_pose.raiseSkipCnt & 0xF) && ((size_t)(((cfg_pose.raiseSkipCnt & 0xF0) >> 4) + 1) <= nrConsecStablePoses))
        maxSkipPairsNew = (size_t)std::ceil((double)cfg_pose.maxSkipPairs * (1.0 + (double)(cfg_pose.raiseSkipCnt & 0xF) * 0.25));
    else
        maxSkipPairsNew = cfg_pose.maxSkipPairs;
}

int StereoRefine::Impl::checkPoseStability() {  // :3131-3298
    CV_Assert(nrEstimation == pose_history.size());
    const size_t minPoolSizeToBeStable = 1000;
    const int err = getNearToMeanPose();
    if (err) {
        self->poseIsStable = false;
        self->mostLikelyPose_stable = false;
        self->R_mostLikely = mat33(R_);
        self->t_mostLikely = mat31(t_);
        self->E_mostLikely = mat33(E_);
        if (err != -2) nr_tries = 0;
        return -1;
    }
    if (nrEstimation < cfg_pose.minContStablePoses || correspondencePool.size() < minPoolSizeToBeStable) {
        self->poseIsStable = false;
        self->mostLikelyPose_stable = false;
        nr_tries = 0;
        return -1;
    }
    size_t count = 2, stable_poses = 2;
    const double lo = pose_history_rating.back() - cfg_pose.absThRankingStable, hi = pose_history_rating.back() + cfg_pose.absThRankingStable;
    while (count <= cfg_pose.minContStablePoses) {
        const double r = pose_history_rating[nrEstimation - count];
        if (r > lo && r < hi && r > cfg_pose.minNormDistStable) {
            stable_poses++;
        } else {
            stable_poses--;
            break;
        }
        count++;
    }
    if (mostLikelyPoseIdxs.size() >= cfg_pose.minContStablePoses) {
        const int min_idx = (int)(mostLikelyPoseIdxs.size() - cfg_pose.minContStablePoses);
        const size_t last_idx = mostLikelyPoseIdxs.back();
        if (pose_history_rating[last_idx] > cfg_pose.minNormDistStable) {
            int cnt = (int)mostLikelyPoseIdxs.size() - 2;
            for (; cnt >= min_idx; cnt--)
                if (mostLikelyPoseIdxs[(size_t)cnt] != last_idx) break;
            self->mostLikelyPose_stable = cnt < min_idx;
        } else {
            self->mostLikelyPose_stable = false;
        }
    }
    double ratio3DPtsFar = 0;
    if (nr_Qs) ratio3DPtsFar = (double)nr_Q_tooFar / (double)nr_Qs;
    if (stable_poses == count && ratio3DPtsFar < 0.95) {
        self->poseIsStable = true;
        nrConsecStablePoses++;
        if (maxSkipPairsNew <= cfg_pose.maxSkipPairs) updateMaxSkipPairs();
        if (nr_tries) nr_tries--;
        return 0;
    }
    self->poseIsStable = false;
    nr_tries++;
    if (nr_tries > cfg_pose.minContStablePoses && maxPoolSizeReached && ratio3DPtsFar < cfg_pose.maxRat3DPtsFar) {
        const double minOverlap = 0.8;
        const size_t m = cfg_pose.minContStablePoses;
        std::vector<std::pair<double, double>> err_ranges(m);
        double mean_error = 0;
        for (count = 0; count < m; count++) {
            const StatVals &s = errorStatistic_history[nrEstimation - count - 1];
            err_ranges[count] = std::make_pair(s.arithErr - 2.0 * s.arithStd, s.arithErr + 2.0 * s.arithStd);
            mean_error += s.arithErr;
        }
        mean_error /= (double)count;
        double min_lo = err_ranges[0].first, max_lo = err_ranges[0].first, min_hi = err_ranges[0].second, max_hi = err_ranges[0].second;
        for (auto &r : err_ranges) {
            min_lo = std::min(min_lo, r.first);
            max_lo = std::max(max_lo, r.first);
            min_hi = std::min(min_hi, r.second);
            max_hi = std::max(max_hi, r.second);
        }
        if (min_hi <= min_lo || max_lo >= max_hi) {
            self->poseIsStable = false;
            nrConsecStablePoses = 0;
            return 0;
        }
        const double er0 = mean_error - min_lo, er1 = max_hi - mean_error, full = er0 + er1;
        const double pct0 = er0 / full, pct1 = er1 / full;
        for (count = 0; count < m; count++) {
            const double right_overlap = pct1 * (err_ranges[count].second - mean_error) / er1;
            const double left_overlap = pct0 * (mean_error - err_ranges[count].first) / er0;
            if (right_overlap + left_overlap < minOverlap) {
                self->poseIsStable = false;
                nrConsecStablePoses = 0;
                return 0;
            }
        }
        self->poseIsStable = true;
        nrConsecStablePoses++;
    } else {
        nrConsecStablePoses = 0;
    }
    if (self->poseIsStable && maxSkipPairsNew <= cfg_pose.maxSkipPairs) updateMaxSkipPairs();
    return 0;
}

// ---- addNewCorrespondences (:416-957) ----------------------------------------------------------------------------------------------------

int StereoRefine::Impl::addNewCorrespondences(std::vector<cv::DMatch> &matches, std::vector<cv::KeyPoint> &kp1,
                                              std::vector<cv::KeyPoint> &kp2, const ConfigUSAC &cfg) {
    cfg_usac = cfg;
    // the caller's pointers describe the matches of this call (tests/poselib-test/main.cpp:1944-1957); the by-value copies this function
    // works on are the same data and stay alive for the whole call, where the reference re-points cfg_usac.matches at them (:625-649)
    if (cfg.matches) cfg_usac.matches = &matches, cfg_usac.keypoints1 = &kp1, cfg_usac.keypoints2 = &kp2;
    // ... and die with it: on every exit the member copy forgets them
    struct Unpoint {
        ConfigUSAC &c;
        ~Unpoint() { c.matches = nullptr, c.keypoints1 = nullptr, c.keypoints2 = nullptr; }
    } unpoint{cfg_usac};
    nr_corrs_new = matches.size();
    const int n0 = (int)nr_corrs_new;
    std::vector<float> a((size_t)n0 * 2), b((size_t)n0 * 2);
    for (int i = 0; i < n0; ++i) {
        const cv::Point2f pa = kp1[(size_t)matches[i].queryIdx].pt, pb = kp2[(size_t)matches[i].trainIdx].pt;
        a[2 * i] = pa.x, a[2 * i + 1] = pa.y, b[2 * i] = pb.x, b[2 * i + 1] = pb.y;
    }
    const cv::Mat &K0 = *cfg_pose.K0, &K1 = *cfg_pose.K1;
    const double k0[4] = {K0.at<double>(0, 0), K0.at<double>(1, 1), K0.at<double>(0, 2), K0.at<double>(1, 2)};
    const double k1[4] = {K1.at<double>(0, 0), K1.at<double>(1, 1), K1.at<double>(0, 2), K1.at<double>(1, 2)};
    mlpl_ctx *ctx = mlpl_facade_default_ctx();
    // ImgToCamCoordTrans + Remove_LensDist (pose_helper.cpp:1100-1109, 1169-1279) on the GPU
    if (n0 > 0 && (mlpl_img_to_cam(ctx, a.data(), n0, k0) != MLPL_OK || mlpl_img_to_cam(ctx, b.data(), n0, k1) != MLPL_OK))
        throw cv::Exception(std::string("mlpl_img_to_cam: ") + mlpl_last_error());
    double d0[8] = {0}, d1[8] = {0};
    auto read_dist = [](const cv::Mat *m, double *out) {
        if (!m || m->empty()) return;
        CV_Assert(m->rows * m->cols == 8 && m->type() == CV_64F);
        for (int i = 0; i < 8; ++i) out[i] = m->rows == 1 ? m->at<double>(0, i) : m->at<double>(i, 0);
    };
    read_dist(cfg_pose.dist0_8, d0);
    read_dist(cfg_pose.dist1_8, d1);
    int n_left = n0;
    const int rcd = n0 > 0 ? mlpl_remove_lens_dist(ctx, a.data(), b.data(), n0, d0, d1, &n_left) : MLPL_E_FAILED;
    if (rcd == MLPL_E_FAILED) {
        std::cout << "Removing lens distortion failed or too less matches!" << std::endl;
        return -1;
    }
    if (rcd != MLPL_OK) throw cv::Exception(std::string("mlpl_remove_lens_dist: ") + mlpl_last_error());
    // (Remove_LensDist drops correspondences whose undistortion fails from the point vectors only; the reference keeps indexing
    // `matches` by position afterwards, so a drop misaligns its bookkeeping -- with n_left == n0, the normal case, nothing moves)
    const int n = n_left;
    std::vector<cv::DMatch> matches_left;
    if (n_left != n0 && cfg_usac.matches) {
        // USAC asserts one match per correspondence (usac_estimations.cpp:316).  After a drop the reference's own bookkeeping pairs the
        // surviving correspondences with the leading matches by position; the PROSAC order and the SPRT start values get the same view
        // here instead of an assertion out of addNewCorrespondences (every other RobMethod carries on)
        matches_left.assign(matches.begin(), matches.begin() + n_left);
        cfg_usac.matches = &matches_left;
    }
    points1new.resize((size_t)n);
    points2new.resize((size_t)n);
    p1new.resize((size_t)n * 2);
    p2new.resize((size_t)n * 2);
    for (int i = 0; i < n; ++i) {
        points1new[i] = cv::Point2f(a[2 * i], a[2 * i + 1]);
        points2new[i] = cv::Point2f(b[2 * i], b[2 * i + 1]);
        p1new[2 * i] = (double)a[2 * i], p1new[2 * i + 1] = (double)a[2 * i + 1];
        p2new[2 * i] = (double)b[2 * i], p2new[2 * i + 1] = (double)b[2 * i + 1];
    }
    p1new_tmp = p1new;
    p2new_tmp = p2new;

    double inlier_ratio_new1 = 0;
    if (nrEstimation == 0) {
        const int err = robustInitialization(inlier_ratio_new1, matches, kp1, kp2);
        if (err == -1) return -1;
        if (err == -2) return -2;
        return 0;  // 0 and -3
    }

    std::vector<uint8_t> mask;
    std::vector<double> errorNew;
    bool addToPool = false;
    size_t nr_inliers_tmp = getInliers(E_, p1new, p2new, mask, errorNew);
    double inlier_ratio_new = (double)nr_inliers_tmp / (double)nr_corrs_new;
    if (inlier_ratio_new < (1.0 - cfg_pose.relInlRatThLast) * inlier_ratio_history.back()) {
        // has the pose changed?  robust estimation on the new pair alone
        if (robustPoseEstimation()) return -1;
        inlier_ratio_new1 = (double)nr_inliers_new / (double)nr_corrs_new;
        if (inlier_ratio_new < inlier_ratio_new1 * (1.0 - cfg_pose.relInlRatThNew)) {
            if (inlier_ratio_new1 >= cfg_pose.minInlierRatioReInit && inlier_ratio_new < cfg_pose.minInlierRatioReInit) {
                if (!reinitializeSystem(inlier_ratio_new1, matches, kp1, kp2)) return -2;
                std::cout << "The pose has changed! System is reinitialized!" << std::endl;
                return 0;
            }
            if (inlier_ratio_new1 < cfg_pose.minInlierRatSkip &&
                inlier_ratio_new1 < cfg_pose.relMinInlierRatSkip * inlier_ratio_history.back()) {
                std::memcpy(E_, pose_history.back().E, 72);
                std::memcpy(R_, pose_history.back().R, 72);
                std::memcpy(t_, pose_history.back().t, 24);
                std::cout << "It seems that the new image pair is really bad. Restoring last valid pose! "
                             "Be aware that the 3D points might not be valid!"
                          << std::endl;
            } else {
                std::cout << "Either the pose has changed or the image pair has bad quality! "
                             "Robustly estimating new pose from pool which might be wrong!"
                          << std::endl;
                double E_old[9], R_old[9], t_old[3];
                std::memcpy(E_old, E_, 72), std::memcpy(R_old, R_, 72), std::memcpy(t_old, t_, 24);
                const std::vector<uint8_t> mE = mask_E_new, mQ = mask_Q_new;
                const std::vector<double> Q_old = Qv;
                const bool hq = have_Q;
                const size_t nr_old = nr_inliers_new;
                if (robustEstimationOnPool()) {
                    std::cout << "Robust estimation on pool correspondences failed! Reinitializing system!" << std::endl;
                    std::memcpy(E_, E_old, 72), std::memcpy(R_, R_old, 72), std::memcpy(t_, t_old, 24);
                    mask_E_new = mE, mask_Q_new = mQ, Qv = Q_old, have_Q = hq;
                    nr_inliers_new = nr_old;
                    if (!reinitializeSystem(inlier_ratio_new1, matches, kp1, kp2)) return -2;
                    return 0;
                }
                mask_E_new = mE, mask_Q_new = mQ, Qv = Q_old, have_Q = hq;
                nr_inliers_new = nr_old;
                self->poseIsStable = false;
                self->mostLikelyPose_stable = false;
            }
            skipCount++;
        } else {
            std::cout << "Low inlier ratio detected! Bad image pair!" << std::endl;
            mask_Q_new.clear();
            mask_E_new = mask;
            have_Q = false;
            Qv.clear();
            nr_inliers_new = nr_inliers_tmp;
            inlier_ratio_new1 = inlier_ratio_new;
            std::memcpy(E_, pose_history.back().E, 72);
            std::memcpy(R_, pose_history.back().R, 72);
            std::memcpy(t_, pose_history.back().t, 24);
            addToPool = true;
        }
    } else {
        addToPool = true;
        mask_E_new = mask;
        nr_inliers_new = nr_inliers_tmp;
        inlier_ratio_new1 = inlier_ratio_new;
    }

    if (addToPool) {
        if (filterNewCorrespondences(matches, kp1, kp2, errorNew)) {
            reinitializeSystem(inlier_ratio_new1, matches, kp1, kp2);
            return -2;
        }
        if (matches.size() + correspondencePool.size() > cfg_pose.maxPoolCorrespondences) {
            const long long keep = (long long)cfg_pose.maxPoolCorrespondences - (long long)matches.size();
            if (checkPoolSize(keep)) {
                reinitializeSystem(inlier_ratio_new1, matches, kp1, kp2);
                return -2;
            }
        }
        if (addCorrespondencesToPool(matches, kp1, kp2)) return -2;

        double E_old[9], R_old[9], t_old[3];
        std::memcpy(E_old, E_, 72), std::memcpy(R_old, R_, 72), std::memcpy(t_old, t_, 24);
        double minRelRemainingCorrsRef = 0.75;
        // robust estimation on the pool or refinement of the last pose on it (:680-820)
        if (cfg_pose.checkPoolPoseRobust == 1 || nr_since_robust > checkPoolPoseRobust_tmp ||
            (!maxPoolSizeReached && checkPoolPoseRobust_tmp * initNumberInliers < correspondencePool.size())) {
            const std::vector<uint8_t> mE = mask_E_new;
            const size_t nr_old = nr_inliers_new;
            mask_Q_new.clear();
            have_Q = false;
            Qv.clear();
            if (robustEstimationOnPool()) {
                std::cout << "Robust estimation on pool correspondences failed! Reinitializing system with last pose!" << std::endl;
                std::memcpy(E_, E_old, 72), std::memcpy(R_, R_old, 72), std::memcpy(t_, t_old, 24);
                mask_E_new = mE;
                nr_inliers_new = nr_old;
                have_Q = false;
                Qv.clear();
                if (!reinitializeSystem(inlier_ratio_new1, matches, kp1, kp2)) return -2;
                return -3;
            }
            if (cfg_pose.checkPoolPoseRobust > 1) {
                if (maxPoolSizeReached)
                    checkPoolPoseRobust_tmp = cfg_pose.checkPoolPoseRobust > 10 ? cfg_pose.checkPoolPoseRobust : 10;
                else if (checkPoolPoseRobust_tmp > 50)
                    checkPoolPoseRobust_tmp = cfg_pose.maxPoolCorrespondences / initNumberInliers + 2;
                else
                    checkPoolPoseRobust_tmp =
                        (size_t)std::round((double)cfg_pose.checkPoolPoseRobust + std::exp(0.8 + (double)checkPoolPoseRobust_tmp / 6.0));
            }
            nr_since_robust = 0;
            minRelRemainingCorrsRef = 0.7;
        } else {
            if (maxPoolSizeReached) nr_since_robust++;
            else nr_since_robust = 0;
            if (refinePoseFromPool()) {
                std::cout << "Taking old pose!" << std::endl;
                std::memcpy(E_, E_old, 72), std::memcpy(R_, R_old, 72), std::memcpy(t_, t_old, 24);
                skipCount++;
                if (failed_refinements > 0) {
                    failed_refinements = 0;
                    std::cout << "Reinitializing system!" << std::endl;
                    clearHistoryAndPool();
                } else {
                    std::vector<size_t> newIdx(newAddedPoolCorrs);
                    for (size_t i = 0; i < newAddedPoolCorrs; i++) newIdx[i] = corrIdx - 1 - i;
                    if (poolCorrespondenceDelete(newIdx)) {
                        clearHistoryAndPool();
                        failed_refinements = 0;
                        const int err = robustInitialization(inlier_ratio_new1, matches, kp1, kp2);
                        if (err == -1) return -1;
                        if (err == -3) return 0;
                        return -2;
                    }
                    failed_refinements++;
                }
                return -3;
            }
            failed_refinements = 0;
        }
        if ((double)nr_inliers_new < minRelRemainingCorrsRef * (double)correspondencePool.size()) {
            std::cout << "Too less inliers (<75%) after refinement! Reinitializing system and taking old pose!" << std::endl;
            std::memcpy(E_, E_old, 72), std::memcpy(R_, R_old, 72), std::memcpy(t_, t_old, 24);
            clearHistoryAndPool();
            return -3;
        }
        // inliers of the new pair (before the pool filter) with the E of all pairs
        nr_inliers_tmp = getInliers(E_, p1new_tmp, p2new_tmp, mask, errorNew);
        inlier_ratio_new = (double)nr_inliers_tmp / (double)(p1new_tmp.size() / 2);
        if (inlier_ratio_new < inlier_ratio_new1 * (1 - cfg_pose.relInlRatThNew)) {
            std::cout << "Inlier ratio of new image pair calculated with refined E over all image pairs is too small compared to its "
                         "initial inlier ratio! Reinitializing system and taking old pose!"
                      << std::endl;
            std::memcpy(E_, E_old, 72), std::memcpy(R_, R_old, 72), std::memcpy(t_, t_old, 24);
            clearHistoryAndPool();
            return -3;
        }
        inlier_ratio_history.push_back(inlier_ratio_new);
        PoseHist ph;
        std::memcpy(ph.E, E_, 72), std::memcpy(ph.R, R_, 72), std::memcpy(ph.t, t_, 24);
        pose_history.push_back(ph);
        {
            std::vector<double> e_in;
            e_in.reserve(nr_inliers_tmp);
            // (sic) the reference walks only the first nr_inliers_tmp entries of the mask (:836-841)
            std::vector<double> errorNew_tmp(nr_inliers_tmp, 0.0);
            for (size_t i = 0, c = 0; i < nr_inliers_tmp; i++)
                if (mask[i]) errorNew_tmp[c++] = errorNew[i];
            errorStatistic_history.push_back(statsFromVec(errorNew_tmp));
        }
        // delete pool elements marked as outliers by the estimation on the pool
        if (nr_inliers_new < correspondencePool.size()) {
            std::vector<uint8_t> mq;
            std::vector<double> Qt;
            std::unordered_set<size_t> delIdxNew;
            const size_t ps = correspondencePool.size();
            for (size_t i = 0; i < ps; i++) {
                if (!mask_E_new[i]) {
                    delIdxNew.insert(i);
                } else {
                    for (int c = 0; c < 3; ++c) Qt.push_back(Qv[i * 3 + c]);
                    mq.push_back(mask_Q_new[i] ? 1 : 0);
                }
            }
            mask_Q_new = mq;
            Qv = Qt;
            std::vector<size_t> delIdxPool;
            for (auto &c : correspondencePool)
                if (delIdxNew.count(c.ptIdx)) delIdxPool.push_back(c.poolIdx);
            if (poolCorrespondenceDelete(delIdxPool)) {
                reinitializeSystem(inlier_ratio_new1, matches, kp1, kp2);
                return -2;
            }
        }
        // errors with the new E and the 3-D points, per pool element
        std::vector<double> error;
        reprojErrors(p1Cam, p2Cam, E_, error);
        size_t count = 0;
        for (auto &c : correspondencePool) {
            if (have_Q && !Qv.empty()) {
                for (int k = 0; k < 3; ++k) c.Q[k] = Qv[count * 3 + k];
                c.Q_tooFar = !mask_Q_new[count];
                if (c.age <= 1) {
                    nr_Qs++;
                    if (c.Q_tooFar) nr_Q_tooFar++;
                }
            }
            c.SampsonErrors.push_back(error[count++]);
            double s = 0;
            for (double e : c.SampsonErrors) s += e;
            c.meanSampsonError = s / (double)c.SampsonErrors.size();
        }
        nrEstimation++;
        skipCount = 0;
        checkPoseStability();
    }
    if (skipCount > maxSkipPairsNew) {
        if (!reinitializeSystem(inlier_ratio_new1, matches, kp1, kp2)) return -2;
    }
    return 0;
}

// ---- public surface -----------------------------------------------------------------------------------------------------------------------

StereoRefine::StereoRefine(ConfigPoseEstimation cfg_pose_, bool verbose_) : d(new Impl) {
    d->self = this;
    d->cfg_pose = cfg_pose_;
    d->verbose = verbose_;
    d->init();
}
StereoRefine::~StereoRefine() = default;

void StereoRefine::setNewParameters(ConfigPoseEstimation cfg_pose_) {  // :87-185: thresholds follow the new intrinsics / th_pix_user
    d->cfg_pose = cfg_pose_;
    d->noticed = false;
    d->init();
}

int StereoRefine::addNewCorrespondences(std::vector<cv::DMatch> matches, std::vector<cv::KeyPoint> kp1, std::vector<cv::KeyPoint> kp2,
                                        const poselib::ConfigUSAC &cfg) {
    const int rc = d->addNewCorrespondences(matches, kp1, kp2, cfg);
    d->publish();
    return rc;
}

size_t StereoRefine::getCorrespondencePoolSize() { return d->correspondencePool.size(); }
double StereoRefine::inlierThreshold() const { return d->th; }
size_t StereoRefine::nrInliersNew() const { return d->nr_inliers_new; }
size_t StereoRefine::nrCorrsNew() const { return d->nr_corrs_new; }
size_t StereoRefine::nrEstimations() const { return d->nrEstimation; }
size_t StereoRefine::skipCounter() const { return d->skipCount; }
size_t StereoRefine::poseHistorySize() const { return d->pose_history.size(); }
const cv::Mat &StereoRefine::maskENew() const { return d->mask_E_new_mat; }

}  // namespace poselib
