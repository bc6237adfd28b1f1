// facade.cpp -- C++ drop-in facade (reference signatures) over the C ABI of libmlpl_hip.so.  Host glue only: argument
// checks, cv::Mat <-> pointer plumbing and the reference's error behaviour (return codes, cv::Exception, exit(1)).
#include <algorithm>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <ctime>
#include <iostream>
#include <mutex>
#include <string>

#include "matchinglib_poselib/matchinglib_matchers.h"
#include "matchinglib_poselib/pose_estim.h"
#include "matchinglib_poselib/stereo_pose_refinement.h"
#include "facade_internal.h"
#include "mlpl_c.h"

namespace {

struct CtxHolder {
    mlpl_ctx *ctx = nullptr;
    ~CtxHolder() {
        if (ctx) mlpl_ctx_destroy(ctx);
    }
};

mlpl_ctx *default_ctx() {
    static thread_local CtxHolder h;
    if (!h.ctx) {
        int dev = 0;
        if (const char *e = std::getenv("MLPL_DEVICE")) dev = std::atoi(e);
        if (mlpl_ctx_create(dev, &h.ctx) != MLPL_OK) {
            // no CPU fallback: the drop-in fails loudly when the GPU path is unavailable
            throw cv::Exception(std::string("mlpl_ctx_create failed: ") + mlpl_last_error());
        }
        // MLPL_OPTIONS="name=value,name=value": tuning knobs of mlpl_set_option for callers that only see the reference's API
        if (const char *e = std::getenv("MLPL_OPTIONS")) {
            std::string opts(e);
            size_t pos = 0;
            while (pos < opts.size()) {
                const size_t end = std::min(opts.find(',', pos), opts.size());
                const std::string kv = opts.substr(pos, end - pos);
                const size_t eq = kv.find('=');
                if (eq == std::string::npos || mlpl_set_option(h.ctx, kv.substr(0, eq).c_str(), std::atoi(kv.c_str() + eq + 1)) != MLPL_OK)
                    throw cv::Exception("MLPL_OPTIONS: cannot apply '" + kv + "'");
                pos = end + 1;
            }
        }
    }
    return h.ctx;
}

thread_local bool g_seed_fixed = false;
thread_local unsigned g_seed = 0;

// contiguous n x 2 CV_64F copy of a point matrix (the reference converts with convertTo(CV_64F))
std::vector<double> points64(cv::InputArray pa, int &n) {
    const cv::Mat p = pa.getMat();
    CV_Assert(p.cols == 2 && (p.type() == CV_64F || p.type() == CV_32F));
    n = p.rows;
    std::vector<double> out((size_t)n * 2);
    for (int i = 0; i < n; ++i)
        for (int c = 0; c < 2; ++c) out[2 * i + c] = p.type() == CV_64F ? p.at<double>(i, c) : (double)p.at<float>(i, c);
    return out;
}

}  // namespace

mlpl_ctx *mlpl_facade_default_ctx() { return default_ctx(); }

namespace matchinglib {

int getMatches(const std::vector<cv::KeyPoint> &keypoints1, const std::vector<cv::KeyPoint> &keypoints2,
               cv::Mat const &descriptors1, cv::Mat const &descriptors2, cv::Size /*imgSi*/,
               std::vector<cv::DMatch> &finalMatches, std::string const &matcher_name, bool VFCrefine, bool ratioTest,
               std::string const & /*descriptor_name*/, std::string /*idxPars_NMSLIB*/, std::string /*queryPars_NMSLIB*/,
               const size_t /*nr_threads*/) {
    CV_Assert(descriptors1.type() == descriptors2.type());  // matchers.cpp:119
    if (keypoints1.size() < 15 || keypoints2.size() < 15) {
        std::cout << "Too less keypoits!" << std::endl;
        return -4;
    }
    if ((int)keypoints1.size() != descriptors1.rows || (int)keypoints2.size() != descriptors2.rows) {
        std::cout << "Number of descriptors must be equal to the number of keypoints!" << std::endl;
        return -1;
    }
    finalMatches.clear();
    const bool nms = matcher_name == "BRUTEFORCENMS";
    if (matcher_name != "LINEAR" && !nms) {
        std::cout << "Matcher " << matcher_name << " is not supported." << std::endl;
        return -2;
    }
    if (descriptors1.type() != CV_32F && descriptors1.type() != CV_8U) {
        std::cout << (nms ? "Wrong descriptor data type for BRUTEFORCENMS! Must be 32bit float or 8bit unsigned char here."
                          : "Format of descriptors not supported!")
                  << std::endl;
        return -1;
    }
    if (descriptors1.cols != descriptors2.cols) return -1;
    if (VFCrefine) {
        std::cout << "VFC refinement is not part of the MI355X hot path." << std::endl;
        return -2;
    }
    std::vector<mlpl_dmatch> out((size_t)descriptors1.rows);
    int n_out = 0;
    // cv::Mat::step is honoured, so non-continuous Mats (ROIs) are handled (the reference silently mis-reads them,
    // matchers.cpp:567-568 uses .data with rows x cols)
    const int rc = (nms ? mlpl_get_matches_bruteforce_nms : mlpl_get_matches_linear)(
        default_ctx(), (int)keypoints1.size(), (int)keypoints2.size(), descriptors1.data, descriptors1.rows, descriptors1.step,
        descriptors2.data, descriptors2.rows, descriptors2.step, descriptors1.cols, descriptors1.type(), ratioTest ? 1 : 0,
        out.data(), &n_out);
    if (rc != 0 && rc != -3) {
        if (rc == -1 || rc == -4) return rc;
        throw cv::Exception(std::string("mlpl_get_matches_linear: ") + mlpl_last_error());
    }
    finalMatches.resize((size_t)n_out);
    static_assert(sizeof(cv::DMatch) == sizeof(mlpl_dmatch), "DMatch layout");
    if (n_out) std::memcpy((void *)finalMatches.data(), out.data(), (size_t)n_out * sizeof(mlpl_dmatch));
    if (rc == -3) std::cout << "Too less remaining matches using the " << matcher_name << " matcher." << std::endl;
    return rc;
}

}  // namespace matchinglib

namespace poselib {

void setRansacSeed(unsigned seed) {
    g_seed_fixed = true;
    g_seed = seed;
}
void clearRansacSeed() { g_seed_fixed = false; }

namespace {
std::mutex g_arrsac_mutex;
uint64_t g_arrsac_rng[2] = {0xffffffffull, 0xffffffffull};  // cv::RNG's default state
}  // namespace

void setArrsacRngState(uint64_t prosac_state, uint64_t uniform_state) {
    std::lock_guard<std::mutex> lock(g_arrsac_mutex);
    g_arrsac_rng[0] = prosac_state, g_arrsac_rng[1] = uniform_state;
}
void getArrsacRngState(uint64_t *prosac_state, uint64_t *uniform_state) {
    std::lock_guard<std::mutex> lock(g_arrsac_mutex);
    *prosac_state = g_arrsac_rng[0], *uniform_state = g_arrsac_rng[1];
}

bool estimateEssentialMat(cv::OutputArray E, cv::InputArray p1, cv::InputArray p2, const std::string &method,
                          double threshold, bool refine, cv::OutputArray mask) {
    if (method == "RANSAC" || method == "LMEDS" || method == "ARRSAC") {
        if (!E.needed()) return false;  // five-point.cpp:143-144
        int n1 = 0, n2 = 0;
        std::vector<double> a = points64(p1, n1), b = points64(p2, n2);
        CV_Assert(n1 >= 5 && n1 == n2);  // five-point.cpp:81
        if (n1 == 5) {
            // five-point.cpp:108-114: the minimal case runs the solver once and returns up to 10 stacked 3x3 matrices
            const int32_t samples[5] = {0, 1, 2, 3, 4};
            double Es[90];
            int32_t nm = 0;
            if (mlpl_solve_5pt(default_ctx(), a.data(), b.data(), 5, samples, 1, Es, &nm) != MLPL_OK)
                throw cv::Exception(std::string("mlpl_solve_5pt: ") + mlpl_last_error());
            E.create(3 * nm, 3, CV_64F);
            cv::Mat Em = E.getMat();
            for (int i = 0; i < 9 * nm; ++i) Em.at<double>(i / 3, i % 3) = Es[i];
            if (mask.needed()) {
                mask.create(1, 5, CV_8U);
                cv::Mat mm = mask.getMat();
                std::memset(mm.ptr<uint8_t>(0), 1, 5);
            }
            return true;
        }
        double Ev[9];
        std::vector<uint8_t> m((size_t)n1);
        int ninl = 0, iters = 0;
        const unsigned seed = g_seed_fixed ? g_seed : (unsigned)std::time(nullptr);  // modelest.cpp:58
        // RANSAC: 1000 iterations + optional refit (pose_estim.cpp:870-873); LMEDS: 2000 iterations, no refit (:874-877,
        // five-point.cpp:125-129)
        int rc;
        if (method == "ARRSAC") {
            // pose_estim.cpp:866-869: ARRSAC with the pseudo-Huber refinement as its `lesqu` step.  The two cv::RNG streams are
            // process-wide in the reference (function-local statics of the samplers); so is this pair.
            std::lock_guard<std::mutex> lock(g_arrsac_mutex);
            rc = mlpl_arrsac_essential(default_ctx(), a.data(), b.data(), n1, threshold, refine ? 1 : 0, g_arrsac_rng, Ev, m.data(), &ninl);
        } else {
            rc = method == "LMEDS"
                     ? mlpl_lmeds_essential(default_ctx(), a.data(), b.data(), n1, 0.999, 2000, seed, Ev, m.data(), &ninl, nullptr)
                     : mlpl_ransac_essential(default_ctx(), a.data(), b.data(), n1, threshold, 0.999, 1000, refine ? 1 : 0, seed, Ev,
                                             m.data(), &ninl, &iters);
        }
        if (rc == MLPL_E_FAILED) return false;
        if (rc != MLPL_OK) throw cv::Exception(std::string("estimateEssentialMat: ") + mlpl_last_error());
        if (mask.needed()) {
            mask.create(1, n1, CV_8U);
            cv::Mat mm = mask.getMat();
            std::memcpy(mm.ptr<uint8_t>(0), m.data(), (size_t)n1);
        }
        E.create(3, 3, CV_64F);
        cv::Mat Em = E.getMat();
        for (int i = 0; i < 9; ++i) Em.at<double>(i / 3, i % 3) = Ev[i];
        return true;
    }
    if (method == "USAC") {
        std::cout << "USAC must be executed by function estimateEssentialOrPoseUSAC as it needs additional paramters! Exiting."
                  << std::endl;
        std::exit(1);  // pose_estim.cpp:878-882
    }
    std::cout << "Either there is a typo in the specified robust estimation method or the method is not supported. Exiting."
              << std::endl;
    std::exit(1);  // pose_estim.cpp:883-887
}

void robustEssentialRefine(cv::InputArray points1, cv::InputArray points2, cv::InputArray E_init, cv::Mat &E_refined, double th,
                           unsigned int iters, bool makeClosestE, double *sumSqrErr_init, double *sumSqrErr, cv::OutputArray errors,
                           cv::InputOutputArray mask, int model, bool tryOrientedEpipolar, bool normalizeCorrs) {
    if (iters != 0 || !makeClosestE || model != 0 || tryOrientedEpipolar || normalizeCorrs || sumSqrErr_init || sumSqrErr || errors.needed())
        throw cv::Exception("robustEssentialRefine (MI355X hot-path library): built for model 0, iters 0, makeClosestE, without normalisation, "
                            "oriented-epipolar test and error outputs");
    int n1 = 0, n2 = 0;
    std::vector<double> a = points64(points1, n1), b = points64(points2, n2);
    CV_Assert(n1 == n2 && n1 > 0);
    const cv::Mat E0 = E_init.getMat();
    CV_Assert(E0.rows == 3 && E0.cols == 3 && E0.type() == CV_64F);
    double Ei[9], Eo[9];
    for (int i = 0; i < 9; ++i) Ei[i] = E0.at<double>(i / 3, i % 3);
    std::vector<uint8_t> m;
    if (mask.needed() && !mask.empty()) {
        const cv::Mat mm = mask.getMat();
        CV_Assert(mm.rows * mm.cols == n1 && mm.type() == CV_8U);
        m.resize((size_t)n1);
        for (int i = 0; i < n1; ++i) m[i] = mm.rows == 1 ? mm.at<uint8_t>(0, i) : mm.at<uint8_t>(i, 0);
    }
    int info[2] = {0, 0};
    if (mlpl_robust_essential_refine(default_ctx(), a.data(), b.data(), n1, m.empty() ? nullptr : m.data(), Ei, th, Eo, info) != MLPL_OK)
        throw cv::Exception(std::string("robustEssentialRefine: ") + mlpl_last_error());
    if (info[1] == 2) std::cout << "There are too less points for a refinement left!" << std::endl;  // pose_estim.cpp:411-416
    else if (info[1] == 3) std::cout << "Refinement failed!" << std::endl;
    cv::Mat out(3, 3, CV_64F);
    for (int i = 0; i < 9; ++i) out.at<double>(i / 3, i % 3) = Eo[i];
    E_refined = out;
}

// ---- USAC (pose_estim.cpp:1737-2244, pose_helper.cpp:2830-2923) --------------------------------------------------------------------------
namespace {
void usac_history_stats(const std::vector<double> &vals, double *mean, double *stddev);  // getStatsfromVec(vals, &stats, true): defined below
// convex hull (Andrew's monotone chain) + cv::contourArea's shoelace sum in double: the hull of a point set is unique, so is its area
// up to the summation order
double hull_area(std::vector<cv::Point2f> pts) {
    std::sort(pts.begin(), pts.end(), [](const cv::Point2f &a, const cv::Point2f &b) { return a.x < b.x || (a.x == b.x && a.y < b.y); });
    pts.erase(std::unique(pts.begin(), pts.end(), [](const cv::Point2f &a, const cv::Point2f &b) { return a.x == b.x && a.y == b.y; }), pts.end());
    const size_t n = pts.size();
    if (n < 3) return 0.0;
    auto cross = [](const cv::Point2f &o, const cv::Point2f &a, const cv::Point2f &b) {
        return ((double)a.x - o.x) * ((double)b.y - o.y) - ((double)a.y - o.y) * ((double)b.x - o.x);
    };
    std::vector<cv::Point2f> h(2 * n);
    size_t k = 0;
    for (size_t i = 0; i < n; ++i) {
        while (k >= 2 && cross(h[k - 2], h[k - 1], pts[i]) <= 0) k--;
        h[k++] = pts[i];
    }
    for (size_t i = n - 1, t = k + 1; i > 0; --i) {
        while (k >= t && cross(h[k - 2], h[k - 1], pts[i - 1]) <= 0) k--;
        h[k++] = pts[i - 1];
    }
    h.resize(k - 1);
    double a00 = 0;
    cv::Point2f prev = h.back();
    for (const cv::Point2f &p : h) {
        a00 += (double)prev.x * p.y - (double)prev.y * p.x;
        prev = p;
    }
    return std::fabs(a00 * 0.5);
}
bool usac_near_zero(double d) { return (d < 1e-3) && (d > -1e-3); }  // poselib::nearZero (pose_helper.h:82-87)
std::mutex g_usac_mutex;
struct UsacHistory {  // the function-local statics of estimateEssentialOrPoseUSAC (:1755-1762) and estimateEssentialMatUsac (:321-323)
    double sprt_delta_old = 0, sprt_delta_new = 0, sprt_epsilon_old = 0, sprt_epsilon_new = 0;
    double delta_history[20] = {0}, epsilon_history[20] = {0};
    int historyCnt = 0;
    bool historyBufFull = false, statistic_valid = false;
    double delta_mean = 0, delta_std = 0, epsilon_mean = 0, epsilon_std = 0;
    unsigned numhyps = 0, modelcount = 0;
    double avgModels = 6;
} g_usac_hist;
std::once_flag g_usac_notice[4];
}  // namespace

double estimateSprtDeltaInit(const std::vector<cv::DMatch> &matches, const std::vector<cv::KeyPoint> &kp1,
                             const std::vector<cv::KeyPoint> &kp2, const double &th, const cv::Size &imgSize) {
    std::vector<cv::Point2f> points1, points2;
    for (const cv::DMatch &m : matches) {
        points1.push_back(kp1[(size_t)m.queryIdx].pt);
        points2.push_back(kp2[(size_t)m.trainIdx].pt);
    }
    double area[2] = {hull_area(points1), hull_area(points2)};
    area[0] = area[0] > area[1] ? area[1] : area[0];
    double maxEpipoleArea = std::sqrt((double)(imgSize.width * imgSize.width + imgSize.height * imgSize.height));
    maxEpipoleArea *= 2 * th;
    area[0] = area[0] < (6 * maxEpipoleArea) ? (6 * maxEpipoleArea) : area[0];
    double sprt_delta = maxEpipoleArea / area[0];
    return sprt_delta < 0.001 ? 0.001 : sprt_delta;
}

double estimateSprtEpsilonInit(const std::vector<cv::DMatch> &matches, const unsigned int &nrMatchesVfcFiltered) {
    double e = 0.8 * (double)nrMatchesVfcFiltered / (double)matches.size();
    e = e > 0.4 ? 0.4 : e;
    return e < 0.1 ? 0.1 : e;
}

void getSortedMatchIdx(std::vector<cv::DMatch> matches, std::vector<unsigned int> &sortedMatchIdx) {
    size_t i = 0;
    for (i = 0; i < matches.size(); i++)
        if (matches[i].queryIdx != static_cast<int>(i)) break;
    if (i < matches.size())
        for (i = 0; i < matches.size(); i++) matches[i].queryIdx = static_cast<int>(i);
    std::sort(matches.begin(), matches.end(), [](cv::DMatch const &first, cv::DMatch const &second) { return first.distance < second.distance; });
    sortedMatchIdx.resize(matches.size());
    for (i = 0; i < matches.size(); i++) sortedMatchIdx[i] = (unsigned int)matches[i].queryIdx;
}

void resetUsacHistory() {
    std::lock_guard<std::mutex> lock(g_usac_mutex);
    g_usac_hist = UsacHistory();
}

int estimateEssentialOrPoseUSAC(const cv::Mat &p1, const cv::Mat &p2, cv::OutputArray E, double th, ConfigUSAC &cfg, bool &isDegenerate,
                                cv::OutputArray inliers, cv::OutputArray R_degenerate, cv::OutputArray inliers_degenerate_R,
                                cv::OutputArray R, cv::OutputArray t, bool verbose) {
    (void)verbose;
    std::lock_guard<std::mutex> lock(g_usac_mutex);  // the history is process-wide: calls are serialised like the reference's statics imply
    UsacHistory &H = g_usac_hist;
    // Options of ConfigUSAC that this library does not build are REFUSED with the reference's own message and -1 (what the reference
    // returns for a configuration it does not support, pose_estim.cpp:1789-1797, 2001-2003), never served by another algorithm.
    // MLPL_OPTIONS=usac_substitute (environment, read once) brings back the substitution of earlier rounds for callers that ask for it:
    // POSE_EIG_KNEIP -> the 5-point solver, REF_8PT_PSEUDOHUBER / REF_EIG_KNEIP(_WEIGHTS) -> REF_WEIGHTS, DEGEN_QDEGSAC -> no check.
    static const bool substitute = [] {
        const char *o = std::getenv("MLPL_OPTIONS");
        return o && std::strstr(o, "usac_substitute") != nullptr;
    }();
    int estimator, refine;
    switch (cfg.estimator) {
        case POSE_NISTER: estimator = 0; break;
        case POSE_STEWENIUS: estimator = 2; break;
        case POSE_EIG_KNEIP:
            if (!substitute) {
                std::cout << "Estimator not supported!" << std::endl;  // (Kneip's eigensolver as the minimal solver is not built)
                return -1;
            }
            std::call_once(g_usac_notice[0], [] { std::cout << "USAC (MI355X hot-path library, MLPL_OPTIONS=usac_substitute): Kneip's eigensolver is not built; the 5-point solver is used." << std::endl; });
            estimator = 0;
            break;
        default: std::cout << "Estimator not supported!" << std::endl; return -1;
    }
    switch (cfg.refinealg) {
        case REF_WEIGHTS: refine = 0; break;
        case REF_STEWENIUS: refine = 4; break;
        case REF_STEWENIUS_WEIGHTS: refine = 5; break;
        case REF_NISTER: refine = 6; break;
        case REF_NISTER_WEIGHTS: refine = 7; break;
        case REF_8PT_PSEUDOHUBER:
        case REF_EIG_KNEIP:
        case REF_EIG_KNEIP_WEIGHTS:
            if (!substitute) {
                std::cout << "Refinement algorithm not supported!" << std::endl;  // (built: REF_WEIGHTS and the four 5-point refinements)
                return -1;
            }
            std::call_once(g_usac_notice[1], [] { std::cout << "USAC (MI355X hot-path library, MLPL_OPTIONS=usac_substitute): this inner refinement is not built; REF_WEIGHTS is used." << std::endl; });
            refine = 0;
            break;
        default: std::cout << "Refinement algorithm not supported!" << std::endl; return -1;
    }
    CV_Assert(p1.cols == 2 && p2.cols == 2 && p1.rows == p2.rows && p1.type() == CV_64F && p2.type() == CV_64F);  // usac_estimations.cpp:315
    const int n = p1.rows;
    CV_Assert(!cfg.matches || cfg.matches->empty() || (size_t)n == cfg.matches->size());  // :316

    // initial delta / epsilon of the sequential test (:1799-1990)
    double prosac_beta = 0.09, sprt_delta = 0.05, sprt_epsilon = 0.15;
    const double statStdDivTh[2] = {0.1, 0.2}, relativeDifferenceTh[2] = {0.33, 0.4}, relDiffArithToStd = 0.45, ignoreRelDiffATSTh[2] = {0.1, 0.1};
    const bool want_delta = (cfg.automaticSprtInit & SPRT_DELTA_AUTOM_INIT) != 0, want_eps = (cfg.automaticSprtInit & SPRT_EPSILON_AUTOM_INIT) != 0;
    if (cfg.automaticSprtInit < 0 || cfg.automaticSprtInit > 3) {
        std::cout << "Method for SPRT initialization not supported!" << std::endl;
        return -1;
    }
    if ((want_delta || want_eps || !cfg.noAutomaticProsacParamters) && (want_delta || want_eps) &&
        (!cfg.matches || (want_delta && (!cfg.keypoints1 || !cfg.keypoints2)))) {
        std::cout << "USAC: the automatic SPRT initialisation needs cfg.matches and the keypoints!" << std::endl;
        return -1;
    }
    if (want_delta) {
        if (H.sprt_delta_old == 0 || H.sprt_delta_new == 0) {
            sprt_delta = estimateSprtDeltaInit(*cfg.matches, *cfg.keypoints1, *cfg.keypoints2, cfg.th_pixels, cfg.imgSize);
            H.sprt_delta_new = sprt_delta;
        } else if (!H.statistic_valid || (H.delta_std > statStdDivTh[0]) ||
                   ((H.delta_mean > ignoreRelDiffATSTh[0]) && (H.delta_std > relDiffArithToStd * H.delta_mean))) {
            if (std::abs((H.sprt_delta_old - H.sprt_delta_new) / H.sprt_delta_old) < relativeDifferenceTh[0])
                sprt_delta = H.sprt_delta_new;
            else
                sprt_delta = estimateSprtDeltaInit(*cfg.matches, *cfg.keypoints1, *cfg.keypoints2, cfg.th_pixels, cfg.imgSize);
        } else
            sprt_delta = H.delta_mean;
    }
    if (want_eps) {
        if (H.sprt_epsilon_old == 0 || H.sprt_epsilon_new == 0) {
            sprt_epsilon = estimateSprtEpsilonInit(*cfg.matches, cfg.nrMatchesVfcFiltered);
            H.sprt_epsilon_new = sprt_epsilon;
        } else if (!H.statistic_valid || (H.epsilon_std > statStdDivTh[1]) ||
                   ((H.epsilon_mean > ignoreRelDiffATSTh[1]) && (H.epsilon_std > relDiffArithToStd * H.epsilon_mean))) {
            if (std::abs((H.sprt_epsilon_old - H.sprt_epsilon_new) / H.sprt_epsilon_old) < relativeDifferenceTh[1])
                sprt_epsilon = H.sprt_epsilon_new;
            else
                sprt_epsilon = estimateSprtEpsilonInit(*cfg.matches, cfg.nrMatchesVfcFiltered);
        } else
            sprt_epsilon = H.epsilon_mean;
    }
    if (!cfg.noAutomaticProsacParamters) prosac_beta = sprt_delta;
    std::vector<unsigned int> sortedMatchIdx;
    if (cfg.matches) getSortedMatchIdx(*cfg.matches, sortedMatchIdx);
    if (cfg.degeneracyCheck != DEGEN_NO_CHECK && cfg.degeneracyCheck != DEGEN_QDEGSAC && cfg.degeneracyCheck != DEGEN_USAC_INTERNAL) {
        std::cout << "Mothod for checking degeneracy not available!" << std::endl;
        return -1;
    }
    if (cfg.degeneracyCheck == DEGEN_QDEGSAC) {
        if (!substitute) {
            std::cout << "Mothod for checking degeneracy not available!" << std::endl;  // (QDEGSAC is not built)
            return -1;
        }
        std::call_once(g_usac_notice[2], [] { std::cout << "USAC (MI355X hot-path library, MLPL_OPTIONS=usac_substitute): QDEGSAC is not built; the estimation runs without a degeneracy check." << std::endl; });
    }
    const bool eight_point = refine == 0;
    if (cfg.degeneracyCheck == DEGEN_USAC_INTERNAL && eight_point)
        std::call_once(g_usac_notice[3], [] { std::cout << "USAC (MI355X hot-path library): the rotation-only / no-motion tests and the model upgrade are built; the homography test the reference adds for the 8-point refinements is not." << std::endl; });

    // estimateEssentialMatUsac (usac_estimations.cpp:283-470)
    mlpl_usac_params P;
    mlpl_usac_default_params(&P, th);
    P.estimator = estimator, P.refine = refine;
    P.seed = g_seed_fixed ? g_seed : (unsigned)std::time(nullptr);
    P.prosac_beta = prosac_beta, P.sprt_delta = sprt_delta, P.sprt_epsilon = sprt_epsilon;
    if (H.numhyps == 0 || H.modelcount == 0)
        P.sprt_mS = estimator == 0 ? 8.5 : H.avgModels;
    else {
        H.avgModels = (double)H.modelcount / (double)H.numhyps;
        P.sprt_mS = H.avgModels;
    }
    P.sprt_tM = cfg.estimator == POSE_STEWENIUS ? 2736.0 : 2314.0;
    P.sorted_idx = sortedMatchIdx.empty() ? nullptr : sortedMatchIdx.data();
    if (cfg.degeneracyCheck == DEGEN_USAC_INTERNAL) {  // usac_estimations.cpp:367-375, 443-456
        P.check_degeneracy = eight_point ? 3 : 1;
        P.th_pixels = cfg.th_pixels, P.focal_length = cfg.focalLength;
    }
    std::vector<double> a((size_t)n * 2), b((size_t)n * 2);
    for (int i = 0; i < n; ++i) {
        a[2 * i] = p1.at<double>(i, 0), a[2 * i + 1] = p1.at<double>(i, 1);
        b[2 * i] = p2.at<double>(i, 0), b[2 * i + 1] = p2.at<double>(i, 1);
    }
    double Ev[9], res[12];
    std::vector<uint8_t> m((size_t)std::max(n, 1));
    const int rc = mlpl_usac_essential(default_ctx(), a.data(), b.data(), n, &P, Ev, m.data(), res);
    if (rc == MLPL_E_FAILED) {
        std::cout << "USAC failed!" << std::endl;
        return -2;
    }
    if (rc != MLPL_OK) throw cv::Exception(std::string("estimateEssentialOrPoseUSAC: ") + mlpl_last_error());
    H.numhyps += (unsigned)res[1];
    H.modelcount += (unsigned)res[2];
    const double sprt_epsilon_res = (res[1] > 2000 && res[9] > 0.2) ? res[9] / 2.0 : res[9];  // :460-467
    const double sprt_delta_res = res[8];
    if (E.needed()) {
        E.create(3, 3, CV_64F);
        cv::Mat Em = E.getMat();
        for (int i = 0; i < 9; ++i) Em.at<double>(i / 3, i % 3) = Ev[i];
    }
    if (inliers.needed()) {
        inliers.create(1, n, CV_8U);
        cv::Mat mm = inliers.getMat();
        std::memcpy(mm.ptr<uint8_t>(0), m.data(), (size_t)n);
    }
    isDegenerate = false;
    if (R.needed()) R.release();  // :2036-2042: only Kneip's eigensolver delivers R, t
    if (t.needed()) t.release();
    if (P.check_degeneracy) {  // the decision of DEGEN_USAC_INTERNAL (:2101-2133) on what estimateEssentialMatUsac reports (:564-636)
        double info[16];
        std::vector<uint8_t> fr((size_t)std::max(n, 1)), fn((size_t)std::max(n, 1));
        if (mlpl_usac_last_degeneracy(default_ctx(), info, fr.data(), fn.data(), n) != MLPL_OK)
            throw cv::Exception(std::string("estimateEssentialOrPoseUSAC: ") + mlpl_last_error());
        const double nrInliers = res[5], fraction_inliers = nrInliers / (double)n;
        const double fraction_R = info[1] > 2 ? (nrInliers > 0 ? info[1] / nrInliers : 0.0) : 0.0;
        const double fraction_noMot = info[2] > 1 ? (nrInliers > 0 ? info[2] / nrInliers : 0.0) : 0.0;
        auto put_mask = [&](const std::vector<uint8_t> &f) {
            if (!inliers_degenerate_R.needed()) return;
            inliers_degenerate_R.create(1, n, CV_8U);
            cv::Mat mm = inliers_degenerate_R.getMat();
            std::memcpy(mm.ptr<uint8_t>(0), f.data(), (size_t)n);
        };
        if (cfg.degenDecisionTh * fraction_inliers < fraction_R) {
            isDegenerate = true;
            if (R_degenerate.needed() && info[1] > 2) {
                R_degenerate.create(3, 3, CV_64F);
                cv::Mat Rm = R_degenerate.getMat();
                for (int i = 0; i < 9; ++i) Rm.at<double>(i / 3, i % 3) = info[4 + i];
            }
            put_mask(fr);
        } else if (cfg.degenDecisionTh * fraction_inliers < fraction_noMot) {
            isDegenerate = true;
            if (R_degenerate.needed()) {
                R_degenerate.create(3, 3, CV_64F);
                cv::Mat Rm = R_degenerate.getMat();
                for (int i = 0; i < 9; ++i) Rm.at<double>(i / 3, i % 3) = (i % 4 == 0) ? 1.0 : 0.0;
            }
            put_mask(fn);
        }
    }

    // the carried-over statistics (:2201-2224)
    H.sprt_delta_old = H.sprt_delta_new;
    H.sprt_delta_new = sprt_delta_res;
    H.sprt_epsilon_old = H.sprt_epsilon_new;
    H.sprt_epsilon_new = sprt_epsilon_res;
    if (!usac_near_zero(sprt_delta_res) && !usac_near_zero(sprt_epsilon_res)) {
        H.delta_history[H.historyCnt] = sprt_delta_res;
        H.epsilon_history[H.historyCnt] = sprt_epsilon_res;
        H.historyCnt = (H.historyCnt + 1) % 20;
        if (H.historyCnt == 0) H.historyBufFull = true;
        const int cnt = H.historyBufFull ? 20 : (H.historyCnt > 5 ? H.historyCnt : 0);
        if (cnt) {
            usac_history_stats(std::vector<double>(H.delta_history, H.delta_history + cnt), &H.delta_mean, &H.delta_std);
            usac_history_stats(std::vector<double>(H.epsilon_history, H.epsilon_history + cnt), &H.epsilon_mean, &H.epsilon_std);
            if (!H.historyBufFull) H.statistic_valid = true;
        }
    }
    return 0;
}

// ---- AutoThEpi (pose_estim.cpp:81-300) ------------------------------------------------------------------------------------------------
namespace {
struct FullStats {
    double medErr = 0, arithErr = 0, arithStd = 0, medStd = 0;
};
// getStatsfromVec (pose_helper.cpp:358-413) with its quartile rejection and the rounding of a tiny variance
FullStats fullStatsFromVec(const std::vector<double> &vals, bool rejQuartiles, bool roundStd = true) {
    FullStats st;
    if (vals.empty()) return st;
    int n = (int)vals.size();
    const int qrt_si = (int)std::floor(0.25 * (double)n);
    std::vector<double> v(vals);
    std::sort(v.begin(), v.end());
    st.medErr = (n % 2) ? v[(n - 1) / 2] : (v[n / 2] + v[n / 2 - 1]) / 2.0;
    double sum = 0, sum2 = 0;
    std::vector<double> mad;
    for (int i = rejQuartiles ? qrt_si : 0; i < (rejQuartiles ? (n - qrt_si) : n); i++) {
        sum += v[i];
        sum2 += v[i] * v[i];
        mad.push_back(std::abs(v[i] - st.medErr));
    }
    if (rejQuartiles) n -= 2 * qrt_si;
    st.arithErr = sum / (double)n;
    std::sort(mad.begin(), mad.end());
    st.medStd = (n % 2) ? 1.4826 * mad[(n - 1) / 2] : 1.4826 * (mad[n / 2] + mad[n / 2 - 1]) / 2.0;
    const double hlp = sum2 - (double)n * st.arithErr * st.arithErr;
    st.arithStd = (roundStd && std::abs(hlp) < 1e-6) ? 0.0 : std::sqrt(hlp / ((double)n - 1.0));
    return st;
}
void usac_history_stats(const std::vector<double> &vals, double *mean, double *stddev) {
    const FullStats st = fullStatsFromVec(vals, true);
    *mean = st.arithErr, *stddev = st.arithStd;
}
}  // namespace

double AutoThEpi::setCorrTH(double thresh, bool useImgCoordSystem, bool storeGlobally) {  // :279-300
    double pix = useImgCoordSystem ? thresh : thresh / pixToCamFact;
    if (pix < corr_filt_min_pix_th) pix = corr_filt_min_pix_th;
    else if (pix > MAX_PIX_TH) pix = MAX_PIX_TH;
    const double cam = pix * pixToCamFact;
    if (storeGlobally) corr_filt_pix_th = pix, corr_filt_cam_th = cam;
    return useImgCoordSystem ? pix : cam;
}

double AutoThEpi::estimateThresh(cv::InputArray p1, cv::InputArray p2, cv::InputArray E_, bool useImgCoordSystem, bool storeGlobally) {  // :196-262
    if (useImgCoordSystem) throw cv::Exception("AutoThEpi::estimateThresh: image coordinates need camera matrices the class does not hold");
    int n1 = 0, n2 = 0;
    std::vector<double> a = points64(p1, n1), b = points64(p2, n2);
    CV_Assert(n1 == n2 && n1 > 0);
    const cv::Mat E = E_.getMat();
    double Ev[9];
    for (int i = 0; i < 9; ++i) Ev[i] = E.at<double>(i / 3, i % 3);
    std::vector<double> error((size_t)n1);
    std::vector<uint8_t> m((size_t)n1);
    if (mlpl_get_inliers_strict(default_ctx(), a.data(), b.data(), n1, Ev, 1.0, error.data(), m.data()) < 0)  // computeReprojError2 on the device
        throw cv::Exception(std::string("AutoThEpi::estimateThresh: ") + mlpl_last_error());
    for (double &e : error) e = std::sqrt(e);
    double th = corr_filt_cam_th;
    double maxInlDist = 4.0 * th;
    maxInlDist = maxInlDist > 5.0 * pixToCamFact ? (5.0 * pixToCamFact) : maxInlDist;
    const double r1 = *std::max_element(error.begin(), error.end()) - *std::min_element(error.begin(), error.end());
    const FullStats qp = fullStatsFromVec(error, r1 > maxInlDist);
    double th_tmp;
    if ((qp.arithErr / qp.medErr > 2.0) || (qp.arithErr / qp.medErr < 0.5)) th_tmp = qp.medErr + 3.0 * qp.medStd;
    else th_tmp = qp.arithErr + 3.0 * qp.arithStd;
    if ((th_tmp < 5.0 * th) || (th_tmp < 4.0 * PIX_MIN_GOOD_TH)) {  // (sic) the second test compares camera units with pixels: always true
        th = setCorrTH(th_tmp, false, storeGlobally);
    } else if (th < (MAX_PIX_TH / 2) * pixToCamFact) {
        th = setCorrTH(th * 2.0, false, storeGlobally);
    } else {
        th = setCorrTH(corr_filt_min_pix_th, true, storeGlobally) * pixToCamFact;
    }
    return th;
}

int AutoThEpi::estimateEVarTH(cv::InputArray p1, cv::InputArray p2, cv::OutputArray E, cv::OutputArray mask, double *th, int *nrgoodPts) {  // :81-178
    bool th_sem[3] = {true, true, false};
    int th_fail_cnt = 2;
    double th_failed = *th, th_old;
    cv::Mat mask_, E_;
    int n1 = 0;
    {
        int n2 = 0;
        points64(p1, n1);
        (void)n2;
    }
    auto count = [](const cv::Mat &m) {
        int c = 0;
        for (int i = 0; i < m.rows * m.cols; ++i) c += (m.rows == 1 ? m.at<uint8_t>(0, i) : m.at<uint8_t>(i, 0)) != 0;
        return c;
    };
    // findEssentialMat(E, p1, p2, ARRSAC, 0.99, th, mask, true, robustEssentialRefine)
    auto find = [&](double t) {
        cv::Mat e, m;
        if (!estimateEssentialMat(e, p1, p2, "ARRSAC", t, true, m)) return false;
        E_ = e, mask_ = m;
        return true;
    };
    do {
        mask_ = cv::Mat();
        th_old = *th;
        if (!find(*th)) {
            if ((*th < PIX_MIN_GOOD_TH * pixToCamFact) && !th_stable && !th_sem[2]) {
                th_failed = *th;
                *th = PIX_MIN_GOOD_TH * pixToCamFact;
                th_old = *th;
                if (!find(*th)) return -1;
                th_sem[2] = true;
            } else if (th_sem[2]) {
                th_fail_cnt *= 2;
            } else {
                return -1;
            }
        } else {
            if (th_sem[2] && (th_fail_cnt > 2)) corr_filt_min_pix_th = *th / pixToCamFact;
            th_sem[2] = false;
        }
        if ((th_fail_cnt <= 2) || !th_sem[2]) *nrgoodPts = mask_.empty() ? 0 : count(mask_);
        if (!th_stable) {
            if ((th_fail_cnt <= 2) || !th_sem[2]) *th = estimateThresh(p1, p2, E_);
            else *th = th_failed * (double)th_fail_cnt;
            if (th_sem[2] && (th_failed >= *th)) *th = th_failed * (double)th_fail_cnt;
            if (!th_sem[2]) {
                if (th_old / *th > 1.0 + 1e-6) th_sem[0] = false;
                else if (th_old / *th < 1.0 - 1e-6) th_sem[1] = false;
            }
        }
    } while (((th_old / *th < 0.9) || (*th / th_old < 0.9)) &&
             ((((float)*nrgoodPts / (float)n1 < 0.67) && (th_sem[0] || th_sem[1])) || th_sem[2]));
    if (!E.needed()) return -2;
    E.create(3, 3, CV_64F);
    cv::Mat Eo = E.getMat();
    for (int i = 0; i < 9; ++i) Eo.at<double>(i / 3, i % 3) = E_.at<double>(i / 3, i % 3);
    if (mask.needed()) {
        mask.create(1, n1, CV_8U);
        cv::Mat mo = mask.getMat();
        for (int i = 0; i < n1; ++i) mo.at<uint8_t>(0, i) = mask_.rows == 1 ? mask_.at<uint8_t>(0, i) : mask_.at<uint8_t>(i, 0);
    }
    return 0;
}

int getPoseTriangPts(cv::InputArray E_, cv::InputArray p1, cv::InputArray p2, cv::OutputArray R_, cv::OutputArray t_,
                     cv::OutputArray Q_, cv::InputOutputArray mask_, const double dist, bool translatE) {
    if (!R_.needed() || !t_.needed() || !Q_.needed()) return -1;  // pose_estim.cpp:925-926
    const cv::Mat E = E_.getMat();
    CV_Assert(E.rows == 3 && E.cols == 3 && E.type() == CV_64F);
    int n1 = 0, n2 = 0;
    std::vector<double> a = points64(p1, n1), b = points64(p2, n2);
    CV_Assert(n1 == n2);
    double Ev[9], Rv[9], tv[3];
    for (int i = 0; i < 9; ++i) Ev[i] = E.at<double>(i / 3, i % 3);
    std::vector<double> Qv((size_t)std::max(n1, 1) * 3);
    std::vector<uint8_t> m;
    const bool use_mask = mask_.needed();
    if (use_mask) {
        m.assign((size_t)n1, 1);  // an empty mask is created as all ones (five-point.cpp:275-280)
        if (!mask_.empty()) {
            const cv::Mat mask = mask_.getMat();
            CV_Assert(mask.rows * mask.cols == n1 && mask.type() == CV_8U);
            for (int i = 0; i < n1; ++i) m[i] = mask.rows == 1 ? mask.at<uint8_t>(0, i) : mask.at<uint8_t>(i, 0);
        }
    }
    int rc;
    if (translatE) {
        // getTfromTransEssential (pose_helper.cpp:422-433)
        double t0[3] = {E.at<double>(1, 2), E.at<double>(2, 0), E.at<double>(0, 1)};
        const double nrm = std::sqrt(t0[0] * t0[0] + t0[1] * t0[1] + t0[2] * t0[2]);
        if (std::abs(nrm - 1.0) > 1e-3)
            for (double &v : t0) v /= nrm;
        rc = mlpl_recover_pose_translation(default_ctx(), t0, a.data(), b.data(), n1, dist, Rv, tv, Qv.data(),
                                           use_mask ? m.data() : nullptr);
    } else {
        rc = mlpl_recover_pose(default_ctx(), Ev, a.data(), b.data(), n1, dist, Rv, tv, Qv.data(), use_mask ? m.data() : nullptr);
    }
    if (rc < 0) throw cv::Exception(std::string("mlpl_recover_pose: ") + mlpl_last_error());
    R_.create(3, 3, CV_64F);
    t_.create(3, 1, CV_64F);
    Q_.create(n1, 3, CV_64F);
    cv::Mat R = R_.getMat(), t = t_.getMat(), Q = Q_.getMat();
    for (int i = 0; i < 9; ++i) R.at<double>(i / 3, i % 3) = Rv[i];
    for (int i = 0; i < 3; ++i) t.at<double>(i, 0) = tv[i];
    for (int i = 0; i < n1; ++i)
        for (int c = 0; c < 3; ++c) Q.at<double>(i, c) = Qv[(size_t)i * 3 + c];
    if (use_mask) {
        mask_.create(1, n1, CV_8U);
        cv::Mat mo = mask_.getMat();
        std::memcpy(mo.ptr<uint8_t>(0), m.data(), (size_t)n1);
    }
    return rc;
}

bool estimateRelativePose(cv::InputArray p1, cv::InputArray p2, cv::OutputArray E, cv::OutputArray R, cv::OutputArray t,
                          cv::OutputArray Q, cv::OutputArray mask, double threshold, bool refine, double dist) {
    cv::Mat m;
    if (!estimateEssentialMat(E, p1, p2, "RANSAC", threshold, refine, m)) return false;
    const cv::Mat Em = E.getMat();
    const int ng = getPoseTriangPts(Em, p1, p2, R, t, Q, m, dist, false);
    if (mask.needed()) {
        mask.create(m.rows, m.cols, CV_8U);
        cv::Mat mo = mask.getMat();
        for (int r = 0; r < m.rows; ++r) std::memcpy(mo.ptr<uint8_t>(r), m.ptr<uint8_t>(r), (size_t)m.cols);
    }
    return ng >= 0;
}

}  // namespace poselib
