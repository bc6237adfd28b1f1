// pose_estim.h -- drop-in for the hot-path part of the reference's poselib/include/poselib/pose_estim.h (defines :56-59, enums :61-92,
// ConfigUSAC :94-132, getPoseTriangPts :192-200, estimateEssentialMat :204-210).  Same names, argument order, defaults and error
// behaviour; the work runs on the MI355X through libmlpl_hip.so (include/mlpl_c.h).
#pragma once
#include <cstdint>
#include <string>
#include <vector>

#include "matchinglib_poselib/cv_compat.h"

// reference pose_estim.h:56-59
#define PIX_TH_START 0.5
#define MIN_PIX_TH ((0.25 < PIX_TH_START) ? 0.25 : PIX_TH_START)
#define MAX_PIX_TH 2.0
#define PIX_MIN_GOOD_TH 1.6

namespace poselib {

// reference pose_estim.h:61-92: the configuration vocabulary of estimateEssentialOrPoseUSAC (below).
enum UsacChkDegenType { DEGEN_NO_CHECK, DEGEN_QDEGSAC, DEGEN_USAC_INTERNAL };
enum PoseEstimator { POSE_NISTER, POSE_EIG_KNEIP, POSE_STEWENIUS };
enum RefineAlg {
    REF_WEIGHTS,
    REF_8PT_PSEUDOHUBER,
    REF_EIG_KNEIP,
    REF_EIG_KNEIP_WEIGHTS,
    REF_STEWENIUS,
    REF_STEWENIUS_WEIGHTS,
    REF_NISTER,
    REF_NISTER_WEIGHTS
};
enum RefinePostAlg {
    PR_NO_REFINEMENT = 0x0,
    PR_8PT = 0x1,
    PR_NISTER = 0x2,
    PR_STEWENIUS = 0x3,
    PR_KNEIP = 0x4,
    PR_TORR_WEIGHTS = 0x10,
    PR_PSEUDOHUBER_WEIGHTS = 0x20,
    PR_NO_WEIGHTS = 0x30
};
enum SprtInit { SPRT_DEFAULT_INIT = 0x0, SPRT_DELTA_AUTOM_INIT = 0x1, SPRT_EPSILON_AUTOM_INIT = 0x2 };

// reference pose_estim.h:94-132: every field, the reference's defaults.
struct ConfigUSAC {
    ConfigUSAC()
        : focalLength(800),
          th_pixels(0.8),
          degeneracyCheck(DEGEN_USAC_INTERNAL),
          estimator(POSE_STEWENIUS),
          refinealg(REF_STEWENIUS_WEIGHTS),
          prevalidateSample(false),
          noAutomaticProsacParamters(false),
          automaticSprtInit(SPRT_DELTA_AUTOM_INIT | SPRT_EPSILON_AUTOM_INIT),
          matches(nullptr),
          keypoints1(nullptr),
          keypoints2(nullptr),
          nrMatchesVfcFiltered(0),
          imgSize(800, 600),
          degenDecisionTh(0.85) {}

    double focalLength;
    double th_pixels;
    UsacChkDegenType degeneracyCheck;
    PoseEstimator estimator;
    RefineAlg refinealg;
    bool prevalidateSample;
    bool noAutomaticProsacParamters;
    int automaticSprtInit;
    std::vector<cv::DMatch> *matches;
    std::vector<cv::KeyPoint> *keypoints1;
    std::vector<cv::KeyPoint> *keypoints2;
    unsigned int nrMatchesVfcFiltered;
    cv::Size imgSize;
    double degenDecisionTh;
};

// RANSAC / LMedS / ARRSAC seed control.  The reference seeds std::srand(std::time(nullptr)) in the estimator constructor
// (five-point-nister/modelest.cpp:58) and its setSeed() is never called on this path, so its results are time-seeded.
// Default here: the same (time-seeded).  setRansacSeed(s) fixes the glibc rand() stream for reproducible runs;
// clearRansacSeed() returns to time seeding.  Thread-local.
void setRansacSeed(unsigned seed);
void clearRansacSeed();

// ARRSAC draws from two cv::RNG streams that are function-local statics in the reference (include/arrsac/prosac_sampler.h:115,
// random_sampler.h:65): process-wide, default-seeded (0xffffffff), never reset -- the n-th ARRSAC call of a program continues where the
// (n-1)-th stopped.  The library keeps the same process-wide pair; these two functions let a test or a caller that wants reproducible
// runs read and set it.  Consequence, as in the reference: ARRSAC through this facade is SINGLE-FLIGHT -- concurrent calls (also from
// StereoRefine and AutoThEpi) are serialised on that pair for their whole duration, and their order decides who draws what.  Callers
// that want concurrent ARRSAC runs use mlpl_arrsac_essential (include/mlpl_c.h) with one context and one rng_state pair per thread.
void setArrsacRngState(uint64_t prosac_state, uint64_t uniform_state);
void getArrsacRngState(uint64_t *prosac_state, uint64_t *uniform_state);

// poselib::estimateEssentialMat (pose_estim.h:204-210, pose_estim.cpp:857-890).  p1, p2: n x 2 camera coordinates (CV_64F or CV_32F).
// "RANSAC": 1000 iterations, confidence 0.999, `refine` = least-squares refit on the inliers; "LMEDS": 2000 iterations, no refit,
// `threshold` unused (as in the reference); "ARRSAC" (the default): the reference's preemptive estimator (modelest.cpp:197-341) with
// `refine` = robustEssentialRefine on the winner's inliers (pose_estim.cpp:866-869).  "USAC" / unknown method: prints the reference's
// message and calls exit(1), like the reference.
bool estimateEssentialMat(cv::OutputArray E, cv::InputArray p1, cv::InputArray p2, const std::string &method = "ARRSAC",
                          double threshold = PIX_MIN_GOOD_TH, bool refine = true, cv::OutputArray mask = cv::noArray());

// poselib::estimateEssentialOrPoseUSAC (pose_estim.h:212-223, pose_estim.cpp:1737-2244): the USAC framework -- PROSAC sampling in the
// order of the matching costs (cfg.matches), Wald's sequential test with automatically initialised and carried-over delta / epsilon
// (cfg.automaticSprtInit; the history of the last 20 calls lives in function-local statics in the reference, and so it does here:
// process-wide), local optimisation, SPRT-aware stopping -- on the MI355X (mlpl_usac_essential, include/mlpl_c.h).
// Returns 0, -1 = configuration not supported, -2 = USAC failed.  E: 3 x 3, inliers: 1 x n CV_8U.
// Built: PoseEstimator POSE_NISTER and POSE_STEWENIUS (one exact 5-point solver serves both); RefineAlg REF_WEIGHTS (8-point fit with
// Torr weights: the harness default, tests/poselib-test/main.cpp cfgUSAC "311220") and the four 5-point refinements REF_STEWENIUS,
// REF_STEWENIUS_WEIGHTS (ConfigUSAC's own default), REF_NISTER, REF_NISTER_WEIGHTS; UsacChkDegenType DEGEN_NO_CHECK and
// DEGEN_USAC_INTERNAL -- the rotation-only / no-motion tests after every new best model (with REF_WEIGHTS also after every local
// optimisation, usac_estimations.cpp:368-375), the upgrade of a degenerate model (no motion -> t, R -> R + t) and the verdict:
// isDegenerate, R_degenerate and inliers_degenerate_R are filled as the reference fills them (pose_estim.cpp:2101-2133).  Of that check
// only the homography test the reference adds for the 8-point refinements is missing (its upgrade branch has no defined behaviour,
// EssentialMatEstimator.h:1958-2015).  R and t are only ever filled by Kneip's eigensolver in the reference: left empty.
// NOT built, and REFUSED with the reference's message and -1 rather than served by another algorithm: POSE_EIG_KNEIP,
// REF_8PT_PSEUDOHUBER, REF_EIG_KNEIP, REF_EIG_KNEIP_WEIGHTS, DEGEN_QDEGSAC.  (MLPL_OPTIONS=usac_substitute in the environment opts
// into the substitution of earlier releases: Kneip -> 5-point solver, those refinements -> REF_WEIGHTS, QDEGSAC -> no check.)
// The reference seeds srand(time(nullptr)) per call; setRansacSeed() fixes the seed here as for RANSAC.
int estimateEssentialOrPoseUSAC(const cv::Mat &p1, const cv::Mat &p2, cv::OutputArray E, double th, ConfigUSAC &cfg, bool &isDegenerate,
                                cv::OutputArray inliers = cv::noArray(), cv::OutputArray R_degenerate = cv::noArray(),
                                cv::OutputArray inliers_degenerate_R = cv::noArray(), cv::OutputArray R = cv::noArray(),
                                cv::OutputArray t = cv::noArray(), bool verbose = false);
// The process-wide SPRT history of estimateEssentialOrPoseUSAC (its function-local statics in the reference) -- for tests and for
// callers that want the first-call behaviour again.
void resetUsacHistory();
// estimateSprtDeltaInit / estimateSprtEpsilonInit / getSortedMatchIdx (pose_helper.cpp:2830-2923): the SPRT start values from the convex
// hull of the matched keypoints / from the share of matches a flow filter kept, and the PROSAC order.
double estimateSprtDeltaInit(const std::vector<cv::DMatch> &matches, const std::vector<cv::KeyPoint> &kp1,
                             const std::vector<cv::KeyPoint> &kp2, const double &th, const cv::Size &imgSize);
double estimateSprtEpsilonInit(const std::vector<cv::DMatch> &matches, const unsigned int &nrMatchesVfcFiltered);
void getSortedMatchIdx(std::vector<cv::DMatch> matches, std::vector<unsigned int> &sortedMatchIdx);

// poselib::robustEssentialRefine (pose_estim.h:225-228, pose_estim.cpp:337-792): pseudo-Huber re-weighted linear refinement of an
// essential matrix on the device.  Built for what the library's own callers use: model 0 (essential matrix), iters = 0 (run to the
// reference's stopping tests), makeClosestE = true, no normalisation, no oriented-epipolar test; other argument values throw
// cv::Exception.  `mask` (1 x n or n x 1 CV_8U) selects the correspondences and is left as it is; `errors` receives the final Sampson
// errors (n x 1 CV_64F) when requested.
void robustEssentialRefine(cv::InputArray points1, cv::InputArray points2, cv::InputArray E_init, cv::Mat &E_refined, double th = 0.005,
                           unsigned int iters = 0, bool makeClosestE = true, double *sumSqrErr_init = nullptr, double *sumSqrErr = nullptr,
                           cv::OutputArray errors = cv::noArray(), cv::InputOutputArray mask = cv::noArray(), int model = 0,
                           bool tryOrientedEpipolar = false, bool normalizeCorrs = false);

// poselib::AutoThEpi (pose_estim.h:137-190, pose_estim.cpp:81-300): ARRSAC with an inlier threshold estimated from the error statistics
// of its own result -- estimate, derive a threshold (mean + 3 sigma of the Sampson distances, or median + 3 MAD-sigma when the two
// disagree, clamped to [0.25, 2] pixels), re-estimate until the threshold settles.  Thresholds in camera units unless stated.
class AutoThEpi {
   public:
    explicit AutoThEpi(double pixToCamFact_, bool thStable = false)
        : corr_filt_cam_th(PIX_TH_START * pixToCamFact_),
          corr_filt_pix_th(PIX_TH_START),
          corr_filt_min_pix_th(MIN_PIX_TH),
          th_stable(thStable),
          pixToCamFact(pixToCamFact_) {}
    double getThCam() { return corr_filt_cam_th; }
    double getThPix() { return corr_filt_pix_th; }
    void setThStable(bool isStable) { th_stable = isStable; }
    // 0 ok; -1 no essential matrix at any threshold tried; -2 E not requested.  *th is read and updated.
    int estimateEVarTH(cv::InputArray p1, cv::InputArray p2, cv::OutputArray E, cv::OutputArray mask, double *th, int *nrgoodPts);
    // useImgCoordSystem = true needs camera matrices the class does not hold (the reference asserts there): camera units only.
    double estimateThresh(cv::InputArray p1, cv::InputArray p2, cv::InputArray E, bool useImgCoordSystem = false, bool storeGlobally = false);
    double setCorrTH(double thresh, bool useImgCoordSystem = false, bool storeGlobally = true);

   private:
    double corr_filt_cam_th, corr_filt_pix_th, corr_filt_min_pix_th;
    bool th_stable;
    double pixToCamFact;
};

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
