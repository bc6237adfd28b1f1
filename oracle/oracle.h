/*
 * oracle.h -- CPU restatement of the reference's descriptor-matching + robust-pose hot path.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE.
 * Only tests/, __graft_entry__.smoke() and bench.py's `cpu_baseline` leg may load or call it,
 * and there only as the checker / the timed CPU baseline.  Nothing under matchinglib_poselib_amd/
 * or include/ may include, link or dlopen anything from oracle/.
 *
 * Parity pinning status (see DESIGN.md "Oracle"):
 *   - matching (knn/ratio): pinned against the reference's vendored NMSLIB seq_search built from
 *     /root/reference by oracle/Makefile into oracle/_ref/ (distances and tie-free indices) and
 *     against numpy brute force; the reference ships no golden vectors of its own for this path.
 *   - 5-point solver: pinned against the reference's vendored OpenGV fivept_nister built into
 *     oracle/_ref/ (E-sets up to sign).
 *   - ARRSAC (arrsac_oracle.cpp): Eigen::JacobiSVD<Matrix3d> pinned against the Eigen 3.2.0 the reference vendors (oracle/_ref/eigen_svd3,
 *     tests/golden/eigen_svd3.npz); cv::RNG, cv::findFundamentalMat(FM_8POINT), Eigen::EigenSolver restated, unpinned; the sign and the
 *     order of the 5-point solutions are fixed by convention (artefacts of cv::SVD's null-space basis, see that file's header).
 *   - USAC (usac_oracle.cpp): control flow pinned by the reference's own USAC.h compiled in place (oracle/_ref/usac_ref), incl. the
 *     degeneracy handling (testSolutionDegeneracyRot / NoMot, upgradeDegenerateModel: oracle_usac_essential_degen), whose driver runs
 *     the restated estimator members over the reference's own OpenGV and PoseTools functions (tests/golden/usac_degen_trace.npz);
 *     oracle/ref_drivers/opengv_degen.cpp pins the 3 x 3 numerics one function at a time (tests/golden/usac_degen_math.npz).
 *     Solver + control flow at once: `usac_ref --stewenius` runs the reference's DEFAULT estimator (POSE_STEWENIUS, OpenGV's
 *     fivept_stewenius) and oracle_usac_essential -- with its own five-point solver -- follows it event by event in 21 of 24 runs
 *     (tests/golden/usac_stewenius_trace.npz).
 *   - The arithmetic of cvflann::LinearIndex / cv::SVD / cv::solvePoly / cv::triangulatePoints lives in
 *     OpenCV 4.2.0 (pinned in ci/make_opencv.sh:6), which is NOT vendored under /root/reference and is
 *     not installed here; those steps restate the published algorithms and are "parity unpinned" at
 *     that boundary (no reference test holds a vector for them).
 *
 * Citations are relative to /root/reference/matchinglib_poselib/source/ :
 *   M/ = matchinglib/   P/ = poselib/
 */
#ifndef MLPL_ORACLE_H
#define MLPL_ORACLE_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* cv::DMatch layout (OpenCV core/types.hpp): 16 bytes. */
typedef struct {
    int32_t queryIdx;
    int32_t trainIdx;
    int32_t imgIdx;
    float distance;
} oracle_dmatch;

/* ---- matching: M/source/matchers.cpp:525-714 (LINEAR branch) ---------------------------------- */

/* cvflann::Index<HammingLUT>(LinearIndexParams).knnSearch (matchers.cpp:584-588).
 * idx/dist are nq*k int32, row-major; k in {1,2}.  Returns 0, or -1 on bad arguments. */
int oracle_knn_hamming(const uint8_t *q, int nq, size_t q_stride, const uint8_t *t, int nt, size_t t_stride,
                       int nbytes, int k, int32_t *idx, int32_t *dist);

/* cvflann::Index<L2<float>>(LinearIndexParams).knnSearch (matchers.cpp:660-664): squared L2. */
int oracle_knn_l2sq_f32(const float *q, int nq, size_t q_stride_elems, const float *t, int nt,
                        size_t t_stride_elems, int dim, int k, int32_t *idx, float *dist);

/* Ratio / emit loop (matchers.cpp:601-625 int distances, :677-701 float distances).
 * k==2: keep q iff d0 < 0.75f*d1; k==1: keep all.  out must hold nq entries.  Returns #matches. */
int oracle_ratio_filter_i32(const int32_t *idx, const int32_t *dist, int nq, int k, oracle_dmatch *out);
int oracle_ratio_filter_f32(const int32_t *idx, const float *dist, int nq, int k, oracle_dmatch *out);

/* getMatches(...,"LINEAR",...) end to end (matchers.cpp:115-135, 525-714): return codes 0/-1/-3/-4.
 * desc_type: 0 = CV_8U (cols = bytes), 5 = CV_32F (cols = floats).  n_kp1/n_kp2 = keypoint counts. */
int oracle_get_matches_linear(int n_kp1, int n_kp2, const void *desc1, int rows1, const void *desc2, int rows2,
                              int cols, int desc_type, int ratio_test, oracle_dmatch *out, int *n_out);

/* getMatches(...,"BRUTEFORCENMS",...) (matchers.cpp:476-519 -> nmslib_matchers.h:159-424), incl. the 240-bit quirk for
 * CV_8U and true (sqrt) L2 for CV_32F.  desc_type 0/5, dense rows of `cols` elements. */
int oracle_get_matches_bruteforce_nms(const void *desc1, int rows1, const void *desc2, int rows2, int cols, int desc_type,
                                      int ratio_test, oracle_dmatch *out, int *n_out);

/* BASELINE.md "CPU-best" tier: identical results, hardware popcount + OpenMP over queries (k = 2, nbytes % 8 == 0).
 * Returns the number of threads used. */
int oracle_knn_hamming_fast(const uint8_t *q, int nq, size_t q_stride, const uint8_t *t, int nt, size_t t_stride, int nbytes,
                            int32_t *idx, int32_t *dist, int threads);

/* ---- robust pose: P/source/five-point-nister/{five-point,modelest}.cpp, P/source/pose_estim.cpp -- */

/* glibc srand/rand (TYPE_3 additive feedback generator) restated, so that sampling is reproducible
 * without touching the process-global stream (modelest.cpp:58,585 use std::srand/std::rand). */
typedef struct {
    int32_t r[34];
    int f, b; /* front/rear indices */
} oracle_glibc_rand;
void oracle_srand(oracle_glibc_rand *st, unsigned seed);
int oracle_rand(oracle_glibc_rand *st);

/* CvModelEstimator3::getSubset (modelest.cpp:567-610) + checkSubset (:613-650).
 * p1,p2: n x 2 doubles.  idx5 receives the 5 sample indices.  Returns 1 if found. */
int oracle_get_subset(oracle_glibc_rand *st, const double *p1, const double *p2, int n, int max_attempts,
                      int *idx5);

/* CvEMEstimator::run5Point (five-point.cpp:366-471).  q1,q2: n x 2 doubles (n>=5).
 * E_out: up to 10 row-major 3x3 matrices.  Returns the number of solutions. */
int oracle_run5point(const double *q1, const double *q2, int n, double *E_out);
/* The same solver on an n x 9 system given row by row (row i = s_i * (u2_i (x) u1_i), entry 3 a + b): OpenGV's fivept_nister /
 * fivept_stewenius on n >= 5 unit bearing vectors (P/thirdparty/opengv/src/relative_pose/methods.cpp:183-268) and the reference's
 * weighted forms (P/source/usac/utils/weightingEssential.cpp:62-148) build this system; the essential matrices are those of the four
 * right singular vectors of the smallest singular values. */
int oracle_run5point_rows(const double *rows, int n, double *E_out);
/* The same plus diagnostics: c_out[11] = the degree-10 polynomial (ascending powers), roots_out[20] = solvePoly's roots as
 * (re, im) in root order, xy1z_out[10] = third component of the SVD::solveZ vector per root (NaN = rejected as complex). */
int oracle_run5point_dbg(const double *q1, const double *q2, int n, double *E_out, double *c_out, double *roots_out,
                         double *xy1z_out);

/* Building blocks of run5Point exposed for pinning against the reference's own expressions: the 10 x 20 constraint matrix in the
 * reference's column order (getCoeffMat, five-point.cpp:603-824; EE[b * 9 + k] = basis matrix b) and the determinant polynomial of
 * B(z) (five-point.cpp:416-428; b = 3 x 13, c[0..10] ascending). */
void oracle_coeff_matrix(const double *EE, double *A);
void oracle_detpoly(const double *b, double *c);

/* computeReprojError3 (five-point.cpp:476-503): Sampson error in fp64 stored as float. */
void oracle_sampson_err(const double *p1, const double *p2, int n, const double *E, float *err);

/* findInliers (modelest.cpp:69-83): mask[i] = err[i] <= thresh^2; returns count; *err_sum = cv::sum(err). */
int oracle_find_inliers(const double *p1, const double *p2, int n, const double *E, double thresh,
                        float *err, uint8_t *mask, double *err_sum);

/* cvRANSACUpdateNumIters1 (modelest.cpp:86-109). */
int oracle_ransac_update_num_iters(double p, double ep, int model_points, int max_iters);

/* Optional per-iteration trace of runRANSAC. */
typedef struct {
    int32_t idx[5];
    int32_t nmodels;
    int32_t good[10];
    double err_sum[10];
    int32_t niters_after; /* niters after processing this iteration */
    int32_t best_taken;   /* index of the model taken as new best in this iteration, or -1 */
} oracle_ransac_trace;

/* CvModelEstimator3::runRANSAC (modelest.cpp:343-474) driven as findEssentialMat does
 * (five-point.cpp:69-148) but with seed / confidence / max_iters exposed.
 * mask: n bytes (0/1).  trace may be NULL; if not, it must hold max_iters entries.
 * Returns 1 on success, 0 on failure.  *iters_run = iterations actually executed. */
int oracle_ransac_essential(const double *p1, const double *p2, int n, double thresh, double confidence,
                            int max_iters, int lesqu, unsigned seed, double *E, uint8_t *mask,
                            int *n_inliers, int *iters_run, oracle_ransac_trace *trace);

/* CvModelEstimator3::runLMeDS (modelest.cpp:483-564) with the 5-point kernel; returns 1 on success. */
int oracle_lmeds_essential(const double *p1, const double *p2, int n, double confidence, int max_iters, unsigned seed,
                           double *E, uint8_t *mask, int *n_inliers, double *min_median);

/* decomposeEssentialMat (five-point.cpp:340-352). */
void oracle_decompose_essential(const double *E, double *R1, double *R2, double *t);

/* recoverPose / getPoseTriangPts (five-point.cpp:150-338, pose_estim.cpp:913-946) with t_only empty.
 * mask_inout: n bytes, nonzero = use (may be NULL = all ones).  Q: n x 3.  Returns #good points. */
int oracle_recover_pose(const double *E, const double *p1, const double *p2, int n, double dist, double *R,
                        double *t, double *Q, uint8_t *mask_inout);

int oracle_recover_pose_translation(const double *Et, const double *p1, const double *p2, int n, double dist, double *R,
                                    double *t, double *Q, uint8_t *mask_inout);

/* cv::triangulatePoints for one correspondence with P0=[I|0], P1=[R|t] (unit-norm homogeneous X). */
void oracle_triangulate_point(const double *P0, const double *P1, const double *x1, const double *x2, double *X4);

/* one-sided Jacobi SVD helper exposed for tests: A is m x n row-major (m may be < n).
 * On return w[n] (descending) and V (n x n row-major, columns = right singular vectors). */
void oracle_jacobi_svd(const double *A, int m, int n, double *w, double *V);

/* Durand-Kerner as cv::solvePoly; coeffs[0..deg] ascending powers; roots as (re,im) pairs. Returns #roots. */
int oracle_solve_poly(const double *coeffs, int deg, double *roots_re_im, int max_iters);

/* ---- pre/post steps: P/source/pose_helper.cpp:1100-1109, 1169-1279, 639-664, 3030-3045 -------- */
void oracle_img_to_cam(float *pts, int n, const double K[4]);
int oracle_remove_lens_dist(float *points1, float *points2, int n, const double dist1[8], const double dist2[8], int *n_out);
int oracle_get_inliers_strict(const double *p1, const double *p2, int n, const double *E, double th2, double *err,
                              unsigned char *mask);

/* ---- ARRSAC (arrsac_oracle.cpp; SURVEY 8(f) rank 4, first half): P/source/five-point-nister/modelest.cpp:111-341,
 *      P/include/arrsac/{arrsac,prosac_sampler,random_sampler,sequential_probability_ratio}.h, P/source/pose_estim.cpp:337-792 ---- */
/* cv::RNG: `count` values of next(), *state updated. */
void oracle_cv_rng_stream(uint64_t *state, int count, uint32_t *out);
/* Eigen::JacobiSVD<Matrix3d>(M, ComputeFullU | ComputeFullV): M = U diag(sv) V^T, row-major, Eigen's column signs. */
void oracle_eigen_svd3(const double *M, double *sv, double *U, double *V);
/* CvEMEstimator::ValidModel (five-point.cpp:534-601) on the m sample correspondences q1, q2 (m x 2). */
int oracle_valid_model(const double *q1, const double *q2, int m, const double *E);
/* cv::findFundamentalMat(q1, q2, FM_8POINT), m >= 8 (float32 input rounding as OpenCV); returns 0 when degenerate. */
int oracle_cv_fm_8point(const double *q1, const double *q2, int m, double *F);
double oracle_sprt_threshold(double sigma, double epsilon, double time_ratio, int num_models_verified);
int oracle_robust_essential_refine(const double *p1, const double *p2, int n, const double *E_init, double th, double *E_refined,
                                   double *err2);
int oracle_arrsac_essential(const double *p1, const double *p2, int n, double thresh, int refine, uint64_t *rng_state, double *E,
                            uint8_t *mask, int *n_inliers, int64_t *stats);
void oracle_std_sort_desc(const double *score, int n, int32_t *perm);
int oracle_arrsac_trace(int32_t *buf, int cap);

/* ---- USAC with the Nister minimal solver (usac_oracle.cpp; SURVEY 8(f) rank 4, second half): P/include/usac/estimators/USAC.h,
 *      EssentialMatEstimator.h, P/source/usac/usac_estimations.cpp:283-470 (estimateEssentialMatUsac without degeneracy tests).
 * th = inlier threshold (not squared); seed = the srand() seed (reference: time(nullptr)); refine 0 = REF_WEIGHTS, 6 = REF_NISTER;
 * sorted_idx NULL = uniform sampling, else PROSAC over these indices (best match first).  results[12] = ok, hypotheses, models,
 * rejected samples, rejected models, best inlier count, points verified, local optimisations, SPRT delta / epsilon of the newest
 * history entry (what the reference returns), delta / epsilon at the end.  events: optional trace, 16 doubles per record
 * (type 1 sample, 2 evaluation, 3 refined model, 4 stored model, 5 minimal model, 6 model rejected by the oriented constraint);
 * *n_events = records produced (may exceed event_cap: only event_cap are written).  Returns 1 if solve() ran. */
int oracle_usac_essential(const double *p1, const double *p2, int n, double th, unsigned seed, int refine, const uint32_t *sorted_idx,
                          int max_hyp, double conf, double prosac_beta, double sprt_delta, double sprt_epsilon, double sprt_mS,
                          double sprt_tM, double *E, uint8_t *inlier_flags, double *results, double *events, int event_cap, int *n_events);

/* The same with the degeneracy handling of ConfigUSAC's DEGEN_USAC_INTERNAL (EssentialMatEstimator.h:1334-1362, 1511-1663, 1838-1911,
 * 2098-2361; no homography test): check_degeneracy 1 = after every new best model, 3 = also after every local optimisation; additional
 * event types 7 degeneracy test, 8 rotation on all correspondences, 9 upgrade, 10 upgrade candidate (oracle/ref_drivers/usac_ref.cpp);
 * degen[16] = {1, inliers of the rotation-only model, of "no motion", degeneracy type, R_degenerate[9]}. */
int oracle_usac_essential_degen(const double *p1, const double *p2, int n, double th, unsigned seed, int refine, const uint32_t *sorted_idx,
                                int max_hyp, double conf, double prosac_beta, double sprt_delta, double sprt_epsilon, double sprt_mS,
                                double sprt_tM, int check_degeneracy, double th_pixels, double focal_length, double *E,
                                uint8_t *inlier_flags, double *results, double *events, int event_cap, int *n_events, double *degen,
                                uint8_t *flags_rot, uint8_t *flags_nomot);

/* The eigenvalues of a 3 x 3 matrix (row-major) in the order Eigen::EigenSolver<Matrix3d> returns them (OpenGV's eigensolver takes its
 * translation from position 0): Eigen::RealSchur restated; pinned by tests/golden/eigen_order3.npz. */
void oracle_eigen_order3(const double *M, double *d);

#ifdef __cplusplus
}
#endif
#endif
