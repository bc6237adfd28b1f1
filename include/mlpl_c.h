/*
 * mlpl_c.h -- C ABI of the MI355X-native (gfx950) descriptor-matching + robust-pose hot path.
 *
 * This is the drop-in boundary: plain pointers and sizes, no C++/torch types.  The reference
 * (josefmaierfl/matchinglib_poselib) has no FFI of its own for this path -- the path sits behind exported
 * C++ functions -- so each entry point below names the reference code it replaces.  Paths are relative to
 * /root/reference/matchinglib_poselib/source/ (M/ = matchinglib/, P/ = poselib/).
 *
 * Conventions
 *   - return value: 0 = ok, negative = error.  Codes -1..-4 keep the meaning of the reference's getMatches
 *     (M/source/matchers.cpp:109-114); MLPL_E_* below are this library's own.
 *   - no exceptions, no exit(), no stdout.  mlpl_last_error() returns a thread-local message.
 *   - the caller owns every buffer; outputs are caller-allocated.
 *   - `*_dev` entry points take DEVICE pointers plus a hipStream_t (as void*; NULL = HIP's null stream, use
 *     mlpl_ctx_stream() for the context's own), enqueue work and return without synchronising; the plain
 *     entry points take HOST pointers, copy in/out on the context's stream and synchronise.
 *   - a context (mlpl_ctx) owns the device workspace and a private stream; use one per host thread.
 *   - there is NO CPU fallback: every compute entry point fails with MLPL_E_NO_DEVICE when no gfx950
 *     device is usable.
 */
#ifndef MLPL_C_H
#define MLPL_C_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define MLPL_OK 0
#define MLPL_E_BAD_INPUT (-1)     /* getMatches: "Wrong input data"        (matchers.cpp:110) */
#define MLPL_E_UNSUPPORTED (-2)   /* getMatches: "Matcher not supported"   (matchers.cpp:111) */
#define MLPL_E_FAILED (-3)        /* getMatches: "Matching algorithm failed" / < 2 matches (matchers.cpp:112,709-713) */
#define MLPL_E_FEW_KEYPOINTS (-4) /* getMatches: "Too less keypoits"       (matchers.cpp:113,123-127) */
#define MLPL_E_NO_DEVICE (-100)
#define MLPL_E_HIP (-101)
#define MLPL_E_NOMEM (-102)
#define MLPL_E_INTERNAL (-103)    /* an internal capacity limit was hit: NOT an estimator outcome (in/out states are left untouched) */

/* cv::DMatch layout {int queryIdx; int trainIdx; int imgIdx; float distance;} -- 16 bytes. */
typedef struct mlpl_dmatch {
    int32_t queryIdx;
    int32_t trainIdx;
    int32_t imgIdx;
    float distance;
} mlpl_dmatch;

typedef struct mlpl_ctx mlpl_ctx;

/* ---- context ------------------------------------------------------------------------------------------ */
int mlpl_ctx_create(int device_ordinal, mlpl_ctx **out);
void mlpl_ctx_destroy(mlpl_ctx *ctx);
const char *mlpl_last_error(void);
const char *mlpl_version(void);
/* Number of visible HIP devices (0 when none); does not create a context. */
int mlpl_device_count(void);
/* The context's private stream (hipStream_t) and device ordinal. */
void *mlpl_ctx_stream(mlpl_ctx *ctx);
int mlpl_ctx_device(mlpl_ctx *ctx);
int mlpl_ctx_synchronize(mlpl_ctx *ctx);

/* Tuning knobs (performance only; "solver_polish", below, is the one numerical option):
 *   Hamming: "hamming_variant" 3 = fp4 matrix-core kernels (default; descriptors above 64 bytes fall back to 0), 0 = LDS-tiled VALU
 *     kernel, 1 = scalar-operand VALU kernel, 2 = one-wave-per-block VALU kernel; "hamming_mfma_lds" 1 (default) = LDS-ring kernel for
 *     32-byte descriptors, 0 = register-prefetch kernel, 2 = dynamic train splits; "hamming_mfma_weighted" (default 1) = age-aware split
 *     sizes; "hamming_mfma_prio" 0|1|2 (diagnostics) | 3 (wave-uniform skip of the running top-2 update: measured slower, round 6); "hamming_mfma_blocks_per_cu" and "hamming_mfma_qt" (query tiles per wave,
 *     0 = automatic) size the matrix-core grid; "hamming_qpl" queries per lane 1|2 and "hamming_blocks_per_cu" size the VALU grids;
 *     "hamming_stamps" 1 = per-wave / per-workgroup clock stamps (mlpl_debug_hamming_stamps), 2 = one clock record per launch
 *       (mlpl_debug_hamming_clock).  "hamming_train01" 1 = {0, +1} instead of +-1 train fragments in the matrix-core Hamming kernel
 *       (same results; see knn_hamming_mfma.hip; measured no faster, default 0).  "hamming_merge_emit" 1 = with one image pair per call the
 *       merge kernel writes the DMatch rows itself (no ratio_write launch; measured no faster, default 0; TEST-ONLY: its chained look-back
 *       assumes that the whole merge grid is resident, which other streams on the same device can break).
 *       "hamming_split_rows" 0 (default) | 8192 | 4096 = cap on the train rows one workgroup scans (4096 = rounds 1-4: every 8192-row train
 *       set was cut in two even with the chip full); "hamming_mfma_waves" 0 (automatic) | 4 | 8 | 16 waves per workgroup and
 *       "hamming_mfma_prefetch" 0 | 2 | 4 | 6 tiles of prefetch distance in the LDS-ring kernel.
 *   L2: "l2_mfma_waves" 0|4|8 and "l2_mfma_blocks_per_cu" shape the int8 matrix-core kernel of the forced mode (see mlpl_set_l2_path);
 *     "l2_float_mfma" 0|1|2 decides when the fp16 candidate path serves non-integer float descriptors (mlpl_set_l2_path, mode 0).
 *   RANSAC: "ransac_device_draw" (default 1) = large passes draw their samples on the device (mlpl_debug_ransac_draw); "ransac_chunk" hypotheses per device pass (0 = 32768 = the maximum; the sequential best/niters rule is replayed across
 *     passes); "ransac_lazy_sums" (default 1) = the passes count inliers without the division and compute error sums only for the
 *     models that can still win, 0 = sums for every model; "ransac_f32_filter" (default 1) = the counting kernels decide in packed single
 *     precision outside a rigorous error band and in fp64 inside it (same counts), 0 = fp64 only; "ransac_overlap" (default 1) = the
 *     root kernels of a large pass run on helper streams beside the elimination kernels; "ransac_host_table" 1 = build the
 *     iteration-bound table T[g] on the host for every call (default 0: the device evaluates the few bounds it needs and the host
 *     verifies exactly those against its libm, falling back to the table when one differs); "ransac_event_cap" (tests) = capacity of
 *     the record-event list of the replay kernels.
 *   Sequential estimators (round 4): "eig_inverse_iteration" (default 1) = the one eigenvector the re-weighted 9 x 9 fits need (USAC
 *     REF_WEIGHTS, robustEssentialRefine) by inverse iteration, the Jacobi decomposition as fallback, 0 = Jacobi always;
 *     "usac_lo_warm_start" / "arrsac_refine_warm_start" (default 1) = a chain's eigen-iterations start from its previous fit;
 *     "usac_sprt_fast", "usac_lo_stepwise" (tests).  Batches of them: "hub_lanes" (cohorts of runs in flight, 1..8; 0 = the estimator's own choice: six for USAC with REF_WEIGHTS, four otherwise),
 *     "usac_lo5_fused_fit" (default 1: a fit of the 5-point refinements' chains -- solve, roots, choice -- is one launch, 0 = three),
 *     "usac_first_batch" (samples in a run's first speculative batch, 1..128; 0 = default = as later batches: up to 128),
 *     "hub_cohort" (runs per cohort, 0 = 128), "hub_workers" (worker threads per cohort, 0 = 16: the runs are fibers on them),
 *     "hub_blocking_sync" (default 1: a cohort's thread sleeps at the end of a round instead of spinning), "pair_batch" /
 *     "pair_batch_seq" (image pairs per internal batch of mlpl_pair_pose_batch_dev, 0 = 256 / of its USAC and ARRSAC forms, 0 = 512).
 *     Hamming: "hamming_fused_merge" (default 1) = the LDS-ring kernel folds its train splits, evaluates the ratio predicate and counts
 *     itself, 0 = separate merge launch; RANSAC: "ransac_count_mpl" 1|2 models per lane of the counting kernel (A/B, default 2);
 *     "ransac_count_threads" 256 (default since round 6: 4-wave workgroups at 96 VGPRs, five per CU) | 512; "ransac_count_tiles" 2 (default) | 1
 *     tiles of 512 correspondences per counting workgroup.
 *   "solver_polish" (default 1) = every 5-point solution is finished by <= 4 Gauss-Newton steps on the ten cubic constraints (a step is
 *     kept only while the residual falls).  It is the accuracy safeguard of THIS solver, not a departure from the reference: the device's
 *     elimination (like the CPU code's, five-point.cpp:366-471, but on other samples) is ill conditioned on ~0.4 % of minimal samples and its
 *     plain result is then off by up to 4e-3.  Measured against the CPU path on 43 760 models (tools/polish_default_ab.py, round 6): with the
 *     safeguard 158 models differ by more than 1e-8 -- every one a sample on which the CPU path's OWN model violates the essential-matrix
 *     constraints; without it 318 differ, half of them because the device's model is the inaccurate one, and ten parity tests against the
 *     oracle fail (USAC decisions part).  RANSAC runs: identical to the CPU path on 126 of 126 seeds either way.  0 = the plain elimination +
 *     root path, kept for A/B (mlpl_set_option(ctx, "solver_polish", 0) or MLPL_OPTIONS=solver_polish=0). */
int mlpl_set_option(mlpl_ctx *ctx, const char *name, int value);
/* The current value of a tuning knob (the names mlpl_set_option takes; a subset: the Hamming knobs, "solver_polish", "ransac_count_mpl",
 * "hub_workers", "hub_lanes").  Returns 0, MLPL_E_BAD_INPUT for a name it does not know. */
int mlpl_get_option(mlpl_ctx *ctx, const char *name, int *value);

/* ---- in-library kernel timing (for roofline accounting) ------------------------------------------------------
 * When enabled, the launches of the dominant kernel of each path are bracketed with hipEvents on the stream
 * they run on.  mlpl_profile_read() synchronises those events and returns the summed duration and the
 * launch count since the last reset.  kernel_id: 0 = Hamming partial top-2 (matrix-core or VALU kernel), 1 = knn_l2 (exact or
 * MFMA), 2 = 5-point solver, 3 = Sampson scoring, 4 = recover_pose.  mlpl_profile_enable(ctx, N): 0 = off, 1 = every launch,
 * N > 1 = every Nth launch (sampling: the two event records cost ~10 us of stream time per bracketed launch). */
#define MLPL_PROF_KNN_HAMMING 0
#define MLPL_PROF_KNN_L2 1
#define MLPL_PROF_SOLVE_5PT 2
#define MLPL_PROF_SCORE 3
#define MLPL_PROF_RECOVER_POSE 4
#define MLPL_PROF_COUNT 5         /* the inlier-counting kernel of a RANSAC pass alone (MLPL_PROF_SCORE brackets the whole scoring pass: it, the candidate selection and the candidates' error sums) */
#define MLPL_PROF_NUM 6
int mlpl_profile_enable(mlpl_ctx *ctx, int on);
int mlpl_profile_reset(mlpl_ctx *ctx);
int mlpl_profile_read(mlpl_ctx *ctx, int kernel_id, double *total_ms, int *launches);

/* ---- brute-force k-NN (k = 1 or 2) ---------------------------------------------------------------------
 * Replaces cvflann::Index<HammingLUT>(dataset, LinearIndexParams()).knnSearch(query, indices, dists, nn, ...)
 * at M/source/matchers.cpp:567-588 (CV_8U) and cvflann::Index<L2<float>> at :634-664 (CV_32F).
 * q = descriptors1 (query, nq rows), t = descriptors2 (train, nt rows).  Strides are in BYTES for the
 * Hamming entry and in ELEMENTS for the float entry.  idx/dist are nq*k row-major.
 * Result = the k lexicographically smallest (distance, trainIdx) pairs, ascending (cvflann
 * KNNUniqueResultSet semantics).  Hamming distances are exact integers; L2 is SQUARED, summed in fp32 in
 * the reference's order (4 differences per step), so both are bit-exact against the CPU path.
 * Requires nt >= k, nbytes in [1,256] / dim in [1,1024].
 */
int mlpl_knn2_hamming(mlpl_ctx *ctx, const uint8_t *q, int nq, size_t q_stride, const uint8_t *t, int nt,
                      size_t t_stride, int nbytes, int k, int32_t *idx, int32_t *dist);
int mlpl_knn2_l2sq_f32(mlpl_ctx *ctx, const float *q, int nq, size_t q_stride, const float *t, int nt,
                       size_t t_stride, int dim, int k, int32_t *idx, float *dist);

/* Device-pointer, batched forms.  `batch` independent problems of identical shape; problem b reads
 * q + b*q_batch_stride (bytes for Hamming / elements for float), writes idx + b*nq*k etc. */
int mlpl_knn2_hamming_dev(mlpl_ctx *ctx, const uint8_t *d_q, int nq, size_t q_stride, size_t q_batch_stride,
                          const uint8_t *d_t, int nt, size_t t_stride, size_t t_batch_stride, int nbytes, int k,
                          int batch, int32_t *d_idx, int32_t *d_dist, void *stream);
int mlpl_knn2_l2sq_f32_dev(mlpl_ctx *ctx, const float *d_q, int nq, size_t q_stride, size_t q_batch_stride,
                           const float *d_t, int nt, size_t t_stride, size_t t_batch_stride, int dim, int k,
                           int batch, int32_t *d_idx, float *d_dist, void *stream);
/* L2 path selector for the *_dev/host float entries (results are identical on every path):
 *   0 = auto: int8 matrix-core distance-GEMM when every element is an integer in [0,255] (OpenCV SIFT layout); otherwise fp16
 *       matrix-core candidate passes + exact fp32 re-rank (dim <= 128; option "l2_float_mfma": 1 (default) = once the previous call of
 *       this context saw non-integer data -- the first such call runs the exact kernel --, 2 = enqueued on every call, 0 = never), or the
 *       exact fp32 VALU kernel in cvflann's summation order;
 *   1 = force the exact fp32 VALU kernel;
 *   2 = force the int8 matrix-core path (MLPL_E_BAD_INPUT if the data are not integer-valued 0..255);
 *   3 = force the fp16 candidate path + exact re-rank (MLPL_E_BAD_INPUT if dim > 128 or a row is outside its range: non-finite,
 *       |x| > 1e15, or largest element below 1e-12). */
int mlpl_set_l2_path(mlpl_ctx *ctx, int mode);

/* ---- ratio test + DMatch emission ----------------------------------------------------------------------
 * Replaces the loops at M/source/matchers.cpp:601-625 (int distances) and :677-701 (float distances):
 * k==2: keep q iff (float)d0 < ratio*(float)d1 (reference ratio = 0.75f); k==1: keep every q.
 * Matches are emitted in ascending queryIdx, imgIdx = -1, distance = (float)d0.
 * out must hold nq entries (per batch item); *n_out receives the count.
 */
int mlpl_ratio_compact_i32(mlpl_ctx *ctx, const int32_t *idx, const int32_t *dist, int nq, int k, float ratio,
                           mlpl_dmatch *out, int *n_out);
int mlpl_ratio_compact_f32(mlpl_ctx *ctx, const int32_t *idx, const float *dist, int nq, int k, float ratio,
                           mlpl_dmatch *out, int *n_out);
int mlpl_ratio_compact_i32_dev(mlpl_ctx *ctx, const int32_t *d_idx, const int32_t *d_dist, int nq, int k,
                               int batch, float ratio, mlpl_dmatch *d_out, int32_t *d_n_out, void *stream);
int mlpl_ratio_compact_f32_dev(mlpl_ctx *ctx, const int32_t *d_idx, const float *d_dist, int nq, int k,
                               int batch, float ratio, mlpl_dmatch *d_out, int32_t *d_n_out, void *stream);

/* ---- getMatches(..., "LINEAR", ...) in one call ---------------------------------------------------------
 * Replaces the whole LINEAR branch, M/source/matchers.cpp:115-135 + 525-714.  desc_type: 0 = CV_8U
 * (cols bytes per row), 5 = CV_32F (cols floats per row); step1/step2 = cv::Mat::step in bytes.
 * Returns the reference's codes (0, -1, -3, -4).  out must hold rows1 entries.
 */
int mlpl_get_matches_linear(mlpl_ctx *ctx, int n_keypoints1, int n_keypoints2, const void *desc1, int rows1,
                            size_t step1, const void *desc2, int rows2, size_t step2, int cols, int desc_type,
                            int ratio_test, mlpl_dmatch *out, int *n_out);
/* getMatches(..., "BRUTEFORCENMS", ...): M/source/matchers.cpp:476-519 -> nmslibMatching<dist_t>(..., "seq_search",
 * "bit_hamming" | "l2") (M/include/nmslib/nmslib_matchers.h:159-424) on the vendored NMSLIB, including its quirks:
 * CV_8U rows are compared WITHOUT their last two bytes (one for odd widths; the wrapper omits the length word that
 * SpaceBitHamming::HiddenDistance strips), CV_32F uses the TRUE L2 distance sqrt(sum) summed in NMSLIB's 4-lane order and
 * the ratio test runs on those, K = 2 always, without ratio test ties emit the larger train id, and there is no
 * minimum-match check.  desc_type 0 = CV_8U, 5 = CV_32F (CV_64F: MLPL_E_UNSUPPORTED). */
int mlpl_get_matches_bruteforce_nms(mlpl_ctx *ctx, int n_keypoints1, int n_keypoints2, const void *desc1, int rows1,
                                    size_t step1, const void *desc2, int rows2, size_t step2, int cols, int desc_type,
                                    int ratio_test, mlpl_dmatch *out, int *n_out);
/* Device-resident, batched: knn + ratio + compaction, nothing leaves the device. */
int mlpl_match_hamming_dev(mlpl_ctx *ctx, const uint8_t *d_q, int nq, size_t q_stride, size_t q_batch_stride,
                           const uint8_t *d_t, int nt, size_t t_stride, size_t t_batch_stride, int nbytes,
                           int ratio_test, float ratio, int batch, int32_t *d_idx, int32_t *d_dist,
                           mlpl_dmatch *d_out, int32_t *d_n_out, void *stream);

/* ---- correspondence gather (pre-step of the pose path) -------------------------------------------------------
 * Replaces the gather + ImgToCamCoordTrans of StereoRefine::addNewCorrespondences (P/source/stereo_pose_refinement.cpp:
 * 428-455, P/source/pose_helper.cpp:1100-1109): p1[i] = ((double)kp1[m.queryIdx] - c0) / f0 rounded to float and widened
 * to double, p2 likewise with kp2[m.trainIdx].  kp1/kp2: device arrays of (x,y) float pairs; K = {fx, fy, cx, cy}.
 * All pointers are device pointers; n = number of matches (known to the host). */
int mlpl_gather_match_points_dev(mlpl_ctx *ctx, const mlpl_dmatch *d_matches, int n, const float *d_kp1, const float *d_kp2,
                                 const double K0[4], const double K1[4], double *d_p1, double *d_p2, void *stream);

/* ---- pre/post steps of the pose path (host-pointer forms) ----------------------------------------------------------
 * mlpl_img_to_cam: ImgToCamCoordTrans (P/source/pose_helper.cpp:1100-1109), pts = n x (x,y) floats rewritten in place,
 *   K = {fx, fy, cx, cy}.
 * mlpl_remove_lens_dist: Remove_LensDist + LensDist_Oulu (pose_helper.cpp:1169-1279): 10 fixed-point iterations per
 *   point and view, correspondences failing the 0.25 proof gate are dropped (order kept), points rewritten in place,
 *   *n_out = remaining count.  No-op when both coefficient sums are within 1e-3 of zero.  MLPL_E_FAILED (-3) = the
 *   reference's `false` (fewer than 16 correspondences left).
 * mlpl_get_inliers_strict: computeReprojError2 + getInlierMask (pose_helper.cpp:639-664, 3030-3045) as used by
 *   StereoRefine::getInliers (stereo_pose_refinement.cpp:2085-2090): err[i] = fp64 Sampson error (kept double),
 *   mask[i] = err[i] < th2 (STRICT, unlike RANSAC's <=).  Returns the inlier count (>= 0) or a negative error. */
int mlpl_img_to_cam(mlpl_ctx *ctx, float *pts, int n, const double K[4]);
int mlpl_remove_lens_dist(mlpl_ctx *ctx, float *points1, float *points2, int n, const double dist1[8], const double dist2[8],
                          int *n_out);
int mlpl_get_inliers_strict(mlpl_ctx *ctx, const double *p1, const double *p2, int n, const double E[9], double th2, double *err,
                            uint8_t *mask);

/* ---- robust essential matrix ----------------------------------------------------------------------------
 * Replaces poselib::estimateEssentialMat(E,p1,p2,"RANSAC",th,refine,mask) (P/source/pose_estim.cpp:857-890)
 * = findEssentialMat (P/source/five-point-nister/five-point.cpp:69-148) = CvModelEstimator3::runRANSAC
 * (modelest.cpp:343-474) with the Nister solver CvEMEstimator::run5Point (five-point.cpp:366-471) and
 * Sampson scoring computeReprojError3 (five-point.cpp:476-503).
 * p1,p2: n x 2 doubles (camera-normalised).  The reference hard-codes max_iters=1000, confidence=0.999
 * and seeds std::srand(time) (modelest.cpp:58); here they are parameters, and `seed` reproduces the glibc
 * srand(seed)/rand() sample stream so that a fixed seed gives the CPU path's hypotheses.
 * refit != 0 = the reference's `lesqu` least-squares step on all inliers (modelest.cpp:420-464).
 * mask: n bytes, 1 = inlier.  Returns 0 on success, MLPL_E_FAILED when no model was found
 * (reference returns false), -1 on bad input.
 */
int mlpl_ransac_essential(mlpl_ctx *ctx, const double *p1, const double *p2, int n, double thresh,
                          double confidence, int max_iters, int refit, uint32_t seed, double E[9],
                          uint8_t *mask, int *n_inliers, int *iters_used);
/* Device-resident points; E/mask/n_inliers/iters_used are HOST outputs (the replay of the sequential
 * best/niters logic needs one device->host hop).  d_p1/d_p2: n x 2 doubles on the device. */
int mlpl_ransac_essential_dev(mlpl_ctx *ctx, const double *d_p1, const double *d_p2, int n, double thresh,
                              double confidence, int max_iters, int refit, uint32_t seed, double E[9],
                              uint8_t *d_mask, int *n_inliers, int *iters_used, void *stream);

/*
 * CvModelEstimator3::runLMeDS (modelest.cpp:483-564) as findEssentialMat drives it for method LMEDS (five-point.cpp:125-129;
 * estimateEssentialMat passes prob = 0.999, pose_estim.cpp:874-877, findEssentialMat maxIters = 2000):
 * niters = clamp(round(log(1-confidence)/log(1-0.55^5)), 3, max_iters) samples from the same glibc stream as RANSAC, the 5-point
 * solver on each, the MEDIAN of the float Sampson errors per model (exact radix select over the bit patterns, the order the
 * reference's int sort gives), the first model with the smallest median, sigma = max(2.5*1.4826*(1+5/(n-5))*sqrt(median), 0.001),
 * mask = err <= sigma^2.  Returns 0, MLPL_E_FAILED when no model was found or fewer than 5 correspondences pass (the mask and
 * *n_inliers are still written in the latter case), -1 on bad input (n must exceed 5).  *min_median may be NULL.
 */
int mlpl_lmeds_essential(mlpl_ctx *ctx, const double *p1, const double *p2, int n, double confidence, int max_iters,
                         uint32_t seed, double E[9], uint8_t *mask, int *n_inliers, double *min_median);
int mlpl_lmeds_essential_dev(mlpl_ctx *ctx, const double *d_p1, const double *d_p2, int n, double confidence, int max_iters,
                             uint32_t seed, double E[9], uint8_t *d_mask, int *n_inliers, double *min_median, void *stream);

/*
 * CvModelEstimator3::runARRSAC (modelest.cpp:197-341) as findEssentialMat drives it for method ARRSAC, the DEFAULT method of
 * poselib::estimateEssentialMat (pose_estim.h:204-210, pose_estim.cpp:866-869; five-point.cpp:120-124): theia::Arrsac(5, thresh^2, 500
 * hypotheses, blocks of 100, inner RANSAC on 14 / at least 8 points) (include/arrsac/arrsac.h:236-547) over the 5-point solver with
 * CvEMEstimator::ValidModel (five-point.cpp:534-601), PROSAC and uniform sampling from two cv::RNG streams, Wald's sequential test per
 * hypothesis, then the preemptive breadth-first stage; the mask is findInliers of the winner; `refine` != 0 runs
 * robustEssentialRefine (pose_estim.cpp:337-792) on its inliers (>= 50 needed) and returns the refined matrix with the unrefined
 * model's mask, as the reference does (modelest.cpp:280-318).  Solver, validity test, all error evaluations and the refinement run on
 * the device in speculative batches; the sequential decisions are taken on the host.
 * rng_state: in/out, the states of the reference's two function-local `static cv::RNG rng;` (prosac_sampler.h:115, random_sampler.h:65),
 * which live as long as the PROCESS there: both are 0xffffffff before the first ARRSAC call of a program and carry over from call to
 * call -- the caller keeps them (the C++ facade keeps one pair per process).  Returns 0, MLPL_E_FAILED when no hypothesis passed or the
 * winner has < 15 inliers (< 50 when n > 200) (modelest.cpp:275-278; the mask and *n_inliers are still written then), -1 on bad input
 * (n must exceed 5; with exactly 5 correspondences findEssentialMat never reaches runARRSAC, use mlpl_solve_5pt).
 */
int mlpl_arrsac_essential(mlpl_ctx *ctx, const double *p1, const double *p2, int n, double thresh, int refine, uint64_t rng_state[2],
                          double E[9], uint8_t *mask, int *n_inliers);
int mlpl_arrsac_essential_dev(mlpl_ctx *ctx, const double *d_p1, const double *d_p2, int n, double thresh, int refine,
                              uint64_t rng_state[2], double E[9], uint8_t *d_mask, int *n_inliers, void *stream);
/* A batch of ARRSAC problems in one call (estimateEssentialMat's default method, pose_estim.h:207): problem b = correspondences d_p1 / d_p2 +
 * b * stride * 2 doubles (device), counts[b] in [6, stride] of them (host), its own pair of cv::RNG states rng_states[2 b], [2 b + 1] (host,
 * in / out -- the reference's samplers draw from process-wide streams; here every problem owns a pair, so problems do not depend on each
 * other).  Every problem runs the sequential program of mlpl_arrsac_essential_dev on its own stack (a fiber on a worker thread) and the launches of all runs that
 * stand at the same point of their control flow are merged into one launch per kernel (csrc/batch_hub.h), 128 problems at a time.
 * Outputs per problem: status[b] (0; MLPL_E_FAILED; other < 0 end the call), E + 9 b, n_inliers[b] (optional), its inlier mask at d_masks +
 * b * stride (device) -- what mlpl_arrsac_essential_dev returns for the problem alone with the same stream states. */
int mlpl_arrsac_essential_batch_dev(mlpl_ctx *ctx, int n_problems, const double *d_p1, const double *d_p2, int stride, const int32_t *counts, double thresh,
                                    int refine, uint64_t *rng_states, double *E, uint8_t *d_masks, int32_t *n_inliers, int32_t *status, void *stream);
/*
 * poselib::robustEssentialRefine(points1, points2, E_init, E_refined, th, 0, true, ..., mask, 0) (pose_estim.h:225-228,
 * pose_estim.cpp:337-792) for the essential-matrix model without normalisation: iteratively re-weighted (pseudo-Huber on the Sampson
 * distance, threshold th) 9 x 9 eigenproblem over the correspondences with mask != 0 (mask == NULL: all), the closest essential matrix
 * after every round, the reference's stopping tests; one workgroup on the device.  Fewer than 50 correspondences or a rank-deficient
 * system return E_init.  info (may be NULL) = {the reference's loop counter at the end (index of the round a stopping
 * test fired in -- one less than the rounds executed -- or 50), status: 0 converged or exhausted, 1 stopped on an invalid matrix (last
 * valid one returned), 2 rejected: fewer than 50 correspondences, 3 rejected: rank-deficient system (E_init returned by both)}.  StereoRefine's refineRTold step (stereo_pose_refinement.cpp:1460-1474) and ARRSAC's
 * `refine` are this.
 */
int mlpl_robust_essential_refine(mlpl_ctx *ctx, const double *p1, const double *p2, int n, const uint8_t *mask, const double E_init[9],
                                 double th, double E_refined[9], int info[2]);

/* ARRSAC's model estimators on ONE sample (EssentialMatEstimatorTheia::EstimateModel / EstimateModelNonminimal, modelest.cpp:111-178), a
 * building block exposed for parity tests: idx = m indices into p1/p2; kind 0 = the 5-point solver on 5..7 correspondences (the reference
 * runs run5Point on them: cv::SVD of an m x 9 system, its four last right singular vectors), kind 1 = cv::findFundamentalMat(FM_8POINT)
 * on 8..14.  E_out: 10 x 9 doubles, the first *n_models are models (solver order, library sign convention); valid[i] = ValidModel
 * (five-point.cpp:534-601) on the sample's correspondences.  thresh only sizes internal buffers' inlier rows. */
int mlpl_arrsac_sample_models(mlpl_ctx *ctx, const double *p1, const double *p2, int n, const int32_t *idx, int m, int kind, double thresh,
                              double *E_out, int32_t *n_models, uint8_t *valid);

/* Statistics of the last mlpl_arrsac_essential[_dev] call: {k of the initial hypothesis set, hypotheses after it, PROSAC samples,
 * inner-RANSAC samples, inner-RANSAC restarts, samples of the preemptive stage, correspondence index where that stage ended,
 * hypotheses left there, device batches, samples solved on the device, samples the control flow consumed, refinement status
 * (-1 not run, 0 converged, 1 stopped on an invalid matrix, 2 / 3 rejected: too few points / rank-deficient system)}. */
int mlpl_arrsac_last_stats(mlpl_ctx *ctx, long long stats[12]);

/*
 * USAC essential-matrix estimation with the Nister minimal solver -- poselib::estimateEssentialOrPoseUSAC (pose_estim.h:212-223,
 * pose_estim.cpp:1737-2244) -> estimateEssentialMatUsac (source/usac/usac_estimations.cpp:283-735) -> USAC<EssentialMatEstimator>::solve
 * (include/usac/estimators/USAC.h:335-620, EssentialMatEstimator.h): PROSAC or uniform sampling, sample pre-validation, the oriented
 * epipolar constraint on every model, Wald's sequential test over a shuffled evaluation order with re-estimated delta / epsilon,
 * local optimisation (5 inner repetitions of a 14-point fit + 4 re-weighted refits; `refine` below) on every new best model, and the
 * SPRT-aware stopping criterion.  The harness default estimator (tests/poselib-test/main.cpp:734).  Minimal solves, validity tests and
 * every error evaluation run on the device in speculative batches, the refits as one launch (REF_WEIGHTS) or one chain of launches
 * (5-point refinements) per local optimisation; the sequential
 * decisions are taken on the host.  params->check_degeneracy switches the degeneracy handling of
 * UsacChkDegenType::DEGEN_USAC_INTERNAL on: EssentialMatEstimator::testSolutionDegeneracyRot / NoMot and upgradeDegenerateModel
 * (EssentialMatEstimator.h:1334-1362, 1511-1663, 1838-1911, 2098-2361); results through mlpl_usac_last_degeneracy.  Not built: the
 * homography test the reference adds with the 8-point refinements (:1368-1505; its upgrade branch :1958-2015 writes past a vector and
 * reads a translation nothing has set), DEGEN_QDEGSAC.
 * The reference seeds srand(time(nullptr)) and shuffles its evaluation order on the process-wide stream; `seed` is that seed.
 * estimator: PoseEstimator value, 0 = POSE_NISTER, 2 = POSE_STEWENIUS (both are exact 5-point solvers with the same real solution set;
 * the device solver serves both, solutions ordered by the library's convention); refine: RefineAlg value of the local optimisation's
 * fits (EssentialMatEstimator::generateRefinedModel :526-850, findWeights :2366-2428): 0 = REF_WEIGHTS (8-point fit with Torr weights),
 * 4 = REF_STEWENIUS, 5 = REF_STEWENIUS_WEIGHTS (ConfigUSAC's default, pose_estim.h:99-100), 6 = REF_NISTER, 7 = REF_NISTER_WEIGHTS -- the
 * five-point solver on all points of the fit set (unit bearing vectors; the _WEIGHTS forms scale the rows of a re-weighted step by the
 * pseudo-Huber weights of P/source/usac/utils/weightingEssential.cpp:190-206), of its solutions the one with the smallest Sampson-error
 * sum over the inliers of the best model so far.  Other values (REF_8PT_PSEUDOHUBER, REF_EIG_KNEIP(_WEIGHTS), POSE_EIG_KNEIP):
 * MLPL_E_UNSUPPORTED.  check_degeneracy = 3 goes with refine = 0 only (usac_estimations.cpp:368-375).
 * results[12] = {1, hypotheses, models, samples rejected by pre-validation, models rejected by the oriented constraint, inliers of the
 * best model, correspondences verified, local optimisations, SPRT delta and epsilon the reference reports back (newest history entry,
 * usac_estimations.cpp:459-468 halves epsilon itself), delta and epsilon at the end}.  mask: n bytes, 1 = inlier of the returned model.
 * Returns 0; MLPL_E_FAILED when solve() refuses (fewer than 5 correspondences, or fewer than 20 with PROSAC).
 */
typedef struct {
    double th;             /* inlier threshold in camera coordinates (not squared) */
    double conf;           /* 0.99 */
    int32_t max_hyp;       /* 50000 */
    int32_t estimator;     /* poselib::PoseEstimator */
    int32_t refine;        /* poselib::RefineAlg */
    uint32_t seed;
    double prosac_beta;    /* 0.09, or the SPRT delta when the automatic PROSAC parameter is on */
    double sprt_delta;     /* 0.05 */
    double sprt_epsilon;   /* 0.15 */
    double sprt_mS;        /* 8.5 for Nister on the first call of a process, then models / hypotheses so far */
    double sprt_tM;        /* 2314 (Nister), 2736 (Stewenius) */
    const uint32_t *sorted_idx; /* HOST pointer: NULL = uniform sampling; else n indices, best match first = PROSAC */
    int32_t check_degeneracy; /* 0 = DEGEN_NO_CHECK; 1 = the rotation-only / no-motion tests after every new best model and the
                                 upgrade R -> R + t, no motion -> t (DEGEN_USAC_INTERNAL); 3 = also after every local optimisation
                                 (what estimateEssentialMatUsac switches on with the 8-point refinements, usac_estimations.cpp:368-371) */
    int32_t reserved;
    double th_pixels;      /* ConfigUSAC::th_pixels (0.8): with focal_length the angular threshold of the degeneracy tests */
    double focal_length;   /* ConfigUSAC::focalLength (800) */
} mlpl_usac_params;
void mlpl_usac_default_params(mlpl_usac_params *p, double th);
int mlpl_usac_essential(mlpl_ctx *ctx, const double *p1, const double *p2, int n, const mlpl_usac_params *params, double E[9],
                        uint8_t *mask, double results[12]);
int mlpl_usac_essential_dev(mlpl_ctx *ctx, const double *d_p1, const double *d_p2, int n, const mlpl_usac_params *params, double E[9],
                            uint8_t *d_mask, double results[12], void *stream);
/*
 * A batch of USAC problems in one call -- the harness runs estimateEssentialOrPoseUSAC once per image pair (T/poselib-test/main.cpp:
 * 1440-2072, RobMethod "USAC" is its default, :734); a rank's share of a batch of pairs is many such problems.  Problem b: correspondences
 * d_p1 / d_p2 + b * stride * 2 doubles (device, camera coordinates), counts[b] <= stride of them (host), parameters params[b] (host; its own
 * seed, thresholds, PROSAC order).  Every problem runs the sequential program of mlpl_usac_essential_dev on its own stack (a fiber on a worker thread), and the
 * launches of all runs that stand at the same point of their control flow are merged into ONE launch per kernel (problem = grid
 * dimension; csrc/batch_hub.h), 128 problems at a time.  Outputs per problem: status[b] (0; MLPL_E_FAILED = solve() refused; < 0 other
 * errors, which also end the call), E + 9 b, results + 12 b, its inlier mask at d_masks + b * stride (device, optional), degen + 16 b
 * (optional: what mlpl_usac_last_degeneracy's info[] returns), its decision trace at trace + b * trace_cap * 16 with trace_lens[b]
 * records (optional, diagnostics).  Every output equals what mlpl_usac_essential_dev returns for that problem alone
 * (tests/test_gpu_usac_batch.py: records and event traces of 64 problems).
 */
int mlpl_usac_essential_batch_dev(mlpl_ctx *ctx, int n_problems, const double *d_p1, const double *d_p2, int stride, const int32_t *counts,
                                  const mlpl_usac_params *params, double *E, uint8_t *d_masks, double *results, int32_t *status, double *degen,
                                  double *trace, int trace_cap, int32_t *trace_lens, void *stream);
/* Statistics of the last mlpl_usac_essential[_dev] call: {device batches, samples solved on the device, samples the control flow
 * consumed, local-optimisation launches / chain runs, of which chain resumes, rechecks of the solution choices after a repetition
 * stored a new best model (5-point refinements), chains re-run because a choice changed, Jacobi sweeps (REF_WEIGHTS)}.
 * After a BATCHED call (mlpl_usac_essential_batch_dev, mlpl_pair_pose_batch_usac_dev): {hub rounds, merged launches, microseconds the hub
 * waited for host work (summed over the lanes), microseconds of device work (summed), microseconds spent starting the runs, microseconds
 * of the whole call, lanes (cohorts served side by side) the call used, cohorts}. */
int mlpl_usac_last_stats(mlpl_ctx *ctx, long long stats[8]);
/* What the degeneracy tests of the last mlpl_usac_essential[_dev] call found -- the quantities estimateEssentialMatUsac hands to
 * estimateEssentialOrPoseUSAC (usac_estimations.cpp:564-636, 689-726; pose_estim.cpp:2044-2133 takes the decision "degenerate" from
 * them): info[16] = {1 if the tests ran, inliers of the best rotation-only model (degen_inlier_count_rot), inliers of "no motion"
 * (degen_inlier_count_noMot), degeneracy type of the last test (EssentialMatEstimator.h:168-175 bit set), R_degenerate[9] row-major,
 * 0...}; flags_rot / flags_nomot (n bytes each, may be NULL): the inlier masks of the two degenerate models.  Returns 0, or
 * MLPL_E_BAD_INPUT when n differs from the last call's correspondence count (masks requested) or no call has been made. */
int mlpl_usac_last_degeneracy(mlpl_ctx *ctx, double info[16], uint8_t *flags_rot, uint8_t *flags_nomot, int n);

/*
 * One image pair through the whole hot path, device-resident (the per-pair body of the reference harness loop,
 * tests/poselib-test/main.cpp:1440-2072, and of StereoRefine's first call): Hamming 2-NN + 0.75 ratio test -> gather of the matched
 * keypoints with ImgToCamCoordTrans -> RANSAC essential matrix -> cheirality.  d_q/d_t: dense nq/nt x nbytes descriptors,
 * d_kp1/d_kp2: (x, y) float pixel coordinates per keypoint, K = {fx, fy, cx, cy}, thresh in camera units.  Two host hops.
 * status: 0 = pose found, -1 = fewer than 16 matches, -2 = RANSAC found no model.
 */
typedef struct {
    int32_t n_matches, n_inliers, n_good, status, iters, pad;
    double E[9], R[9], t[3];
} mlpl_pair_result;
int mlpl_pair_pose_dev(mlpl_ctx *ctx, const uint8_t *d_q, int nq, const uint8_t *d_t, int nt, int nbytes, const float *d_kp1,
                       const float *d_kp2, const double K0[4], const double K1[4], double thresh, int max_iters, double confidence,
                       int refit, uint32_t seed, double dist, mlpl_pair_result *out, void *stream);

/*
 * A BATCH of image pairs through the same pipeline with the pair as a grid dimension of every launch (the reference harness loop over
 * image pairs, tests/poselib-test/main.cpp:1440-2072; BASELINE config 5): d_q / d_t / d_kp1 / d_kp2 hold n_pairs contiguous items
 * ([n_pairs][nq][nbytes], [n_pairs][nt][nbytes], [n_pairs][nq][2], [n_pairs][nt][2]), seeds[n_pairs] and out[n_pairs] are HOST arrays.
 * Every pair's record equals what mlpl_pair_pose_dev returns for it with the same seed.  Host hops per internal batch of 256 pairs
 * (option "pair_batch"): the match counts, one per RANSAC pass (the first 324 iterations of every pair, then the rest for the pairs
 * the adaptive bound has not stopped), the results.  refit != 0 runs the pairs one by one through mlpl_pair_pose_dev.
 */
int mlpl_pair_pose_batch_dev(mlpl_ctx *ctx, int n_pairs, const uint8_t *d_q, int nq, const uint8_t *d_t, int nt, int nbytes, const float *d_kp1,
                             const float *d_kp2, const double K0[4], const double K1[4], double thresh, int max_iters, double confidence,
                             int refit, const uint32_t *seeds, double dist, mlpl_pair_result *out, mlpl_dmatch *d_matches_out, void *stream);
/* mlpl_pair_pose_batch_dev (refit = 0) with `lanes` calls in flight on one GPU: lane l = (ctxs[l], streams[l]: a context and a stream of
 * its own, same device) runs the l-th contiguous share of the batch on its own host thread inside this call, which returns when every
 * lane has finished AND its stream is idle.  Inputs must be complete before the call (they were produced on another stream).
 * lane_span_ms: NULL or 2 * lanes doubles {start, end} of each lane's call in milliseconds since entry (diagnostics). */
int mlpl_pair_pose_batch_lanes_dev(mlpl_ctx *const *ctxs, void *const *streams, int lanes, int n_pairs, const uint8_t *d_q, int nq, const uint8_t *d_t, int nt,
                                   int nbytes, const float *d_kp1, const float *d_kp2, const double K0[4], const double K1[4], double thresh, int max_iters,
                                   double confidence, const uint32_t *seeds, double dist, mlpl_pair_result *out, mlpl_dmatch *d_matches_out,
                                   double *lane_span_ms);
/*
 * The same batch with USAC -- the reference harness' default RobMethod (T/poselib-test/main.cpp:734) -- as the robust estimator:
 * matching, match counts, gather + ImgToCamCoordTrans as above, then mlpl_usac_essential_batch_dev on the pairs' correspondences (every
 * pair's sequential program on its own stack, every launch merged over the pairs), then the batched cheirality step.  *usac:
 * the parameters every pair runs with (th, conf, max_hyp, estimator, refine, SPRT start values, degeneracy tests; its seed and sorted_idx
 * are ignored); seeds[n_pairs]: the pairs' seeds; prosac != 0: PROSAC sampling in the order of the matching costs -- poselib::
 * getSortedMatchIdx' std::sort of the pair's matches (pose_helper.cpp:2896-2923) -- else uniform sampling.  Record per pair:
 * status 0, -1 (fewer than 16 matches) or -2 (USAC failed); iters = hypotheses, n_inliers = inliers of the USAC model, n_good, E, R, t --
 * what mlpl_match_hamming_dev + mlpl_gather_match_points_dev -> mlpl_usac_essential_dev -> mlpl_recover_pose_dev return for the pair.
 */
int mlpl_pair_pose_batch_usac_dev(mlpl_ctx *ctx, int n_pairs, const uint8_t *d_q, int nq, const uint8_t *d_t, int nt, int nbytes, const float *d_kp1,
                                  const float *d_kp2, const double K0[4], const double K1[4], const mlpl_usac_params *usac, int prosac,
                                  const uint32_t *seeds, double dist, mlpl_pair_result *out, mlpl_dmatch *d_matches_out, void *stream);
/* ... and with ARRSAC, estimateEssentialMat's default method (pose_estim.h:207; `refine` = robustEssentialRefine on the winner's inliers):
 * mlpl_arrsac_essential_batch_dev on the pairs' correspondences.  rng_states[2 * n_pairs]: a pair of cv::RNG states per image pair (in / out).
 * Record per pair: status 0 / -1 / -2 (ARRSAC failed), iters = 0, n_inliers = inliers of the unrefined winner, n_good, E, R, t. */
int mlpl_pair_pose_batch_arrsac_dev(mlpl_ctx *ctx, int n_pairs, const uint8_t *d_q, int nq, const uint8_t *d_t, int nt, int nbytes, const float *d_kp1,
                                    const float *d_kp2, const double K0[4], const double K1[4], double thresh, int refine, uint64_t *rng_states, double dist,
                                    mlpl_pair_result *out, mlpl_dmatch *d_matches_out, void *stream);
/* poselib::getSortedMatchIdx (P/source/pose_helper.cpp:2896-2923) on a HOST match list: the indices of the matches in the order
 * std::sort leaves them when comparing the distances -- the PROSAC order estimateEssentialOrPoseUSAC and the batch entry above use. */
int mlpl_sorted_match_idx(const mlpl_dmatch *matches, int n, uint32_t *sorted_idx);
/* The same batch behind the matching -- the batched form of mlpl_ransac_essential_dev (refit = 0) followed, with recover_pose != 0, by
 * mlpl_recover_pose_dev on the RANSAC inliers: problem b's correspondences are d_p1 / d_p2 + b * stride * 2 (camera coordinates, n x 2
 * doubles, device), counts[b] <= stride of them (host array), seeds[b] its srand() seed.  Records as above with n_matches = counts[b];
 * status -1 for fewer than 6 correspondences, -2 when RANSAC finds no model; R, t, n_good stay zero without recover_pose.  d_masks: NULL or
 * a device block [n_problems][stride] that receives the inlier masks (after cheirality when recover_pose is set, as mlpl_recover_pose_dev
 * leaves them).  Every record equals what the single-problem entries return for that problem (tests/test_gpu_batch.py). */
int mlpl_ransac_essential_batch_dev(mlpl_ctx *ctx, int n_problems, const double *d_p1, const double *d_p2, int stride, const int32_t *counts, double thresh,
                                    int max_iters, double confidence, const uint32_t *seeds, int recover_pose, double dist, mlpl_pair_result *out,
                                    uint8_t *d_masks, void *stream);
/* d_matches_out: NULL, or a device block [n_pairs][nq] that receives every pair's match list (out[i].n_matches valid entries each,
 * ascending queryIdx) -- what a caller gathers beside the pose records (needs refit = 0).
 * Statistics of the last mlpl_pair_pose_batch_dev call: {RANSAC passes, pair slots summed over the passes, pairs redone by the
 * single-pair pipeline (a device-evaluated iteration bound differed from the host's libm, or the pair's raw rand() stream ran out on the
 * device), host microseconds spent generating the raw rand() streams (the samples themselves are drawn on the device), essential matrices scored, Sampson evaluations (matrices x correspondences), RANSAC iterations executed, 0}. */
int mlpl_pair_batch_last_stats(mlpl_ctx *ctx, long long stats[8]);

/* Building blocks, exposed for parity tests and for callers that schedule the phases themselves. */
/* 5-point minimal solver, one wavefront per sample: samples = n_samples x 5 indices into p1/p2 (host).
 * E_out: n_samples x 10 x 9 doubles, n_models: n_samples ints (host). Replaces run5Point (five-point.cpp:366-471). */
int mlpl_solve_5pt(mlpl_ctx *ctx, const double *p1, const double *p2, int n, const int32_t *samples, int n_samples,
                   double *E_out, int32_t *n_models);
/* Sampson scoring of n_models 3x3 matrices against n correspondences: count[i] = #{err <= thresh^2},
 * err_sum[i] = sum of the float-rounded errors (findInliers + cv::sum(err), modelest.cpp:69-83,407). */
int mlpl_score_models(mlpl_ctx *ctx, const double *p1, const double *p2, int n, const double *E, int n_models,
                      double thresh, int32_t *count, double *err_sum);

/* Inlier counts only, by the division-free predicate the RANSAC passes use (exactly (double)(float)(N / D) <= thresh2, see
 * sampson_inlier in ransac_5pt.hip); thresh2 is the SQUARED threshold as the reference holds it (modelest.cpp:79).
 * shape: 0 = automatic, 1 = 4-lanes-per-model kernel, 2 = block-per-model kernel (for tests). */
int mlpl_count_models(mlpl_ctx *ctx, const double *p1, const double *p2, int n, const double *E, int n_models, double thresh2, int shape,
                      int32_t *count);

/* Median of the float Sampson errors per model, as runLMeDS takes it (modelest.cpp:540-544: errors sorted as int bit patterns;
 * even n: float sum of the two middle values * 0.5).  median: n_models doubles (host). */
int mlpl_median_models(mlpl_ctx *ctx, const double *p1, const double *p2, int n, const double *E, int n_models, double *median);

/* Statistics of the last mlpl_ransac_essential[_dev] call on this context: {iterations executed, essential matrices
 * scored}.  Used by bench.py to turn the scoring kernel's time into algorithmic FLOP/s. */
int mlpl_ransac_last_stats(mlpl_ctx *ctx, long long stats[2]);




/* ---- cheirality / pose recovery --------------------------------------------------------------------------
 * Replaces poselib::getPoseTriangPts (P/source/pose_estim.cpp:913-946) = recoverPose
 * (five-point.cpp:150-338) with t_only empty: decomposeEssentialMat (:340-352), four triangulations,
 * masks z*w>0, (P*Q)z*w>0, z<dist, AND with mask_inout (NULL = all ones), candidate choice by the
 * reference's if-chain.  R: 9, t: 3, Q: n x 3 doubles, mask_inout: n bytes (nonzero = keep; rewritten as
 * 0/255 like the reference's comparison masks).  Returns the number of good points (>=0) or a negative error.
 */
int mlpl_recover_pose(mlpl_ctx *ctx, const double E[9], const double *p1, const double *p2, int n, double dist,
                      double R[9], double t[3], double *Q, uint8_t *mask_inout);
/* The same with `t_only` given (getPoseTriangPts(..., translatE = true), five-point.cpp:178-193): R = I and only the two
 * candidates [I|t], [I|-t] compete.  t_only = getTfromTransEssential(E) (P/source/pose_helper.cpp:422-433). */
int mlpl_recover_pose_translation(mlpl_ctx *ctx, const double t_only[3], const double *p1, const double *p2, int n,
                                  double dist, double R[9], double t[3], double *Q, uint8_t *mask_inout);
/* mlpl_recover_pose on device-resident correspondences: d_p1, d_p2 n x 2 doubles, d_Q n x 3 doubles or NULL, d_mask_inout n bytes
 * (nonzero = use; rewritten with the chosen candidate's mask) or NULL -- all device pointers; E, R, t are host.  One host hop:
 * the four candidate counts (five-point.cpp:299-336) decide whose outputs are kept.  Returns the number of valid 3-D points. */
int mlpl_recover_pose_dev(mlpl_ctx *ctx, const double E[9], const double *d_p1, const double *d_p2, int n, double dist,
                          double R[9], double t[3], double *d_Q, uint8_t *d_mask_inout, void *stream);


/* The diagnostics entry points (mlpl_debug_*: traces, clock stamps, solver statistics, self-tests -- exported by the same library, used by
 * the tests and by bench.py, not part of the drop-in surface) are declared in mlpl_debug.h. */

#ifdef __cplusplus
}
#endif
#endif /* MLPL_C_H */
