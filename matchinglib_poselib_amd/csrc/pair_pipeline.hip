// pair_pipeline.hip -- one image pair, device-resident, through the whole hot path in one call:
//   Hamming 2-NN + ratio test (matchers.cpp:525-631)  ->  gather of the matched keypoints + ImgToCamCoordTrans
//   (stereo_pose_refinement.cpp:428-455, pose_helper.cpp:1100-1109)  ->  RANSAC essential matrix (five-point.cpp:69-148)  ->
//   cheirality / pose (pose_estim.cpp:913-946).
// This is the per-pair body of the reference harness loop (tests/poselib-test/main.cpp:1440-2072) and of StereoRefine's first
// call, as one C-ABI entry so that a caller (or one host thread per stream) pays two host hops per pair -- the match count
// and the final state -- and nothing else leaves the device.  Every step is the library's own *_dev entry point.

#include "mlpl_internal.h"

#include <cstring>

using namespace mlpl;

extern "C" int mlpl_pair_pose_dev(mlpl_ctx *ctx, const uint8_t *d_q, int nq, const uint8_t *d_t, int nt, int nbytes, const float *d_kp1,
                                  const float *d_kp2, const double K0[4], const double K1[4], double thresh, int max_iters,
                                  double confidence, int refit, uint32_t seed, double dist, mlpl_pair_result *out, void *stream) {
    if (!ctx || !d_q || !d_t || !d_kp1 || !d_kp2 || !K0 || !K1 || !out || nq < 1 || nt < 2 || nbytes < 1) {
        set_error("mlpl_pair_pose_dev: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = pick_stream(ctx, stream);
    std::memset(out, 0, sizeof(*out));

    // one block of the arena for everything that lives across the steps
    const size_t n = (size_t)nq;
    const size_t off_idx = 0, off_dist = off_idx + n * 8, off_match = off_dist + n * 8, off_p1 = off_match + n * 16,
                 off_p2 = off_p1 + n * 16, off_mask = off_p2 + n * 16, off_cnt = (off_mask + n + 255) / 256 * 256;
    void *blk = nullptr;
    int rc = ws_get(ctx, WS_PIPE, off_cnt + 256, &blk);
    if (rc) return rc;
    char *b = (char *)blk;
    int32_t *d_cnt = (int32_t *)(b + off_cnt);
    mlpl_dmatch *d_m = (mlpl_dmatch *)(b + off_match);

    rc = mlpl_match_hamming_dev(ctx, d_q, nq, (size_t)nbytes, 0, d_t, nt, (size_t)nbytes, 0, nbytes, 1, 0.75f, 1, (int32_t *)(b + off_idx),
                                (int32_t *)(b + off_dist), d_m, d_cnt, s);
    if (rc) return rc;
    int32_t cnt = 0;
    MLPL_HIP_TRY(hipMemcpyAsync(&cnt, d_cnt, 4, hipMemcpyDeviceToHost, s));
    MLPL_HIP_TRY(hipStreamSynchronize(s));  // host hop 1: the match count sizes everything that follows
    out->n_matches = cnt;
    if (cnt < 16) {  // below the reference's working minimum (Remove_LensDist / StereoRefine refuse fewer than 16)
        out->status = -1;
        return MLPL_OK;
    }
    double *d_p1 = (double *)(b + off_p1), *d_p2 = (double *)(b + off_p2);
    uint8_t *d_mask = (uint8_t *)(b + off_mask);
    rc = mlpl_gather_match_points_dev(ctx, d_m, cnt, d_kp1, d_kp2, K0, K1, d_p1, d_p2, s);
    if (rc) return rc;
    int ninl = 0, iters = 0;
    rc = mlpl_ransac_essential_dev(ctx, d_p1, d_p2, cnt, thresh, confidence, max_iters, refit, seed, out->E, d_mask, &ninl, &iters, s);
    out->iters = iters;
    if (rc == MLPL_E_FAILED) {
        out->status = -2;
        return MLPL_OK;
    }
    if (rc) return rc;
    out->n_inliers = ninl;
    rc = mlpl_recover_pose_dev(ctx, out->E, d_p1, d_p2, cnt, dist, out->R, out->t, nullptr, d_mask, s);
    if (rc < 0) return rc;
    out->n_good = rc;
    return MLPL_OK;
}
