// pose_prepost.hip -- the embarrassingly parallel steps either side of the robust estimation.
//
// Replaces, in reference poselib/source/pose_helper.cpp:
//   :1100-1109  ImgToCamCoordTrans  -> img_to_cam_kernel        (also fused into gather_match_points_kernel)
//   :1169-1279  Remove_LensDist / LensDist_Oulu -> undistort_kernel + ordered compaction (drops failing pairs)
//   :639-664    computeReprojError2 + :3030-3045 getInlierMask (strict <) -> inliers_strict_kernel
// One thread per correspondence; the same double->float rounding points as the reference, no FMA contraction.

#include "mlpl_internal.h"

namespace mlpl {
namespace {

__global__ void img_to_cam_kernel(float *__restrict__ pts, int n, double fx, double fy, double cx, double cy) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    pts[2 * i] = (float)__ddiv_rn(__dsub_rn((double)pts[2 * i], cx), fx);
    pts[2 * i + 1] = (float)__ddiv_rn(__dsub_rn((double)pts[2 * i + 1], cy), fy);
}

struct Dist8 {
    double k[8];
};

// LensDist_Oulu (pose_helper.cpp:1241-1279), 10 iterations; c = corrected (in/out), d = distorted
__device__ __forceinline__ void oulu_terms(float cx, float cy, const Dist8 &D, double &rad_corr, double &d0, double &d1) {
    const double k1 = D.k[0], k2 = D.k[1], p1 = D.k[2], p2 = D.k[3], k3 = D.k[4], k4 = D.k[5], k5 = D.k[6], k6 = D.k[7];
    const double x = (double)cx, y = (double)cy;
    const double r2 = __dadd_rn(__dmul_rn(x, x), __dmul_rn(y, y));
    const double _2xy = __dmul_rn(__dmul_rn(2.0, x), y);
    const double num = __dadd_rn(1.0, __dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(k3, r2), k2), r2), k1), r2));
    const double den = __dadd_rn(1.0, __dmul_rn(__dadd_rn(__dmul_rn(__dadd_rn(__dmul_rn(k6, r2), k5), r2), k4), r2));
    rad_corr = __ddiv_rn(num, den);
    d0 = __dadd_rn(__dmul_rn(p1, _2xy), __dmul_rn(p2, __dadd_rn(r2, __dmul_rn(__dmul_rn(2.0, x), x))));
    d1 = __dadd_rn(__dmul_rn(p1, __dadd_rn(r2, __dmul_rn(__dmul_rn(2.0, y), y))), __dmul_rn(p2, _2xy));
}

__device__ __forceinline__ bool lens_dist_oulu(float dx, float dy, float &cx, float &cy, const Dist8 &D) {
    double rc, d0, d1;
    for (int it = 0; it < 10; ++it) {
        oulu_terms(cx, cy, D, rc, d0, d1);
        cx = (float)__ddiv_rn(__dsub_rn((double)dx, d0), rc);
        cy = (float)__ddiv_rn(__dsub_rn((double)dy, d1), rc);
    }
    oulu_terms(cx, cy, D, rc, d0, d1);
    const float px = (float)__dsub_rn(__dadd_rn(__dmul_rn((double)cx, rc), d0), (double)dx);
    const float py = (float)__dsub_rn(__dadd_rn(__dmul_rn((double)cy, rc), d1), (double)dy);
    return !(sqrtf(__fadd_rn(__fmul_rn(px, px), __fmul_rn(py, py))) > 0.25f);
}

__global__ void undistort_kernel(float *__restrict__ p1, float *__restrict__ p2, int n, Dist8 D1, Dist8 D2,
                                 uint8_t *__restrict__ valid, int32_t *__restrict__ group_counts) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool ok = false;
    if (i < n) {
        float c1x = p1[2 * i], c1y = p1[2 * i + 1];
        ok = lens_dist_oulu(c1x, c1y, c1x, c1y, D1);
        p1[2 * i] = c1x;
        p1[2 * i + 1] = c1y;
        if (ok) {  // the reference skips the second view once the first failed (pose_helper.cpp:1189-1193)
            float c2x = p2[2 * i], c2y = p2[2 * i + 1];
            ok = lens_dist_oulu(c2x, c2y, c2x, c2y, D2);
            p2[2 * i] = c2x;
            p2[2 * i + 1] = c2y;
        }
        valid[i] = ok ? 1 : 0;
    }
    const unsigned long long bal = __ballot(ok);
    if ((threadIdx.x & 63) == 0) group_counts[(blockIdx.x * blockDim.x + threadIdx.x) >> 6] = __popcll(bal);
}

// stable compaction of the valid correspondences (64 per count group, one wave per group)
__global__ void compact_pairs_kernel(const float *__restrict__ p1, const float *__restrict__ p2, const uint8_t *__restrict__ valid,
                                     const int32_t *__restrict__ group_counts, int n, float *__restrict__ o1, float *__restrict__ o2,
                                     int32_t *__restrict__ n_out) {
    __shared__ int red[4];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = blockIdx.x * 4 + wave;  // 4 groups per 256-thread block
    const int ngroups = (n + 63) / 64;
    // exclusive prefix of the counts of all groups before this block's first group
    int part = 0;
    for (int j = tid; j < blockIdx.x * 4; j += 256) part += group_counts[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) red[wave] = part;
    __syncthreads();
    int base = red[0] + red[1] + red[2] + red[3];
    for (int w = 0; w < wave; ++w) base += (blockIdx.x * 4 + w < ngroups) ? group_counts[blockIdx.x * 4 + w] : 0;
    const int i = grp * 64 + lane;
    const bool ok = (i < n) && valid[i];
    const unsigned long long bal = __ballot(ok);
    const int pos = base + __popcll(bal & ((1ull << lane) - 1ull));
    if (ok) {
        o1[2 * pos] = p1[2 * i];
        o1[2 * pos + 1] = p1[2 * i + 1];
        o2[2 * pos] = p2[2 * i];
        o2[2 * pos + 1] = p2[2 * i + 1];
    }
    if (grp == ngroups - 1 && lane == 0) n_out[0] = base + __popcll(bal);
}

__global__ void inliers_strict_kernel(const double *__restrict__ p1, const double *__restrict__ p2, int n, const double *__restrict__ E,
                                      double th2, double *__restrict__ err, uint8_t *__restrict__ mask, int32_t *__restrict__ count) {
    __shared__ int wave_cnt[4];
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool in = false;
    if (i < n) {
        const double x1 = p1[2 * i], y1 = p1[2 * i + 1], x2 = p2[2 * i], y2 = p2[2 * i + 1];
        const double Ex1_0 = __dadd_rn(__dadd_rn(__dmul_rn(E[0], x1), __dmul_rn(E[1], y1)), E[2]);
        const double Ex1_1 = __dadd_rn(__dadd_rn(__dmul_rn(E[3], x1), __dmul_rn(E[4], y1)), E[5]);
        const double Ex1_2 = __dadd_rn(__dadd_rn(__dmul_rn(E[6], x1), __dmul_rn(E[7], y1)), E[8]);
        const double x2tEx1 = __dadd_rn(__dadd_rn(__dmul_rn(x2, Ex1_0), __dmul_rn(y2, Ex1_1)), Ex1_2);
        const double Etx2_0 = __dadd_rn(__dadd_rn(__dmul_rn(E[0], x2), __dmul_rn(E[3], y2)), E[6]);
        const double Etx2_1 = __dadd_rn(__dadd_rn(__dmul_rn(E[1], x2), __dmul_rn(E[4], y2)), E[7]);
        const double a = __dmul_rn(Ex1_0, Ex1_0), b = __dmul_rn(Ex1_1, Ex1_1), c = __dmul_rn(Etx2_0, Etx2_0), d = __dmul_rn(Etx2_1, Etx2_1);
        const double e = __ddiv_rn(__dmul_rn(x2tEx1, x2tEx1), __dadd_rn(__dadd_rn(__dadd_rn(a, b), c), d));
        err[i] = e;
        in = e < th2;  // strict, pose_helper.cpp:3038
        mask[i] = in ? 1 : 0;
    }
    const unsigned long long bal = __ballot(in);
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(count, wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3]);
}

}  // namespace
}  // namespace mlpl

using namespace mlpl;

extern "C" {

int mlpl_img_to_cam(mlpl_ctx *ctx, float *pts, int n, const double K[4]) {
    if (!ctx || !pts || !K || n < 0) return MLPL_E_BAD_INPUT;
    if (n == 0) return MLPL_OK;
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    void *d;
    int rc = ws_get(ctx, WS_AUX0, (size_t)n * 8, &d);
    if (rc) return rc;
    hipStream_t s = ctx->stream;
    MLPL_HIP_TRY(hipMemcpyAsync(d, pts, (size_t)n * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(img_to_cam_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (float *)d, n, K[0], K[1], K[2], K[3]);
    MLPL_HIP_TRY(hipMemcpyAsync(pts, d, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    MLPL_HIP_TRY(hipStreamSynchronize(s));
    return MLPL_OK;
}

int mlpl_remove_lens_dist(mlpl_ctx *ctx, float *points1, float *points2, int n, const double dist1[8], const double dist2[8],
                          int *n_out) {
    if (!ctx || !points1 || !points2 || !dist1 || !dist2 || !n_out || n < 0) return MLPL_E_BAD_INPUT;
    *n_out = n;
    double s1 = 0, s2 = 0;
    for (int i = 0; i < 8; ++i) s1 += dist1[i], s2 += dist2[i];
    if ((s1 < 1e-3 && s1 > -1e-3) && (s2 < 1e-3 && s2 > -1e-3)) return MLPL_OK;  // pose_helper.cpp:1176-1178
    if (n == 0) return MLPL_E_FAILED;
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    const int ngroups = (n + 63) / 64;
    void *d1, *d2, *o1, *o2, *dv, *dg;
    int rc;
    if ((rc = ws_get(ctx, WS_AUX0, (size_t)n * 8, &d1))) return rc;
    if ((rc = ws_get(ctx, WS_AUX1, (size_t)n * 8, &d2))) return rc;
    if ((rc = ws_get(ctx, WS_AUX3, (size_t)n * 8, &o1))) return rc;
    if ((rc = ws_get(ctx, WS_AUX4, (size_t)n * 8, &o2))) return rc;
    if ((rc = ws_get(ctx, WS_AUX5, (size_t)n, &dv))) return rc;
    if ((rc = ws_get(ctx, WS_AUX6, (size_t)(ngroups + 4) * 4 + 16, &dg))) return rc;
    int32_t *dcnt = (int32_t *)dg + ngroups + 2;
    Dist8 D1, D2;
    for (int i = 0; i < 8; ++i) D1.k[i] = dist1[i], D2.k[i] = dist2[i];
    MLPL_HIP_TRY(hipMemcpyAsync(d1, points1, (size_t)n * 8, hipMemcpyHostToDevice, s));
    MLPL_HIP_TRY(hipMemcpyAsync(d2, points2, (size_t)n * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(undistort_kernel, dim3(ngroups), dim3(64), 0, s, (float *)d1, (float *)d2, n, D1, D2, (uint8_t *)dv,
                       (int32_t *)dg);
    hipLaunchKernelGGL(compact_pairs_kernel, dim3((ngroups + 3) / 4), dim3(256), 0, s, (const float *)d1, (const float *)d2,
                       (const uint8_t *)dv, (const int32_t *)dg, n, (float *)o1, (float *)o2, dcnt);
    MLPL_HIP_TRY(hipGetLastError());
    int32_t n1 = 0;
    MLPL_HIP_TRY(hipMemcpyAsync(&n1, dcnt, 4, hipMemcpyDeviceToHost, s));
    MLPL_HIP_TRY(hipStreamSynchronize(s));
    if (n1 < 16) {  // pose_helper.cpp:1200-1201: return false; the (uncompacted) undistorted values stay in the inputs
        MLPL_HIP_TRY(hipMemcpy(points1, d1, (size_t)n * 8, hipMemcpyDeviceToHost));
        MLPL_HIP_TRY(hipMemcpy(points2, d2, (size_t)n * 8, hipMemcpyDeviceToHost));
        return MLPL_E_FAILED;
    }
    MLPL_HIP_TRY(hipMemcpy(points1, o1, (size_t)n1 * 8, hipMemcpyDeviceToHost));
    MLPL_HIP_TRY(hipMemcpy(points2, o2, (size_t)n1 * 8, hipMemcpyDeviceToHost));
    *n_out = n1;
    return MLPL_OK;
}

int mlpl_get_inliers_strict(mlpl_ctx *ctx, const double *p1, const double *p2, int n, const double E[9], double th2, double *err,
                            uint8_t *mask) {
    if (!ctx || !p1 || !p2 || !E || !err || !mask || n < 0) return MLPL_E_BAD_INPUT;
    if (n == 0) return 0;
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    void *d1, *d2, *de, *dm, *dsmall;
    int rc;
    if ((rc = ws_get(ctx, WS_AUX0, (size_t)n * 16, &d1))) return rc;
    if ((rc = ws_get(ctx, WS_AUX1, (size_t)n * 16, &d2))) return rc;
    if ((rc = ws_get(ctx, WS_AUX5, (size_t)n * 8, &de))) return rc;
    if ((rc = ws_get(ctx, WS_AUX6, (size_t)n, &dm))) return rc;
    if ((rc = ws_get(ctx, WS_AUX2, 4096, &dsmall))) return rc;
    double *dE = (double *)dsmall;
    int32_t *dcnt = (int32_t *)(dE + 16);
    MLPL_HIP_TRY(hipMemcpyAsync(d1, p1, (size_t)n * 16, hipMemcpyHostToDevice, s));
    MLPL_HIP_TRY(hipMemcpyAsync(d2, p2, (size_t)n * 16, hipMemcpyHostToDevice, s));
    MLPL_HIP_TRY(hipMemcpyAsync(dE, E, 72, hipMemcpyHostToDevice, s));
    MLPL_HIP_TRY(hipMemsetAsync(dcnt, 0, 4, s));
    hipLaunchKernelGGL(inliers_strict_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const double *)d1, (const double *)d2, n,
                       (const double *)dE, th2, (double *)de, (uint8_t *)dm, dcnt);
    MLPL_HIP_TRY(hipGetLastError());
    int32_t cnt = 0;
    MLPL_HIP_TRY(hipMemcpyAsync(err, de, (size_t)n * 8, hipMemcpyDeviceToHost, s));
    MLPL_HIP_TRY(hipMemcpyAsync(mask, dm, (size_t)n, hipMemcpyDeviceToHost, s));
    MLPL_HIP_TRY(hipMemcpyAsync(&cnt, dcnt, 4, hipMemcpyDeviceToHost, s));
    MLPL_HIP_TRY(hipStreamSynchronize(s));
    return cnt;
}

}  // extern "C"
