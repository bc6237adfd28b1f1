// ratio_compact.hip -- Lowe ratio test + ordered DMatch emission on the device.
//
// Replaces the loops at reference matchinglib/source/matchers.cpp:601-625 (int Hamming distances) and
// :677-701 (float squared-L2 distances):
//     if (dists[q][0] < (0.75f * dists[q][1])) push_back(DMatch{distance=(float)d0, queryIdx=q, trainIdx=idx0})
// Matches must come out in ascending query order (push_back order), so compaction is a stable two-level scan over
// groups of 256 queries: (1) per-group pass counts (ratio_count_kernel, or for Hamming the merge kernel's by-product),
// (2) ratio_write_kernel: each group sums the counts of the groups before it, ballot/popcount-scans its own 256
// predicates and writes its DMatch rows at the right offset; the last group stores the total.

#include "mlpl_internal.h"

namespace mlpl {

namespace {

// nms_emit (BRUTEFORCENMS, nmslib_matchers.h:360-414): with k == 2 and NO ratio test every query is emitted and, when the two
// distances tie, NMSLIB's sorted list starts with the LARGER id (heap pop order kept by std::sort); with the ratio test the
// predicate is the same d0 < ratio*d1.  nms_emit: 0 = LINEAR semantics, 1 = NMS with ratio test, 2 = NMS without.
template <bool kFloat>
__device__ __forceinline__ bool ratio_pred(const void *dist_v, size_t qi, int k, float ratio, float &d0, int nms_emit = 0,
                                           bool *tie = nullptr) {
    float d1 = 0.f;
    if constexpr (kFloat) {
        const float *df = reinterpret_cast<const float *>(dist_v);
        d0 = df[qi * k];
        if (k == 2) d1 = df[qi * k + 1];
    } else {
        const int32_t *di = reinterpret_cast<const int32_t *>(dist_v);
        d0 = (float)di[qi * k];
        if (k == 2) d1 = (float)di[qi * k + 1];
    }
    if (tie) *tie = (k == 2) && (d0 == d1);
    if (nms_emit == 2) return true;
    return (k == 2) ? (d0 < __fmul_rn(ratio, d1)) : true;
}

template <bool kFloat>
__global__ __launch_bounds__(kCountGroup) void ratio_count_kernel(const void *__restrict__ dist_v, int nq, int k, float ratio,
                                                                  int32_t *__restrict__ group_counts, int nms_emit) {
    static_assert(kCountGroup == 64, "one wave per count group");
    const int b = blockIdx.y, tid = threadIdx.x;
    const int qi = blockIdx.x * kCountGroup + tid;
    bool pass = false;
    float d0;
    if (qi < nq) pass = ratio_pred<kFloat>(dist_v, (size_t)b * nq + qi, k, ratio, d0, nms_emit);
    const unsigned long long bal = __ballot(pass);
    if (tid == 0) group_counts[(size_t)b * gridDim.x + blockIdx.x] = __popcll(bal);
}

template <bool kFloat>
__global__ __launch_bounds__(kRatioGroup) void ratio_write_kernel(const int32_t *__restrict__ idx, const void *__restrict__ dist_v,
                                                                  const int32_t *__restrict__ group_counts, int nq, int k,
                                                                  float ratio, mlpl_dmatch *__restrict__ out,
                                                                  int32_t *__restrict__ n_out, int nms_emit) {
    __shared__ int red[kRatioGroup / 64];
    __shared__ int wave_tot[kRatioGroup / 64];
    const int b = blockIdx.y, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int grp = blockIdx.x, ngrp = gridDim.x;
    const int ncnt = (nq + kCountGroup - 1) / kCountGroup;
    const int32_t *gc = group_counts + (size_t)b * ncnt;
    // exclusive prefix: pass counts of all count-groups before this block's first query
    int part = 0;
    for (int j = tid; j < grp * (kRatioGroup / kCountGroup); j += kRatioGroup) part += gc[j];
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) part += __shfl_xor(part, off);
    if (lane == 0) red[wave] = part;

    const int qi = grp * kRatioGroup + tid;
    bool pass = false, tie = false;
    float d0 = 0.f;
    if (qi < nq) pass = ratio_pred<kFloat>(dist_v, (size_t)b * nq + qi, k, ratio, d0, nms_emit, &tie);
    const unsigned long long bal = __ballot(pass);
    const int lane_prefix = __popcll(bal & ((1ull << lane) - 1ull));
    if (lane == 0) wave_tot[wave] = __popcll(bal);
    __syncthreads();
    int base = 0, wave_prefix = 0, total = 0;
#pragma unroll
    for (int w = 0; w < kRatioGroup / 64; ++w) {
        base += red[w];
        const int c = wave_tot[w];
        if (w < wave) wave_prefix += c;
        total += c;
    }
    if (pass) {
        mlpl_dmatch m;
        m.queryIdx = qi;
        m.trainIdx = idx[((size_t)b * nq + qi) * k + ((nms_emit == 2 && tie) ? 1 : 0)];
        m.imgIdx = -1;
        m.distance = d0;
        out[(size_t)b * nq + base + wave_prefix + lane_prefix] = m;
    }
    if (grp == ngrp - 1 && tid == 0) n_out[b] = base + total;
}

// Correspondence gather with the reference's ImgToCamCoordTrans fused in (poselib/source/pose_helper.cpp:1100-1109:
// float result of a double operation), as StereoRefine::addNewCorrespondences does (stereo_pose_refinement.cpp:428-455).
__global__ void gather_match_points_kernel(const mlpl_dmatch *__restrict__ matches, int n, const float *__restrict__ kp1,
                                           const float *__restrict__ kp2, double fx0, double fy0, double cx0, double cy0,
                                           double fx1, double fy1, double cx1, double cy1, double *__restrict__ p1,
                                           double *__restrict__ p2) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    const mlpl_dmatch m = matches[i];
    const float ax = kp1[2 * m.queryIdx], ay = kp1[2 * m.queryIdx + 1];
    const float bx = kp2[2 * m.trainIdx], by = kp2[2 * m.trainIdx + 1];
    p1[2 * i] = (double)(float)(((double)ax - cx0) / fx0);
    p1[2 * i + 1] = (double)(float)(((double)ay - cy0) / fy0);
    p2[2 * i] = (double)(float)(((double)bx - cx1) / fx1);
    p2[2 * i + 1] = (double)(float)(((double)by - cy1) / fy1);
}

}  // namespace

int launch_gather_match_points(const mlpl_dmatch *d_matches, int n, const float *d_kp1, const float *d_kp2, const double K0[4],
                               const double K1[4], double *d_p1, double *d_p2, hipStream_t s) {
    if (n <= 0) return MLPL_OK;
    hipLaunchKernelGGL(gather_match_points_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_matches, n, d_kp1, d_kp2, K0[0],
                       K0[1], K0[2], K0[3], K1[0], K1[1], K1[2], K1[3], d_p1, d_p2);
    MLPL_HIP_TRY(hipGetLastError());
    return MLPL_OK;
}

// batch of pairs: blockIdx.y = pair; matches / coordinates of pair b at b * pair_stride, its keypoints at b * kp*_stride floats
__global__ void gather_match_points_batch_kernel(const mlpl_dmatch *__restrict__ matches, const int32_t *__restrict__ counts, int pair_stride,
                                                 const float *__restrict__ kp1, size_t kp1_stride, const float *__restrict__ kp2,
                                                 size_t kp2_stride, double fx0, double fy0, double cx0, double cy0, double fx1, double fy1,
                                                 double cx1, double cy1, double *__restrict__ p1, double *__restrict__ p2) {
    const int b = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= counts[b]) return;
    const mlpl_dmatch m = matches[(size_t)b * pair_stride + i];
    const float *k1 = kp1 + (size_t)b * kp1_stride, *k2 = kp2 + (size_t)b * kp2_stride;
    const float ax = k1[2 * m.queryIdx], ay = k1[2 * m.queryIdx + 1];
    const float bx = k2[2 * m.trainIdx], by = k2[2 * m.trainIdx + 1];
    const size_t o = ((size_t)b * pair_stride + i) * 2;
    p1[o] = (double)(float)(((double)ax - cx0) / fx0);
    p1[o + 1] = (double)(float)(((double)ay - cy0) / fy0);
    p2[o] = (double)(float)(((double)bx - cx1) / fx1);
    p2[o + 1] = (double)(float)(((double)by - cy1) / fy1);
}

int launch_gather_match_points_batch(const mlpl_dmatch *d_matches, const int32_t *d_counts, int B, int pair_stride, const float *d_kp1,
                                     size_t kp1_stride, const float *d_kp2, size_t kp2_stride, const double K0[4], const double K1[4],
                                     double *d_p1, double *d_p2, hipStream_t s) {
    hipLaunchKernelGGL(gather_match_points_batch_kernel, dim3((pair_stride + 255) / 256, B), dim3(256), 0, s, d_matches, d_counts, pair_stride,
                       d_kp1, kp1_stride, d_kp2, kp2_stride, K0[0], K0[1], K0[2], K0[3], K1[0], K1[1], K1[2], K1[3], d_p1, d_p2);
    MLPL_HIP_TRY(hipGetLastError());
    return MLPL_OK;
}

int launch_ratio_compact(mlpl_ctx *ctx, const int32_t *d_idx, const void *d_dist, int dist_is_float, int nq, int k,
                         int batch, float ratio, mlpl_dmatch *d_out, int32_t *d_n_out, hipStream_t s,
                         int32_t *d_group_counts_ready, int nms_emit) {
    if (!d_idx || !d_dist || !d_out || !d_n_out || nq < 0 || batch < 1 || batch > 65535 || (k != 1 && k != 2)) {
        set_error("ratio_compact: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    if (nq == 0) {
        MLPL_HIP_TRY(hipMemsetAsync(d_n_out, 0, sizeof(int32_t) * batch, s));
        return MLPL_OK;
    }
    const int ngrp = (nq + kRatioGroup - 1) / kRatioGroup;
    const int ncnt = (nq + kCountGroup - 1) / kCountGroup;
    dim3 grid(ngrp, batch);
    int32_t *gc = d_group_counts_ready;
    if (!gc) {
        void *buf = nullptr;
        int rc = ws_get(ctx, WS_COUNT, (size_t)batch * ncnt * sizeof(int32_t), &buf);
        if (rc) return rc;
        gc = (int32_t *)buf;
        dim3 cgrid(ncnt, batch);
        if (dist_is_float)
            hipLaunchKernelGGL(ratio_count_kernel<true>, cgrid, dim3(kCountGroup), 0, s, d_dist, nq, k, ratio, gc, nms_emit);
        else
            hipLaunchKernelGGL(ratio_count_kernel<false>, cgrid, dim3(kCountGroup), 0, s, d_dist, nq, k, ratio, gc, nms_emit);
    }
    if (dist_is_float)
        hipLaunchKernelGGL(ratio_write_kernel<true>, grid, dim3(kRatioGroup), 0, s, d_idx, d_dist, gc, nq, k, ratio, d_out,
                           d_n_out, nms_emit);
    else
        hipLaunchKernelGGL(ratio_write_kernel<false>, grid, dim3(kRatioGroup), 0, s, d_idx, d_dist, gc, nq, k, ratio, d_out,
                           d_n_out, nms_emit);
    MLPL_HIP_TRY(hipGetLastError());
    return MLPL_OK;
}

}  // namespace mlpl
