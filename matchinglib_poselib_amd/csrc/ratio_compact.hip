// ratio_compact.hip -- Lowe ratio test + ordered DMatch emission on the device.
//
// Replaces the loops at reference matchinglib/source/matchers.cpp:601-625 (int Hamming distances) and
// :677-701 (float squared-L2 distances):
//     if (dists[q][0] < (0.75f * dists[q][1])) push_back(DMatch{distance=(float)d0, queryIdx=q, trainIdx=idx0})
// Matches must come out in ascending query order (push_back order), so compaction is a stable block scan:
// one 1024-thread block per batch item walks the queries in chunks of 1024, ballot + popcount inside each
// wave, a 16-entry LDS prefix across waves and a running base across chunks.

#include "mlpl_internal.h"

namespace mlpl {

namespace {

constexpr int kThreads = 1024;

template <bool kFloat>
__global__ __launch_bounds__(kThreads) void ratio_compact_kernel(const int32_t *__restrict__ idx,
                                                                  const void *__restrict__ dist_v, int nq, int k,
                                                                  float ratio, mlpl_dmatch *__restrict__ out,
                                                                  int32_t *__restrict__ n_out) {
    __shared__ int wave_tot[kThreads / 64];
    const int b = blockIdx.x;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    idx += (size_t)b * nq * k;
    out += (size_t)b * nq;
    const int32_t *di = reinterpret_cast<const int32_t *>(dist_v) + (size_t)b * nq * k;
    const float *df = reinterpret_cast<const float *>(dist_v) + (size_t)b * nq * k;

    int running = 0;
    for (int base = 0; base < nq; base += kThreads) {
        const int qi = base + tid;
        bool pass = false;
        float d0 = 0.f;
        int i0 = 0;
        if (qi < nq) {
            float d1 = 0.f;
            if constexpr (kFloat) {
                d0 = df[(size_t)qi * k];
                if (k == 2) d1 = df[(size_t)qi * k + 1];
            } else {
                d0 = (float)di[(size_t)qi * k];
                if (k == 2) d1 = (float)di[(size_t)qi * k + 1];
            }
            i0 = idx[(size_t)qi * k];
            pass = (k == 2) ? (d0 < __fmul_rn(ratio, d1)) : true;
        }
        const unsigned long long bal = __ballot(pass);
        const int lane_prefix = __popcll(bal & ((1ull << lane) - 1ull));
        if (lane == 0) wave_tot[wave] = __popcll(bal);
        __syncthreads();
        int wave_prefix = 0, chunk_total = 0;
#pragma unroll
        for (int w = 0; w < kThreads / 64; ++w) {
            const int c = wave_tot[w];
            if (w < wave) wave_prefix += c;
            chunk_total += c;
        }
        if (pass) {
            mlpl_dmatch m;
            m.queryIdx = qi;
            m.trainIdx = i0;
            m.imgIdx = -1;
            m.distance = d0;
            out[running + wave_prefix + lane_prefix] = m;
        }
        running += chunk_total;
        __syncthreads();
    }
    if (tid == 0) n_out[b] = running;
}

}  // namespace

int launch_ratio_compact(mlpl_ctx *ctx, const int32_t *d_idx, const void *d_dist, int dist_is_float, int nq, int k,
                         int batch, float ratio, mlpl_dmatch *d_out, int32_t *d_n_out, hipStream_t s) {
    (void)ctx;
    if (!d_idx || !d_dist || !d_out || !d_n_out || nq < 0 || batch < 1 || (k != 1 && k != 2)) {
        set_error("ratio_compact: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    if (dist_is_float)
        hipLaunchKernelGGL(ratio_compact_kernel<true>, dim3(batch), dim3(kThreads), 0, s, d_idx, d_dist, nq, k, ratio,
                           d_out, d_n_out);
    else
        hipLaunchKernelGGL(ratio_compact_kernel<false>, dim3(batch), dim3(kThreads), 0, s, d_idx, d_dist, nq, k, ratio,
                           d_out, d_n_out);
    MLPL_HIP_TRY(hipGetLastError());
    return MLPL_OK;
}

}  // namespace mlpl
