// knn_l2_mfma.hip -- squared-L2 2-NN as an fp16 MFMA distance-GEMM with a fused top-2 epilogue (gfx950).
//
// Replaces cvflann::Index<L2<float>>(LinearIndexParams).knnSearch (reference matchinglib/source/matchers.cpp:634-664)
// for float descriptors whose elements are all integers in [0,255] (OpenCV SIFT layout).  For such data
//     d2(q,t) = |q|^2 + (|t|^2 - 2 q.t)
// is exact in this formulation: the train operand is stored as -2 t (exact in fp16: |.| <= 510), the accumulator starts at |t|^2,
// and every value it can pass through -- |t|^2 plus any subset of the products -2 q_k t_k -- is an integer in
// [|t|^2 - 2 q.t, |t|^2] = [d2 - |q|^2, |t|^2], i.e. of magnitude < dim*65025 < 2^24 (dim <= 256), hence exact in fp32 whatever order
// the matrix core adds in; the final d2 = acc + |q|^2 is an integer < 2^24, so that add is exact too (|q|^2 + |t|^2 itself can
// exceed 2^24 for dim > 128 and is never formed).  The reference's own fp32 running sum of integer squares is exact for the same
// reason.  So distances, hence the lexicographic (d2, trainIdx) order, are bit-identical to the CPU path.  Anything else
// (fractional / negative / large values, dim > 256) takes knn_l2_exact_kernel.
//
// Mapping: v_mfma_f32_32x32x16_f16 with A = 32 train rows, B = 32 queries, so a lane's 16 accumulators are 16 train rows of
// ONE query (column = lane & 31): the running top-2 stays per lane, on packed 32-bit keys  d2 << ib | row_in_split
// (d2 is an integer < 2^(32-ib)), and only lanes l / l+32 have to be combined at the end.  Operands are pre-swizzled once
// per call into MFMA fragment order ([tile][kstep][lane] x 16 B), so the train tile is staged into LDS by a straight
// coalesced copy and read back conflict-free with one ds_read_b128 per MFMA; the query fragments live in VGPRs.

#include <algorithm>

#include "mlpl_internal.h"

namespace mlpl {

void launch_knn_l2_merge(const void *part, int nq, int nsplit, int k, int batch, int32_t *d_idx, float *d_dist, hipStream_t s,
                         const int *gate, int gate_want);

namespace {

typedef _Float16 half8 __attribute__((ext_vector_type(8)));
typedef float float16v __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

constexpr int kMaxKS = 16;  // dim <= 256

__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// X: [n][dim] f32 (row stride `stride`) -> frag[(tile*KS + s)*64 + lane] = 8 halfs X[tile*32 + (lane&31)][16 s + 8 (lane>>5) + j]
// (train rows scaled by -2, see above); flags[0] |= 1 when an element is not an integer in [0,255].  One launch for both operands:
// blockIdx.z = 0 queries, 1 train rows.
struct L2PrepArgs {
    const float *X;
    size_t stride, bstride;
    int n, ntiles;
    uint4 *frag;
};

__global__ void l2_prep_kernel(L2PrepArgs qa, L2PrepArgs ta, int dim, int KS, int *__restrict__ flags) {
    const L2PrepArgs A = blockIdx.z ? ta : qa;
    const int b = blockIdx.y;
    const long long total = (long long)A.ntiles * KS * 64;
    const long long g = (long long)blockIdx.x * blockDim.x + threadIdx.x;
    if (g >= total) return;
    const int lane = (int)(g & 63);
    const int s = (int)((g >> 6) % KS);
    const int tile = (int)((g >> 6) / KS);
    const int row = tile * 32 + (lane & 31);
    const int k0 = 16 * s + 8 * (lane >> 5);
    half8 h;
    bool bad = false;
#pragma unroll
    for (int j = 0; j < 8; ++j) {
        float v = 0.f;
        if (row < A.n && k0 + j < dim) v = A.X[(size_t)b * A.bstride + (size_t)row * A.stride + k0 + j];
        bad = bad || !(v >= 0.f && v <= 255.f && v == floorf(v));
        h[j] = (_Float16)(blockIdx.z ? -2.0f * v : v);
    }
    if (bad) atomicOr(flags, 1);
    A.frag[(size_t)b * total + g] = *reinterpret_cast<uint4 *>(&h);
}

// Squared norms of the rows of both sets in one launch: 8 lanes per row (coalesced 32-byte segments), xor-shuffle sum.
// Any summation order is exact here (integer data, sums < 2^24); when the data do not qualify the result is unused.
__global__ void l2_norm_kernel(const float *__restrict__ Xq, size_t q_stride, size_t q_bstride, int nq, int nq_pad,
                               const float *__restrict__ Xt, size_t t_stride, size_t t_bstride, int nt, int nt_pad, int dim,
                               float *__restrict__ qnorm, float *__restrict__ tnorm) {
    const int b = blockIdx.y;
    const int g = blockIdx.x * blockDim.x + threadIdx.x;
    const int row_all = g >> 3, sub = g & 7;
    const bool is_t = row_all >= nq_pad;
    const int i = is_t ? row_all - nq_pad : row_all;
    const int n = is_t ? nt : nq, npad = is_t ? nt_pad : nq_pad;
    float s = 0.f;
    if (i < n) {
        const float *r = is_t ? Xt + (size_t)b * t_bstride + (size_t)i * t_stride : Xq + (size_t)b * q_bstride + (size_t)i * q_stride;
        for (int c = sub; c < dim; c += 8) s += r[c] * r[c];
    }
    s += __shfl_xor(s, 1);
    s += __shfl_xor(s, 2);
    s += __shfl_xor(s, 4);
    if (sub == 0 && i < npad) (is_t ? tnorm : qnorm)[(size_t)b * npad + i] = s;
}

template <int KS>
__global__ __launch_bounds__(256) void knn_l2_mfma_kernel(const uint4 *__restrict__ qfrag, const float *__restrict__ qnorm,
                                                          const uint4 *__restrict__ tfrag, const float *__restrict__ tnorm,
                                                          int nq, int nt, int nq_tiles, int nt_tiles, int nq_pad, int nt_pad,
                                                          int tiles_per_split, int nsplit, int ib,
                                                          ulonglong2 *__restrict__ part, const int *__restrict__ gate) {
    if (*gate != 0) return;  // some element is not an integer in [0,255]: the exact kernel (gated the other way) runs instead
    __shared__ __attribute__((aligned(16))) uint4 tileA[2][KS * 64];
    __shared__ float tileN[2][32];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int b = blockIdx.z, split = blockIdx.y;
    const int qtile = blockIdx.x * 4 + wave;
    const bool wave_active = qtile < nq_tiles;
    qfrag += (size_t)b * nq_tiles * KS * 64;
    tfrag += (size_t)b * nt_tiles * KS * 64;
    qnorm += (size_t)b * nq_pad;
    tnorm += (size_t)b * nt_pad;

    half8 qf[KS];
    float qn = 0.f;
    if (wave_active) {
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const uint4 v = qfrag[((size_t)qtile * KS + s) * 64 + lane];
            qf[s] = *reinterpret_cast<const half8 *>(&v);
        }
        qn = qnorm[qtile * 32 + (lane & 31)];
    } else {
#pragma unroll
        for (int s = 0; s < KS; ++s)
#pragma unroll
            for (int j = 0; j < 8; ++j) qf[s][j] = (_Float16)0;
    }

    const int t_begin = split * tiles_per_split;
    const int t_end = min(nt_tiles, t_begin + tiles_per_split);
    uint32_t k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;

    auto stage = [&](int tile, int buf) {
        const uint4 *src = tfrag + (size_t)tile * KS * 64;
        for (int i = tid; i < KS * 64; i += 256) tileA[buf][i] = src[i];
        if (tid < 32) tileN[buf][tid] = tnorm[tile * 32 + tid];
    };
    if (t_begin < t_end) stage(t_begin, 0);
    __syncthreads();
    for (int t = t_begin; t < t_end; ++t) {
        const int buf = (t - t_begin) & 1;
        if (t + 1 < t_end) stage(t + 1, buf ^ 1);  // the other buffer was released by the barrier ending iteration t-1
        // accumulator reg r of this lane is train row (r&3) + 8 (r>>2) + 4 (lane>>5) of the tile, query lane&31; it starts at |t|^2
        float16v acc;
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[r] = tileN[buf][(r & 3) + 8 * (r >> 2) + 4 * (lane >> 5)];
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const uint4 av = tileA[buf][s * 64 + lane];
            acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(*reinterpret_cast<const half8 *>(&av), qf[s], acc, 0, 0, 0);
        }
        const int lrow0 = (t - t_begin) * 32 + 4 * (lane >> 5);
        const bool partial = (t * 32 + 32 > nt);  // wave-uniform: only the last tile of the set
#pragma unroll
        for (int r = 0; r < 16; ++r) {
            const int m = (r & 3) + 8 * (r >> 2);
            const float d = acc[r] + qn;  // = |t|^2 - 2 q.t + |q|^2, exact: see the header
            uint32_t key = ((uint32_t)d << ib) | (uint32_t)(lrow0 + m);
            if (partial && (t * 32 + m + 4 * (lane >> 5) >= nt)) key = 0xFFFFFFFFu;
            k1 = umed3(k0, k1, key);
            k0 = min(k0, key);
        }
        __syncthreads();
    }
    // global 64-bit keys (float bits of d2 << 32 | train row), combine the two lanes that share a query column
    const uint32_t lmask = (1u << ib) - 1u;
    auto to_global = [&](uint32_t key) -> u64 {
        if (key == 0xFFFFFFFFu) return ~0ull;
        const float d = (float)(key >> ib);
        return ((u64)__float_as_uint(d) << 32) | (u64)((uint32_t)t_begin * 32u + (key & lmask));
    };
    u64 g0 = to_global(k0), g1 = to_global(k1);
    const u64 o0 = __shfl_xor(g0, 32), o1 = __shfl_xor(g1, 32);
    auto upd = [&](u64 g) {
        const bool lt0 = g < g0, lt1 = g < g1;
        g1 = lt0 ? g0 : (lt1 ? g : g1);
        g0 = lt0 ? g : g0;
    };
    upd(o0);
    upd(o1);
    const int qi = qtile * 32 + (lane & 31);
    if (wave_active && lane < 32 && qi < nq) part[((size_t)b * nsplit + split) * nq + qi] = make_ulonglong2(g0, g1);
}

}  // namespace

// Returns 1 when the MFMA path cannot apply at all (caller runs the exact kernel ungated), 0 when the MFMA pipeline was
// enqueued (in auto mode *gate_out then points at the device flag: 0 = the data qualified and the results are final,
// nonzero = the MFMA kernels exited early and the caller's exact kernels, gated on nonzero, produce the results), < 0 on error.
// Nothing here synchronises in auto mode, so the *_dev entry points stay asynchronous.
int launch_knn_l2_mfma(mlpl_ctx *ctx, const float *d_q, int nq, size_t q_stride, size_t q_bstride, const float *d_t, int nt,
                       size_t t_stride, size_t t_bstride, int dim, int k, int batch, int32_t *d_idx, float *d_dist,
                       hipStream_t s, int force, const int **gate_out) {
    const int KS = (dim + 15) / 16;
    int ksel = 0;
    for (int c : {1, 2, 4, 8, 16})
        if (KS <= c) {
            ksel = c;
            break;
        }
    if (!ksel) {
        if (force) {
            set_error("knn_l2 (MFMA): dim %d > %d is not supported on this path", dim, kMaxKS * 16);
            return MLPL_E_BAD_INPUT;
        }
        return 1;
    }
    // bits for d2 <= dim_pad * 255^2 and for the local row index
    int dbits = 1;
    while ((1ull << dbits) <= (unsigned long long)ksel * 16 * 65025ull) ++dbits;
    const int ib = 32 - dbits;
    if (ib < 5) return force ? MLPL_E_BAD_INPUT : 1;

    const int nq_tiles = (nq + 31) / 32, nt_tiles = (nt + 31) / 32;
    const int nq_pad = nq_tiles * 32, nt_pad = nt_tiles * 32;
    void *qf, *tf, *nrm, *flag;
    int rc;
    if ((rc = ws_get(ctx, WS_PACK_Q, (size_t)batch * nq_tiles * ksel * 64 * 16, &qf))) return rc;
    if ((rc = ws_get(ctx, WS_PACK_T, (size_t)batch * nt_tiles * ksel * 64 * 16, &tf))) return rc;
    if ((rc = ws_get(ctx, WS_AUX3, (size_t)batch * (nq_pad + nt_pad) * 4 + 64, &nrm))) return rc;
    if ((rc = ws_get(ctx, WS_L2_FLAG, 4096, &flag))) return rc;
    float *qn = (float *)nrm, *tn = qn + (size_t)batch * nq_pad;
    int *dflag = (int *)flag;
    MLPL_HIP_TRY(hipMemsetAsync(dflag, 0, 4, s));
    {
        const long long tq = (long long)nq_tiles * ksel * 64, tt = (long long)nt_tiles * ksel * 64;
        const L2PrepArgs qa{d_q, q_stride, q_bstride, nq, nq_tiles, (uint4 *)qf}, ta{d_t, t_stride, t_bstride, nt, nt_tiles, (uint4 *)tf};
        hipLaunchKernelGGL(l2_prep_kernel, dim3((unsigned)((std::max(tq, tt) + 255) / 256), batch, 2), dim3(256), 0, s, qa, ta, dim,
                           ksel, dflag);
        const long long rows8 = (long long)(nq_pad + nt_pad) * 8;
        hipLaunchKernelGGL(l2_norm_kernel, dim3((unsigned)((rows8 + 255) / 256), batch), dim3(256), 0, s, d_q, q_stride, q_bstride,
                           nq, nq_pad, d_t, t_stride, t_bstride, nt, nt_pad, dim, qn, tn);
    }
    if (force) {  // forcing is a test/diagnostic mode: report non-qualifying data as an error (one host hop)
        int hflag = 0;
        MLPL_HIP_TRY(hipMemcpyAsync(&hflag, dflag, 4, hipMemcpyDeviceToHost, s));
        MLPL_HIP_TRY(hipStreamSynchronize(s));
        if (hflag) {
            set_error("knn_l2 (MFMA): descriptors are not integer-valued in [0,255]");
            return MLPL_E_BAD_INPUT;
        }
    }
    if (gate_out) *gate_out = dflag;

    // train tiles per split: one split must fit the local-row field and the grid should hold >= 2 waves per SIMD
    const int qblocks = (nq_tiles + 3) / 4;
    const int max_tps = std::max(1, (int)(((1u << ib) - 2u) / 32u));
    long long want_splits = (2LL * ctx->num_cus + (long long)qblocks * batch - 1) / ((long long)qblocks * batch);
    int tps = (int)std::max<long long>(1, nt_tiles / std::max<long long>(1, want_splits));
    tps = std::min(tps, max_tps);
    int nsplit = (nt_tiles + tps - 1) / tps;
    if (nsplit > 65535) {
        tps = std::min(max_tps, (nt_tiles + 65534) / 65535);
        nsplit = (nt_tiles + tps - 1) / tps;
        if (nsplit > 65535) return force ? MLPL_E_BAD_INPUT : 1;
    }
    void *part = nullptr;
    if ((rc = ws_get(ctx, WS_AUX4, (size_t)batch * nsplit * nq * sizeof(ulonglong2), &part))) return rc;

    dim3 grid(qblocks, nsplit, batch);
    prof_mark(ctx, MLPL_PROF_KNN_L2, 0, s);
#define MLPL_MFMA_LAUNCH(K)                                                                                                  \
    hipLaunchKernelGGL(knn_l2_mfma_kernel<K>, grid, dim3(256), 0, s, (const uint4 *)qf, (const float *)qn, (const uint4 *)tf, \
                       (const float *)tn, nq, nt, nq_tiles, nt_tiles, nq_pad, nt_pad, tps, nsplit, ib, (ulonglong2 *)part, \
                       (const int *)dflag)
    switch (ksel) {
        case 1: MLPL_MFMA_LAUNCH(1); break;
        case 2: MLPL_MFMA_LAUNCH(2); break;
        case 4: MLPL_MFMA_LAUNCH(4); break;
        case 8: MLPL_MFMA_LAUNCH(8); break;
        default: MLPL_MFMA_LAUNCH(16); break;
    }
#undef MLPL_MFMA_LAUNCH
    prof_mark(ctx, MLPL_PROF_KNN_L2, 1, s);
    launch_knn_l2_merge(part, nq, nsplit, k, batch, d_idx, d_dist, s, dflag, 0);
    MLPL_HIP_TRY(hipGetLastError());
    return MLPL_OK;
}

}  // namespace mlpl
