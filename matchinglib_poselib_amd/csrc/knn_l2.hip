// knn_l2.hip -- exact brute-force 2-NN under squared L2 distance on float descriptors (gfx950).
//
// Replaces cvflann::Index<L2<float>>(LinearIndexParams).knnSearch as called by
// matchinglib::getMatches(...,"LINEAR",...) -- reference matchinglib/source/matchers.cpp:634-664.
//
// Two device paths, both bit-exact against the CPU path:
//   (1) knn_l2_exact_kernel  -- fp32 VALU, reproduces cvflann::L2<float>'s summation order
//       (per 4 elements: result += ((d0*d0 + d1*d1) + d2*d2) + d3*d3, then a scalar tail) with explicit
//       round-to-nearest mul/add (no FMA contraction: the reference is built -msse4.2, no FMA).
//   (2) knn_l2_mfma_*        -- fp16 MFMA distance-GEMM  |a|^2 + |b|^2 - 2 a.b  with fused top-2, used only
//       when every element is an integer in [0,255] (OpenCV SIFT layout): then every product, every partial
//       sum and d^2 are integers < 2^24, exact in the fp32 accumulator AND in the reference's fp32 sum, so
//       the distances are identical.  See knn_l2_mfma.hip.
// Keys are 64-bit (float_bits(d2) << 32 | trainIdx): non-negative floats order like unsigned ints, so the
// running top-2 is a lexicographic (distance, trainIdx) min exactly like cvflann's KNNUniqueResultSet.

#include <algorithm>
#include <cstring>

#include "knn_l2_common.h"
#include "knn_l2_mfma_body.h"
#include "mlpl_internal.h"

namespace mlpl {

int launch_knn_l2_mfma(mlpl_ctx *ctx, const float *d_q, int nq, size_t q_stride, size_t q_bstride, const float *d_t, int nt,
                       size_t t_stride, size_t t_bstride, int dim, int batch, hipStream_t s, int force, l2mfma::L2MfmaPlan *plan);

int launch_knn_l2_f16(mlpl_ctx *ctx, const float *d_q, int nq, size_t q_stride, size_t q_bstride, const float *d_t, int nt, size_t t_stride,
                      size_t t_bstride, int dim, int k, int batch, int gen, int *d_flags, int *d_hint, int32_t *d_idx, float *d_dist, hipStream_t s);

namespace {

constexpr int kQPB = 256;   // queries per block, one per lane
constexpr int kMaxTileRows = 32;

typedef unsigned long long u64;

__device__ __forceinline__ void top2_update(u64 &k0, u64 &k1, u64 key) { l2_top2_update(k0, k1, key); }  // (knn_l2_common.h, with l2_group4)

// DIM4 = number of float4 groups held in registers per query (dim = 4*DIM4 + tail, tail < 4 handled via LDS/global).
// Generic variant (DIM4 == 0) keeps the query in LDS as well.
// NMSLIB's L2SqrSIMD (similarity_search/src/distcomp_lp.cc:399-451) restated for one 4-chunk: four lane accumulators.
__device__ __forceinline__ void l2_group4_nms(float4 &t, float4 a, float4 b) {
    const float d0 = __fsub_rn(a.x, b.x), d1 = __fsub_rn(a.y, b.y), d2 = __fsub_rn(a.z, b.z), d3 = __fsub_rn(a.w, b.w);
    t.x = __fadd_rn(t.x, __fmul_rn(d0, d0));
    t.y = __fadd_rn(t.y, __fmul_rn(d1, d1));
    t.z = __fadd_rn(t.z, __fmul_rn(d2, d2));
    t.w = __fadd_rn(t.w, __fmul_rn(d3, d3));
}

// NMS = false: cvflann::L2<float> order, squared distance (LINEAR).  NMS = true: NMSLIB "l2" space order and TRUE distance
// sqrt(sum) (BRUTEFORCENMS, reference matchers.cpp:476-519); there the query is the first operand (x - y with x = query).
struct L2ExactArgs {
    const float *q;
    size_t q_stride, q_bstride;
    const float *t;
    size_t t_stride, t_bstride;
    int nq, nt, dim, rows_per_split, nsplit, tile_rows;
    ulonglong2 *part;
};

// smem: [tile_rows][dim_pad] floats; (bx, by, bz) = (query block, split, batch item).  Every thread of the 256-thread workgroup calls.
template <int DIM4, bool NMS>
__device__ __forceinline__ void l2_exact_body(const L2ExactArgs &a, float *smem, int bx, int by, int bz) {
    const float *__restrict__ q = a.q;
    const float *__restrict__ t = a.t;
    const size_t q_stride = a.q_stride, q_bstride = a.q_bstride, t_stride = a.t_stride, t_bstride = a.t_bstride;
    const int nq = a.nq, nt = a.nt, dim = a.dim, rows_per_split = a.rows_per_split, nsplit = a.nsplit, tile_rows = a.tile_rows;
    ulonglong2 *__restrict__ part = a.part;
    const int dim_pad = (dim + 3) & ~3;
    const int tid = threadIdx.x;
    const int split = by, b = bz;
    const int qi = bx * kQPB + tid;
    q += (size_t)b * q_bstride;
    t += (size_t)b * t_bstride;
    const int ngroups = dim / 4;
    const int tail = dim - ngroups * 4;

    float4 qa[DIM4 > 0 ? DIM4 : 1];
    float qtail[3] = {0.f, 0.f, 0.f};
    const float *qrow = q + (size_t)(qi < nq ? qi : 0) * q_stride;
    if constexpr (DIM4 > 0) {
#pragma unroll
        for (int g = 0; g < DIM4; ++g) qa[g] = make_float4(qrow[4 * g], qrow[4 * g + 1], qrow[4 * g + 2], qrow[4 * g + 3]);
    }
    for (int j = 0; j < tail; ++j) qtail[j] = qrow[ngroups * 4 + j];

    u64 k0 = ~0ull, k1 = ~0ull;
    const int r_begin = split * rows_per_split;
    const int r_end = min(nt, r_begin + rows_per_split);
    // cvflann order with the query in registers: TWO train rows per instruction on packed fp32 ops (v_pk_add_f32 / v_pk_mul_f32; no
    // FMA, the per-row operation order is exactly l2_group4's), the tile stored pair-interleaved: element c of rows 2p, 2p+1 side by side
    constexpr bool kPacked = DIM4 > 0 && !NMS;
    for (int base = r_begin; base < r_end; base += tile_rows) {
        const int rows = min(tile_rows, r_end - base);
        __syncthreads();
        for (int i = tid; i < rows * dim_pad; i += kQPB) {
            const int r = i / dim_pad, c = i - r * dim_pad;
            const float v = (c < dim) ? t[(size_t)(base + r) * t_stride + c] : 0.f;
            if constexpr (kPacked) smem[((size_t)(r >> 1) * dim_pad + c) * 2 + (r & 1)] = v;
            else smem[i] = v;
        }
        __syncthreads();
        if constexpr (kPacked) {
            typedef float f32x2 __attribute__((ext_vector_type(2)));
            for (int rp = 0; 2 * rp < rows; ++rp) {
                const float4 *tp = reinterpret_cast<const float4 *>(smem + (size_t)rp * dim_pad * 2);
                f32x2 res = {0.f, 0.f};
#pragma unroll
                for (int g = 0; g < DIM4; ++g) {
                    // the query already holds 4 * DIM4 registers: keep at most eight 16-byte LDS loads in flight (without the fence
                    // the scheduler hoists all 2 * DIM4 of them and the kernel drops to one wave per SIMD)
                    if ((g & 3) == 0) asm volatile("" ::: "memory");
                    const float4 v0 = tp[2 * g], v1 = tp[2 * g + 1];
                    // t - q with the query element broadcast to both halves by op_sel straight from the register pair it lives in
                    // (written as (t, t') - (q, q) the compiler materialises every (q, q) pair outside the row loop: 256 more registers)
                    const f32x2 qxy = {qa[g].x, qa[g].y}, qzw = {qa[g].z, qa[g].w};
                    f32x2 d0, d1, d2, d3;
                    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d0) : "v"(f32x2{v0.x, v0.y}), "v"(qxy));
                    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d1) : "v"(f32x2{v0.z, v0.w}), "v"(qxy));
                    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,0] op_sel_hi:[1,0] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d2) : "v"(f32x2{v1.x, v1.y}), "v"(qzw));
                    asm("v_pk_add_f32 %0, %1, %2 op_sel:[0,1] op_sel_hi:[1,1] neg_lo:[0,1] neg_hi:[0,1]" : "=v"(d3) : "v"(f32x2{v1.z, v1.w}), "v"(qzw));
                    f32x2 sg = d0 * d0 + d1 * d1;  // (this file is compiled with -ffp-contract=off: separate multiplies and adds)
                    sg = sg + d2 * d2;
                    sg = sg + d3 * d3;
                    res = res + sg;
                }
                float ra = res.x, rb = res.y;
                const float *tt = smem + ((size_t)rp * dim_pad + ngroups * 4) * 2;  // scalar tail: result += diff*diff, one element at a time
                for (int j = 0; j < tail; ++j) {
                    const float da = __fsub_rn(tt[2 * j], qtail[j]), db = __fsub_rn(tt[2 * j + 1], qtail[j]);
                    ra = __fadd_rn(ra, __fmul_rn(da, da));
                    rb = __fadd_rn(rb, __fmul_rn(db, db));
                }
                const int r = 2 * rp;
                top2_update(k0, k1, ((u64)__float_as_uint(ra) << 32) | (u64)(uint32_t)(base + r));
                if (r + 1 < rows) top2_update(k0, k1, ((u64)__float_as_uint(rb) << 32) | (u64)(uint32_t)(base + r + 1));
            }
        } else
        for (int r = 0; r < rows; ++r) {
            const float4 *trow = reinterpret_cast<const float4 *>(smem + (size_t)r * dim_pad);
            float res = 0.f;
            if constexpr (NMS) {
                float4 acc4 = make_float4(0.f, 0.f, 0.f, 0.f);
                if constexpr (DIM4 > 0) {
#pragma unroll
                    for (int g = 0; g < DIM4; ++g) l2_group4_nms(acc4, qa[g], trow[g]);
                } else {
                    for (int g = 0; g < ngroups; ++g) {
                        const float4 qv = make_float4(qrow[4 * g], qrow[4 * g + 1], qrow[4 * g + 2], qrow[4 * g + 3]);
                        l2_group4_nms(acc4, qv, trow[g]);
                    }
                }
                res = __fadd_rn(__fadd_rn(__fadd_rn(acc4.x, acc4.y), acc4.z), acc4.w);
            } else if constexpr (DIM4 > 0) {
#pragma unroll
                for (int g = 0; g < DIM4; ++g) res = l2_group4(res, trow[g], qa[g]);
            } else {
                for (int g = 0; g < ngroups; ++g) {
                    const float4 qv = make_float4(qrow[4 * g], qrow[4 * g + 1], qrow[4 * g + 2], qrow[4 * g + 3]);
                    res = l2_group4(res, trow[g], qv);
                }
            }
            // scalar tail: result += diff*diff, one element at a time
            const float *tt = smem + (size_t)r * dim_pad + ngroups * 4;
            for (int j = 0; j < tail; ++j) {
                const float d = NMS ? __fsub_rn(qtail[j], tt[j]) : __fsub_rn(tt[j], qtail[j]);
                res = __fadd_rn(res, __fmul_rn(d, d));
            }
            if constexpr (NMS) res = sqrtf(res);
            const u64 key = ((u64)__float_as_uint(res) << 32) | (u64)(uint32_t)(base + r);
            top2_update(k0, k1, key);
        }
    }
    if (qi < nq) part[((size_t)b * nsplit + split) * nq + qi] = make_ulonglong2(k0, k1);
}

template <int DIM4, bool NMS>
__global__ __launch_bounds__(kQPB) void knn_l2_exact_kernel(L2ExactArgs a) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    l2_exact_body<DIM4, NMS>(a, smem, blockIdx.x, blockIdx.y, blockIdx.z);
}

// The auto path in one launch: the operand-preparation kernel has decided (device flag) whether the descriptors are integer-valued
// in [0,255]; if so the workgroups run the int8 matrix-core tile loop (knn_l2_mfma_body.h), otherwise the exact fp32 kernel body --
// bit-identical results either way, different partial tables (the merge reads the same flag).  1-D grid = the larger of the two
// grids; the surplus workgroups of the path taken exit at once.
template <int DIM4, int KS>
__global__ __launch_bounds__(kQPB, 3) void knn_l2_auto_kernel(l2mfma::L2MfmaArgs m, unsigned grid_m, L2ExactArgs e, int qtiles_e,
                                                           unsigned grid_e, L2Gate gate, int *__restrict__ hint) {
    extern __shared__ __attribute__((aligned(16))) float smem[];
    const bool nonint = *gate.flag == gate.gen;
    // host-visible word for the NEXT call of this context (never waited for): which kind of data the last finished call saw
    if (hint && blockIdx.x == 0 && threadIdx.x == 0) *hint = 2 * gate.gen + (nonint ? 1 : 0);
    if (!nonint) {
        if (blockIdx.x < grid_m) l2mfma::l2_mfma_body<KS, 4>(m, reinterpret_cast<l2mfma::v4i *>(smem), blockIdx.x);
        return;
    }
    if (blockIdx.x >= grid_e) return;
    const int bx = blockIdx.x % qtiles_e, rest = blockIdx.x / qtiles_e;
    l2_exact_body<DIM4, false>(e, smem, bx, rest % e.nsplit, rest / e.nsplit);
}

// LANES lanes per query: each lane folds every LANES-th split, xor-shuffles combine the lanes (64-bit (dist bits, row) keys).
// Two partial tables may be offered: the exact kernel's (part, nsplit) and the MFMA kernel's (part_m, nsplit_m); the gate says which
// one this call filled (one merge launch serves both outcomes of the auto path).
template <int LANES>
__global__ __launch_bounds__(256) void knn_l2_merge_kernel(const ulonglong2 *__restrict__ part, int nsplit,
                                                           const ulonglong2 *__restrict__ part_m, int nsplit_m, L2Gate gate, int nq,
                                                           int k, int32_t *__restrict__ idx, float *__restrict__ dist) {
    if (part_m && !(gate.flag && *gate.flag == gate.gen)) {
        part = part_m;
        nsplit = nsplit_m;
    }
    const int b = blockIdx.y;
    const int sub = threadIdx.x & (LANES - 1);
    const int qi = blockIdx.x * (256 / LANES) + threadIdx.x / LANES;
    u64 b0 = ~0ull, b1 = ~0ull;
    if (qi < nq) {
        for (int s = sub; s < nsplit; s += LANES) {
            const ulonglong2 p = part[((size_t)b * nsplit + s) * nq + qi];
            if (p.x != ~0ull) top2_update(b0, b1, p.x);
            if (p.y != ~0ull) top2_update(b0, b1, p.y);
        }
    }
#pragma unroll
    for (int off = 1; off < LANES; off <<= 1) {
        const u64 o0 = __shfl_xor(b0, off), o1 = __shfl_xor(b1, off);
        top2_update(b0, b1, o0);
        top2_update(b0, b1, o1);
    }
    if (sub != 0 || qi >= nq) return;
    const size_t o = ((size_t)b * nq + qi) * k;
    idx[o] = (int32_t)(b0 & 0xFFFFFFFFull);
    dist[o] = __uint_as_float((uint32_t)(b0 >> 32));
    if (k == 2) {
        idx[o + 1] = (int32_t)(b1 & 0xFFFFFFFFull);
        dist[o + 1] = __uint_as_float((uint32_t)(b1 >> 32));
    }
}

void launch_knn_l2_merge(const void *part, int nsplit, const void *part_m, int nsplit_m, L2Gate gate, int nq, int k, int batch,
                         int32_t *d_idx, float *d_dist, hipStream_t s) {
    if (std::max(nsplit, nsplit_m) <= 4) {
        dim3 mgrid((nq + 255) / 256, batch);
        hipLaunchKernelGGL(knn_l2_merge_kernel<1>, mgrid, dim3(256), 0, s, (const ulonglong2 *)part, nsplit, (const ulonglong2 *)part_m,
                           nsplit_m, gate, nq, k, d_idx, d_dist);
    } else {
        dim3 mgrid((nq + 15) / 16, batch);
        hipLaunchKernelGGL(knn_l2_merge_kernel<16>, mgrid, dim3(256), 0, s, (const ulonglong2 *)part, nsplit, (const ulonglong2 *)part_m,
                           nsplit_m, gate, nq, k, d_idx, d_dist);
    }
}

}  // namespace

int launch_knn_l2(mlpl_ctx *ctx, const float *d_q, int nq, size_t q_stride, size_t q_bstride, const float *d_t, int nt,
                  size_t t_stride, size_t t_bstride, int dim, int k, int batch, int32_t *d_idx, float *d_dist,
                  hipStream_t s, int nms_order) {
    if (!d_q || !d_t || !d_idx || !d_dist || nq < 0 || batch < 1 || batch > 65535 || (k != 1 && k != 2) || nt < k ||
        dim < 1 || dim > 1024 || q_stride < (size_t)dim || t_stride < (size_t)dim) {
        set_error("knn_l2: bad arguments (nq=%d nt=%d dim=%d k=%d batch=%d)", nq, nt, dim, k, batch);
        return MLPL_E_BAD_INPUT;
    }
    if (nq == 0) return MLPL_OK;

    L2Gate gate{nullptr, 0};
    const void *part_m = nullptr;
    int nsplit_m = 0;
    l2mfma::L2MfmaPlan plan{};
    bool fused = false;
    // The fp16 candidate path + exact re-rank (knn_l2_f16.hip) is exact for ANY float data; the int8 path is faster on integer-valued data.
    // Auto mode picks by a hint, without a host hop: the last kernel of either path leaves "2 * generation + (data were not integer-valued)"
    // in a pinned word; the next call reads whatever is there (a stale or missing hint costs speed, never results): odd -> fp16 path, even ->
    // int8 / exact path as before.  Option l2_float_mfma: 0 = never, 2 = the fp16 path for every float call.
    const bool f16_ok = !nms_order && knn_l2_f16_applicable(dim, nt, k);
    if (ctx->l2_mode == 3 && !f16_ok) {
        set_error("knn_l2 (fp16 path): not applicable (dim %d > 128 or the NMSLIB order)", dim);
        return MLPL_E_BAD_INPUT;
    }
    if (ctx->l2_mode == 0 && ctx->opt_l2_float_mfma == 1 && !ctx->l2_hint_host) {  // the hint word lives with the context
        void *h = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped) == hipSuccess) {
            std::memset(h, 0, 64);
            ctx->l2_hint_host = (int *)h;
            if (hipHostGetDevicePointer((void **)&ctx->l2_hint_dev, h, 0) != hipSuccess) ctx->l2_hint_dev = nullptr;
        }
    }
    const bool hinted = ctx->l2_mode == 0 && ctx->opt_l2_float_mfma == 1 && ctx->l2_hint_host && (*(volatile int *)ctx->l2_hint_host & 1);
    if (ctx->l2_mode == 3 || (ctx->l2_mode == 0 && f16_ok && (ctx->opt_l2_float_mfma == 2 || hinted))) {
        void *flag;
        int rc = ws_get(ctx, WS_L2_FLAG, 4096, &flag);
        if (rc) return rc;
        int *dflag = (int *)flag;
        if (ctx->l2_flag_ptr != flag || ctx->l2_gen == 0x3FFFFFFF) {  // a fresh (or wrapped) flag block starts below every generation
            MLPL_HIP_TRY(hipMemsetAsync(dflag, 0, 16, s));
            ctx->l2_flag_ptr = flag;
            ctx->l2_gen = 0;
        }
        const int gen = ++ctx->l2_gen;
        if ((rc = launch_knn_l2_f16(ctx, d_q, nq, q_stride, q_bstride, d_t, nt, t_stride, t_bstride, dim, k, batch, gen, dflag,
                                    ctx->l2_mode == 0 ? ctx->l2_hint_dev : nullptr, d_idx, d_dist, s)))
            return rc;
        if (ctx->l2_mode == 3) {  // forcing is a test / diagnostic mode: report out-of-range data as an error (one host hop; the result is still exact)
            int hbad = 0;
            MLPL_HIP_TRY(hipMemcpyAsync(&hbad, dflag + 1, 4, hipMemcpyDeviceToHost, s));
            MLPL_HIP_TRY(hipStreamSynchronize(s));
            if (hbad == gen) {
                set_error("knn_l2 (fp16 path): a descriptor row is outside the path's range (non-finite, |x| > 1e15 or largest element < 1e-12)");
                return MLPL_E_BAD_INPUT;
            }
        }
        return MLPL_OK;
    }
    if (ctx->l2_mode != 1 && !nms_order) {
        // int8 matrix-core distance-GEMM when the data qualify (auto) or when forced.  In auto mode ONE kernel holds both paths and a
        // device flag written by the operand-preparation kernel decides which one its workgroups run and which partial table the one
        // merge reads: three launches, no host round trip.
        int rc = launch_knn_l2_mfma(ctx, d_q, nq, q_stride, q_bstride, d_t, nt, t_stride, t_bstride, dim, batch, s, ctx->l2_mode == 2, &plan);
        if (rc < 0) return rc;
        if (rc == 0) {
            part_m = plan.args.part;
            nsplit_m = plan.args.nsplit;
            if (ctx->l2_mode == 2) {
                launch_knn_l2_merge(nullptr, 0, part_m, nsplit_m, L2Gate{nullptr, 0}, nq, k, batch, d_idx, d_dist, s);
                MLPL_HIP_TRY(hipGetLastError());
                return MLPL_OK;
            }
            gate = plan.gate;
            fused = true;
        }
        // rc == 1: the matrix-core path is not applicable at all: the exact kernel runs unconditionally
    }

    const int dim_pad = (dim + 3) & ~3;
    const int kTileRows = std::max(1, std::min(kMaxTileRows, 8192 / dim_pad));  // <= 32 KiB of LDS per block
    const int qtiles = (nq + kQPB - 1) / kQPB;
    const long long target_blocks = 8LL * ctx->num_cus;
    const int max_split = (nt + kTileRows - 1) / kTileRows;
    long long want = (target_blocks + (long long)qtiles * batch - 1) / ((long long)qtiles * batch);
    int nsplit = (int)std::max<long long>(1, std::min<long long>(want, max_split));
    int rps = (nt + nsplit - 1) / nsplit;
    rps = ((rps + kTileRows - 1) / kTileRows) * kTileRows;
    nsplit = (nt + rps - 1) / rps;
    if (nsplit > 65535) {
        rps = ((nt + 65534) / 65535 + kTileRows - 1) / kTileRows * kTileRows;
        nsplit = (nt + rps - 1) / rps;
    }
    void *part = nullptr;
    int rc = ws_get(ctx, WS_PARTIAL, (size_t)batch * nsplit * nq * sizeof(ulonglong2), &part);
    if (rc) return rc;

    const size_t shmem = (size_t)kTileRows * dim_pad * sizeof(float);
    const L2ExactArgs ea{d_q, q_stride, q_bstride, d_t, t_stride, t_bstride, nq, nt, dim, rps, nsplit, kTileRows, (ulonglong2 *)part};
    const int g4 = dim / 4;
    if (fused) {
        const long long grid_e = (long long)qtiles * nsplit * batch;
        if (grid_e > 0x7FFFFFFFLL) {
            set_error("knn_l2: problem too large (%lld workgroups)", grid_e);
            return MLPL_E_BAD_INPUT;
        }
        const unsigned grid = std::max<unsigned>(plan.grid, (unsigned)grid_e);
        const size_t lds = std::max<size_t>(shmem, 18432);  // l2_mfma_lds_bytes<KS>() <= 18432 for every KS
        prof_mark(ctx, MLPL_PROF_KNN_L2, 0, s);
#define MLPL_L2_AUTO(D4, KS)                                                                                                         \
    hipLaunchKernelGGL((knn_l2_auto_kernel<D4, KS>), dim3(grid), dim3(kQPB), lds, s, plan.args, plan.grid, ea, qtiles, (unsigned)grid_e, \
                       gate, ctx->l2_mode == 0 ? ctx->l2_hint_dev : (int *)nullptr)
        if (g4 == 32 && plan.ksel == 4) MLPL_L2_AUTO(32, 4);
        else if (g4 == 16 && plan.ksel == 2) MLPL_L2_AUTO(16, 2);
        else if (g4 == 8 && plan.ksel == 1) MLPL_L2_AUTO(8, 1);
        else if (plan.ksel == 1) MLPL_L2_AUTO(0, 1);
        else if (plan.ksel == 2) MLPL_L2_AUTO(0, 2);
        else if (plan.ksel == 4) MLPL_L2_AUTO(0, 4);
        else MLPL_L2_AUTO(0, 8);
#undef MLPL_L2_AUTO
        prof_mark(ctx, MLPL_PROF_KNN_L2, 1, s);
    } else {
        dim3 grid(qtiles, nsplit, batch);
#define MLPL_L2_LAUNCH(D4)                                                                              \
    if (nms_order)                                                                                      \
        hipLaunchKernelGGL((knn_l2_exact_kernel<D4, true>), grid, dim3(kQPB), shmem, s, ea);            \
    else                                                                                                \
        hipLaunchKernelGGL((knn_l2_exact_kernel<D4, false>), grid, dim3(kQPB), shmem, s, ea)
        if (g4 == 32) MLPL_L2_LAUNCH(32);
        else if (g4 == 16) MLPL_L2_LAUNCH(16);
        else if (g4 == 8) MLPL_L2_LAUNCH(8);
        else MLPL_L2_LAUNCH(0);
#undef MLPL_L2_LAUNCH
    }
    launch_knn_l2_merge(part, nsplit, part_m, nsplit_m, gate, nq, k, batch, d_idx, d_dist, s);
    MLPL_HIP_TRY(hipGetLastError());
    return MLPL_OK;
}

}  // namespace mlpl
