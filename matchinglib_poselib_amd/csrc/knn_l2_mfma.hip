// knn_l2_mfma.hip -- squared-L2 2-NN as an int8 matrix-core distance-GEMM with a fused top-2 epilogue (gfx950).
//
// Replaces cvflann::Index<L2<float>>(LinearIndexParams).knnSearch (reference matchinglib/source/matchers.cpp:634-664)
// for float descriptors whose elements are all integers in [0,255] (OpenCV SIFT layout).  For such data everything is integer
// arithmetic, hence exact: with the operands stored as  qs = q - 128  and  ts = 127 - t  (both fit int8) the i32 accumulator holds
//     acc = sum qs ts = 127 sum q - q.t + 128 sum t - 127*128*dim
// and  d2 = |q|^2 + |t|^2 - 2 q.t = 2 acc + Tcb(t) + Qd(q)  with the per-row constants
//     Tcb = |t|^2 - 256 sum t + 2*127*128*dim + 256 dim   (> 0),     Qd = |q|^2 - 254 sum q - 256 dim,
// so for one query the order of d2 over the train rows is the order of  e = 2 acc + Tcb >= 0.  The accumulator starts at Tcb >> 1 and
// the parity bit of Tcb travels with the row index, which makes the packed 32-bit key  e << ib | row  ONE instruction per candidate:
//     key = (accC << (ib + 1)) | ((Tcb & 1) << ib | row)          (v_lshl_or_b32)
// d2 < 2^24 (dim <= 256) converts to float exactly, and the reference's own fp32 running sum of integer squares is exact for the same
// reason, so distances, hence the lexicographic (d2, trainIdx) order, are bit-identical to the CPU path.  Anything else (fractional /
// negative / large values, dim > 256) takes knn_l2_exact_kernel.
//
// Mapping: v_mfma_i32_32x32x32_i8 with A = 32 train rows, B = 32 queries, so a lane's 16 accumulators are 16 train rows of ONE query
// (column = lane & 31): the running top-2 stays per lane on the packed keys, per group of train tiles (the row field only has to number
// the rows of one group); after each group the pair is folded into 64-bit (d2, train row) keys.  Operands are pre-swizzled once per
// call into fragment order ([tile][kstep][lane] x 16 B: row = lane & 31, the 16 k of half lane >> 5 -- any k assignment works as long
// as both operands use the same one), so a group of train tiles is staged into LDS by a straight coalesced copy and read back
// conflict-free with one ds_read_b128 per MFMA; the query fragments live in VGPRs.

#include <algorithm>

#include "mlpl_internal.h"

namespace mlpl {

namespace {

typedef int v4i __attribute__((ext_vector_type(4)));
typedef int v16i __attribute__((ext_vector_type(16)));
typedef unsigned long long u64;

constexpr int kMaxKS = 8;  // dim <= 256

__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t umin3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_min3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Operand preparation, one launch for both sets (blockIdx.z = 0 queries, 1 train rows), one block per 32-row tile:
//   frag[(tile*KS + s)*64 + lane] = 16 int8: enc(X[tile*32 + (lane&31)][32 s + 16 (lane>>5) + j]), enc = x - 128 (queries) / 127 - x
//                                   (train rows); 0 past the end of the row so that padding adds nothing to the product
//   cst[tile*32 + r]              = Qd (queries) / Tcb (train rows), see the header
//   *flag = gen (atomicMax) when an element is not an integer in [0,255]: `gen` is the context's call counter, so the flag needs no
//   reset between calls -- a consumer tests *flag == gen.
// Thread (r = tid & 31, c = tid >> 5) converts the 16 elements of half-step u = c, c + 8, ... of row r (one fragment entry each).
struct L2PrepArgs {
    const float *X;
    size_t stride, bstride;
    int n, ntiles;
    uint4 *frag;
    int *cst;
};

__global__ __launch_bounds__(256) void l2_prep_kernel(L2PrepArgs qa, L2PrepArgs ta, int dim, int KS, int *__restrict__ flag, int gen) {
    const L2PrepArgs A = blockIdx.z ? ta : qa;
    const int tile = blockIdx.x;
    if (tile >= A.ntiles) return;
    __shared__ int part[2][8][32];
    const int b = blockIdx.y, r = threadIdx.x & 31, c = threadIdx.x >> 5;
    const int row = tile * 32 + r;
    const bool train = blockIdx.z != 0;
    const float *x = A.X + (size_t)b * A.bstride + (size_t)row * A.stride;
    uint4 *frag = A.frag + ((size_t)b * A.ntiles + tile) * KS * 64;
    int s1 = 0, s2 = 0;
    bool bad = false;
    for (int u = c; u < 2 * KS; u += 8) {
        const int k0 = 16 * u;
        uint32_t w[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            float v = 0.f;
            const bool in = row < A.n && k0 + j < dim;
            if (in) v = x[k0 + j];
            bad = bad || !(v >= 0.f && v <= 255.f && v == floorf(v));
            const int iv = (int)v;
            s1 += iv;
            s2 += iv * iv;
            const int e = in ? (train ? 127 - iv : iv - 128) : 0;
            w[j >> 2] |= (uint32_t)(e & 0xFF) << (8 * (j & 3));
        }
        frag[(u >> 1) * 64 + (u & 1) * 32 + r] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    if (bad) atomicMax(flag, gen);
    part[0][c][r] = s1;
    part[1][c][r] = s2;
    __syncthreads();
    if (c == 0) {
#pragma unroll
        for (int i = 1; i < 8; ++i) s1 += part[0][i][r], s2 += part[1][i][r];
        A.cst[((size_t)b * A.ntiles + tile) * 32 + r] = train ? s2 - 256 * s1 + (2 * 127 * 128 + 256) * dim : s2 - 254 * s1 - 256 * dim;
    }
}

// One workgroup = W waves = W query tiles (32 queries each) against the train tiles of one split, consumed in groups of G = 16/KS tiles
// (16 KiB of LDS; 8 tiles for KS = 1).  A whole group is requested from memory at once and the next group's loads are in flight (16 VGPRs per thread)
// while this one is multiplied.  Work items are numbered so that each XCD (block id mod 8) owns a contiguous range of
// (batch, split, query block): neighbouring splits share an L2, so each XCD pulls every query fragment but only an eighth of the
// train fragments across the fabric.
template <int KS, int W>
__global__ __launch_bounds__(64 * W, 2) void knn_l2_mfma_kernel(const uint4 *__restrict__ qfrag, const int *__restrict__ qcst,
                                                             const uint4 *__restrict__ tfrag, const int *__restrict__ tcst, int nq,
                                                             int nt, int nq_tiles, int nt_tiles, int tiles_per_split, int nsplit,
                                                             int qblocks, int batch, int ib, ulonglong2 *__restrict__ part, L2Gate gate,
                                                             unsigned long long *__restrict__ stamps) {
    unsigned long long st0 = 0, st1 = 0;
    if (stamps) st0 = __builtin_readcyclecounter(), st1 = __builtin_amdgcn_s_memrealtime();
    constexpr int G = KS == 1 ? 8 : 16 / KS;  // tiles per group (at most 256 rows: one thread per row in commit)
    constexpr int NT = 64 * W;               // threads
    constexpr int kPre = G * KS * 64 / NT;   // uint4 per thread and group
    // (native vector types throughout the staging path: an array of HIP_vector_type structs is not split into registers and ends up
    // in scratch, with a wait per load)
    __shared__ __attribute__((aligned(16))) v4i lds[G * KS * 64 + 2 * G * 8];  // fragments | accumulator starts | row words
    v4i *tileA = lds;
    int *tileC = reinterpret_cast<int *>(lds + G * KS * 64);
    uint32_t *tileR = reinterpret_cast<uint32_t *>(lds + G * KS * 64 + G * 8);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware work numbering
    const long long total = (long long)batch * nsplit * qblocks, per_xcd = (total + 7) / 8;
    const long long j = blockIdx.x >> 3, w = (long long)(blockIdx.x & 7) * per_xcd + j;
    if (j >= per_xcd || w >= total) return;
    const int qb = (int)(w % qblocks), split = (int)((w / qblocks) % nsplit), b = (int)(w / ((long long)qblocks * nsplit));
    const int qtile = qb * W + wave;
    const bool wave_active = qtile < nq_tiles;
    qfrag += (size_t)b * nq_tiles * KS * 64;
    tfrag += (size_t)b * nt_tiles * KS * 64;
    qcst += (size_t)b * nq_tiles * 32;
    tcst += (size_t)b * nt_tiles * 32;

    const int t_begin = split * tiles_per_split;
    const int t_end = min(nt_tiles, t_begin + tiles_per_split);
    v4i pre[kPre];
    int pre_c = 0;
    // a whole group is always moved (no per-load predicates: they would serialise the loads); what lies past t_end is the next
    // split's tiles or the padding the launcher allocates behind the last tile, and is never multiplied
    auto fetch = [&](int t0) {
        const v4i *src = reinterpret_cast<const v4i *>(tfrag) + (size_t)t0 * KS * 64;
#pragma unroll
        for (int i = 0; i < kPre; ++i) pre[i] = src[tid + NT * i];
        pre_c = tcst[t0 * 32 + (tid & (G * 32 - 1))];
    };
    auto commit = [&](int t0) {
#pragma unroll
        for (int i = 0; i < kPre; ++i) tileA[tid + NT * i] = pre[i];
        if (G * 32 >= NT || tid < G * 32) {
            tileC[tid] = pre_c >> 1;
            // rows past the end of the train set get an all-ones row word: their keys come out as 0xFFFFFFFF = "none"
            tileR[tid] = (t0 * 32 + tid < nt) ? (((uint32_t)pre_c & 1u) << ib) | (uint32_t)tid : 0xFFFFFFFFu;
        }
    };
    // every load of the prologue is issued before the first wait (the gate test below), so the block pays one round trip, not three
    if (t_begin < t_end) fetch(t_begin);
    v4i qf[KS];
    const int qtile_ld = wave_active ? qtile : 0;
#pragma unroll
    for (int s = 0; s < KS; ++s) qf[s] = reinterpret_cast<const v4i *>(qfrag)[((size_t)qtile_ld * KS + s) * 64 + lane];
    const int qd = qcst[qtile_ld * 32 + (lane & 31)];
    if (*gate.flag == gate.gen) return;  // some element is not an integer in [0,255]: the exact kernel (gated the other way) runs instead

    u64 g0 = ~0ull, g1 = ~0ull;  // (float bits of d2) << 32 | train row
    auto upd = [&](u64 g) {
        const bool lt0 = g < g0, lt1 = g < g1;
        g1 = lt0 ? g0 : (lt1 ? g : g1);
        g0 = lt0 ? g : g0;
    };
    const uint32_t lmask = (1u << ib) - 1u;
    for (int t0 = t_begin; t0 < t_end; t0 += G) {
        if (t0 != t_begin) __syncthreads();  // every wave is done with the previous group
        commit(t0);
        if (t0 + G < t_end) fetch(t0 + G);
        __syncthreads();
        const int g_end = min(t0 + G, t_end);
        uint32_t k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;
        for (int t = t0; t < g_end; ++t) {
            const v4i *A = tileA + (t - t0) * KS * 64;
            // accumulator reg r of this lane is train row (r&3) + 8 (r>>2) + 4 (lane>>5) of the tile, query lane&31; it starts at Tcb >> 1
            v16i acc;
            uint32_t rw[16];
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const v4i c4 = *reinterpret_cast<const v4i *>(&tileC[(t - t0) * 32 + 8 * g + 4 * (lane >> 5)]);
                const v4i r4 = *reinterpret_cast<const v4i *>(&tileR[(t - t0) * 32 + 8 * g + 4 * (lane >> 5)]);
                acc[4 * g] = c4[0], acc[4 * g + 1] = c4[1], acc[4 * g + 2] = c4[2], acc[4 * g + 3] = c4[3];
                rw[4 * g] = (uint32_t)r4[0], rw[4 * g + 1] = (uint32_t)r4[1], rw[4 * g + 2] = (uint32_t)r4[2], rw[4 * g + 3] = (uint32_t)r4[3];
            }
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                acc = __builtin_amdgcn_mfma_i32_32x32x32_i8(A[s * 64 + lane], qf[s], acc, 0, 0, 0);
            }
            uint32_t key[16];
#pragma unroll
            for (int r = 0; r < 16; ++r) key[r] = ((uint32_t)acc[r] << (ib + 1)) | rw[r];
            // four candidates per step, 5 ops: with T = {k0, a, b} the second smallest of T + {k1} is min(med3(T), k1) because k1 >= k0
#pragma unroll
            for (int r = 0; r < 16; r += 4) {
                const uint32_t s0 = umed3(k0, key[r], key[r + 1]);
                const uint32_t u0 = umin3(k0, key[r], key[r + 1]);
                const uint32_t s1 = umed3(u0, key[r + 2], key[r + 3]);
                k0 = umin3(u0, key[r + 2], key[r + 3]);
                k1 = umin3(k1, s0, s1);
            }
        }
        // fold the group's pair into the 64-bit keys: d2 = e + Qd is exact in float (< 2^24)
        auto to_global = [&](uint32_t key) -> u64 {
            if (key == 0xFFFFFFFFu) return ~0ull;
            const float d = (float)((int)(key >> ib) + qd);
            return ((u64)__float_as_uint(d) << 32) | (u64)((uint32_t)t0 * 32u + (key & lmask));
        };
        upd(to_global(k0));
        upd(to_global(k1));
    }
    // combine the two lanes that share a query column
    const u64 o0 = __shfl_xor(g0, 32), o1 = __shfl_xor(g1, 32);
    upd(o0);
    upd(o1);
    const int qi = qtile * 32 + (lane & 31);
    if (wave_active && lane < 32 && qi < nq) part[((size_t)b * nsplit + split) * nq + qi] = make_ulonglong2(g0, g1);
    if (stamps && tid == 0) {  // diagnostics: {cycles, start tick (100 MHz), end tick, hardware id} per workgroup
        unsigned hw = 0, xcc = 0;
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID)" : "=s"(hw));
        asm volatile("s_getreg_b32 %0, hwreg(HW_REG_XCC_ID)" : "=s"(xcc));
        unsigned long long *o = stamps + 4 * (size_t)w;
        o[0] = __builtin_readcyclecounter() - st0;
        o[1] = st1;
        o[2] = __builtin_amdgcn_s_memrealtime();
        o[3] = (unsigned long long)hw | ((unsigned long long)(xcc & 0xF) << 32);
    }
}

}  // namespace

// Returns 1 when the matrix-core path cannot apply at all (caller runs the exact kernel ungated), 0 when the operand preparation and
// the MFMA kernel were enqueued: *part_out / *nsplit_out describe its partial table and *gate_out the device flag (flag == gen: the
// data did not qualify, the MFMA kernel exited early and the caller's exact kernel, gated the other way, produces the partials).  The
// caller launches the one merge.  < 0 on error.  Nothing here synchronises in auto mode, so the *_dev entry points stay asynchronous.
int launch_knn_l2_mfma(mlpl_ctx *ctx, const float *d_q, int nq, size_t q_stride, size_t q_bstride, const float *d_t, int nt,
                       size_t t_stride, size_t t_bstride, int dim, int batch, hipStream_t s, int force, L2Gate *gate_out,
                       const void **part_out, int *nsplit_out) {
    const int KS = (dim + 31) / 32;
    int ksel = 0;
    for (int c : {1, 2, 4, 8})
        if (KS <= c) {
            ksel = c;
            break;
        }
    if (!ksel) {
        if (force) {
            set_error("knn_l2 (MFMA): dim %d > %d is not supported on this path", dim, kMaxKS * 32);
            return MLPL_E_BAD_INPUT;
        }
        return 1;
    }
    // bits of e = d2 - Qd <= (65025 + 16129 + 256) * dim, the rest of the 32-bit key numbers the rows of one group of 16/ksel tiles
    int ebits = 1;
    while ((1ull << ebits) <= 81410ull * (unsigned long long)dim) ++ebits;
    const int ib = 32 - ebits;
    const int group = ksel == 1 ? 8 : 16 / ksel;
    if ((1 << ib) <= group * 32) return force ? MLPL_E_BAD_INPUT : 1;  // cannot happen for dim <= 256 (ib >= 7, group*32 <= 64 there)

    const int nq_tiles = (nq + 31) / 32, nt_tiles = (nt + 31) / 32;
    void *qf, *tf, *cst, *flag;
    int rc;
    if ((rc = ws_get(ctx, WS_PACK_Q, (size_t)batch * nq_tiles * ksel * 64 * 16, &qf))) return rc;
    // + one group (16 KiB of fragments, up to 512 constants) behind the last train tile: the kernel always moves whole groups
    if ((rc = ws_get(ctx, WS_PACK_T, (size_t)batch * nt_tiles * ksel * 64 * 16 + 16384, &tf))) return rc;
    if ((rc = ws_get(ctx, WS_AUX3, (size_t)batch * (nq_tiles + nt_tiles) * 32 * 4 + 2048 + 64, &cst))) return rc;
    if ((rc = ws_get(ctx, WS_L2_FLAG, 4096, &flag))) return rc;
    int *qc = (int *)cst, *tc = qc + (size_t)batch * nq_tiles * 32;
    int *dflag = (int *)flag;
    if (ctx->l2_flag_ptr != flag || ctx->l2_gen == 0x7FFFFFFF) {  // a fresh (or wrapped) flag starts below every generation
        MLPL_HIP_TRY(hipMemsetAsync(dflag, 0, 4, s));
        ctx->l2_flag_ptr = flag;
        ctx->l2_gen = 0;
    }
    const L2Gate gate{dflag, ++ctx->l2_gen};
    {
        const L2PrepArgs qa{d_q, q_stride, q_bstride, nq, nq_tiles, (uint4 *)qf, qc}, ta{d_t, t_stride, t_bstride, nt, nt_tiles, (uint4 *)tf, tc};
        hipLaunchKernelGGL(l2_prep_kernel, dim3((unsigned)std::max(nq_tiles, nt_tiles), batch, 2), dim3(256), 0, s, qa, ta, dim, ksel,
                           dflag, gate.gen);
    }
    if (force) {  // forcing is a test/diagnostic mode: report non-qualifying data as an error (one host hop)
        int hflag = 0;
        MLPL_HIP_TRY(hipMemcpyAsync(&hflag, dflag, 4, hipMemcpyDeviceToHost, s));
        MLPL_HIP_TRY(hipStreamSynchronize(s));
        if (hflag == gate.gen) {
            set_error("knn_l2 (MFMA): descriptors are not integer-valued in [0,255]");
            return MLPL_E_BAD_INPUT;
        }
    }
    *gate_out = gate;

    // Waves (= query tiles) per workgroup and train tiles per split (a whole number of groups).  What bounds this kernel on the
    // sizes of an image pair is the traffic from L2 into the CUs (a workgroup loads W query tiles and tps train tiles for W * tps
    // tile products), so the workgroup is made as square as the grid allows: 8 waves when that still gives every CU a workgroup.
    int waves = ctx->opt_l2_mfma_waves;
    if (waves != 4 && waves != 8) waves = ((long long)((nq_tiles + 7) / 8) * ((nt_tiles + 7) / 8) * batch >= ctx->num_cus) ? 8 : 4;
    const int qblocks = (nq_tiles + waves - 1) / waves;
    const int per_cu = ctx->opt_l2_mfma_blocks_per_cu > 0 ? ctx->opt_l2_mfma_blocks_per_cu : (waves == 8 ? 1 : 4);
    long long want_splits = ((long long)per_cu * ctx->num_cus + (long long)qblocks * batch - 1) / ((long long)qblocks * batch);
    int tps = (int)std::max<long long>(1, (nt_tiles + want_splits - 1) / std::max<long long>(1, want_splits));
    tps = (tps + group - 1) / group * group;
    int nsplit = (nt_tiles + tps - 1) / tps;
    const long long total = (long long)batch * nsplit * qblocks;
    if (total > 0x3FFFFFF0LL) return force ? MLPL_E_BAD_INPUT : 1;
    void *part = nullptr;
    if ((rc = ws_get(ctx, WS_AUX4, (size_t)batch * nsplit * nq * sizeof(ulonglong2), &part))) return rc;

    const unsigned grid = (unsigned)(((total + 7) / 8) * 8);
    unsigned long long *stamps = nullptr;
    ctx->dbg_stamp_items = 0;
    if (ctx->opt_hamming_stamps) {  // the same diagnostics switch as the Hamming kernels (mlpl_debug_hamming_stamps reads them back)
        void *sp = nullptr;
        if ((rc = ws_get(ctx, WS_DEBUG, (size_t)total * 32, &sp))) return rc;
        stamps = (unsigned long long *)sp;
        ctx->dbg_stamp_items = (int)total;
    }
    prof_mark(ctx, MLPL_PROF_KNN_L2, 0, s);
#define MLPL_MFMA_LAUNCH_W(K, WV)                                                                                                    \
    hipLaunchKernelGGL((knn_l2_mfma_kernel<K, WV>), dim3(grid), dim3(64 * WV), 0, s, (const uint4 *)qf, (const int *)qc,                  \
                       (const uint4 *)tf, (const int *)tc, nq, nt, nq_tiles, nt_tiles, tps, nsplit, qblocks, batch, ib, (ulonglong2 *)part, \
                       gate, stamps)
#define MLPL_MFMA_LAUNCH(K)          \
    if (waves == 8)                  \
        MLPL_MFMA_LAUNCH_W(K, 8);    \
    else                             \
        MLPL_MFMA_LAUNCH_W(K, 4)
    switch (ksel) {
        case 1: MLPL_MFMA_LAUNCH(1); break;
        case 2: MLPL_MFMA_LAUNCH(2); break;
        case 4: MLPL_MFMA_LAUNCH(4); break;
        default: MLPL_MFMA_LAUNCH(8); break;
    }
#undef MLPL_MFMA_LAUNCH_W
#undef MLPL_MFMA_LAUNCH
    prof_mark(ctx, MLPL_PROF_KNN_L2, 1, s);
    MLPL_HIP_TRY(hipGetLastError());
    *part_out = part;
    *nsplit_out = nsplit;
    return MLPL_OK;
}

}  // namespace mlpl
