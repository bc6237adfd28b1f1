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

#include "knn_l2_mfma_body.h"
#include "mlpl_internal.h"

namespace mlpl {

namespace {

using namespace l2mfma;

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

// VEC: rows are 16-byte aligned and dim is a multiple of 16 -> four 16-byte loads per half-step instead of sixteen scalar ones.
template <bool VEC>
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
        float xv[16];
        if constexpr (VEC) {
#pragma unroll
            for (int j = 0; j < 4; ++j) {
                float4 f = make_float4(0.f, 0.f, 0.f, 0.f);
                if (row < A.n && k0 < dim) f = *reinterpret_cast<const float4 *>(x + k0 + 4 * j);
                xv[4 * j] = f.x, xv[4 * j + 1] = f.y, xv[4 * j + 2] = f.z, xv[4 * j + 3] = f.w;
            }
        } else {
#pragma unroll
            for (int j = 0; j < 16; ++j) xv[j] = (row < A.n && k0 + j < dim) ? x[k0 + j] : 0.f;
        }
#pragma unroll
        for (int j = 0; j < 16; ++j) {
            const bool in = row < A.n && k0 + j < dim;
            const float v = xv[j];
            bad = bad || !(v >= 0.f && v <= 255.f && v == floorf(v));
            const int iv = (int)v;
            s1 += iv;
            s2 += iv * iv;
            const int e = in ? (train ? 127 - iv : iv - 128) : 0;
            w[j >> 2] |= (uint32_t)(e & 0xFF) << (8 * (j & 3));
        }
        frag[(u >> 1) * 64 + (u & 1) * 32 + r] = make_uint4(w[0], w[1], w[2], w[3]);
    }
    part[0][c][r] = s1;
    part[1][c][r] = s2;
    // (one atomic per block at most, and none once the word is set: with non-integer data every thread would queue on the same address)
    const int any_bad = __syncthreads_or(bad);
    if (threadIdx.x == 0 && any_bad && *(volatile int *)flag != gen) atomicMax(flag, gen);
    if (c == 0) {
#pragma unroll
        for (int i = 1; i < 8; ++i) s1 += part[0][i][r], s2 += part[1][i][r];
        A.cst[((size_t)b * A.ntiles + tile) * 32 + r] = train ? s2 - 256 * s1 + (2 * 127 * 128 + 256) * dim : s2 - 254 * s1 - 256 * dim;
    }
}

// Stand-alone form of the tile loop (forced mode, any W); the auto path runs the same body inside knn_l2_auto_kernel (knn_l2.hip).
template <int KS, int W>
__global__ __launch_bounds__(64 * W, 2) void knn_l2_mfma_kernel(L2MfmaArgs a, L2Gate gate) {
    __shared__ __attribute__((aligned(16))) v4i lds[l2_mfma_lds_bytes<KS>() / 16];
    if (*gate.flag == gate.gen) return;  // some element is not an integer in [0,255]
    l2_mfma_body<KS, W>(a, lds, blockIdx.x);
}

}  // namespace

// Prepares the matrix-core path: enqueues the operand preparation and fills *plan (kernel arguments, K-step class, gate, grid).
// Returns 1 when the path cannot apply at all (caller runs the exact kernel ungated), 0 on success, < 0 on error.  In forced mode the
// stand-alone tile-loop kernel is enqueued here as well (after a host check of the gate); in auto mode the caller launches
// knn_l2_auto_kernel, which runs this plan or the exact kernel's depending on the device flag, and then the one merge.  Nothing here
// synchronises in auto mode, so the *_dev entry points stay asynchronous.
int launch_knn_l2_mfma(mlpl_ctx *ctx, const float *d_q, int nq, size_t q_stride, size_t q_bstride, const float *d_t, int nt,
                       size_t t_stride, size_t t_bstride, int dim, int batch, hipStream_t s, int force, L2MfmaPlan *plan) {
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
    if (ctx->l2_flag_ptr != flag || ctx->l2_gen == 0x3FFFFFFF) {  // a fresh (or wrapped) flag starts below every generation
        MLPL_HIP_TRY(hipMemsetAsync(dflag, 0, 16, s));  // {not integer-valued, outside the fp16 path's range, re-rank overflows, -}
        ctx->l2_flag_ptr = flag;
        ctx->l2_gen = 0;
    }
    const L2Gate gate{dflag, ++ctx->l2_gen};
    {
        const L2PrepArgs qa{d_q, q_stride, q_bstride, nq, nq_tiles, (uint4 *)qf, qc}, ta{d_t, t_stride, t_bstride, nt, nt_tiles, (uint4 *)tf, tc};
        const bool vec = dim % 16 == 0 && ((uintptr_t)d_q | (uintptr_t)d_t) % 16 == 0 &&
                         (q_stride | q_bstride | t_stride | t_bstride) % 4 == 0;
        const dim3 pgrid((unsigned)std::max(nq_tiles, nt_tiles), batch, 2);
        if (vec) hipLaunchKernelGGL(l2_prep_kernel<true>, pgrid, dim3(256), 0, s, qa, ta, dim, ksel, dflag, gate.gen);
        else hipLaunchKernelGGL(l2_prep_kernel<false>, pgrid, dim3(256), 0, s, qa, ta, dim, ksel, dflag, gate.gen);
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
    plan->gate = gate;

    // Waves (= query tiles) per workgroup and train tiles per split (a whole number of groups).  What bounds this kernel on the
    // sizes of an image pair is the traffic from L2 into the CUs (a workgroup loads W query tiles and tps train tiles for W * tps
    // tile products), so the workgroup is made as square as the grid allows: 8 waves when that still gives every CU a workgroup.
    int waves = ctx->opt_l2_mfma_waves;
    if (!force) waves = 4;  // the fused auto-path kernel is a 256-thread kernel
    else if (waves != 4 && waves != 8) waves = ((long long)((nq_tiles + 7) / 8) * ((nt_tiles + 7) / 8) * batch >= ctx->num_cus) ? 8 : 4;
    const int qblocks = (nq_tiles + waves - 1) / waves;
    // (the fused auto-path kernel carries the exact kernel's registers: two 256-thread workgroups per CU for 128-dim descriptors)
    const int per_cu = ctx->opt_l2_mfma_blocks_per_cu > 0 ? ctx->opt_l2_mfma_blocks_per_cu : (waves == 8 ? 1 : (force ? 4 : 2));
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
    if (ctx->opt_hamming_stamps == 1) {  // the same diagnostics switch as the Hamming kernels (mlpl_debug_hamming_stamps reads them back)
        void *sp = nullptr;
        if ((rc = ws_get(ctx, WS_DEBUG, (size_t)total * 32, &sp))) return rc;
        stamps = (unsigned long long *)sp;
        ctx->dbg_stamp_items = (int)total;
    }
    plan->args = L2MfmaArgs{(const uint4 *)qf, (const int *)qc, (const uint4 *)tf, (const int *)tc, nq, nt, nq_tiles, nt_tiles, tps, nsplit,
                            qblocks, batch, ib, (ulonglong2 *)part, stamps};
    plan->ksel = ksel;
    plan->grid = grid;
    if (!force) return MLPL_OK;
    prof_mark(ctx, MLPL_PROF_KNN_L2, 0, s);
#define MLPL_MFMA_LAUNCH_W(K, WV) hipLaunchKernelGGL((knn_l2_mfma_kernel<K, WV>), dim3(grid), dim3(64 * WV), 0, s, plan->args, gate)
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
    return MLPL_OK;
}

}  // namespace mlpl
