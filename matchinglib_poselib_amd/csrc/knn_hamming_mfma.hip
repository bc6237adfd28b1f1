// knn_hamming_mfma.hip -- exact brute-force 2-NN under bit-Hamming distance on the gfx950 matrix cores.
//
// Same contract as knn_hamming.hip (cvflann::Index<HammingLUT>(LinearIndexParams).knnSearch of
// matchinglib/source/matchers.cpp:567-588: the two lexicographically smallest (distance, trainIdx) pairs per query,
// bit-exact), different arithmetic: every descriptor bit b becomes the fp4 (E2M1) value 2b-1 in {-1,+1}, so that
//     sum_k a'_k b'_k = (#equal bits) - (#different bits) = nbits - 2 * hamming(a, b),
// an all-pairs distance table is a dense +-1 GEMM, and v_mfma_scale_f32_32x32x64_f8f6f4 (fp4 x fp4, unit scales, fp32
// accumulate) produces 32x32 of them per K=64 step.  All sums are small integers: exact in fp32.
//
// The VALU kernels of knn_hamming.hip spend 2*NW+3 issue slots per descriptor pair (76 cycles per 64 pairs at NW = 8,
// measured issue floor); here a pair costs KS/1024 MFMAs plus TWO vector instructions:
//   * queries sit on the lanes (B operand, column = lane & 31), a 32-row train tile is the A operand, so after the K loop
//     a lane holds 16 distances of ITS query to 16 train rows;
//   * the row index rides in the accumulator: the first MFMA of a tile reads C = -(local row) * 2^-14 from 16 constant
//     registers, so v = (nbits - 2 d) - row * eps orders candidates by (distance asc, row asc) as a plain float, and the
//     running top-2 per lane is  m2 = med3(m1, m2, v); m1 = max(m1, v)  -- two VALU ops per pair, no key packing;
//   * between tiles the running pair is re-based by +32 eps (two adds per 512 pairs) instead of re-building the 16
//     constants; values stay exactly representable (|int| <= 512, fraction a multiple of 2^-14 below 1/4);
//   * operands are pre-expanded once per call into MFMA fragment order ([tile][kstep][lane] x 16 B), so a wave fetches a
//     train tile's fragment with KS perfectly coalesced 1 KiB loads straight from L2 (the expanded train set of a C2
//     image is 1 MiB) -- no LDS, no barriers; the query fragments stay in registers for the whole kernel.
// Output: the same [batch][split][nq] uint2 table of packed (dist << dshift | local row) keys the VALU kernels write,
// merged by knn_hamming_merge_kernel.

#include "mlpl_internal.h"


namespace mlpl {

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr float kEps = 1.0f / 16384.0f;  // 2^-14: row weight in the accumulator
constexpr int kMaxRowsPerSplit = 4096;   // keeps the re-based fraction below 1/4

__device__ __forceinline__ float fmax_raw(float a, float b) {
    float r;
    asm("v_max_f32 %0, %1, %2" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float fmed3_raw(float a, float b, float c) {
    float r;
    asm("v_med3_f32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}
__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// 8 bits -> 8 fp4 nibbles (bit i -> nibble i): 1 -> 0x2 (+1.0), 0 -> 0xA (-1.0)
__device__ __forceinline__ uint32_t expand_byte(uint32_t x) {
    uint32_t t = (x | (x << 12)) & 0x000F000Fu;
    t = (t | (t << 6)) & 0x03030303u;
    t = (t | (t << 3)) & 0x11111111u;
    return 0xAAAAAAAAu ^ (t << 3);
}

// Rows of `nw` 32-bit words (word rows as the VALU path uses them) -> fragment order.  Lane (r = l & 31, h = l >> 5) of
// K-step s holds word h * KS + s of row 32 * tile + r, i.e. each lane owns KS CONSECUTIVE words of its row (one 16-byte
// load at KS = 4, and the wave reads one contiguous KiB); which 32 bits go to which K position is free as long as both
// operands use the same rule, and they do -- this kernel expands both.  Rows >= n and words >= nw read as zero bits (the
// same in both operands, so they add nothing to a distance).  `tiles` covers the padded row count.
// One launch: blockIdx.z = 0 queries, 1 train rows; one thread per (tile, lane), KS x 16 bytes out.
struct ExpandArgs {
    const uint32_t *src;
    size_t src_batch_words;
    int n, tiles;
    uint4 *dst;
};

template <int KS>
__global__ __launch_bounds__(256) void hamming_expand_kernel(ExpandArgs qa, ExpandArgs ta, int nw) {
    const ExpandArgs A = blockIdx.z ? ta : qa;
    const int b = blockIdx.y;
    const int i = blockIdx.x * 256 + threadIdx.x;  // tile * 64 + lane
    if (i >= A.tiles * 64) return;
    const int l = i & 63, tile = i >> 6;
    const int row = tile * 32 + (l & 31);
    const int w0 = (l >> 5) * KS;
    uint32_t v[KS];
    const uint32_t *p = A.src + (size_t)b * A.src_batch_words + (size_t)row * nw + w0;
    if (row < A.n && nw == 2 * KS) {  // whole words, KS * 4-byte aligned (rows are nw words): one vector load
        if constexpr (KS == 4) {
            const uint4 x = *reinterpret_cast<const uint4 *>(p);
            v[0] = x.x, v[1] = x.y, v[2] = x.z, v[3] = x.w;
        } else if constexpr (KS == 8) {
            const uint4 x = reinterpret_cast<const uint4 *>(p)[0], y = reinterpret_cast<const uint4 *>(p)[1];
            v[0] = x.x, v[1] = x.y, v[2] = x.z, v[3] = x.w, v[4] = y.x, v[5] = y.y, v[6] = y.z, v[7] = y.w;
        } else if constexpr (KS == 2) {
            const uint2 x = *reinterpret_cast<const uint2 *>(p);
            v[0] = x.x, v[1] = x.y;
        } else {
            v[0] = p[0];
        }
    } else {
#pragma unroll
        for (int s = 0; s < KS; ++s) v[s] = (row < A.n && w0 + s < nw) ? p[s] : 0u;
    }
    uint4 *out = A.dst + ((size_t)b * A.tiles + tile) * KS * 64 + l;
#pragma unroll
    for (int s = 0; s < KS; ++s) {
        typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
        const u32x4 o = {expand_byte(v[s] & 255u), expand_byte((v[s] >> 8) & 255u), expand_byte((v[s] >> 16) & 255u),
                         expand_byte(v[s] >> 24)};
        __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(out + s * 64));
    }
}

__device__ __forceinline__ v16f mfma_fp4(uint4 a, uint4 b, v16f c) {
    const v8i av = {(int)a.x, (int)a.y, (int)a.z, (int)a.w, 0, 0, 0, 0};
    const v8i bv = {(int)b.x, (int)b.y, (int)b.z, (int)b.w, 0, 0, 0, 0};
    // cbsz = blgp = 4: fp4 operands (4 registers each); E8M0 scale 0x7F = 2^0 for both
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}

// KS K-steps of 64 bits, QT query tiles (32 queries each) per wave.  A wave's work item is (batch item, train split, query
// group of QT tiles); the 4 waves of a workgroup are independent (no LDS, no barriers) and take consecutive items, i.e.
// neighbouring query groups of the same train split, so they stream the same train fragments through the CU's L1.
template <int KS, int QT>
__global__ __launch_bounds__(256, (KS >= 8 ? 2 : 3)) void knn_hamming_mfma_kernel(
    const uint4 *__restrict__ qfrag, size_t q_batch_u4, const uint4 *__restrict__ tfrag, size_t t_batch_u4, int nq, int nt,
    int rows_per_split, int nsplit, int dshift, int qgroups, int n_items, uint2 *__restrict__ part) {
    const int l = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    // XCD-aware work assignment.  Workgroups go round-robin to the 8 XCDs (each with its own L2), so workgroup L runs on
    // XCD L % 8.  The work items are ordered (batch item, split, query group) and XCD x takes the x-th contiguous eighth of
    // that order: with 8 image pairs per launch every XCD streams ONE pair's train fragments (1 MiB at C2) through its L2
    // instead of all eight; with one pair an XCD sees an eighth of the train splits.
    const int per_xcd = (int)(gridDim.x >> 3);  // the grid is padded to a multiple of 8
    const int item = ((int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3)) * 4 + w;
    if (item >= n_items) return;  // wave-uniform; the kernel has no barriers
    const int qg = item % qgroups;
    const int split = (item / qgroups) % nsplit;
    const int b = item / (qgroups * nsplit);
    const int qt0 = qg * QT;  // first query tile of this wave (the fragment buffer is padded to whole groups)
    const uint4 *qf = qfrag + (size_t)b * q_batch_u4 + (size_t)qt0 * KS * 64 + l;
    const uint4 *tf = tfrag + (size_t)b * t_batch_u4 + l;

    uint4 bq[QT][KS];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int s = 0; s < KS; ++s) bq[t][s] = qf[(size_t)(t * KS + s) * 64];

    // C of the first MFMA of every tile: minus the local row of accumulator register `reg` in this lane, times eps
    const int h = l >> 5;
    v16f cinit;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) cinit[reg] = -(float)((reg & 3) + 8 * (reg >> 2) + 4 * h) * kEps;

    const int row0 = split * rows_per_split;
    const int row1 = min(nt, row0 + rows_per_split);
    const int tile0 = row0 >> 5;
    const int ntiles = (row1 - row0 + 31) >> 5;  // >= 1: the host never launches an empty split

    float m1[QT], m2[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) m1[t] = m2[t] = -INFINITY;

    // fragment stream of this split; the buffer carries one spare tile, so the prefetch past the last tile needs no clamp
    const uint4 *tp = tf + (size_t)tile0 * KS * 64;
    uint4 fa[KS], fb[KS];
#pragma unroll
    for (int s = 0; s < KS; ++s) fa[s] = tp[s * 64];

    // one train tile against the wave's QT query tiles: KS MFMAs + 22 VALU ops per query tile
    auto tile_body = [&](const uint4 (&a)[KS], const v16f &c0) {
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            v16f acc = mfma_fp4(a[0], bq[t][0], c0);
#pragma unroll
            for (int s = 1; s < KS; ++s) acc = mfma_fp4(a[s], bq[t][s], acc);
            // re-base the running pair to this tile's row origin (exact; -inf stays -inf)
            m1[t] += 32.0f * kEps;
            m2[t] += 32.0f * kEps;
            // four candidates per step, 5 VALU ops: with T = {m1, a, b} the second largest of T + {m2} is max(med3(T), m2)
            // because m2 <= m1; two such pairs share one max3 for m2 (values are distinct, or -inf)
#pragma unroll
            for (int reg = 0; reg < 16; reg += 4) {
                const float s0 = __builtin_amdgcn_fmed3f(m1[t], acc[reg], acc[reg + 1]);
                const float t0 = __builtin_fmaxf(__builtin_fmaxf(m1[t], acc[reg]), acc[reg + 1]);
                const float s1 = __builtin_amdgcn_fmed3f(t0, acc[reg + 2], acc[reg + 3]);
                m1[t] = __builtin_fmaxf(__builtin_fmaxf(t0, acc[reg + 2]), acc[reg + 3]);
                m2[t] = __builtin_fmaxf(__builtin_fmaxf(m2[t], s0), s1);
            }
        }
    };

    // only the last tile of the train set can be ragged; it gets its own C (rows >= nt start at -inf and stay there)
    const bool ragged = row0 + ntiles * 32 > nt;
    const int nfull = ragged ? ntiles - 1 : ntiles;
    int it = 0;
    for (; it + 2 <= nfull; it += 2) {  // ping-pong between the two fragment sets: no register copies in the steady state
#pragma unroll
        for (int s = 0; s < KS; ++s) fb[s] = tp[(KS + s) * 64];
        tile_body(fa, cinit);
        tp += 2 * KS * 64;
#pragma unroll
        for (int s = 0; s < KS; ++s) fa[s] = tp[s * 64];
        tile_body(fb, cinit);
    }
    if (it < nfull) {
#pragma unroll
        for (int s = 0; s < KS; ++s) fb[s] = tp[(KS + s) * 64];
        tile_body(fa, cinit);
#pragma unroll
        for (int s = 0; s < KS; ++s) fa[s] = fb[s];
    }
    if (ragged) {
        const int tile_row0 = row0 + nfull * 32;
        v16f cl;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int lr = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            cl[reg] = (tile_row0 + lr < nt) ? -(float)lr * kEps : -INFINITY;
        }
        tile_body(fa, cl);
    }

    // decode (frame = last tile): v = (64 KS - 2 d) + (32 (ntiles - 1) - local_row) * eps
    const float frame = (float)(32 * (ntiles - 1));
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        uint32_t k[2];
        const float mm[2] = {m1[t], m2[t]};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (mm[j] == -INFINITY) {
                k[j] = 0xFFFFFFFFu;
            } else {
                const float ip = rintf(mm[j]);
                const int d = (64 * KS - (int)ip) >> 1;
                const int lrow = (int)(frame - (mm[j] - ip) * 16384.0f);
                k[j] = ((uint32_t)d << dshift) | (uint32_t)lrow;
            }
        }
        // the two lane halves hold disjoint rows of the same query: top-2 of the four keys
        const uint32_t o0 = __shfl_xor(k[0], 32), o1 = __shfl_xor(k[1], 32);
        uint32_t k0 = k[0], k1 = k[1];
        k1 = umed3(k0, k1, o0);
        k0 = min(k0, o0);
        k1 = umed3(k0, k1, o1);
        k0 = min(k0, o1);
        const int q = (qt0 + t) * 32 + (l & 31);
        if (h == 0 && q < nq) part[((size_t)b * nsplit + split) * nq + q] = make_uint2(k0, k1);
    }
}

template <int KS>
void launch_mfma(int qt, dim3 grid, hipStream_t s, const uint4 *qf, size_t qb, const uint4 *tf, size_t tb, int nq, int nt, int rps,
                 int nsplit, int dshift, int qgroups, int n_items, uint2 *part) {
#define MLPL_MFMA_LAUNCH(QT_)                                                                                                    \
    hipLaunchKernelGGL((knn_hamming_mfma_kernel<KS, QT_>), grid, dim3(256), 0, s, qf, qb, tf, tb, nq, nt, rps, nsplit, dshift, qgroups, \
                       n_items, part)
    if constexpr (KS <= 4) {
        if (qt == 4) {
            MLPL_MFMA_LAUNCH(4);
            return;
        }
    }
    if (qt >= 2)
        MLPL_MFMA_LAUNCH(2);
    else
        MLPL_MFMA_LAUNCH(1);
#undef MLPL_MFMA_LAUNCH
}

}  // namespace

// Called by launch_knn_hamming for descriptors of at most 64 bytes.  qw/tw: word rows (nw words per row, zero padded).
int launch_knn_hamming_mfma(mlpl_ctx *ctx, const uint32_t *qw, size_t q_batch_words, const uint32_t *tw, size_t t_batch_words,
                            int nq, int nt, int nw, int batch, int dshift, hipStream_t s, int *rps_out, int *nsplit_out,
                            uint2 **part_out) {
    int ks = 1;
    while (ks * 2 < nw) ks *= 2;  // 64-bit K-steps: nw <= 2 -> 1, 4 -> 2, 8 -> 4, 16 -> 8
    if (nw > 16) {
        set_error("knn_hamming (mfma): descriptors above 64 bytes take the VALU kernels");
        return MLPL_E_BAD_INPUT;
    }
    // query tiles per wave: fill the chip first, then amortise the train fragment loads over more queries
    const int nqt = (nq + 31) / 32;
    const int max_qt = ks <= 4 ? 4 : 2;
    int qt = max_qt;
    if (ctx->opt_hamming_mfma_qt > 0) qt = std::min(ctx->opt_hamming_mfma_qt, max_qt);
    while (ctx->opt_hamming_mfma_qt <= 0 && qt > 1 && (long long)((nqt + qt - 1) / qt) * batch < (long long)ctx->num_cus) qt >>= 1;
    const int qgroups = (nqt + qt - 1) / qt;  // wave-level work: one group of qt query tiles against one train split
    const int q_tiles_padded = qgroups * qt;
    const int t_tiles = (nt + 31) / 32;

    // train splits: ~4 * opt waves per CU in flight, whole tiles, bounded so that the re-based fraction stays exact
    const long long target_waves = 4LL * std::max(1, ctx->opt_hamming_mfma_blocks_per_cu) * ctx->num_cus;
    long long want = (target_waves + (long long)qgroups * batch - 1) / ((long long)qgroups * batch);
    int nsplit = (int)std::max<long long>(1, std::min<long long>(want, t_tiles));
    int rps = ((nt + nsplit - 1) / nsplit + 31) / 32 * 32;
    rps = std::min(rps, kMaxRowsPerSplit);
    nsplit = (nt + rps - 1) / rps;
    if (nsplit > 65535) {
        set_error("knn_hamming: train set too large (nt=%d)", nt);
        return MLPL_E_BAD_INPUT;
    }
    const long long items = (long long)qgroups * nsplit * batch;
    if (items > (1LL << 30)) {
        set_error("knn_hamming: problem too large for one launch (nq=%d nt=%d batch=%d)", nq, nt, batch);
        return MLPL_E_BAD_INPUT;
    }

    void *qf = nullptr, *tf = nullptr, *part = nullptr;
    int rc;
    const size_t q_u4 = (size_t)q_tiles_padded * ks * 64, t_u4 = (size_t)t_tiles * ks * 64;
    if ((rc = ws_get(ctx, WS_FRAG_Q, (size_t)batch * q_u4 * 16, &qf))) return rc;
    if ((rc = ws_get(ctx, WS_FRAG_T, ((size_t)batch * t_u4 + (size_t)ks * 64) * 16, &tf))) return rc;  // + one spare tile
    if ((rc = ws_get(ctx, WS_PARTIAL, (size_t)batch * nsplit * nq * sizeof(uint2), &part))) return rc;
    const ExpandArgs qa{qw, q_batch_words, nq, q_tiles_padded, (uint4 *)qf}, ta{tw, t_batch_words, nt, t_tiles, (uint4 *)tf};
    const dim3 egrid((unsigned)((std::max(q_tiles_padded, t_tiles) * 64 + 255) / 256), batch, 2);
    switch (ks) {
        case 1: hipLaunchKernelGGL(hamming_expand_kernel<1>, egrid, dim3(256), 0, s, qa, ta, nw); break;
        case 2: hipLaunchKernelGGL(hamming_expand_kernel<2>, egrid, dim3(256), 0, s, qa, ta, nw); break;
        case 4: hipLaunchKernelGGL(hamming_expand_kernel<4>, egrid, dim3(256), 0, s, qa, ta, nw); break;
        default: hipLaunchKernelGGL(hamming_expand_kernel<8>, egrid, dim3(256), 0, s, qa, ta, nw); break;
    }
    // 1-D grid, remapped in the kernel (XCD-aware); padded so that every XCD gets the same number of workgroups
    const long long blocks = (items + 3) / 4;
    dim3 grid((unsigned)((blocks + 7) / 8 * 8));
    prof_mark(ctx, MLPL_PROF_KNN_HAMMING, 0, s);
    switch (ks) {
        case 1: launch_mfma<1>(qt, grid, s, (const uint4 *)qf, q_u4, (const uint4 *)tf, t_u4, nq, nt, rps, nsplit, dshift, qgroups, (int)items, (uint2 *)part); break;
        case 2: launch_mfma<2>(qt, grid, s, (const uint4 *)qf, q_u4, (const uint4 *)tf, t_u4, nq, nt, rps, nsplit, dshift, qgroups, (int)items, (uint2 *)part); break;
        case 4: launch_mfma<4>(qt, grid, s, (const uint4 *)qf, q_u4, (const uint4 *)tf, t_u4, nq, nt, rps, nsplit, dshift, qgroups, (int)items, (uint2 *)part); break;
        default: launch_mfma<8>(qt, grid, s, (const uint4 *)qf, q_u4, (const uint4 *)tf, t_u4, nq, nt, rps, nsplit, dshift, qgroups, (int)items, (uint2 *)part); break;
    }
    prof_mark(ctx, MLPL_PROF_KNN_HAMMING, 1, s);
    *rps_out = rps;
    *nsplit_out = nsplit;
    *part_out = (uint2 *)part;
    return MLPL_OK;
}

}  // namespace mlpl
