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

#include <algorithm>
#include <cmath>
#include <vector>

#include "mlpl_internal.h"


namespace mlpl {

namespace {

typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

constexpr float kEps = 1.0f / 16384.0f;  // 2^-14: row weight in the accumulator
// Rows of one split.  A candidate's value is I - (row - frame) * 2^-14 with I = nbits - 2 d an integer of magnitude <= 512 and the frame the
// first row of the tile being folded, so over a split of R rows the fraction lies in (-32 eps, R eps): R <= 8192 keeps it inside
// (-1/2, 1/2) -- rint() recovers I -- and a multiple of 2^-14 below 512 is exact in fp32.  (Up to round 4 the cap was 4096, which cut every
// 8192-row train set of a full batch in two although the chip was already full: partials, tickets and a fold for nothing.)
constexpr int kMaxRowsPerSplit = 8192;

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

// The {0, +1} form of the STREAMED (train) operand (option hamming_train01): bit 1 -> 0x2 (+1.0), bit 0 -> 0x0 (+0.0).  With the query
// operand still +-1,  sum_k t_k q'_k = pop(t & q) - pop(t & ~q) = pop(q) - hamming(t, q):  for a fixed query (a lane's column) the
// accumulator still orders the train rows by distance, all values are exact integers in [-256, 256], and half of the A operand's nibbles
// are zero (fewer toggling multiplier inputs: the sustained matrix-core clock is data dependent).  The decode adds pop(q).
__device__ __forceinline__ uint32_t expand_byte01(uint32_t x) {
    uint32_t t = (x | (x << 12)) & 0x000F000Fu;
    t = (t | (t << 6)) & 0x03030303u;
    t = (t | (t << 3)) & 0x11111111u;
    return t << 1;
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
__global__ __launch_bounds__(256) void hamming_expand_kernel(ExpandArgs qa, ExpandArgs ta, int nw, int *__restrict__ counters,
                                                             int counters_per_batch, int train01) {
    const ExpandArgs A = blockIdx.z ? ta : qa;
    const int b = blockIdx.y;
    if (counters && blockIdx.x == 0 && blockIdx.z == 0)  // chunk counters of the dynamic-split kernel that follows in the stream
        for (int i = threadIdx.x; i < counters_per_batch; i += 256) counters[(size_t)b * counters_per_batch + i] = 0;
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
        u32x4 o = {expand_byte(v[s] & 255u), expand_byte((v[s] >> 8) & 255u), expand_byte((v[s] >> 16) & 255u),
                   expand_byte(v[s] >> 24)};
        if (train01)  // (only ever set for the train operand of the static LDS-ring kernel, which expands its queries itself)
            o = u32x4{expand_byte01(v[s] & 255u), expand_byte01((v[s] >> 8) & 255u), expand_byte01((v[s] >> 16) & 255u),
                      expand_byte01(v[s] >> 24)};
        __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(out + s * 64));
    }
}

// The same expansion of a TRAIN set of 32-byte descriptors with one thread per (tile, K-step, lane) instead of per (tile, lane): for ONE
// image pair hamming_expand_kernel<4> is 64 workgroups with 16 expansions per thread -- a quarter of the chip, one long thread each; this
// form is one 256-thread workgroup per tile (256 for 8192 rows).  Same bytes at the same addresses.  (Round 6, the latency shape.)
__global__ __launch_bounds__(256) void hamming_expand_fine_kernel(ExpandArgs A, int train01) {
    constexpr int KS = 4;
    const int b = blockIdx.y;
    const int tile = blockIdx.x;   // gridDim.x == A.tiles
    const int l = threadIdx.x & 63, st = threadIdx.x >> 6;
    const int row = tile * 32 + (l & 31);
    const int w = (l >> 5) * KS + st;
    const uint32_t v = row < A.n ? A.src[(size_t)b * A.src_batch_words + (size_t)row * (2 * KS) + w] : 0u;
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    u32x4 o = {expand_byte(v & 255u), expand_byte((v >> 8) & 255u), expand_byte((v >> 16) & 255u), expand_byte(v >> 24)};
    if (train01) o = u32x4{expand_byte01(v & 255u), expand_byte01((v >> 8) & 255u), expand_byte01((v >> 16) & 255u), expand_byte01(v >> 24)};
    __builtin_nontemporal_store(o, reinterpret_cast<u32x4 *>(A.dst + ((size_t)b * A.tiles + tile) * KS * 64 + st * 64 + l));
}

// Round 5: the UNSCALED form.  v_mfma_scale_f32_32x32x64_f8f6f4 is a two-part instruction (a scale-load prefix + the MFMA) that reads two
// more registers; with unit scales it computes what v_mfma_f32_32x32x64_f8f6f4 computes without them.  The compiler selects the unscaled
// opcode when both scale operands are the constant 0 (its encoding of "no scales").  Rounds 1-4 issued the scaled form with E8M0 0x7F
// (2^0) in a register: 35 cycles per MFMA in the probes against the nominal 32.  MLPL_MFMA_SCALED=1 rebuilds the old form for the A/B.
#ifndef MLPL_MFMA_SCALED
#define MLPL_MFMA_SCALED 0
#endif
__device__ __forceinline__ v16f mfma_fp4(uint4 a, uint4 b, v16f c) {
    const v8i av = {(int)a.x, (int)a.y, (int)a.z, (int)a.w, 0, 0, 0, 0};
    const v8i bv = {(int)b.x, (int)b.y, (int)b.z, (int)b.w, 0, 0, 0, 0};
    // cbsz = blgp = 4: fp4 operands (4 registers each)
#if MLPL_MFMA_SCALED
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);  // E8M0 scale 0x7F = 2^0 for both
#else
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 4, 4, 0, 0, 0, 0);
#endif
}

// KS K-steps of 64 bits, QT query tiles (32 queries each) per wave.  A wave's work item is (batch item, train split, query
// group of QT tiles); the 4 waves of a workgroup are independent (no LDS, no barriers) and take consecutive items, i.e.
// neighbouring query groups of the same train split, so they stream the same train fragments through the CU's L1.
template <int KS, int QT>
__global__ __launch_bounds__(256, (KS >= 8 ? 2 : 3)) void knn_hamming_mfma_kernel(
    const uint4 *__restrict__ qfrag, size_t q_batch_u4, const uint4 *__restrict__ tfrag, size_t t_batch_u4, int nq, int nt,
    int rows_per_split, int nsplit, int dshift, int qgroups, int n_items, uint2 *__restrict__ part,
    unsigned long long *__restrict__ stamps) {
    const int l = threadIdx.x & 63;
    const int w = threadIdx.x >> 6;
    // diagnostics only (stamps != nullptr, "hamming_stamps" option): shader-clock and 100 MHz real-time deltas per wave
    unsigned long long st_c = 0, st_r = 0;
    if (stamps) {
        st_c = __builtin_amdgcn_s_memtime();
        st_r = __builtin_amdgcn_s_memrealtime();
    }
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
    if (stamps && l == 0) {
        unsigned long long *o = stamps + (size_t)item * 4;
        o[0] = __builtin_amdgcn_s_memtime() - st_c;
        o[1] = __builtin_amdgcn_s_memrealtime() - st_r;
        // HW_REG_HW_ID (4): wave/simd/cu/sh/se ids; HW_REG_XCC_ID (20): the XCD
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
        o[2] = (unsigned long long)ntiles * QT | ((unsigned long long)(hw & 0xFFFFFu) << 32) | ((unsigned long long)(xcc & 15u) << 56);
        o[3] = st_r;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// KS = 4 (descriptors of 17..32 bytes: ORB-256) with the train tiles SHARED by the 4 waves of a workgroup through an LDS ring.
// Why (tools/hamming_unit_probe3.hip, in-kernel clock stamps): a 1 KiB global_load_dwordx4 occupies the CU's one vector-memory/L1 path
// (64 B/clk) for 16 cycles; with every wave fetching its own 4 KiB per tile the four SIMDs of a CU need 64 of the ~130 cycles a unit
// costs them, and the fragment double buffer (32 VGPRs) holds the kernel at 3 waves per SIMD, where a wave's ~44-cycle MFMA issue
// interval is not hidden (178-193 cycles per unit per SIMD measured; MFMA and VALU do not overlap on a SIMD, they add: ~90 + ~40).
// Here wave w copies K-step w of every tile straight into LDS (global_load_lds_dwordx4: no staging registers, a quarter of the L1
// traffic), all four read the tile back with ds_read_b128 (LDS: 256 B/clk), and 120 VGPRs give 4 waves per SIMD.
// Ring protocol (NB = 4 slots of 4 KiB, prefetch distance D = 2, one s_barrier per tile): in iteration `it` a wave issues the copy of
// tile it + 2, waits for its own piece of tile `it` (vmcnt counts LDS-DMA in issue order), and the barrier makes all four pieces
// visible.  Slot (it + 2) % 4 was last read in iteration it - 2, and a wave that has passed the barrier of iteration it - 1 knows that
// every wave has finished iteration it - 2 (its MFMAs consumed those reads), so the copy cannot overtake a reader.
// ---------------------------------------------------------------------------------------------------------------------------------
// Fused epilogue (`fuse.idx` != nullptr; QT = 2 or 4): the merge of the splits, the ratio predicate and the per-64-query pass counts -- the
// work of knn_hamming_merge_kernel -- happen HERE.  One split: straight from the registers that hold the final top-2 (no partial table at
// all).  Several splits: every workgroup writes its partial top-2 as before, then takes a ticket of its (image pair, query block); the
// workgroup that draws the last ticket folds the splits' partials of its 128 * QT queries (64-bit (distance, global row) keys, as the
// merge kernel) and resets the ticket counter.  Same outputs as the two-kernel path bit for bit (min over keys is order-independent).
struct HammingFuse {
    int32_t *idx, *dist, *group_counts;
    int *tickets;   // [batch][qblocks], zero before the launch; left zero by the last workgroup
    int k;
    float ratio;
    int train01;    // the train fragments are {0, +1} (hamming_expand_kernel train01): accumulator = pop(query) - distance
    // Clock record of this launch (option hamming_stamps = 2; nullptr otherwise): thread 0 of work item 0 leaves {shader-clock cycles,
    // 100 MHz ticks, start tick, launch number} of its workgroup's lifetime -- one 32-byte store per launch, the cost of the facility.
    unsigned long long *clk;
    unsigned long long launch_no;
};

// NW (round 5) = waves per workgroup, 4 or 8: waves 0..3 copy one K-step of every tile each, ALL NW read it -- with 8 waves a tile is
// fetched and a barrier passed once per eight units' worth of query tiles instead of four (tools/hamming_unit_probe3.hip `ringnw`: 167.9 ->
// 159.1 cycles per unit per SIMD).  A workgroup then serves NW * QT query tiles of one (image pair, train split).
// PD (round 5) = prefetch distance in tiles (2, 4 or 6): the copy of tile it + PD is issued in iteration it; the ring has NB >= PD + 2 slots.
template <int QT, int PRIO, int NW = 4, int PD = 2>
__global__ __launch_bounds__(64 * NW, 4) void knn_hamming_mfma_lds_kernel(
    const uint32_t *__restrict__ qw, size_t q_batch_words, const uint4 *__restrict__ tfrag, size_t t_batch_u4, int nq, int nt,
    int rows_per_split, int nsplit, int dshift, int qblocks, int n_items, uint2 *__restrict__ part,
    unsigned long long *__restrict__ stamps, const int32_t *__restrict__ split_tile0, HammingFuse fuse) {
    constexpr int KS = 4, NB = PD == 2 ? 4 : 8;
    static_assert(PD == 2 || PD == 4 || PD == 6, "prefetch distance");
    __shared__ __attribute__((aligned(16))) uint4 ring[NB][KS * 64];
    const int l = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned long long st_c = 0, st_r = 0;
    if (stamps || fuse.clk) {
        st_c = __builtin_amdgcn_s_memtime();
        st_r = __builtin_amdgcn_s_memrealtime();
    }
    // XCD-aware, as above, at workgroup granularity: item = (batch item, train split, block of 4 query groups)
    const int per_xcd = (int)(gridDim.x >> 3);
    const int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (item >= n_items) return;  // workgroup-uniform: no wave of this workgroup reaches a barrier
    const int qb = item % qblocks;
    const int split = (item / qblocks) % nsplit;
    const int b = item / (qblocks * nsplit);
    const int qt0 = (qb * NW + w) * QT;
    const int h = l >> 5;

    uint4 bq[QT][KS];   // the wave's query fragments (filled below, behind the first tile copies)
    v16f cinit;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) cinit[reg] = -(float)((reg & 3) + 8 * (reg >> 2) + 4 * h) * kEps;

    // split geometry: equal splits, or the age-aware table of the launcher (tiles [tab[s], tab[s + 1]) per image pair)
    int row0 = split * rows_per_split;
    int row1 = min(nt, row0 + rows_per_split);
    if (split_tile0) {
        row0 = 32 * split_tile0[(size_t)b * (nsplit + 1) + split];
        row1 = min(nt, 32 * split_tile0[(size_t)b * (nsplit + 1) + split + 1]);
    }
    const int tile0 = row0 >> 5;
    const int ntiles = (row1 - row0 + 31) >> 5;

    float m1[QT], m2[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) m1[t] = m2[t] = -INFINITY;

    // my piece (K-step w) of the split's first tile; a tile is KS * 64 uint4 = 4 KiB, contiguous
    const uint4 *tbase = tfrag + (size_t)b * t_batch_u4 + (size_t)tile0 * KS * 64 + (size_t)w * 64;  // wave-uniform
    if (PRIO == 2) tbase = tfrag + (size_t)w * 64;  // diagnostics: every workgroup streams the SAME tiles (wrong results, L1/L2-hot source)
    auto copy_tile = [&](int t_rel) {
        if (NW > KS && w >= KS) return;  // (wave-uniform: the K-steps are copied by the first KS waves)
        __builtin_amdgcn_global_load_lds((const void *)(tbase + (size_t)t_rel * KS * 64 + l),
                                         (__attribute__((address_space(3))) void *)&ring[t_rel & (NB - 1)][w * 64], 16, 0, 0);
    };
    // The tile is read back with hand-written ds_read_b128: for a compiler-visible LDS load the waitcnt pass would first drain EVERY
    // outstanding LDS-DMA (vmcnt(0): it cannot know that the copies in flight target other ring slots), which serialises the prefetch.
    // The reads return in order, so K-step s is complete once at most 3 - s of them are outstanding.
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4 *)&ring[0][0] + (uint32_t)l * 16u;
    auto tile_body = [&](int slot, const v16f &c0) {
        u32x4 r[KS];
        const uint32_t addr = ring_lds + (uint32_t)slot * (KS * 1024u);
        asm volatile(
            "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
            : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
            : "v"(addr)
            : "memory");
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(r[0]));
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(r[1]));
        asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(r[2]));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[3]));
        uint4 a[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) a[s] = make_uint4(r[s].x, r[s].y, r[s].z, r[s].w);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            v16f acc = mfma_fp4(a[0], bq[t][0], c0);
#pragma unroll
            for (int s = 1; s < KS; ++s) acc = mfma_fp4(a[s], bq[t][s], acc);
            m1[t] += 32.0f * kEps;
            m2[t] += 32.0f * kEps;
#pragma unroll
            for (int reg = 0; reg < 16; reg += 4) {
                if constexpr (PRIO == 3) {
                    // Experiment (round 6; option hamming_mfma_prio = 3): wave-uniform skip.  A group of four rows changes a lane's
                    // running pair only if its largest value beats the lane's m2; after the first eighth of a split fewer than half of the
                    // groups do so in ANY lane.  3 ops (max3, max, compare) + a scalar branch instead of the 5-op update when none does.
                    // MEASURED NEGATIVE (gpurun_out/r6/hamming_skip_ab*.log, profiles/README.md): same outputs, the shader clock RISES from
                    // 1.84-1.97 to 2.06-2.17 GHz (less switching per cycle: the launch is power-limited), yet the kernel takes 0.376-0.383 ms
                    // against 0.363-0.379 -- the four VCC round trips per unit cost more cycles than the skipped updates save.  Off.
                    const float g = __builtin_fmaxf(__builtin_fmaxf(__builtin_fmaxf(acc[reg], acc[reg + 1]), acc[reg + 2]), acc[reg + 3]);
                    if (__builtin_amdgcn_ballot_w64(g > m2[t]) == 0) continue;
                }
                const float s0 = __builtin_amdgcn_fmed3f(m1[t], acc[reg], acc[reg + 1]);
                const float t0 = __builtin_fmaxf(__builtin_fmaxf(m1[t], acc[reg]), acc[reg + 1]);
                const float s1 = __builtin_amdgcn_fmed3f(t0, acc[reg + 2], acc[reg + 3]);
                m1[t] = __builtin_fmaxf(__builtin_fmaxf(t0, acc[reg + 2]), acc[reg + 3]);
                m2[t] = __builtin_fmaxf(__builtin_fmaxf(m2[t], s0), s1);
            }
        }
    };

#pragma unroll
    for (int t = 0; t < PD; ++t)
        if (t < ntiles) copy_tile(t);
    // (round 5: the first two tile copies are in flight BEFORE the query rows are fetched and expanded -- the two global latencies of a
    // workgroup's prologue overlap instead of adding; vmcnt retires in issue order, so the query loads' own wait also covers these copies,
    // which the first arrive() would wait for anyway)
    // The query operand is expanded to fp4 HERE, once per wave, from the raw 32-byte descriptors (lane (r, h) owns words 4h..4h+3 of its
    // row: one 16-byte load per tile; the same word -> K-position rule as hamming_expand_kernel, which now runs for the train set only --
    // that one has to exist in fragment order in memory because it is streamed into LDS by DMA).  Rows >= nq read as zero bits.
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        const int row = (qt0 + t) * 32 + (l & 31);
        uint4 v = make_uint4(0u, 0u, 0u, 0u);
        if (row < nq) v = *reinterpret_cast<const uint4 *>(qw + (size_t)b * q_batch_words + (size_t)row * (2 * KS) + (size_t)h * KS);
        const uint32_t vs[KS] = {v.x, v.y, v.z, v.w};
#pragma unroll
        for (int s = 0; s < KS; ++s)
            bq[t][s] = make_uint4(expand_byte(vs[s] & 255u), expand_byte((vs[s] >> 8) & 255u), expand_byte((vs[s] >> 16) & 255u),
                                  expand_byte(vs[s] >> 24));
    }

    // only the last tile of the train set can be ragged; it runs after the loop with its own C (rows >= nt start at -inf and stay there)
    const bool ragged = row0 + ntiles * 32 > nt;
    const int nfull = ragged ? ntiles - 1 : ntiles;
    // own piece of tile `it` landed <=> at most the copies of the tiles after it are still in flight (LDS-DMA retires in issue order)
    // (slot (it + PD) % NB was last read in iteration it + PD - NB <= it - 2, and a wave that has passed the barrier of iteration it - 1 knows
    // that every wave has finished iteration it - 2)
    auto arrive = [&](int it) {
        if (it + PD < ntiles) {
            copy_tile(it + PD);
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(PD) : "memory");
        } else {
            const int ahead = ntiles - 1 - it;  // copies still allowed in flight: those of the tiles behind this one (< PD)
            if (PD > 4 && ahead == 5) asm volatile("s_waitcnt vmcnt(5)" ::: "memory");
            else if (PD > 4 && ahead == 4) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else if (PD > 2 && ahead == 3) asm volatile("s_waitcnt vmcnt(3)" ::: "memory");
            else if (PD > 2 && ahead == 2) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (ahead == 1) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        __builtin_amdgcn_s_barrier();
    };
    // Optional (PRIO, off by default): the hardware arbitrates oldest-first among equal priorities, so with identical work items the four
    // resident waves of a SIMD finish one after the other (28 -> 50 us measured).  Rotating s_setprio on a clock slice (2048 shader
    // cycles ~ 1 us) by (time + wave slot) makes them finish together (38 -> 50 us) -- but the launch is no shorter: the SIMD's
    // throughput is the same either way.  Kept as a measured negative result and a diagnostic.  (Round 5, eight-wave workgroups, one split per
    // 8192-row train set, i.e. exactly two workgroups resident per CU for the whole launch: unrotated the older workgroup finishes at 468k
    // cycles and the younger at 697k; rotated both finish at ~740k -- 6 % MORE cycles.  Strict oldest-first is the better schedule.
    // Priority by a workgroup's OWN progress instead -- high and low alternating every 16 / 32 / 64 tiles, one s_setprio per segment, no clock
    // read -- lets the younger workgroup catch up at every boundary: the older one then finishes at 288-306 us instead of 247 and the launch
    // takes 375.2-376.6 us against 375.7 (gpurun_out/r5/prio_progress_ab.log).  So the half-occupied last third is NOT slower per unit: the
    // SIMD delivers the same units per cycle with two resident waves as with four, and the staggered finish costs nothing.)
    const int wslot = (int)(__builtin_amdgcn_s_getreg(4 | (0 << 6) | (3 << 11)) & 3u);  // HW_ID.wave_id: slot within the SIMD
    auto rotate_prio = [&]() {
        if (PRIO != 1) return;
        const int p = (int)(((unsigned)(__builtin_amdgcn_s_memtime() >> 11) + (unsigned)wslot) & 3u);
        if (p == 0) __builtin_amdgcn_s_setprio(0);
        else if (p == 1) __builtin_amdgcn_s_setprio(1);
        else if (p == 2) __builtin_amdgcn_s_setprio(2);
        else __builtin_amdgcn_s_setprio(3);
    };
    for (int it = 0; it < nfull; ++it) {
        rotate_prio();
        arrive(it);
        if (stamps && l == 0 && it < 48) stamps[(size_t)n_items * NW * 4 + ((size_t)item * NW + w) * 48 + it] = __builtin_amdgcn_s_memtime();
        tile_body(it & (NB - 1), cinit);
    }
    if (ragged) {
        arrive(nfull);
        const int tile_row0 = row0 + nfull * 32;
#pragma unroll
        for (int reg = 0; reg < 16; ++reg) {
            const int lr = (reg & 3) + 8 * (reg >> 2) + 4 * h;
            cinit[reg] = (tile_row0 + lr < nt) ? -(float)lr * kEps : -INFINITY;
        }
        tile_body(nfull & (NB - 1), cinit);
    }

    // (the epilogue's addresses are formed from copies of the lane and tile indices the compiler cannot see through: formed in the prologue
    // they would have to live -- as spills: 128 registers are all in use -- across the main loop)
    int le = l, qt0e = qt0;
    asm volatile("" : "+v"(le), "+s"(qt0e));
    const int he = le >> 5;
    const float frame = (float)(32 * (ntiles - 1));
    int pass_acc = 0;
#pragma unroll
    for (int t = 0; t < QT; ++t) {
        // {0, +1} train operand (fuse.train01): the accumulator is pop(query) - distance.  The query's popcount is formed HERE, from the raw
        // words once more (both lane halves' shares of the row), so that the option costs the main loop no registers.
        int qpop = 0;
        if (fuse.train01) {
            const int row = (qt0e + t) * 32 + (le & 31);
            uint4 v = make_uint4(0u, 0u, 0u, 0u);
            if (row < nq) v = *reinterpret_cast<const uint4 *>(qw + (size_t)b * q_batch_words + (size_t)row * (2 * KS) + (size_t)he * KS);
            const int pc = __popc(v.x) + __popc(v.y) + __popc(v.z) + __popc(v.w);
            qpop = pc + __shfl_xor(pc, 32);
        }
        uint32_t k[2];
        const float mm[2] = {m1[t], m2[t]};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (mm[j] == -INFINITY) {
                k[j] = 0xFFFFFFFFu;
            } else {
                const float ip = rintf(mm[j]);
                const int d = fuse.train01 ? qpop - (int)ip : (64 * KS - (int)ip) >> 1;
                const int lrow = (int)(frame - (mm[j] - ip) * 16384.0f);
                k[j] = ((uint32_t)d << dshift) | (uint32_t)lrow;
            }
        }
        const uint32_t o0 = __shfl_xor(k[0], 32), o1 = __shfl_xor(k[1], 32);
        uint32_t k0 = k[0], k1 = k[1];
        k1 = umed3(k0, k1, o0);
        k0 = min(k0, o0);
        k1 = umed3(k0, k1, o1);
        k0 = min(k0, o1);
        const int q = (qt0e + t) * 32 + (le & 31);
        if (fuse.idx && nsplit == 1) {  // the final top-2 of the query: outputs straight from the registers
            const uint32_t lmask = (1u << dshift) - 1u;
            bool pass = false;
            if (he == 0 && q < nq) {
                const size_t o = ((size_t)b * nq + q) * fuse.k;
                const int d0 = k0 == 0xFFFFFFFFu ? -1 : (int)(k0 >> dshift);
                fuse.idx[o] = k0 == 0xFFFFFFFFu ? -1 : (int)(row0 + (k0 & lmask));
                fuse.dist[o] = d0;
                if (fuse.k == 2) {
                    const int d1 = k1 == 0xFFFFFFFFu ? -1 : (int)(k1 >> dshift);
                    fuse.idx[o + 1] = k1 == 0xFFFFFFFFu ? -1 : (int)(row0 + (k1 & lmask));
                    fuse.dist[o + 1] = d1;
                    pass = (float)d0 < __fmul_rn(fuse.ratio, (float)d1);
                } else {
                    pass = true;
                }
            }
            const int c = __popcll(__ballot(pass));
            pass_acc = (t & 1) ? pass_acc + c : c;   // a group of 64 queries = tiles (t even, t + 1) of this wave (qt0 is a multiple of QT)
            if (fuse.group_counts && (t & 1) && le == 0 && (qt0e + t - 1) * 32 < nq) fuse.group_counts[(size_t)b * ((nq + 63) >> 6) + ((qt0e + t) >> 1)] = pass_acc;
        } else if (he == 0 && q < nq) {
            if (fuse.idx)  // read back by another workgroup of THIS launch: written through to the coherence point (see below)
                __hip_atomic_store(reinterpret_cast<unsigned long long *>(&part[((size_t)b * nsplit + split) * nq + q]),
                                   (unsigned long long)k0 | ((unsigned long long)k1 << 32), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else
                part[((size_t)b * nsplit + split) * nq + q] = make_uint2(k0, k1);
        }
    }
    if (fuse.idx && nsplit > 1) {
        // The last workgroup of this (pair, query block) folds the splits.  No fences: an agent-scope release / acquire pair writes back
        // and INVALIDATES the XCD's L2 -- with a workgroup finishing every few microseconds the train fragments the other workgroups
        // stream would never stay cached (measured: 431 -> 633 us per 64-pair launch).  Instead the partials themselves are relaxed
        // agent-scope atomics (written through / read from the coherence point), every wave waits for its own stores to be acknowledged
        // (vmcnt(0)) before the workgroup's barrier, and only then does thread 0 draw the ticket.
        __shared__ int s_last;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (threadIdx.x == 0) {
            const int old = __hip_atomic_fetch_add(&fuse.tickets[(size_t)b * qblocks + qb], 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            s_last = old == nsplit - 1;
            if (s_last) __hip_atomic_store(&fuse.tickets[(size_t)b * qblocks + qb], 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (s_last) {
            const uint32_t lmask = (1u << dshift) - 1u;
            const int q_first = qb * NW * QT * 32;
#pragma unroll 1
            for (int rr = 0; rr * (64 * NW) < NW * QT * 32; ++rr) {  // (QT = 1: the first half of the threads)
                const int ql = rr * (64 * NW) + (int)threadIdx.x;
                const int q = ql < NW * QT * 32 ? q_first + ql : nq;
                unsigned long long b0 = ~0ull, b1 = ~0ull;
                auto upd = [&](unsigned long long g) {
                    const bool lt0 = g < b0, lt1 = g < b1;
                    b1 = lt0 ? b0 : (lt1 ? g : b1);
                    b0 = lt0 ? g : b0;
                };
                bool pass = false;
                if (q < nq) {
                    for (int sp = 0; sp < nsplit; ++sp) {
                        const unsigned long long pw = __hip_atomic_load(reinterpret_cast<const unsigned long long *>(&part[((size_t)b * nsplit + sp) * nq + q]),
                                                                        __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        const uint2 pv = make_uint2((uint32_t)pw, (uint32_t)(pw >> 32));
                        const unsigned long long base = split_tile0 ? 32ull * (unsigned long long)split_tile0[(size_t)b * (nsplit + 1) + sp]
                                                                    : (unsigned long long)sp * rows_per_split;
                        if (pv.x != 0xFFFFFFFFu) upd(((unsigned long long)(pv.x >> dshift) << 32) | (base + (pv.x & lmask)));
                        if (pv.y != 0xFFFFFFFFu) upd(((unsigned long long)(pv.y >> dshift) << 32) | (base + (pv.y & lmask)));
                    }
                    const size_t o = ((size_t)b * nq + q) * fuse.k;
                    const int d0 = (int32_t)(b0 >> 32);
                    fuse.idx[o] = (int32_t)(b0 & 0xFFFFFFFFull);
                    fuse.dist[o] = d0;
                    if (fuse.k == 2) {
                        const int d1 = (int32_t)(b1 >> 32);
                        fuse.idx[o + 1] = (int32_t)(b1 & 0xFFFFFFFFull);
                        fuse.dist[o + 1] = d1;
                        pass = (float)d0 < __fmul_rn(fuse.ratio, (float)d1);
                    } else {
                        pass = true;
                    }
                }
                const int c = __popcll(__ballot(pass));
                const int q_wave = q_first + rr * (64 * NW) + (int)(threadIdx.x & ~63u);   // first query of this wave's 64
                if (fuse.group_counts && le == 0 && q_wave < nq && rr * (64 * NW) + (int)(threadIdx.x & ~63u) < NW * QT * 32) fuse.group_counts[(size_t)b * ((nq + 63) >> 6) + (q_wave >> 6)] = c;
            }
        }
    }
    if (stamps && l == 0) {
        unsigned long long *o = stamps + ((size_t)item * NW + w) * 4;
        o[0] = __builtin_amdgcn_s_memtime() - st_c;
        o[1] = __builtin_amdgcn_s_memrealtime() - st_r;
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
        o[2] = (unsigned long long)ntiles * QT | ((unsigned long long)(hw & 0xFFFFFu) << 32) | ((unsigned long long)(xcc & 15u) << 56);
        o[3] = st_r;
    }
    if (fuse.clk && item == 0 && threadIdx.x == 0) {
        fuse.clk[0] = __builtin_amdgcn_s_memtime() - st_c;
        fuse.clk[1] = __builtin_amdgcn_s_memrealtime() - st_r;
        fuse.clk[2] = st_r;
        fuse.clk[3] = fuse.launch_no;
    }
}

// ---------------------------------------------------------------------------------------------------------------------------------
// The LDS-ring kernel with DYNAMIC train splits.  With fixed, equal work items the hardware's oldest-first arbitration lets the four
// workgroups of a CU finish one after the other (28 ... 50 us for identical items) and the CU runs the second half of the launch
// under-occupied.  Here the workgroups that serve one (image pair, block of 4 query groups, row span) form a TEAM and draw chunks of
// `chunk_tiles` (>= 2) train tiles from the team's counter: a fast workgroup simply takes more chunks, and all of them run dry
// within a chunk of each other.  A workgroup still writes ONE partial top-2 per query (slot = its member index), so the partial
// table and the merge are unchanged; the rows it has seen are not contiguous, so the running pair is re-based by the actual tile
// distance and local rows are relative to the span (<= 8192 rows: (row offset) * 2^-14 < 1/2 keeps rint() exact, see the decode).
// Ring: NB = 3 slots, prefetch distance 1 (the copy of the next tile runs under the current tile's ~2400 cycles).  The chunk after
// the current one is requested (one returning atomic by lane 0 of wave 0, issued BEFORE this iteration's copy so that the iteration's
// own vmcnt wait covers it) at the first tile of every chunk and published to the other waves through LDS behind the barrier.
// The ragged last tile of the train set is the last tile of the last chunk, hence always the last tile of its workgroup.
// ---------------------------------------------------------------------------------------------------------------------------------
template <int QT>
__global__ __launch_bounds__(256, 4) void knn_hamming_mfma_dyn_kernel(
    const uint4 *__restrict__ qfrag, size_t q_batch_u4, const uint4 *__restrict__ tfrag, size_t t_batch_u4, int nq, int nt,
    int span_tiles, int nspan, int nmem, int chunk_tiles, int dshift, int qblocks, int n_items, uint2 *__restrict__ part,
    int *__restrict__ counters, unsigned long long *__restrict__ stamps) {
    constexpr int KS = 4, NB = 3;
    __shared__ __attribute__((aligned(16))) uint4 ring[NB][KS * 64];
    __shared__ int s_claim, s_first;
    const int l = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    unsigned long long st_c = 0, st_r = 0;
    if (stamps) {
        st_c = __builtin_amdgcn_s_memtime();
        st_r = __builtin_amdgcn_s_memrealtime();
    }
    const int per_xcd = (int)(gridDim.x >> 3);
    const int item = (int)(blockIdx.x & 7) * per_xcd + (int)(blockIdx.x >> 3);
    if (item >= n_items) return;  // workgroup-uniform
    const int slots = nspan * nmem;
    const int qb = item % qblocks;
    const int slot_id = (item / qblocks) % slots;
    const int b = item / (qblocks * slots);
    const int span = slot_id / nmem;
    int *ctr = counters + ((size_t)b * qblocks + qb) * nspan + span;
    const int qt0 = (qb * 4 + w) * QT;
    const uint4 *qf = qfrag + (size_t)b * q_batch_u4 + (size_t)qt0 * KS * 64 + l;

    // first chunk (the latency of this one atomic hides under the query fragment loads)
    if (threadIdx.x == 0) s_first = atomicAdd(ctr, 1);

    uint4 bq[QT][KS];
#pragma unroll
    for (int t = 0; t < QT; ++t)
#pragma unroll
        for (int s = 0; s < KS; ++s) bq[t][s] = qf[(size_t)(t * KS + s) * 64];

    const int h = l >> 5;
    v16f cinit;
#pragma unroll
    for (int reg = 0; reg < 16; ++reg) cinit[reg] = -(float)((reg & 3) + 8 * (reg >> 2) + 4 * h) * kEps;

    const int total_tiles = nt >> 5;  // FULL tiles only: the < 32 rows behind them are folded in by the merge kernel
    const int span_tile0 = span * span_tiles;
    const int ntl = min(span_tiles, total_tiles - span_tile0);  // tiles of this span (>= 1)
    const int nchunks = (ntl + chunk_tiles - 1) / chunk_tiles;

    float m1[QT], m2[QT];
#pragma unroll
    for (int t = 0; t < QT; ++t) m1[t] = m2[t] = -INFINITY;

    const uint4 *tbase = tfrag + (size_t)b * t_batch_u4 + (size_t)w * 64;  // wave-uniform; tile T at + T * KS * 64
    const uint32_t l16 = (uint32_t)l * 16u;
    auto copy_tile = [&](int tile, int slot) {  // scalar base + 32-bit lane offset
        __builtin_amdgcn_global_load_lds((const void *)(reinterpret_cast<const char *>(tbase + (size_t)tile * KS * 64) + l16),
                                         (__attribute__((address_space(3))) void *)&ring[slot][w * 64], 16, 0, 0);
    };
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t ring_base = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4 *)&ring[0][0];
    auto tile_body = [&](int slot, float rebase) {
        u32x4 r[KS];
        const uint32_t addr = l16 + (ring_base + (uint32_t)slot * (KS * 1024u));
        asm volatile(
            "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
            : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
            : "v"(addr)
            : "memory");
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(r[0]));
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(r[1]));
        asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(r[2]));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[3]));
        uint4 a[KS];
#pragma unroll
        for (int s = 0; s < KS; ++s) a[s] = make_uint4(r[s].x, r[s].y, r[s].z, r[s].w);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            v16f acc = mfma_fp4(a[0], bq[t][0], cinit);
#pragma unroll
            for (int s = 1; s < KS; ++s) acc = mfma_fp4(a[s], bq[t][s], acc);
            m1[t] += rebase;  // to this tile's row origin (exact; -inf stays -inf)
            m2[t] += rebase;
#pragma unroll
            for (int reg = 0; reg < 16; reg += 4) {
                const float s0 = __builtin_amdgcn_fmed3f(m1[t], acc[reg], acc[reg + 1]);
                const float t0 = __builtin_fmaxf(__builtin_fmaxf(m1[t], acc[reg]), acc[reg + 1]);
                const float s1 = __builtin_amdgcn_fmed3f(t0, acc[reg + 2], acc[reg + 3]);
                m1[t] = __builtin_fmaxf(__builtin_fmaxf(t0, acc[reg + 2]), acc[reg + 3]);
                m2[t] = __builtin_fmaxf(__builtin_fmaxf(m2[t], s0), s1);
            }
            // one accumulator tile alive at a time: without this tie the scheduler interleaves the next unit's MFMA chain on a second
            // accumulator (which buys nothing: MFMA and VALU do not overlap on a SIMD) and pays for it by spilling query fragments.
            // (An empty volatile asm that "rewrites" this unit's result and the next unit's first operand orders the two.)
            asm volatile("" : "+v"(m1[t]), "+v"(m2[t]), "+v"(a[0].x));
        }
    };

    __syncthreads();  // s_first
    int c_cur = __builtin_amdgcn_readfirstlane(s_first);
    int last_tile = span_tile0;  // frame of the decode (stays there when this workgroup got no chunk: everything is -inf)
    if (c_cur < nchunks) {
        int c_nxt = nchunks;  // unknown until requested
        int pos = 0;
        int len = min(chunk_tiles, ntl - c_cur * chunk_tiles);
        int tile = span_tile0 + c_cur * chunk_tiles;
        int prev_tile = tile;
        int slot = 0;
        bool first = true;
        copy_tile(tile, 0);
        for (;;) {
            // request the chunk after this one at the chunk's first tile (not needed when this is the span's last chunk)
            uint32_t req = 0;
            const bool ask = first && (c_cur + 1 < nchunks);
            if (ask && threadIdx.x == 0) {
                const uint32_t zero = 0, one = 1;  // scalar base + zero vector offset: no live 64-bit address register
                asm volatile("global_atomic_add %0, %1, %2, %3 sc0" : "=v"(req) : "v"(zero), "v"(one), "s"(ctr) : "memory");
            }
            // where is the next tile?  inside the chunk, or at the start of the chunk requested one chunk ago
            int next_tile = -1;
            if (pos + 1 < len) next_tile = tile + 1;
            else if (!first && c_nxt < nchunks) next_tile = span_tile0 + c_nxt * chunk_tiles;
            // (first && pos + 1 == len  <=>  a one-tile chunk, which is always the span's last chunk: no next tile)
            const int nslot = slot == NB - 1 ? 0 : slot + 1;
            if (next_tile >= 0) {
                copy_tile(next_tile, nslot);
                asm volatile("s_waitcnt vmcnt(1)" ::: "memory");  // this tile's piece (and the atomic, which is older than the copy)
            } else {
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            if (ask && threadIdx.x == 0) s_claim = (int)req;
            __builtin_amdgcn_s_barrier();
            tile_body(slot, (float)(32 * (tile - prev_tile)) * kEps);
            last_tile = tile;
            if (ask) c_nxt = __builtin_amdgcn_readfirstlane(s_claim);  // written before this iteration's barrier
            prev_tile = tile;
            if (pos + 1 < len) {
                ++pos;
                ++tile;
                first = false;
            } else if (next_tile >= 0) {
                c_cur = c_nxt;
                c_nxt = nchunks;
                pos = 0;
                len = min(chunk_tiles, ntl - c_cur * chunk_tiles);
                tile = next_tile;
                first = true;
            } else {
                break;
            }
            slot = nslot;
        }
    }

    // decode (frame = the last tile processed): v = (64 KS - 2 d) + (32 (last_tile - span_tile0) - local_row) * eps, |fraction| < 1/2
    const float frame = (float)(32 * (last_tile - span_tile0));
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
        const uint32_t o0 = __shfl_xor(k[0], 32), o1 = __shfl_xor(k[1], 32);
        uint32_t k0 = k[0], k1 = k[1];
        k1 = umed3(k0, k1, o0);
        k0 = min(k0, o0);
        k1 = umed3(k0, k1, o1);
        k0 = min(k0, o1);
        const int q = (qt0 + t) * 32 + (l & 31);
        if (h == 0 && q < nq) part[((size_t)b * slots + slot_id) * nq + q] = make_uint2(k0, k1);
    }
    if (stamps && l == 0) {
        unsigned long long *o = stamps + ((size_t)item * 4 + w) * 4;
        o[0] = __builtin_amdgcn_s_memtime() - st_c;
        o[1] = __builtin_amdgcn_s_memrealtime() - st_r;
        const unsigned hw = __builtin_amdgcn_s_getreg(4 | (0 << 6) | (31 << 11)), xcc = __builtin_amdgcn_s_getreg(20 | (0 << 6) | (3 << 11));
        o[2] = 1ull | ((unsigned long long)(hw & 0xFFFFFu) << 32) | ((unsigned long long)(xcc & 15u) << 56);
        o[3] = st_r;
    }
}

template <int KS>
void launch_mfma(int qt, dim3 grid, hipStream_t s, const uint4 *qf, size_t qb, const uint4 *tf, size_t tb, int nq, int nt, int rps,
                 int nsplit, int dshift, int qgroups, int n_items, uint2 *part, unsigned long long *stamps) {
#define MLPL_MFMA_LAUNCH(QT_)                                                                                                    \
    hipLaunchKernelGGL((knn_hamming_mfma_kernel<KS, QT_>), grid, dim3(256), 0, s, qf, qb, tf, tb, nq, nt, rps, nsplit, dshift, qgroups, \
                       n_items, part, stamps)
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
// k, ratio, d_idx, d_dist, d_group_counts: the outputs of the merge step; when the kernel chosen can produce them itself (static LDS-ring
// kernel, QT >= 2) *fused_out = 1 and the caller skips knn_hamming_merge_kernel.
int launch_knn_hamming_mfma(mlpl_ctx *ctx, const uint32_t *qw, size_t q_batch_words, const uint32_t *tw, size_t t_batch_words,
                            int nq, int nt, int nw, int batch, int dshift, hipStream_t s, int *rps_out, int *nsplit_out,
                            int *sps_out, int *tail_row0_out, const int32_t **split_tab_out, uint2 **part_out, int k, float ratio,
                            int32_t *d_idx, int32_t *d_dist, int32_t *d_group_counts, int *fused_out) {
    *fused_out = 0;
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
    const bool lds_ring = ks == 4 && ctx->opt_hamming_mfma_lds != 0;  // workgroup-level work: NW query groups share the train tiles
    const bool dyn = lds_ring && ctx->opt_hamming_mfma_lds == 2 && nt >= 32;  // ... and the train splits are drawn dynamically
    // waves per workgroup of the static ring kernel (option hamming_mfma_waves: 0 = automatic, 4, 8): eight when every wave carries four query
    // tiles (the throughput shape) -- a tile fetched and a barrier passed once per eight waves' units
    const int nwv = (lds_ring && !dyn && qt == 4 && ctx->opt_hamming_mfma_waves != 4) ? (ctx->opt_hamming_mfma_waves == 16 ? 16 : 8) : 4;
    const int qblocks = (qgroups + nwv - 1) / nwv;
    const int q_tiles_padded = lds_ring ? qblocks * nwv * qt : qgroups * qt;
    const int t_tiles = (nt + 31) / 32;

    // train splits: ~4 * opt waves per CU in flight, whole tiles, bounded so that the re-based fraction stays exact
    const int bpc = lds_ring ? std::max(4, ctx->opt_hamming_mfma_blocks_per_cu) : std::max(1, ctx->opt_hamming_mfma_blocks_per_cu);
    const long long target_waves = 4LL * bpc * ctx->num_cus;
    const long long wave_groups = lds_ring ? (long long)nwv * qblocks : (long long)qgroups;
    long long want = (target_waves + wave_groups * batch - 1) / (wave_groups * batch);
    int nsplit = (int)std::max<long long>(1, std::min<long long>(want, t_tiles));
    int rps = ((nt + nsplit - 1) / nsplit + 31) / 32 * 32;
    const int max_rps = ctx->opt_hamming_split_rows == 4096 ? 4096 : kMaxRowsPerSplit;
    rps = std::min(rps, max_rps);
    nsplit = (nt + rps - 1) / rps;
    if (nsplit > 65535) {
        set_error("knn_hamming: train set too large (nt=%d)", nt);
        return MLPL_E_BAD_INPUT;
    }
    // dynamic splits: spans of <= 8192 rows, `nmem` workgroups per (pair, query block, span) drawing chunks of 2 tiles
    int nspan = 1, span_tiles = t_tiles, nmem = 1, sps = 1;
    const int chunk_tiles = 2;
    if (dyn) {
        const int t_full = nt >> 5;  // whole tiles; rows [32 t_full, nt) are folded in by the merge kernel
        nspan = (t_full + 255) / 256;
        span_tiles = (t_full + nspan - 1) / nspan;
        const long long teams = (long long)qblocks * nspan * batch;
        const long long want_mem = ((long long)bpc * ctx->num_cus + teams - 1) / teams;
        nmem = (int)std::max<long long>(1, std::min<long long>(want_mem, (span_tiles + chunk_tiles - 1) / chunk_tiles));
        nsplit = nspan * nmem;
        rps = span_tiles * 32;
        sps = nmem;
        if (nsplit > 65535) {
            set_error("knn_hamming: train set too large (nt=%d)", nt);
            return MLPL_E_BAD_INPUT;
        }
    }
    const long long items = (lds_ring ? (long long)qblocks : (long long)qgroups) * nsplit * batch;  // workgroups (ring) or waves
    if (items > (1LL << 30)) {
        set_error("knn_hamming: problem too large for one launch (nq=%d nt=%d batch=%d)", nq, nt, batch);
        return MLPL_E_BAD_INPUT;
    }

    void *qf = nullptr, *tf = nullptr, *part = nullptr;
    int rc;
    const size_t q_u4 = (size_t)q_tiles_padded * ks * 64, t_u4 = (size_t)t_tiles * ks * 64;
    if ((rc = ws_get(ctx, WS_FRAG_Q, (size_t)batch * q_u4 * 16, &qf))) return rc;
    if ((rc = ws_get(ctx, WS_FRAG_T, ((size_t)batch * t_u4 + (size_t)ks * 64) * 16, &tf))) return rc;  // + one spare tile (register prefetch)
    if ((rc = ws_get(ctx, WS_PARTIAL, (size_t)batch * nsplit * nq * sizeof(uint2), &part))) return rc;
    int *counters = nullptr;
    const int counters_per_batch = dyn ? qblocks * nspan : 0;
    if (dyn) {
        void *cp = nullptr;
        if ((rc = ws_get(ctx, WS_COUNTERS, (size_t)batch * counters_per_batch * sizeof(int), &cp))) return rc;
        counters = (int *)cp;
    }
    // the static LDS-ring kernel expands its query operand itself (in registers): only the train set goes through the expansion kernel
    const bool expand_q = !(lds_ring && !dyn);
    const int train01 = (lds_ring && !dyn && ctx->opt_hamming_train01) ? 1 : 0;  // {0, +1} train fragments: that kernel only (it has the raw query words)
    const ExpandArgs ta{tw, t_batch_words, nt, t_tiles, (uint4 *)tf};
    const ExpandArgs qa = expand_q ? ExpandArgs{qw, q_batch_words, nq, q_tiles_padded, (uint4 *)qf} : ta;  // blockIdx.z == 0
    const dim3 egrid((unsigned)(((expand_q ? std::max(q_tiles_padded, t_tiles) : t_tiles) * 64 + 255) / 256), batch, expand_q ? 2 : 1);
    // (one thread per K-step when the launch is small: the single-pair latency shape)
    if (!expand_q && ks == 4 && nw == 8 && !counters && (long long)t_tiles * batch <= 4LL * ctx->num_cus && ctx->opt_hamming_expand_fine) {
        hipLaunchKernelGGL(hamming_expand_fine_kernel, dim3((unsigned)t_tiles, batch), dim3(256), 0, s, ta, train01);
    } else
    switch (ks) {
        case 1: hipLaunchKernelGGL(hamming_expand_kernel<1>, egrid, dim3(256), 0, s, qa, ta, nw, counters, counters_per_batch, train01); break;
        case 2: hipLaunchKernelGGL(hamming_expand_kernel<2>, egrid, dim3(256), 0, s, qa, ta, nw, counters, counters_per_batch, train01); break;
        case 4: hipLaunchKernelGGL(hamming_expand_kernel<4>, egrid, dim3(256), 0, s, qa, ta, nw, counters, counters_per_batch, train01); break;
        default: hipLaunchKernelGGL(hamming_expand_kernel<8>, egrid, dim3(256), 0, s, qa, ta, nw, counters, counters_per_batch, train01); break;
    }
    // 1-D grid, remapped in the kernel (XCD-aware); padded so that every XCD gets the same number of workgroups
    const long long blocks = lds_ring ? items : (items + 3) / 4;
    dim3 grid((unsigned)((blocks + 7) / 8 * 8));
    // Age-aware split sizes (static LDS-ring kernel).  With exactly R = 4 workgroups per CU the dispatcher deals workgroups breadth
    // first, so workgroup L is the (L / num_cus)-th oldest wave set on its CU, and the SIMD arbitrates oldest-first: measured per-tile
    // costs with all four ranks active are 1700 / 2100 / 3200 / 5300 cycles (tools/hamming_trace.py), i.e. equal splits finish at
    // 62k / 75k / 90k / 105k cycles and the CU runs the second half of the launch under-occupied.  Splits sized by the ranks' rates
    // (weights 1/cost) let the four finish together.  Correctness never depends on this: the table only moves split boundaries, and
    // it is used only when every query block of a (pair, split) has the same rank.
    const int32_t *split_tab = nullptr;
    if (lds_ring && !dyn && nwv == 4 && ctx->opt_hamming_mfma_weighted && blocks == 4LL * ctx->num_cus && nsplit >= 4 && nsplit <= 256 &&
        batch * (nsplit + 1) <= 4096) {
        const int per_xcd = (int)(grid.x >> 3);
        std::vector<int> rank((size_t)batch * nsplit, -1);
        bool ok = true;
        for (long long bid = 0; bid < blocks && ok; ++bid) {
            const long long item = (bid & 7) * per_xcd + (bid >> 3);
            if (item >= items) continue;
            const int sp = (int)((item / qblocks) % nsplit), bb = (int)(item / ((long long)qblocks * nsplit));
            const int r = (int)(bid / ctx->num_cus);
            int &slot = rank[(size_t)bb * nsplit + sp];
            if (slot < 0) slot = r;
            else if (slot != r) ok = false;
        }
        if (ok) {
            static const double kRate[4] = {1.0 / 1700, 1.0 / 2100, 1.0 / 3200, 1.0 / 5300};
            const long long key = ((long long)nt << 32) ^ ((long long)nsplit << 20) ^ ((long long)batch << 8) ^ qblocks;
            void *tabp = nullptr;
            if ((rc = ws_get(ctx, WS_SPLIT_TAB, 4096 * sizeof(int32_t), &tabp))) return rc;
            if (ctx->split_tab_key != key || ctx->split_tab_ptr != tabp) {
                // staging in pageable memory of this call: the context's pinned block belongs to the entry point that called us (the pair
                // batch entries keep pointers into it across the matching call, and pinned_get frees the block when it grows)
                std::vector<int32_t> hv((size_t)batch * (nsplit + 1));
                int32_t *h = hv.data();
                for (int bb = 0; bb < batch; ++bb) {
                    double wsum = 0;
                    for (int sp = 0; sp < nsplit; ++sp) wsum += kRate[rank[(size_t)bb * nsplit + sp]];
                    int acc_tiles = 0;
                    double acc_w = 0;
                    for (int sp = 0; sp < nsplit; ++sp) {
                        h[bb * (nsplit + 1) + sp] = acc_tiles;
                        acc_w += kRate[rank[(size_t)bb * nsplit + sp]];
                        int upto = (int)std::llround(t_tiles * acc_w / wsum);
                        upto = std::max(upto, acc_tiles + 1);                        // at least one tile per split
                        upto = std::min(upto, t_tiles - (nsplit - 1 - sp));          // and one left for every later split
                        upto = std::min(upto, acc_tiles + max_rps / 32);          // exact re-basing bound
                        acc_tiles = upto;
                    }
                    h[bb * (nsplit + 1) + nsplit] = t_tiles;
                    if (t_tiles - h[bb * (nsplit + 1) + nsplit - 1] > max_rps / 32) ok = false;
                }
                if (ok) {
                    MLPL_HIP_TRY(hipStreamSynchronize(s));  // the previous table may still be read by kernels in flight
                    MLPL_HIP_TRY(hipMemcpyAsync(tabp, h, (size_t)batch * (nsplit + 1) * sizeof(int32_t), hipMemcpyHostToDevice, s));
                    MLPL_HIP_TRY(hipStreamSynchronize(s));  // the staging vector dies with this scope
                    ctx->split_tab_key = key;
                    ctx->split_tab_ptr = tabp;
                }
            }
            if (ok) split_tab = (const int32_t *)tabp;
        }
    }
    unsigned long long *stamps = nullptr;
    ctx->dbg_stamp_items = 0;
    if (ctx->opt_hamming_stamps == 1) {
        void *sp = nullptr;
        const long long waves = lds_ring ? items * nwv : items;
        // per-wave records (4 x u64), then a per-tile clock trace of 48 u64 per wave (LDS-ring kernel, static splits)
        if ((rc = ws_get(ctx, WS_DEBUG, (size_t)waves * (32 + 48 * 8), &sp))) return rc;
        MLPL_HIP_TRY(hipMemsetAsync(sp, 0, (size_t)waves * (32 + 48 * 8), s));
        stamps = (unsigned long long *)sp;
        ctx->dbg_stamp_items = (int)waves;
    }
    HammingFuse fuse{nullptr, nullptr, nullptr, nullptr, k, ratio, train01, nullptr, 0ull};
    if (lds_ring && !dyn && ctx->opt_hamming_stamps == 2) {  // clock ring: one 32-byte record per launch, kClockRing launches deep
        void *cp = nullptr;
        if ((rc = ws_get(ctx, WS_CLOCK, (size_t)kClockRing * 32, &cp))) return rc;
        fuse.clk = (unsigned long long *)cp + (size_t)(ctx->hamming_clock_launches % kClockRing) * 4;
        fuse.launch_no = (unsigned long long)ctx->hamming_clock_launches++;
    }
    // (one query tile per wave = a single image pair with sixteen splits: measured with the fused epilogue 26.8 us per pair against 21.2 with
    // the separate merge launch -- the last-arriving workgroup's fold of sixteen written-through partials is a serial tail on a 15 us kernel)
    if (lds_ring && !dyn && qt >= 2 && d_idx && d_dist && (ctx->opt_hamming_fused_merge & 1)) {
        fuse.idx = d_idx, fuse.dist = d_dist, fuse.group_counts = d_group_counts;
        if (nsplit > 1) {  // ticket counters of the (pair, query block)s: zero when (re)allocated, left zero by every launch
            void *tp = nullptr;
            const size_t tb = (size_t)batch * qblocks * sizeof(int);
            if ((rc = ws_get(ctx, WS_TICKETS, tb, &tp))) return rc;   // a slot of its own: nothing else writes it (not the dyn kernel's counters)
            if (ctx->hamming_tickets_ptr != tp || ctx->hamming_tickets_bytes < tb) {
                MLPL_HIP_TRY(hipMemsetAsync(tp, 0, ctx->ws_bytes[WS_TICKETS], s));
                ctx->hamming_tickets_ptr = tp, ctx->hamming_tickets_bytes = ctx->ws_bytes[WS_TICKETS];
            }
            fuse.tickets = (int *)tp;
        }
        *fused_out = 1;
    }
    prof_mark(ctx, MLPL_PROF_KNN_HAMMING, 0, s);
    if (dyn) {
#define MLPL_DYN_LAUNCH(QT_)                                                                                                          \
    hipLaunchKernelGGL((knn_hamming_mfma_dyn_kernel<QT_>), grid, dim3(256), 0, s, (const uint4 *)qf, q_u4, (const uint4 *)tf, t_u4, nq, nt, \
                       span_tiles, nspan, nmem, chunk_tiles, dshift, qblocks, (int)items, (uint2 *)part, counters, stamps)
        if (qt == 4) MLPL_DYN_LAUNCH(4);
        else if (qt == 2) MLPL_DYN_LAUNCH(2);
        else MLPL_DYN_LAUNCH(1);
#undef MLPL_DYN_LAUNCH
    } else if (lds_ring) {
#define MLPL_RING_LAUNCH(QT_)                                                                                                          \
    do {                                                                                                                               \
        if (ctx->opt_hamming_mfma_prio == 2)                                                                                           \
            hipLaunchKernelGGL((knn_hamming_mfma_lds_kernel<QT_, 2>), grid, dim3(256), 0, s, qw, q_batch_words, (const uint4 *)tf, \
                               t_u4, nq, nt, rps, nsplit, dshift, qblocks, (int)items, (uint2 *)part, stamps, split_tab, fuse);         \
        else if (ctx->opt_hamming_mfma_prio)                                                                                           \
            hipLaunchKernelGGL((knn_hamming_mfma_lds_kernel<QT_, 1>), grid, dim3(256), 0, s, qw, q_batch_words, (const uint4 *)tf, \
                               t_u4, nq, nt, rps, nsplit, dshift, qblocks, (int)items, (uint2 *)part, stamps, split_tab, fuse);         \
        else                                                                                                                           \
            hipLaunchKernelGGL((knn_hamming_mfma_lds_kernel<QT_, 0>), grid, dim3(256), 0, s, qw, q_batch_words, (const uint4 *)tf, \
                               t_u4, nq, nt, rps, nsplit, dshift, qblocks, (int)items, (uint2 *)part, stamps, split_tab, fuse);         \
    } while (0)
        if (qt == 4 && nwv == 16)
            hipLaunchKernelGGL((knn_hamming_mfma_lds_kernel<4, 0, 16>), grid, dim3(1024), 0, s, qw, q_batch_words, (const uint4 *)tf, t_u4, nq, nt, rps, nsplit,
                               dshift, qblocks, (int)items, (uint2 *)part, stamps, split_tab, fuse);
        else if (qt == 4 && nwv == 8 && ctx->opt_hamming_mfma_prefetch == 4)
            hipLaunchKernelGGL((knn_hamming_mfma_lds_kernel<4, 0, 8, 4>), grid, dim3(512), 0, s, qw, q_batch_words, (const uint4 *)tf, t_u4, nq, nt, rps, nsplit,
                               dshift, qblocks, (int)items, (uint2 *)part, stamps, split_tab, fuse);
        else if (qt == 4 && nwv == 8 && ctx->opt_hamming_mfma_prefetch == 6)
            hipLaunchKernelGGL((knn_hamming_mfma_lds_kernel<4, 0, 8, 6>), grid, dim3(512), 0, s, qw, q_batch_words, (const uint4 *)tf, t_u4, nq, nt, rps, nsplit,
                               dshift, qblocks, (int)items, (uint2 *)part, stamps, split_tab, fuse);
        else if (qt == 4 && nwv == 8 && ctx->opt_hamming_mfma_prio == 3)
            hipLaunchKernelGGL((knn_hamming_mfma_lds_kernel<4, 3, 8>), grid, dim3(512), 0, s, qw, q_batch_words, (const uint4 *)tf, t_u4, nq, nt, rps, nsplit,
                               dshift, qblocks, (int)items, (uint2 *)part, stamps, split_tab, fuse);
        else if (qt == 4 && nwv == 8 && ctx->opt_hamming_mfma_prio == 1)
            hipLaunchKernelGGL((knn_hamming_mfma_lds_kernel<4, 1, 8>), grid, dim3(512), 0, s, qw, q_batch_words, (const uint4 *)tf, t_u4, nq, nt, rps, nsplit,
                               dshift, qblocks, (int)items, (uint2 *)part, stamps, split_tab, fuse);
        else if (qt == 4 && nwv == 8)
            hipLaunchKernelGGL((knn_hamming_mfma_lds_kernel<4, 0, 8>), grid, dim3(512), 0, s, qw, q_batch_words, (const uint4 *)tf, t_u4, nq, nt, rps, nsplit,
                               dshift, qblocks, (int)items, (uint2 *)part, stamps, split_tab, fuse);
        else if (qt == 4) MLPL_RING_LAUNCH(4);
        else if (qt == 2) MLPL_RING_LAUNCH(2);
        else MLPL_RING_LAUNCH(1);
#undef MLPL_RING_LAUNCH
    } else
    switch (ks) {
        case 1: launch_mfma<1>(qt, grid, s, (const uint4 *)qf, q_u4, (const uint4 *)tf, t_u4, nq, nt, rps, nsplit, dshift, qgroups, (int)items, (uint2 *)part, stamps); break;
        case 2: launch_mfma<2>(qt, grid, s, (const uint4 *)qf, q_u4, (const uint4 *)tf, t_u4, nq, nt, rps, nsplit, dshift, qgroups, (int)items, (uint2 *)part, stamps); break;
        case 4: launch_mfma<4>(qt, grid, s, (const uint4 *)qf, q_u4, (const uint4 *)tf, t_u4, nq, nt, rps, nsplit, dshift, qgroups, (int)items, (uint2 *)part, stamps); break;
        default: launch_mfma<8>(qt, grid, s, (const uint4 *)qf, q_u4, (const uint4 *)tf, t_u4, nq, nt, rps, nsplit, dshift, qgroups, (int)items, (uint2 *)part, stamps); break;
    }
    prof_mark(ctx, MLPL_PROF_KNN_HAMMING, 1, s);
    *rps_out = rps;
    *nsplit_out = nsplit;
    *sps_out = sps;
    *split_tab_out = split_tab;
    *tail_row0_out = dyn ? (nt >> 5) << 5 : nt;
    *part_out = (uint2 *)part;
    return MLPL_OK;
}

}  // namespace mlpl
