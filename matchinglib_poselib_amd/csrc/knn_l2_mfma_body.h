// knn_l2_mfma_body.h -- the int8 matrix-core tile loop of the squared-L2 2-NN (see knn_l2_mfma.hip for the arithmetic), as a device
// function shared by the stand-alone kernel (forced mode) and the fused auto-path kernel in knn_l2.hip.
#pragma once
#include "mlpl_internal.h"

namespace mlpl {
namespace l2mfma {

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

// One workgroup = W waves = W query tiles (32 queries each) against the train tiles of one split, consumed in groups of G = 16/KS tiles
// (16 KiB of LDS; 8 tiles for KS = 1).  A whole group is requested from memory at once and the next group's loads are in flight (16 VGPRs per thread)
// while this one is multiplied.  Work items are numbered so that each XCD (block id mod 8) owns a contiguous range of
// (batch, split, query block): neighbouring splits share an L2, so each XCD pulls every query fragment but only an eighth of the
// train fragments across the fabric.
struct L2MfmaArgs {
    const uint4 *qfrag;
    const int *qcst;
    const uint4 *tfrag;
    const int *tcst;
    int nq, nt, nq_tiles, nt_tiles, tiles_per_split, nsplit, qblocks, batch, ib;
    ulonglong2 *part;
    unsigned long long *stamps;
};

// What launch_knn_l2_mfma prepared for one call.
struct L2MfmaPlan {
    L2MfmaArgs args;
    L2Gate gate;
    int ksel;       // K-step class: 1, 2, 4 or 8 (dim <= 32 * ksel)
    unsigned grid;  // workgroups of 256 threads
};

template <int KS>
constexpr int l2_mfma_group() { return KS == 1 ? 8 : 16 / KS; }  // tiles per group (at most 256 rows: one thread per row in commit)
template <int KS>
constexpr int l2_mfma_lds_bytes() { return (l2_mfma_group<KS>() * KS * 64 + 2 * l2_mfma_group<KS>() * 8) * 16; }

// `lds`: l2_mfma_lds_bytes<KS>() bytes of 16-byte aligned LDS; `block` = linear workgroup id.  Every thread of the workgroup calls.
template <int KS, int W>
__device__ __forceinline__ void l2_mfma_body(const L2MfmaArgs &a, v4i *lds, unsigned block) {
    const uint4 *__restrict__ qfrag = a.qfrag;
    const int *__restrict__ qcst = a.qcst;
    const uint4 *__restrict__ tfrag = a.tfrag;
    const int *__restrict__ tcst = a.tcst;
    const int nq = a.nq, nt = a.nt, nq_tiles = a.nq_tiles, nt_tiles = a.nt_tiles, tiles_per_split = a.tiles_per_split, nsplit = a.nsplit,
              qblocks = a.qblocks, batch = a.batch, ib = a.ib;
    ulonglong2 *__restrict__ part = a.part;
    unsigned long long *__restrict__ stamps = a.stamps;
    unsigned long long st0 = 0, st1 = 0;
    if (stamps) st0 = __builtin_readcyclecounter(), st1 = __builtin_amdgcn_s_memrealtime();
    constexpr int G = l2_mfma_group<KS>();
    constexpr int NT = 64 * W;               // threads
    constexpr int kPre = G * KS * 64 / NT;   // uint4 per thread and group
    // (native vector types throughout the staging path: an array of HIP_vector_type structs is not split into registers and ends up
    // in scratch, with a wait per load)
    v4i *tileA = lds;  // fragments | accumulator starts | row words
    int *tileC = reinterpret_cast<int *>(lds + G * KS * 64);
    uint32_t *tileR = reinterpret_cast<uint32_t *>(lds + G * KS * 64 + G * 8);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    // XCD-aware work numbering
    const long long total = (long long)batch * nsplit * qblocks, per_xcd = (total + 7) / 8;
    const long long j = block >> 3, w = (long long)(block & 7) * per_xcd + j;
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


}  // namespace l2mfma
}  // namespace mlpl
