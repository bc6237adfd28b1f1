// knn_l2_f16.hip -- squared-L2 2-NN for float descriptors that are NOT integer-valued (RootSIFT, SURF, KAZE ...): two fp16 matrix-core
// candidate passes and an exact fp32 re-rank in cvflann's summation order (gfx950).
//
// Replaces cvflann::Index<L2<float>>(LinearIndexParams).knnSearch (reference matchinglib/source/matchers.cpp:634-689) for such data;
// the result is the exact kernel's (knn_l2.hip: every distance of the answer is computed in cvflann::L2<float>'s order, ties go to
// the lower train index), only the SEARCH runs on the matrix cores.
//
// Arithmetic.  A query row is scaled by a power of two 2^kq that puts its largest element into [2^12, 2^13), the 32 train rows of a tile by
// one common 2^kt chosen the same way for the tile's largest element M, and every element is rounded to fp16: s x = h + delta,
// |delta| <= 2^-11 |s x| + 2^-25.  The dot product is accumulated on v_mfma_f32_32x32x16_f16 (fp32 accumulator), and with the row norms
// N = sum x^2 (double)
//     d~ = Nq + Nt - 2 acc 2^-(kq + kt).
// Error of d~ against the reference's fp32 value D_ref (its own rounding included): |d~ - D_ref| <= c (Nq + Nt) + kappa (Nq + M^2) with
//     c = 2^-10 1.002          (representation: 2 x 2^-11 per product, sum |q_i t_i| <= (Nq + Nt) / 2, doubled by the factor 2 of d~)
//       + 16 KS 2^-23 1.01     (16 KS accumulations, each off by at most one unit in the last place of the running magnitude <= |q||t|)
//       + (8 + dim/4) 2^-23    (D_ref: (6 + dim/4) roundings of non-negative terms, D <= 2 (Nq + Nt))
//       + 2^-20                (the float constants and the fp32 operations of the epilogue, the absolute floor of the query side)
//     kappa = sqrt(dim) 2^-36  (absolute floor 2^-25 / 2^kt of the train elements: rows far below their tile's largest row)
// (KS = 16-element steps; c = 1.0e-3 for 128 dimensions.)  With U(t) = d~ + bound and L(t) = d~ - bound:
//   pass 1: the two smallest U(t) per query -> U2.  At least two rows have D_ref <= U2, so the second smallest D_ref is <= U2;
//   pass 2: the same products again (same instructions, same order: the same acc), every row with L(t) <= U2 is a candidate.  A row of
//           the true top-2 has L <= D_ref <= (second smallest D_ref) <= U2: it is a candidate, and so is every row tied with it;
//   re-rank: D_ref of the candidates in cvflann's order (l2_group4 of knn_l2_common.h), lexicographic (D_ref, row) top-2.
// The bound is loose on purpose (one product per step instead of a hi / lo split with three: a third of the matrix-core work and half the
// bytes): the second nearest neighbour is an extreme value of the distance distribution, so few rows fall into a window of 1e-3 of the
// norms -- 2.2 .. 2.4 candidates per query on RootSIFT-like and on normal data.  A query with more than 64 candidates (tight clusters of
// near-identical train rows) is re-ranked against every train row, and so is every query when a row is outside the path's range
// (non-finite or > 1e15 element, or a largest element below 1e-12): correct, only slow.

#include <algorithm>
#include <cmath>

#include "knn_l2_common.h"
#include "mlpl_internal.h"

namespace mlpl {

namespace {

typedef _Float16 v8h __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));
typedef float v4f __attribute__((ext_vector_type(4)));
typedef int v4i __attribute__((ext_vector_type(4)));
typedef unsigned long long u64;

constexpr int kCandCap = 64;   // candidates kept per query
constexpr int kQTilesPerWave = 2;
constexpr int kQTilesPerBlock = 4 * kQTilesPerWave;

// flag words of a call (generation `gen`): flags[1] == gen: a row is outside the path's range; flags[2]: statistics (queries re-ranked
// against every train row); flags[3] == gen: some element is not an integer in [0, 255] (feeds the hint of the auto mode)

struct L2fPrepArgs {
    const float *X;
    size_t stride, bstride;
    int n, ntiles;
    uint4 *frag;  // [batch][tile][KS][64 lanes] x 16 B (8 halfs: row lane & 31, elements 16 s + 8 (lane >> 5) + j)
    float *cst;   // [batch][tile][128]: train {U add [32], L add [32], fac, ...}, query {2^-kq [32], threshold add [32], ...}
};

// One block per 32-row tile, blockIdx.z = 0 queries / 1 train rows.  Thread (r = tid & 31, c = tid >> 5) owns the 8-element pieces
// u = c, c + 8, ... of row r: fragment entry (step u >> 1, lane 32 (u & 1) + r).
template <int KS>
__global__ __launch_bounds__(256) void l2f_prep_kernel(L2fPrepArgs qa, L2fPrepArgs ta, int dim, float cerr, float kappa, int gen, int *__restrict__ flags,
                                                       int *__restrict__ cand_cnt) {
    const L2fPrepArgs A = blockIdx.z ? ta : qa;
    const int tile = blockIdx.x;
    if (tile >= A.ntiles) return;
    __shared__ float pmax[8][32];
    __shared__ double psum[8][32];
    __shared__ float rmax[32];
    const int b = blockIdx.y, r = threadIdx.x & 31, c = threadIdx.x >> 5;
    const int row = tile * 32 + r;
    const bool train = blockIdx.z != 0;
    const float *x = A.X + (size_t)b * A.bstride + (size_t)row * A.stride;
    constexpr int NU = KS / 4;
    float xv[NU][8];
    float mx = 0.f;
    double s2 = 0.0;
    bool bad = false, nonint = false;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int k0 = 8 * (c + 8 * i);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const float v = (row < A.n && k0 + j < dim) ? x[k0 + j] : 0.f;
            xv[i][j] = v;
            const float a = fabsf(v);
            bad = bad || !(a <= 1e15f);  // NaN, infinity, out of range
            nonint = nonint || !(v >= 0.f && v <= 255.f && v == floorf(v));
            mx = fmaxf(mx, a);
            s2 += (double)v * (double)v;
        }
    }
    pmax[c][r] = mx;
    psum[c][r] = s2;
    __syncthreads();
    mx = pmax[0][r], s2 = psum[0][r];
#pragma unroll
    for (int i = 1; i < 8; ++i) mx = fmaxf(mx, pmax[i][r]), s2 += psum[i][r];
    bad = bad || (mx > 0.f && mx < 1e-12f);
    if (c == 0) rmax[r] = mx;
    __syncthreads();
    float tmax = mx;  // queries: the row's own largest element; train rows: the tile's
    if (train) {
        tmax = rmax[0];
#pragma unroll
        for (int i = 1; i < 32; ++i) tmax = fmaxf(tmax, rmax[i]);
    }
    int k = 0;
    if (tmax > 0.f && tmax <= 1e15f) {
        int e;
        (void)frexpf(tmax, &e);  // tmax = m 2^e, m in [0.5, 1)
        k = 13 - e;
    }
    uint4 *frag = A.frag + ((size_t)b * A.ntiles + tile) * KS * 64;
#pragma unroll
    for (int i = 0; i < NU; ++i) {
        const int u = c + 8 * i;
        v8h h;
#pragma unroll
        for (int j = 0; j < 8; ++j) h[j] = (_Float16)(bad ? 0.f : ldexpf(xv[i][j], k));
        frag[(size_t)(u >> 1) * 64 + 32 * (u & 1) + r] = *reinterpret_cast<uint4 *>(&h);
    }
    // one atomic per block at most, and none once the word is set (every thread queueing on one address costs more than the conversion)
    const int any_bad = __syncthreads_or(bad), any_nonint = __syncthreads_or(nonint);
    if (threadIdx.x == 0) {
        if (any_bad && *(volatile int *)&flags[1] != gen) atomicMax(&flags[1], gen);
        if (any_nonint && *(volatile int *)&flags[3] != gen) atomicMax(&flags[3], gen);
    }
    if (c == 0) {
        float *cst = A.cst + ((size_t)b * A.ntiles + tile) * 128;
        const double km2 = (double)kappa * (double)tmax * (double)tmax;
        if (train) {
            const bool in = row < A.n;
            // rounded outwards by one float (the conversion itself rounds to nearest)
            cst[r] = in ? nextafterf((float)(s2 * (1.0 + (double)cerr) + km2), INFINITY) : INFINITY;        // U = fma(m, fac, this) (+ the query's share)
            cst[32 + r] = in ? nextafterf((float)(s2 * (1.0 - (double)cerr) - km2), -INFINITY) : INFINITY;  // L = fma(m, fac, this) (- the query's share)
            cst[64 + r] = -ldexpf(2.f, -k);                                                                 // fac (the same for the whole tile)
            cst[96 + r] = 0.f;
        } else {
            cst[r] = ldexpf(1.f, -k);
            cst[32 + r] = nextafterf((float)(2.0 * ((double)cerr + (double)kappa) * s2 * 1.0001), INFINITY);  // L <= U2 + this for a candidate
            cst[64 + r] = 0.f, cst[96 + r] = 0.f;
            if (row < A.n) cand_cnt[(size_t)b * A.n + row] = 0;
        }
    }
}

struct L2fArgs {
    const uint4 *qfrag;
    const float *qcst;
    const uint4 *tfrag;
    const float *tcst;
    int nq, nt, nq_tiles, nt_tiles, tiles_per_split, nsplit, qblocks, batch;
    float2 *part;  // [batch][split][nq]: the two smallest U of the split (without the query's share)
    int *cand_cnt;
    int *cand;     // [batch][nq][kCandCap]
};

__device__ __forceinline__ void two_smallest(float &u0, float &u1, float v) {
    u1 = fminf(u1, fmaxf(u0, v));
    u0 = fminf(u0, v);
}

// One workgroup = 4 waves x 2 query tiles = 256 queries against the train tiles of one split, staged through LDS in groups of G = 32 / KS
// tiles (32 KiB); the next group's loads are in flight while this one is multiplied.  Two query tiles per wave: every train fragment read
// from LDS feeds two matrix-core instructions (with one, the LDS reads of the four waves take longer than the products).
// EMIT = false: pass 1 (two smallest U per query and split); EMIT = true: pass 2 (candidates with L <= U2 + the query's share).
template <int KS, bool EMIT>
__global__ __launch_bounds__(256, 2) void l2f_pass_kernel(L2fArgs a, int gen, const int *__restrict__ flags) {
    constexpr int G = 32 / KS;
    constexpr int NT = 256;
    constexpr int kPre = G * KS * 64 / NT;  // = 8
    constexpr int kPreC = G * 64 / NT;      // constants per thread and group (1 or 2)
    constexpr int QT = kQTilesPerWave;
    __shared__ __attribute__((aligned(16))) v4i tileA[G * KS * 64];
    __shared__ __attribute__((aligned(16))) float tileC[G * 64];  // per tile: 32 adds (U or L), fac, padding
    if (flags[1] == gen) return;  // a row outside the path's range: the re-rank scans everything
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const long long total = (long long)a.batch * a.nsplit * a.qblocks, per_xcd = (total + 7) / 8;
    const long long jx = blockIdx.x >> 3, w = (long long)(blockIdx.x & 7) * per_xcd + jx;
    if (jx >= per_xcd || w >= total) return;
    const int qb = (int)(w % a.qblocks), split = (int)((w / a.qblocks) % a.nsplit), b = (int)(w / ((long long)a.qblocks * a.nsplit));
    const v4i *tfrag = reinterpret_cast<const v4i *>(a.tfrag) + (size_t)b * a.nt_tiles * KS * 64;
    const float *tcst = a.tcst + (size_t)b * a.nt_tiles * 128;

    const int t_begin = split * a.tiles_per_split;
    const int t_end = min(a.nt_tiles, t_begin + a.tiles_per_split);
    v4i pre[kPre];
    float pre_c[kPreC];
    // whole groups are always moved; what lies past t_end is the next split's tiles or the padding the launcher allocates, never multiplied
    auto fetch = [&](int t0) {
        const v4i *src = tfrag + (size_t)t0 * KS * 64;
#pragma unroll
        for (int i = 0; i < kPre; ++i) pre[i] = src[tid + NT * i];
#pragma unroll
        for (int i = 0; i < kPreC; ++i) {
            const int e = tid + NT * i, tl = e >> 6, j = e & 63;
            pre_c[i] = tcst[(size_t)(t0 + tl) * 128 + (j < 32 ? (EMIT ? 32 : 0) + j : 32 + j)];  // j >= 32: the fac block (64 + (j - 32))
        }
    };
    auto commit = [&]() {
#pragma unroll
        for (int i = 0; i < kPre; ++i) tileA[tid + NT * i] = pre[i];
#pragma unroll
        for (int i = 0; i < kPreC; ++i) tileC[tid + NT * i] = pre_c[i];
    };
    if (t_begin < t_end) fetch(t_begin);
    v8h qh[QT][KS];
    float cq[QT], thr[QT];
    int qi[QT];
    bool q_ok[QT];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        const int qtile = qb * kQTilesPerBlock + wave * QT + u;
        const int qtile_ld = qtile < a.nq_tiles ? qtile : 0;
        const v4i *qfrag = reinterpret_cast<const v4i *>(a.qfrag) + ((size_t)b * a.nq_tiles + qtile_ld) * KS * 64;
        const float *qcst = a.qcst + ((size_t)b * a.nq_tiles + qtile_ld) * 128;
#pragma unroll
        for (int s = 0; s < KS; ++s) {
            const v4i h = qfrag[s * 64 + lane];
            qh[u][s] = *reinterpret_cast<const v8h *>(&h);
        }
        cq[u] = qcst[lane & 31];
        qi[u] = qtile * 32 + (lane & 31);
        q_ok[u] = qtile < a.nq_tiles && qi[u] < a.nq;
        thr[u] = -INFINITY;
    }
    if constexpr (EMIT) {
        // U2 of the block's 256 queries over all splits: thread = query, the partial pairs read eight splits at a time (as a per-lane loop
        // inside each wave -- one dependent round trip per split and query tile -- this prologue was longer than the products)
        __shared__ float thrS[32 * kQTilesPerBlock];
        const int q = qb * 32 * kQTilesPerBlock + tid;
        float th = -INFINITY;
        if (q < a.nq) {
            float m0 = INFINITY, m1 = INFINITY;
            const float2 *pp = a.part + (size_t)b * a.nsplit * a.nq + q;
            for (int s0 = 0; s0 < a.nsplit; s0 += 8) {
                float2 p[8];
#pragma unroll
                for (int i = 0; i < 8; ++i) p[i] = (s0 + i < a.nsplit) ? pp[(size_t)(s0 + i) * a.nq] : make_float2(INFINITY, INFINITY);
#pragma unroll
                for (int i = 0; i < 8; ++i) {
                    two_smallest(m0, m1, p[i].x);
                    two_smallest(m0, m1, p[i].y);
                }
            }
            th = m1 + a.qcst[((size_t)b * a.nq_tiles + (q >> 5)) * 128 + 32 + (q & 31)];
            th += fabsf(th) * 0x1p-20f;  // the sum above rounds to nearest: push it up (inside the 2^-20 term of c)
        }
        thrS[tid] = th;
        __syncthreads();
#pragma unroll
        for (int u = 0; u < QT; ++u) thr[u] = q_ok[u] ? thrS[(wave * QT + u) * 32 + (lane & 31)] : -INFINITY;
    }
    float u0[QT], u1[QT];
    // pass 2 keeps a lane's first four hits in registers and appends them with ONE returning atomic at the end: an atomic per hit, each
    // waited for before the loop goes on, was two thirds of the pass (a wave meets a hit in more than half of its tiles)
    int cl[QT][4], cn[QT];
#pragma unroll
    for (int u = 0; u < QT; ++u) {
        u0[u] = INFINITY, u1[u] = INFINITY, cn[u] = 0;
#pragma unroll
        for (int i = 0; i < 4; ++i) cl[u][i] = 0;
    }
    for (int t0 = t_begin; t0 < t_end; t0 += G) {
        if (t0 != t_begin) __syncthreads();
        commit();
        if (t0 + G < t_end) fetch(t0 + G);
        __syncthreads();
        const int g_end = min(t0 + G, t_end);
        for (int t = t0; t < g_end; ++t) {
            const v4i *A = tileA + (t - t0) * KS * 64;
            v16f acc[QT];
#pragma unroll
            for (int u = 0; u < QT; ++u)
#pragma unroll
                for (int r = 0; r < 16; ++r) acc[u][r] = 0.f;
#pragma unroll
            for (int s = 0; s < KS; ++s) {
                const v4i h4 = A[s * 64 + lane];
                const v8h th = *reinterpret_cast<const v8h *>(&h4);
#pragma unroll
                for (int u = 0; u < QT; ++u) acc[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(th, qh[u][s], acc[u], 0, 0, 0);
            }
            // accumulator r of this lane: train row (r & 3) + 8 (r >> 2) + 4 (lane >> 5) of the tile, query lane & 31
            const float *C = tileC + (t - t0) * 64;
            const float fac = C[32];
            v4f add[4];
#pragma unroll
            for (int g = 0; g < 4; ++g) add[g] = *reinterpret_cast<const v4f *>(&C[8 * g + 4 * (lane >> 5)]);
#pragma unroll
            for (int u = 0; u < QT; ++u) {
                const float f = cq[u] * fac;  // both powers of two
                float v[16];
#pragma unroll
                for (int g = 0; g < 4; ++g)
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[4 * g + e] = __fmaf_rn(acc[u][4 * g + e], f, add[g][e]);
                if constexpr (EMIT) {
                    float vmin = v[0];
#pragma unroll
                    for (int r = 1; r < 16; ++r) vmin = fminf(vmin, v[r]);
                    if (vmin <= thr[u]) {  // rare: a handful of rows per query over the whole train set
#pragma unroll
                        for (int r = 0; r < 16; ++r) {
                            if (v[r] <= thr[u]) {
                                const int row = t * 32 + (r & 3) + 8 * (r >> 2) + 4 * (lane >> 5);
                                if (cn[u] < 4) {
#pragma unroll
                                    for (int i = 0; i < 4; ++i) cl[u][i] = (cn[u] == i) ? row : cl[u][i];
                                    ++cn[u];
                                } else {
                                    const int pos = atomicAdd(&a.cand_cnt[(size_t)b * a.nq + qi[u]], 1);
                                    if (pos < kCandCap) a.cand[((size_t)b * a.nq + qi[u]) * kCandCap + pos] = row;
                                }
                            }
                        }
                    }
                } else {
#pragma unroll
                    for (int r = 0; r < 16; ++r) two_smallest(u0[u], u1[u], v[r]);
                }
            }
        }
    }
    if constexpr (EMIT) {
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            if (cn[u] > 0) {
                const int pos = atomicAdd(&a.cand_cnt[(size_t)b * a.nq + qi[u]], cn[u]);
#pragma unroll
                for (int i = 0; i < 4; ++i)
                    if (i < cn[u] && pos + i < kCandCap) a.cand[((size_t)b * a.nq + qi[u]) * kCandCap + pos + i] = cl[u][i];
            }
        }
    } else {
#pragma unroll
        for (int u = 0; u < QT; ++u) {
            const float o0 = __shfl_xor(u0[u], 32), o1 = __shfl_xor(u1[u], 32);
            two_smallest(u0[u], u1[u], o0);
            two_smallest(u0[u], u1[u], o1);
            if (q_ok[u] && lane < 32) a.part[((size_t)b * a.nsplit + split) * a.nq + qi[u]] = make_float2(u0[u], u1[u]);
        }
    }
}

// Exact re-rank: 16 lanes per query, the candidates lane-strided (all train rows when the list overflowed or the path is out of range).
struct L2fRerankArgs {
    const float *q;
    size_t q_stride, q_bstride;
    const float *t;
    size_t t_stride, t_bstride;
    int nq, nt, dim, k;
    const int *cand_cnt;
    const int *cand;
    int32_t *idx;
    float *dist;
    int *flags;
    int *hint;  // host-visible word of the auto mode: 2 gen + (the call's data were not integer-valued)
};

template <bool VEC>
__device__ __forceinline__ float l2f_exact(const float *__restrict__ t, const float *__restrict__ q, int dim) {
    const int ngroups = dim / 4;
    float res = 0.f;
    int g = 0;
    if constexpr (VEC) {
        for (; g + 8 <= ngroups; g += 8) {  // eight groups of loads in flight, the sum in cvflann's order
            float4 tv[8], qv[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) tv[i] = reinterpret_cast<const float4 *>(t)[g + i], qv[i] = reinterpret_cast<const float4 *>(q)[g + i];
#pragma unroll
            for (int i = 0; i < 8; ++i) res = l2_group4(res, tv[i], qv[i]);
        }
    }
    for (; g < ngroups; ++g) {
        float4 tv, qv;
        if constexpr (VEC) {
            tv = reinterpret_cast<const float4 *>(t)[g];
            qv = reinterpret_cast<const float4 *>(q)[g];
        } else {
            tv = make_float4(t[4 * g], t[4 * g + 1], t[4 * g + 2], t[4 * g + 3]);
            qv = make_float4(q[4 * g], q[4 * g + 1], q[4 * g + 2], q[4 * g + 3]);
        }
        res = l2_group4(res, tv, qv);
    }
    for (int j = ngroups * 4; j < dim; ++j) {  // scalar tail: result += diff * diff
        const float d = __fsub_rn(t[j], q[j]);
        res = __fadd_rn(res, __fmul_rn(d, d));
    }
    return res;
}

template <bool VEC>
__global__ __launch_bounds__(256) void l2f_rerank_kernel(L2fRerankArgs a, int gen) {
    const int b = blockIdx.y, sub = threadIdx.x & 15;
    const int qi = blockIdx.x * 16 + (threadIdx.x >> 4);
    if (a.hint && blockIdx.x == 0 && b == 0 && threadIdx.x == 0) *a.hint = 2 * gen + (a.flags[3] == gen ? 1 : 0);
    const bool live = qi < a.nq;
    const int qs = live ? qi : 0;
    const float *q = a.q + (size_t)b * a.q_bstride + (size_t)qs * a.q_stride;
    const float *t = a.t + (size_t)b * a.t_bstride;
    const int cnt = a.cand_cnt[(size_t)b * a.nq + qs];
    const bool scan_all = a.flags[1] == gen || cnt > kCandCap;
    u64 k0 = ~0ull, k1 = ~0ull;
    if (live) {
        if (!scan_all) {
            for (int c = sub; c < cnt; c += 16) {
                const int row = a.cand[((size_t)b * a.nq + qi) * kCandCap + c];
                const float d = l2f_exact<VEC>(t + (size_t)row * a.t_stride, q, a.dim);
                l2_top2_update(k0, k1, ((u64)__float_as_uint(d) << 32) | (u64)(uint32_t)row);
            }
        } else {
            if (sub == 0) atomicAdd(&a.flags[2], 1);
            for (int row = sub; row < a.nt; row += 16) {
                const float d = l2f_exact<VEC>(t + (size_t)row * a.t_stride, q, a.dim);
                l2_top2_update(k0, k1, ((u64)__float_as_uint(d) << 32) | (u64)(uint32_t)row);
            }
        }
    }
#pragma unroll
    for (int off = 1; off < 16; off <<= 1) {
        const u64 o0 = __shfl_xor(k0, off), o1 = __shfl_xor(k1, off);
        l2_top2_update(k0, k1, o0);
        l2_top2_update(k0, k1, o1);
    }
    if (!live || sub != 0) return;
    const size_t o = ((size_t)b * a.nq + qi) * a.k;
    a.idx[o] = (int32_t)(k0 & 0xFFFFFFFFull);
    a.dist[o] = __uint_as_float((uint32_t)(k0 >> 32));
    if (a.k == 2) {
        a.idx[o + 1] = (int32_t)(k1 & 0xFFFFFFFFull);
        a.dist[o + 1] = __uint_as_float((uint32_t)(k1 >> 32));
    }
}

}  // namespace

bool knn_l2_f16_applicable(int dim, int nt, int k) { return dim >= 1 && dim <= 128 && nt >= k; }

// The whole path: operand preparation, the two candidate passes, the exact re-rank -- four launches, no host hop.  `d_flags`: the four
// flag words of the L2 paths (WS_L2_FLAG), `gen` this call's generation; `d_hint`: device view of the context's hint word or nullptr.
int launch_knn_l2_f16(mlpl_ctx *ctx, const float *d_q, int nq, size_t q_stride, size_t q_bstride, const float *d_t, int nt, size_t t_stride,
                      size_t t_bstride, int dim, int k, int batch, int gen, int *d_flags, int *d_hint, int32_t *d_idx, float *d_dist, hipStream_t s) {
    const int KS = dim <= 64 ? 4 : 8;
    const int G = 32 / KS;
    const int nq_tiles = (nq + 31) / 32, nt_tiles = (nt + 31) / 32;
    void *qf, *tf, *cst, *cand, *part;
    int rc;
    if ((rc = ws_get(ctx, WS_F16_Q, (size_t)batch * nq_tiles * KS * 64 * 16, &qf))) return rc;
    // + one group of fragments and constants behind the last train tile: the pass kernels always move whole groups
    if ((rc = ws_get(ctx, WS_F16_T, ((size_t)batch * nt_tiles + G) * KS * 64 * 16, &tf))) return rc;
    if ((rc = ws_get(ctx, WS_F16_CST, ((size_t)batch * (nq_tiles + nt_tiles) + G) * 128 * 4, &cst))) return rc;
    if ((rc = ws_get(ctx, WS_F16_CAND, (size_t)batch * nq * (kCandCap + 1) * 4 + 64, &cand))) return rc;
    float *qc = (float *)cst, *tc = qc + (size_t)batch * nq_tiles * 128;
    int *cand_cnt = (int *)cand, *cand_rows = cand_cnt + (size_t)batch * nq + 16;
    const double cerr = std::ldexp(1.0, -10) * 1.002 + 16.0 * KS * std::ldexp(1.0, -23) * 1.01 + (8.0 + dim / 4.0) * std::ldexp(1.0, -23) +
                        std::ldexp(1.0, -20);
    const double kappa = std::sqrt((double)dim) * std::ldexp(1.0, -36);
    const L2fPrepArgs qa{d_q, q_stride, q_bstride, nq, nq_tiles, (uint4 *)qf, qc}, ta{d_t, t_stride, t_bstride, nt, nt_tiles, (uint4 *)tf, tc};
    const dim3 pgrid((unsigned)std::max(nq_tiles, nt_tiles), batch, 2);
    if (KS == 4) hipLaunchKernelGGL(l2f_prep_kernel<4>, pgrid, dim3(256), 0, s, qa, ta, dim, (float)cerr, (float)kappa, gen, d_flags, cand_cnt);
    else hipLaunchKernelGGL(l2f_prep_kernel<8>, pgrid, dim3(256), 0, s, qa, ta, dim, (float)cerr, (float)kappa, gen, d_flags, cand_cnt);

    // one workgroup per CU or a little more: both operand sets come out of the L2 once per workgroup
    const int qblocks = (nq_tiles + kQTilesPerBlock - 1) / kQTilesPerBlock;
    const long long want_wgs = (long long)ctx->num_cus * (ctx->opt_l2_mfma_blocks_per_cu > 0 ? ctx->opt_l2_mfma_blocks_per_cu : 1);
    const long long want_splits = std::max<long long>(1, (want_wgs + (long long)qblocks * batch - 1) / ((long long)qblocks * batch));
    int tps = (int)std::max<long long>(1, (nt_tiles + want_splits - 1) / want_splits);
    tps = (tps + G - 1) / G * G;
    const int nsplit = (nt_tiles + tps - 1) / tps;
    const long long total = (long long)batch * nsplit * qblocks;
    if (total > 0x3FFFFFF0LL) {
        set_error("knn_l2 (fp16 path): problem too large");
        return MLPL_E_BAD_INPUT;
    }
    if ((rc = ws_get(ctx, WS_F16_PART, (size_t)batch * nsplit * nq * sizeof(float2), &part))) return rc;
    const unsigned grid = (unsigned)(((total + 7) / 8) * 8);
    const L2fArgs a{(const uint4 *)qf, qc, (const uint4 *)tf, tc, nq, nt, nq_tiles, nt_tiles, tps, nsplit, qblocks, batch, (float2 *)part, cand_cnt, cand_rows};
    prof_mark(ctx, MLPL_PROF_KNN_L2, 0, s);
    if (KS == 4) {
        hipLaunchKernelGGL((l2f_pass_kernel<4, false>), dim3(grid), dim3(256), 0, s, a, gen, (const int *)d_flags);
        hipLaunchKernelGGL((l2f_pass_kernel<4, true>), dim3(grid), dim3(256), 0, s, a, gen, (const int *)d_flags);
    } else {
        hipLaunchKernelGGL((l2f_pass_kernel<8, false>), dim3(grid), dim3(256), 0, s, a, gen, (const int *)d_flags);
        hipLaunchKernelGGL((l2f_pass_kernel<8, true>), dim3(grid), dim3(256), 0, s, a, gen, (const int *)d_flags);
    }
    prof_mark(ctx, MLPL_PROF_KNN_L2, 1, s);
    const bool vec = dim % 4 == 0 && ((uintptr_t)d_q | (uintptr_t)d_t) % 16 == 0 && (q_stride | q_bstride | t_stride | t_bstride) % 4 == 0;
    const L2fRerankArgs ra{d_q, q_stride, q_bstride, d_t, t_stride, t_bstride, nq, nt, dim, k, cand_cnt, cand_rows, d_idx, d_dist, d_flags, d_hint};
    const dim3 rgrid((nq + 15) / 16, batch);
    if (vec) hipLaunchKernelGGL(l2f_rerank_kernel<true>, rgrid, dim3(256), 0, s, ra, gen);
    else hipLaunchKernelGGL(l2f_rerank_kernel<false>, rgrid, dim3(256), 0, s, ra, gen);
    MLPL_HIP_TRY(hipGetLastError());
    return MLPL_OK;
}

}  // namespace mlpl
