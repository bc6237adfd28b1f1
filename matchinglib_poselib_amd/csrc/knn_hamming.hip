// knn_hamming.hip -- exact brute-force 2-NN under bit-Hamming distance for gfx950 (MI355X).
//
// Replaces cvflann::Index<HammingLUT>(LinearIndexParams).knnSearch as called by
// matchinglib::getMatches(...,"LINEAR",...) -- reference matchinglib/source/matchers.cpp:567-588.
// Result per query = the two lexicographically smallest (distance, trainIdx) pairs (cvflann
// KNNUniqueResultSet order), bit-exact.
//
// Mapping (CDNA4):
//   * one query per lane, its descriptor held in NW VGPRs for the whole kernel;
//   * the train set is cut into `nsplit` row ranges; block (x=query tile, y=split, z=batch item) stages its
//     rows TILE_ROWS at a time into LDS with coalesced 16-byte global loads and reads them back as
//     wave-uniform (broadcast, conflict-free) ds_read_b128;
//   * per (query, train row): NW x (v_xor_b32 + v_bcnt_u32_b32 accumulate), then the running top-2 is kept on
//     packed keys  key = dist << dshift | local_row  with  k1 = med3(k0,k1,key); k0 = min(k0,key)
//     -- a min over keys, never "first lane wins", so ties resolve to the smaller train index;
//   * partial top-2 per (split, query) goes to a [batch][split][nq] uint2 table (coalesced), merged by
//     knn_hamming_merge_kernel on 64-bit (dist, global row) keys.
// The kernel is integer-VALU bound (NW*2+3 VALU ops per descriptor pair); HBM traffic is compulsory only.

#include "mlpl_internal.h"

namespace mlpl {

namespace {

constexpr int kQueriesPerBlock = 256;
constexpr int kTileRows = 128;
constexpr int kMergeGroup = 64;                    // queries per merge block (= kCountGroup of ratio_write_kernel)
static_assert(kMergeGroup == kCountGroup, "merge kernel and ratio_write_kernel disagree on the count granularity");

// bits needed for a distance in [0, 32*nw]
constexpr int dist_bits(int nw) {
    int b = 1;
    while ((1 << b) <= nw * 32) ++b;
    return b;
}

// popcount(x) + acc in one VALU op (hipcc otherwise splits it into v_bcnt x,0 + v_add3)
__device__ __forceinline__ uint32_t bcnt_acc(uint32_t x, uint32_t acc) {
    uint32_t r;
    asm("v_bcnt_u32_b32 %0, %1, %2" : "=v"(r) : "v"(x), "v"(acc));
    return r;
}

__device__ __forceinline__ uint32_t umed3(uint32_t a, uint32_t b, uint32_t c) {
    uint32_t r;
    asm("v_med3_u32 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "v"(c));
    return r;
}

// Gathers rows of `nbytes` bytes at arbitrary byte stride into zero-padded rows of `nw` 32-bit words.
__global__ void pack_rows_u8_kernel(const uint8_t *__restrict__ src, size_t stride, size_t bstride, int n, int nbytes,
                                    int nw, uint32_t *__restrict__ dst) {
    const int b = blockIdx.y;
    const long long total = (long long)n * nw;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < total;
         i += (long long)gridDim.x * blockDim.x) {
        const int row = (int)(i / nw);
        const int w = (int)(i - (long long)row * nw);
        const uint8_t *p = src + (size_t)b * bstride + (size_t)row * stride + (size_t)w * 4;
        uint32_t v = 0;
#pragma unroll
        for (int j = 0; j < 4; ++j) {
            if (w * 4 + j < nbytes) v |= (uint32_t)p[j] << (8 * j);
        }
        dst[(size_t)b * total + i] = v;
    }
}

template <int NW>
__global__ __launch_bounds__(kQueriesPerBlock) void knn_hamming_partial_kernel(
    const uint32_t *__restrict__ q, size_t q_bstride_w, const uint32_t *__restrict__ t, size_t t_bstride_w, int nq,
    int nt, int rows_per_split, int nsplit, uint2 *__restrict__ part) {
    constexpr int dshift = 32 - dist_bits(NW);
    constexpr int VW = (NW % 4 == 0) ? 4 : ((NW % 2 == 0) ? 2 : 1);  // words per LDS/global vector access
    __shared__ __attribute__((aligned(16))) uint32_t tile[kTileRows * NW];

    const int tid = threadIdx.x;
    const int split = blockIdx.y;
    const int b = blockIdx.z;
    const int qi = blockIdx.x * kQueriesPerBlock + tid;
    q += (size_t)b * q_bstride_w;
    t += (size_t)b * t_bstride_w;

    uint32_t qa[NW];
    if (qi < nq) {
        const uint32_t *qp = q + (size_t)qi * NW;
        if constexpr (VW == 4) {
#pragma unroll
            for (int w = 0; w < NW; w += 4) {
                const uint4 v = *reinterpret_cast<const uint4 *>(qp + w);
                qa[w] = v.x, qa[w + 1] = v.y, qa[w + 2] = v.z, qa[w + 3] = v.w;
            }
        } else {
#pragma unroll
            for (int w = 0; w < NW; ++w) qa[w] = qp[w];
        }
    } else {
#pragma unroll
        for (int w = 0; w < NW; ++w) qa[w] = 0;
    }

    uint32_t k0 = 0xFFFFFFFFu, k1 = 0xFFFFFFFFu;
    const int r_begin = split * rows_per_split;
    const int r_end = min(nt, r_begin + rows_per_split);

    for (int base = r_begin; base < r_end; base += kTileRows) {
        const int rows = min(kTileRows, r_end - base);
        __syncthreads();  // the previous tile has been consumed by every wave
        {
            const uint32_t *src = t + (size_t)base * NW;
            const int nvec = rows * NW / VW;
            if constexpr (VW == 4) {
                for (int i = tid; i < nvec; i += kQueriesPerBlock)
                    reinterpret_cast<uint4 *>(tile)[i] = reinterpret_cast<const uint4 *>(src)[i];
            } else if constexpr (VW == 2) {
                for (int i = tid; i < nvec; i += kQueriesPerBlock)
                    reinterpret_cast<uint2 *>(tile)[i] = reinterpret_cast<const uint2 *>(src)[i];
            } else {
                for (int i = tid; i < nvec; i += kQueriesPerBlock) tile[i] = src[i];
            }
        }
        __syncthreads();

        const uint32_t lbase = (uint32_t)(base - r_begin);
        auto one_row = [&](int r) {
            uint32_t d = 0;
            if constexpr (VW == 4) {
#pragma unroll
                for (int w = 0; w < NW; w += 4) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(&tile[r * NW + w]);
                    d = bcnt_acc(qa[w] ^ v.x, d);
                    d = bcnt_acc(qa[w + 1] ^ v.y, d);
                    d = bcnt_acc(qa[w + 2] ^ v.z, d);
                    d = bcnt_acc(qa[w + 3] ^ v.w, d);
                }
            } else if constexpr (VW == 2) {
#pragma unroll
                for (int w = 0; w < NW; w += 2) {
                    const uint2 v = *reinterpret_cast<const uint2 *>(&tile[r * NW + w]);
                    d = bcnt_acc(qa[w] ^ v.x, d);
                    d = bcnt_acc(qa[w + 1] ^ v.y, d);
                }
            } else {
#pragma unroll
                for (int w = 0; w < NW; ++w) d = bcnt_acc(qa[w] ^ tile[r * NW + w], d);
            }
            const uint32_t key = (d << dshift) | (lbase + (uint32_t)r);
            k1 = umed3(k0, k1, key);
            k0 = min(k0, key);
        };
        if (rows == kTileRows) {
#pragma unroll 8
            for (int r = 0; r < kTileRows; ++r) one_row(r);
        } else {
            for (int r = 0; r < rows; ++r) one_row(r);
        }
    }
    if (qi < nq) part[((size_t)b * nsplit + split) * nq + qi] = make_uint2(k0, k1);
}

// Variant B: the train rows are wave-uniform operands, so they are fetched with SCALAR loads (s_load_dwordx8 through
// the scalar data cache) straight into SGPRs and fed to v_xor_b32 as the scalar source: no LDS traffic, no barriers.
// (A broadcast ds_read_b128 delivers 1 KiB per wave-instruction for 16 useful bytes and saturates the CU's LDS at
// ~84 % with four SIMDs issuing; the scalar path leaves the VALU as the only bound.)  Q queries per lane.
template <int NW, int Q, int BT>
__global__ __launch_bounds__(BT) void knn_hamming_partial_sgpr_kernel(
    const uint32_t *__restrict__ q, size_t q_bstride_w, const uint32_t *__restrict__ t, size_t t_bstride_w, int nq,
    int nt, int rows_per_split, int nsplit, uint2 *__restrict__ part) {
    constexpr int dshift = 32 - dist_bits(NW);
    constexpr int RB = (NW <= 8) ? 4 : ((NW <= 16) ? 2 : 1);  // rows per scalar-load batch (<= 32 SGPRs)
    const int tid = threadIdx.x;
    const int split = blockIdx.y;
    const int b = blockIdx.z;
    q += (size_t)b * q_bstride_w;

    uint32_t qa[Q][NW];
    int qidx[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) {
        qidx[j] = (blockIdx.x * Q + j) * BT + tid;
        if (qidx[j] < nq) {
            const uint32_t *qp = q + (size_t)qidx[j] * NW;
            if constexpr (NW % 4 == 0) {
#pragma unroll
                for (int w = 0; w < NW; w += 4) {
                    const uint4 v = *reinterpret_cast<const uint4 *>(qp + w);
                    qa[j][w] = v.x, qa[j][w + 1] = v.y, qa[j][w + 2] = v.z, qa[j][w + 3] = v.w;
                }
            } else {
#pragma unroll
                for (int w = 0; w < NW; ++w) qa[j][w] = qp[w];
            }
        } else {
#pragma unroll
            for (int w = 0; w < NW; ++w) qa[j][w] = 0;
        }
    }
    uint32_t k0[Q], k1[Q];
#pragma unroll
    for (int j = 0; j < Q; ++j) k0[j] = k1[j] = 0xFFFFFFFFu;

    const int r_begin = split * rows_per_split;
    const int rows = min(nt, r_begin + rows_per_split) - r_begin;
    const uint32_t *__restrict__ tp = t + (size_t)b * t_bstride_w + (size_t)r_begin * NW;  // wave-uniform

    auto score_row = [&](const uint32_t(&tw)[NW], uint32_t lrow) {
#pragma unroll
        for (int j = 0; j < Q; ++j) {
            uint32_t d = 0;
#pragma unroll
            for (int w = 0; w < NW; ++w) d = bcnt_acc(qa[j][w] ^ tw[w], d);
            const uint32_t key = (d << dshift) | lrow;
            k1[j] = umed3(k0[j], k1[j], key);
            k0[j] = min(k0[j], key);
        }
    };

    int r = 0;
    for (; r + RB <= rows; r += RB) {
        uint32_t tw[RB][NW];
#pragma unroll
        for (int i = 0; i < RB; ++i)
#pragma unroll
            for (int w = 0; w < NW; ++w) tw[i][w] = tp[(size_t)(r + i) * NW + w];
#pragma unroll
        for (int i = 0; i < RB; ++i) score_row(tw[i], (uint32_t)(r + i));
    }
    for (; r < rows; ++r) {
        uint32_t tw[NW];
#pragma unroll
        for (int w = 0; w < NW; ++w) tw[w] = tp[(size_t)r * NW + w];
        score_row(tw, (uint32_t)r);
    }
#pragma unroll
    for (int j = 0; j < Q; ++j)
        if (qidx[j] < nq) part[((size_t)b * nsplit + split) * nq + qidx[j]] = make_uint2(k0[j], k1[j]);
}

struct MergeTail {  // train rows [row0, row0 + n) to be folded in by the merge itself (n = 0: none)
    const uint32_t *qw, *tw;
    size_t q_batch_words, t_batch_words;
    int nw, row0, n;
    const int32_t *split_tile0;  // != NULL: split s of pair b starts at train row 32 * split_tile0[b * (nsplit + 1) + s]
};

// Merge of the per-split partial top-2 lists, LANES lanes per query (each lane folds every LANES-th split, then xor-shuffles
// combine the lanes), on 64-bit (dist << 32 | global row) keys; kMergeGroup queries per block (block = kMergeGroup * LANES threads).
// Also evaluates the ratio predicate and leaves the number of passing queries of this query group in group_counts (consumed by
// ratio_write_kernel), so the fused getMatches path needs no separate counting pass.  LANES = 4 serves the usual handful of
// splits (a quarter of the threads and shuffles of the 16-lane form), LANES = 16 many splits.
// Emission inside the merge (round 5, the latency shape: ONE image pair per call).  When `out` is set the merge kernel also writes the
// DMatch rows -- the work of ratio_write_kernel -- so the step has one launch less: a workgroup (64 queries) publishes its pass count as
// (launch generation << 32 | count) in scan[pair][workgroup], sums the counts of the workgroups before it (spinning on entries that
// still carry an older generation: they belong to workgroups with a smaller linear id, which were dispatched earlier) and writes its
// matches at that offset in query order.  The launcher uses this only for grids that are resident all at once.  Same rows as the
// two-kernel path (same predicate, same order).
struct MergeEmit {
    mlpl_dmatch *out;
    int32_t *n_out;
    unsigned long long *scan;
    uint32_t gen;
};

template <int LANES>
__global__ __launch_bounds__(kMergeGroup * LANES) void knn_hamming_merge_kernel(const uint2 *__restrict__ part, int nq, int nsplit,
                                                                                 int rows_per_split, int sps, int dshift, int k, float ratio,
                                                                                 int32_t *__restrict__ idx, int32_t *__restrict__ dist,
                                                                                 int32_t *__restrict__ group_counts, MergeTail tail, MergeEmit emit) {
    constexpr int kWaves = kMergeGroup * LANES / 64;
    __shared__ int wave_tot[kWaves];
    __shared__ int wave_look[kWaves];
    const int b = blockIdx.y;
    const int tid = threadIdx.x;
    const int sub = tid & (LANES - 1);
    const int qi = blockIdx.x * kMergeGroup + tid / LANES;
    const uint32_t lmask = (1u << dshift) - 1u;
    unsigned long long b0 = ~0ull, b1 = ~0ull;
    auto upd = [&](unsigned long long g) {
        const bool lt0 = g < b0, lt1 = g < b1;
        b1 = lt0 ? b0 : (lt1 ? g : b1);
        b0 = lt0 ? g : b0;
    };
    if (qi < nq) {
        for (int s = sub; s < nsplit; s += LANES) {
            const uint2 p = part[((size_t)b * nsplit + s) * nq + qi];
            // sps partial slots share one row origin (dynamic splits: the members of a team; static splits: sps = 1)
            const unsigned long long base = tail.split_tile0 ? 32ull * (unsigned long long)tail.split_tile0[(size_t)b * (nsplit + 1) + s]
                                                             : (unsigned long long)(sps == 1 ? s : s / sps) * rows_per_split;
            if (p.x != 0xFFFFFFFFu) upd(((unsigned long long)(p.x >> dshift) << 32) | (base + (p.x & lmask)));
            if (p.y != 0xFFFFFFFFu) upd(((unsigned long long)(p.y >> dshift) << 32) | (base + (p.y & lmask)));
        }
    }
    // train rows the partial kernels did not see (the dynamic-split matrix-core kernel works on whole 32-row tiles only): < 32 rows,
    // exact xor/popcount here, the lanes of a query take them in turn
    if (tail.n > 0 && qi < nq) {
        const uint32_t *qrow = tail.qw + (size_t)b * tail.q_batch_words + (size_t)qi * tail.nw;
        for (int r = sub; r < tail.n; r += LANES) {
            const uint32_t *trow = tail.tw + (size_t)b * tail.t_batch_words + (size_t)(tail.row0 + r) * tail.nw;
            uint32_t d = 0;
            for (int w = 0; w < tail.nw; ++w) d += __popc(qrow[w] ^ trow[w]);
            upd(((unsigned long long)d << 32) | (unsigned long long)(tail.row0 + r));
        }
    }
#pragma unroll
    for (int off = 1; off < LANES; off <<= 1) {
        const unsigned long long o0 = __shfl_xor(b0, off), o1 = __shfl_xor(b1, off);
        upd(o0);
        upd(o1);
    }
    bool pass = false;
    if (sub == 0 && qi < nq) {
        const size_t o = ((size_t)b * nq + qi) * k;
        const int d0 = (int32_t)(b0 >> 32);
        idx[o] = (int32_t)(b0 & 0xFFFFFFFFull);
        dist[o] = d0;
        if (k == 2) {
            const int d1 = (int32_t)(b1 >> 32);
            idx[o + 1] = (int32_t)(b1 & 0xFFFFFFFFull);
            dist[o + 1] = d1;
            pass = (float)d0 < __fmul_rn(ratio, (float)d1);
        } else {
            pass = true;
        }
    }
    if (group_counts || emit.out) {
        const unsigned long long bal = __ballot(pass);
        if ((tid & 63) == 0) wave_tot[tid >> 6] = __popcll(bal);
        __syncthreads();
        int tot = 0, wave_prefix = 0;
#pragma unroll
        for (int w = 0; w < kWaves; ++w) {
            if (w < (tid >> 6)) wave_prefix += wave_tot[w];
            tot += wave_tot[w];
        }
        if (tid == 0 && group_counts) group_counts[(size_t)b * gridDim.x + blockIdx.x] = tot;
        if (emit.out) {
            unsigned long long *sc = emit.scan + (size_t)b * gridDim.x;
            if (tid == 0)
                __hip_atomic_store(&sc[blockIdx.x], ((unsigned long long)emit.gen << 32) | (unsigned long long)(uint32_t)tot, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            int look = 0;
            for (int j = tid; j < (int)blockIdx.x; j += kMergeGroup * LANES) {
                unsigned long long v = __hip_atomic_load(&sc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                while ((uint32_t)(v >> 32) != emit.gen) {
                    __builtin_amdgcn_s_sleep(2);
                    v = __hip_atomic_load(&sc[j], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                }
                look += (int)(uint32_t)v;
            }
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) look += __shfl_xor(look, off);
            if ((tid & 63) == 0) wave_look[tid >> 6] = look;
            __syncthreads();
            int base = 0;
#pragma unroll
            for (int w = 0; w < kWaves; ++w) base += wave_look[w];
            if (pass) {  // (only lanes with sub == 0 pass; a wave's queries are in ascending order)
                mlpl_dmatch m;
                m.queryIdx = qi;
                m.trainIdx = (int32_t)(b0 & 0xFFFFFFFFull);
                m.imgIdx = -1;
                m.distance = (float)(int32_t)(b0 >> 32);
                emit.out[(size_t)b * nq + base + wave_prefix + __popcll(bal & ((1ull << (tid & 63)) - 1ull))] = m;
            }
            if (blockIdx.x == gridDim.x - 1 && tid == 0) emit.n_out[b] = base + tot;
        }
    }
}

static void launch_merge(hipStream_t s, const uint2 *part, int nq, int nsplit, int rps, int dshift, int k, float ratio, int batch,
                         int32_t *d_idx, int32_t *d_dist, int32_t *d_group_counts, int sps = 1,
                         MergeTail tail = MergeTail{nullptr, nullptr, 0, 0, 0, 0, 0, nullptr}, MergeEmit emit = MergeEmit{nullptr, nullptr, nullptr, 0u}) {
    dim3 mgrid((nq + kMergeGroup - 1) / kMergeGroup, batch);
    if (nsplit <= 8)
        hipLaunchKernelGGL(knn_hamming_merge_kernel<4>, mgrid, dim3(kMergeGroup * 4), 0, s, part, nq, nsplit, rps, sps, dshift, k, ratio, d_idx,
                           d_dist, d_group_counts, tail, emit);
    else
        hipLaunchKernelGGL(knn_hamming_merge_kernel<16>, mgrid, dim3(kMergeGroup * 16), 0, s, part, nq, nsplit, rps, sps, dshift, k, ratio,
                           d_idx, d_dist, d_group_counts, tail, emit);
}

template <int NW>
void launch_partial(int variant, int qpl, dim3 grid, hipStream_t s, const uint32_t *q, size_t qbw, const uint32_t *t,
                    size_t tbw, int nq, int nt, int rps, int nsplit, uint2 *part) {
    if (variant == 0) {
        hipLaunchKernelGGL(knn_hamming_partial_kernel<NW>, grid, dim3(kQueriesPerBlock), 0, s, q, qbw, t, tbw, nq, nt, rps,
                           nsplit, part);
    } else if (variant == 2) {  // one wave per block: finer scheduling, no co-resident waves needed per block
        hipLaunchKernelGGL((knn_hamming_partial_sgpr_kernel<NW, 1, 64>), grid, dim3(64), 0, s, q, qbw, t, tbw, nq, nt, rps,
                           nsplit, part);
    } else if constexpr (NW <= 16) {
        if (qpl == 2)
            hipLaunchKernelGGL((knn_hamming_partial_sgpr_kernel<NW, 2, kQueriesPerBlock>), grid, dim3(kQueriesPerBlock), 0, s, q,
                               qbw, t, tbw, nq, nt, rps, nsplit, part);
        else
            hipLaunchKernelGGL((knn_hamming_partial_sgpr_kernel<NW, 1, kQueriesPerBlock>), grid, dim3(kQueriesPerBlock), 0, s, q,
                               qbw, t, tbw, nq, nt, rps, nsplit, part);
    } else {
        hipLaunchKernelGGL((knn_hamming_partial_sgpr_kernel<NW, 1, kQueriesPerBlock>), grid, dim3(kQueriesPerBlock), 0, s, q, qbw,
                           t, tbw, nq, nt, rps, nsplit, part);
    }
}

}  // namespace

int launch_knn_hamming_mfma(mlpl_ctx *ctx, const uint32_t *qw, size_t q_batch_words, const uint32_t *tw, size_t t_batch_words,
                            int nq, int nt, int nw, int batch, int dshift, hipStream_t s, int *rps_out, int *nsplit_out,
                            int *sps_out, int *tail_row0_out, const int32_t **split_tab_out, uint2 **part_out, int k, float ratio,
                            int32_t *d_idx, int32_t *d_dist, int32_t *d_group_counts, int *fused_out);

int launch_knn_hamming(mlpl_ctx *ctx, const uint8_t *d_q, int nq, size_t q_stride, size_t q_bstride,
                       const uint8_t *d_t, int nt, size_t t_stride, size_t t_bstride, int nbytes, int k, int batch,
                       int32_t *d_idx, int32_t *d_dist, hipStream_t s, float ratio, int32_t *d_group_counts, HammingEmitOut *emit_out) {
    if (emit_out) emit_out->emitted = 0;
    if (!d_q || !d_t || !d_idx || !d_dist || nq < 0 || batch < 1 || batch > 65535 || (k != 1 && k != 2) || nt < k ||
        nbytes < 1 || nbytes > 256 || q_stride < (size_t)nbytes || t_stride < (size_t)nbytes) {
        set_error("knn_hamming: bad arguments (nq=%d nt=%d nbytes=%d k=%d batch=%d)", nq, nt, nbytes, k, batch);
        return MLPL_E_BAD_INPUT;
    }
    if (nq == 0) return MLPL_OK;

    // descriptor words, padded to an instantiated width (zero padding does not change the distance)
    const int nw_raw = (nbytes + 3) / 4;
    int nw = 1;
    while (nw < nw_raw) nw *= 2;
    const int dshift = 32 - dist_bits(nw);  // distances are in [0, nw*32]

    const uint32_t *qw = nullptr, *tw = nullptr;
    size_t qbw = 0, tbw = 0;
    auto canonical = [&](const uint8_t *p, size_t stride, size_t bstride, int n) {
        return stride == (size_t)nw * 4 && (size_t)nbytes == stride && (reinterpret_cast<uintptr_t>(p) % 16 == 0) &&
               (batch == 1 || (bstride % 16 == 0 && bstride >= (size_t)n * stride));
    };
    if (canonical(d_q, q_stride, q_bstride, nq)) {
        qw = reinterpret_cast<const uint32_t *>(d_q);
        qbw = q_bstride / 4;
    } else {
        void *buf = nullptr;
        int rc = ws_get(ctx, WS_PACK_Q, (size_t)batch * nq * nw * 4, &buf);
        if (rc) return rc;
        const long long total = (long long)nq * nw;
        dim3 g((unsigned)std::min<long long>((total + 255) / 256, 4096), batch);
        hipLaunchKernelGGL(pack_rows_u8_kernel, g, dim3(256), 0, s, d_q, q_stride, q_bstride, nq, nbytes, nw,
                           (uint32_t *)buf);
        qw = (const uint32_t *)buf;
        qbw = (size_t)nq * nw;
    }
    if (canonical(d_t, t_stride, t_bstride, nt)) {
        tw = reinterpret_cast<const uint32_t *>(d_t);
        tbw = t_bstride / 4;
    } else {
        void *buf = nullptr;
        int rc = ws_get(ctx, WS_PACK_T, (size_t)batch * nt * nw * 4, &buf);
        if (rc) return rc;
        const long long total = (long long)nt * nw;
        dim3 g((unsigned)std::min<long long>((total + 255) / 256, 4096), batch);
        hipLaunchKernelGGL(pack_rows_u8_kernel, g, dim3(256), 0, s, d_t, t_stride, t_bstride, nt, nbytes, nw,
                           (uint32_t *)buf);
        tw = (const uint32_t *)buf;
        tbw = (size_t)nt * nw;
    }

    int variant = ctx->opt_hamming_variant;
    if (variant == 3 && nw > 16) variant = 0;  // descriptors above 64 bytes: LDS-tiled VALU kernel
    if (variant == 3) {  // matrix-core kernel (knn_hamming_mfma.hip); wider descriptors take the VALU kernels
        int rps = 0, nsplit = 0, sps = 1, tail_row0 = nt;
        const int32_t *split_tab = nullptr;
        uint2 *part = nullptr;
        int fused = 0;
        int rc = launch_knn_hamming_mfma(ctx, qw, qbw, tw, tbw, nq, nt, nw, batch, dshift, s, &rps, &nsplit, &sps, &tail_row0, &split_tab,
                                         &part, k, ratio, d_idx, d_dist, d_group_counts, &fused);
        if (rc) return rc;
        if (!fused) {  // (the static LDS-ring kernel merges its splits, evaluates the ratio predicate and counts by itself)
            const MergeTail tail{qw, tw, qbw, tbw, nw, tail_row0, nt - tail_row0, split_tab};
            MergeEmit emit{nullptr, nullptr, nullptr, 0u};
            const long long mblocks = (long long)((nq + kMergeGroup - 1) / kMergeGroup) * batch;
            // the merge also emits the DMatch rows when the caller wants them and the whole merge grid is resident at once (the latency
            // shape: one or two image pairs; 1024-thread workgroups at 16 lanes per query: two per CU)
            if (emit_out && emit_out->out && emit_out->n_out && ctx->opt_hamming_merge_emit && mblocks <= 2LL * ctx->num_cus) {
                void *sp = nullptr;
                const size_t sb = (size_t)mblocks * sizeof(unsigned long long);
                if ((rc = ws_get(ctx, WS_SCAN, sb, &sp))) return rc;
                // new block (or the generation wraps): no stale generation in it.  A block that was freed and re-allocated may come back at the SAME
                // address: its size cannot -- a regrow only ever happens to a larger size -- so the size is part of the test.
                if (ctx->hamming_scan_ptr != sp || ctx->hamming_scan_bytes != ctx->ws_bytes[WS_SCAN] || ctx->hamming_scan_gen == 0xFFFFFFFFu) {
                    MLPL_HIP_TRY(hipMemsetAsync(sp, 0, ctx->ws_bytes[WS_SCAN], s));
                    ctx->hamming_scan_ptr = sp, ctx->hamming_scan_bytes = ctx->ws_bytes[WS_SCAN], ctx->hamming_scan_gen = 0;
                }
                emit = MergeEmit{emit_out->out, emit_out->n_out, (unsigned long long *)sp, ++ctx->hamming_scan_gen};
                emit_out->emitted = 1;
            }
            launch_merge(s, (const uint2 *)part, nq, nsplit, rps, dshift, k, ratio, batch, d_idx, d_dist, d_group_counts, sps, tail, emit);
        }
        MLPL_HIP_TRY(hipGetLastError());
        return MLPL_OK;
    }

    // split the train rows so that the grid holds ~tune_blocks_per_cu blocks per CU
    int qpl = (variant == 1 && nw <= 16) ? ctx->opt_hamming_qpl : 1;  // queries per lane
    if (qpl == 2 && nq <= kQueriesPerBlock * 32) qpl = 1;                // too few queries to afford it
    const int qpb = (variant == 2) ? 64 : kQueriesPerBlock * qpl;  // queries per block
    const int qtiles = (nq + qpb - 1) / qpb;
    const long long target_blocks = (long long)ctx->opt_hamming_blocks_per_cu * ctx->num_cus;
    const int max_split = (nt + kTileRows - 1) / kTileRows;
    long long want = (target_blocks + (long long)qtiles * batch - 1) / ((long long)qtiles * batch);
    int nsplit = (int)std::max<long long>(1, std::min<long long>(want, max_split));
    int rps = (nt + nsplit - 1) / nsplit;
    rps = ((rps + kTileRows - 1) / kTileRows) * kTileRows;
    const int max_rps = ((1 << dshift) - 2) / kTileRows * kTileRows;  // local row must fit below the sentinel key
    if (rps > max_rps) rps = max_rps;
    nsplit = (nt + rps - 1) / rps;
    if (nsplit > 65535) {
        set_error("knn_hamming: train set too large (nt=%d)", nt);
        return MLPL_E_BAD_INPUT;
    }

    void *part = nullptr;
    int rc = ws_get(ctx, WS_PARTIAL, (size_t)batch * nsplit * nq * sizeof(uint2), &part);
    if (rc) return rc;

    dim3 grid(qtiles, nsplit, batch);
    prof_mark(ctx, MLPL_PROF_KNN_HAMMING, 0, s);
    switch (nw) {
        case 1: launch_partial<1>(variant, qpl, grid, s, qw, qbw, tw, tbw, nq, nt, rps, nsplit, (uint2 *)part); break;
        case 2: launch_partial<2>(variant, qpl, grid, s, qw, qbw, tw, tbw, nq, nt, rps, nsplit, (uint2 *)part); break;
        case 4: launch_partial<4>(variant, qpl, grid, s, qw, qbw, tw, tbw, nq, nt, rps, nsplit, (uint2 *)part); break;
        case 8: launch_partial<8>(variant, qpl, grid, s, qw, qbw, tw, tbw, nq, nt, rps, nsplit, (uint2 *)part); break;
        case 16: launch_partial<16>(variant, qpl, grid, s, qw, qbw, tw, tbw, nq, nt, rps, nsplit, (uint2 *)part); break;
        case 32: launch_partial<32>(variant, qpl, grid, s, qw, qbw, tw, tbw, nq, nt, rps, nsplit, (uint2 *)part); break;
        case 64: launch_partial<64>(variant, qpl, grid, s, qw, qbw, tw, tbw, nq, nt, rps, nsplit, (uint2 *)part); break;
        default: set_error("knn_hamming: unsupported descriptor width %d bytes", nbytes); return MLPL_E_BAD_INPUT;
    }
    prof_mark(ctx, MLPL_PROF_KNN_HAMMING, 1, s);
    launch_merge(s, (const uint2 *)part, nq, nsplit, rps, dshift, k, ratio, batch, d_idx, d_dist, d_group_counts);
    MLPL_HIP_TRY(hipGetLastError());
    return MLPL_OK;
}

}  // namespace mlpl
