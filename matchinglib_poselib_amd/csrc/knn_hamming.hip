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

__global__ void knn_hamming_merge_kernel(const uint2 *__restrict__ part, int nq, int nsplit, int rows_per_split,
                                         int dshift, int k, int32_t *__restrict__ idx, int32_t *__restrict__ dist) {
    const int b = blockIdx.y;
    const int qi = blockIdx.x * blockDim.x + threadIdx.x;
    if (qi >= nq) return;
    const uint32_t lmask = (1u << dshift) - 1u;
    unsigned long long b0 = ~0ull, b1 = ~0ull;
    for (int s = 0; s < nsplit; ++s) {
        const uint2 p = part[((size_t)b * nsplit + s) * nq + qi];
        const uint32_t keys[2] = {p.x, p.y};
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            if (keys[j] == 0xFFFFFFFFu) continue;
            const unsigned long long d = keys[j] >> dshift;
            const unsigned long long row = (unsigned long long)s * rows_per_split + (keys[j] & lmask);
            const unsigned long long g = (d << 32) | row;
            if (g < b0) {
                b1 = b0;
                b0 = g;
            } else if (g < b1) {
                b1 = g;
            }
        }
    }
    const size_t o = ((size_t)b * nq + qi) * k;
    idx[o] = (int32_t)(b0 & 0xFFFFFFFFull);
    dist[o] = (int32_t)(b0 >> 32);
    if (k == 2) {
        idx[o + 1] = (int32_t)(b1 & 0xFFFFFFFFull);
        dist[o + 1] = (int32_t)(b1 >> 32);
    }
}

template <int NW>
void launch_partial(dim3 grid, hipStream_t s, const uint32_t *q, size_t qbw, const uint32_t *t, size_t tbw, int nq, int nt,
                    int rps, int nsplit, uint2 *part) {
    hipLaunchKernelGGL(knn_hamming_partial_kernel<NW>, grid, dim3(kQueriesPerBlock), 0, s, q, qbw, t, tbw, nq, nt, rps,
                       nsplit, part);
}

}  // namespace

int launch_knn_hamming(mlpl_ctx *ctx, const uint8_t *d_q, int nq, size_t q_stride, size_t q_bstride,
                       const uint8_t *d_t, int nt, size_t t_stride, size_t t_bstride, int nbytes, int k, int batch,
                       int32_t *d_idx, int32_t *d_dist, hipStream_t s) {
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

    // split the train rows so that the grid holds ~8 blocks per CU
    const int qtiles = (nq + kQueriesPerBlock - 1) / kQueriesPerBlock;
    const long long target_blocks = 8LL * ctx->num_cus;
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
        case 1: launch_partial<1>(grid, s, qw, qbw, tw, tbw, nq, nt, rps, nsplit, (uint2 *)part); break;
        case 2: launch_partial<2>(grid, s, qw, qbw, tw, tbw, nq, nt, rps, nsplit, (uint2 *)part); break;
        case 4: launch_partial<4>(grid, s, qw, qbw, tw, tbw, nq, nt, rps, nsplit, (uint2 *)part); break;
        case 8: launch_partial<8>(grid, s, qw, qbw, tw, tbw, nq, nt, rps, nsplit, (uint2 *)part); break;
        case 16: launch_partial<16>(grid, s, qw, qbw, tw, tbw, nq, nt, rps, nsplit, (uint2 *)part); break;
        case 32: launch_partial<32>(grid, s, qw, qbw, tw, tbw, nq, nt, rps, nsplit, (uint2 *)part); break;
        case 64: launch_partial<64>(grid, s, qw, qbw, tw, tbw, nq, nt, rps, nsplit, (uint2 *)part); break;
        default: set_error("knn_hamming: unsupported descriptor width %d bytes", nbytes); return MLPL_E_BAD_INPUT;
    }
    prof_mark(ctx, MLPL_PROF_KNN_HAMMING, 1, s);
    dim3 mgrid((nq + 255) / 256, batch);
    hipLaunchKernelGGL(knn_hamming_merge_kernel, mgrid, dim3(256), 0, s, (const uint2 *)part, nq, nsplit, rps, dshift, k,
                       d_idx, d_dist);
    MLPL_HIP_TRY(hipGetLastError());
    return MLPL_OK;
}

}  // namespace mlpl
