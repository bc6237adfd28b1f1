/*
 * knn_oracle.c -- CPU restatement of matchinglib::getMatches(..., "LINEAR", ...).
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Plain C, single thread, no dependencies.
 *
 * Follows /root/reference/matchinglib_poselib/source/matchinglib/source/matchers.cpp:
 *   :115-135  input validation and return codes
 *   :525-547  nn = ratioTest ? 2 : 1 ; dtype check (CV_8U / CV_32F)
 *   :565-631  CV_8U: cvflann::Index<HammingLUT>(LinearIndexParams) + knnSearch + ratio loop
 *   :632-707  CV_32F: cvflann::Index<L2<float>>(LinearIndexParams) + knnSearch + ratio loop
 *   :709-713  < MIN_FINAL_MATCHES(2) matches -> -3   (match_statOptFlow.h:68)
 *
 * The callee cvflann (OpenCV 4.2.0 modules/flann, pinned by ci/make_opencv.sh:6) is not vendored in
 * the reference tree.  Its published behaviour, restated here:
 *   - NNIndex::knnSearch loops queries serially with a KNNUniqueResultSet(knn): an ordered set of
 *     (dist, index) pairs; addPoint() drops a candidate when dist >= worst_distance once the set is
 *     full, otherwise inserts and evicts the largest pair.  LinearIndex::findNeighbors visits train
 *     rows in ascending order.  Net effect: the k lexicographically smallest (dist, trainIdx) pairs,
 *     sorted ascending.
 *   - HammingLUT: per-byte popcount lookup table summed over the descriptor bytes (int result).
 *   - L2<float>: squared Euclidean distance, four differences per step:
 *         result += d0*d0 + d1*d1 + d2*d2 + d3*d3      (left to right, float)
 *     then a scalar tail; no early exit because worst_dist defaults to -1.
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

static uint8_t g_lut[256];
static int g_lut_ready = 0;

static void lut_init(void) {
    if (g_lut_ready) return;
    for (int v = 0; v < 256; ++v) {
        int c = 0;
        for (int b = 0; b < 8; ++b) c += (v >> b) & 1;
        g_lut[v] = (uint8_t)c;
    }
    g_lut_ready = 1;
}

/* KNNUniqueResultSet for k <= 2 kept as a tiny sorted array. */
typedef struct {
    int cap, size, full;
    double worst; /* DistanceType max until full */
    double d[3];
    int i[3];
} result_set;

static void rs_clear(result_set *rs, int cap) {
    rs->cap = cap;
    rs->size = 0;
    rs->full = 0;
    rs->worst = 1.0e300;
}

static void rs_add(result_set *rs, double dist, int index) {
    if (dist >= rs->worst) return;
    /* insert keeping (dist, index) ascending */
    int pos = rs->size;
    while (pos > 0 && (rs->d[pos - 1] > dist || (rs->d[pos - 1] == dist && rs->i[pos - 1] > index))) {
        rs->d[pos] = rs->d[pos - 1];
        rs->i[pos] = rs->i[pos - 1];
        --pos;
    }
    rs->d[pos] = dist;
    rs->i[pos] = index;
    rs->size++;
    if (rs->full) {
        if (rs->size > rs->cap) {
            rs->size--; /* erase the largest */
            rs->worst = rs->d[rs->size - 1];
        }
    } else if (rs->size == rs->cap) {
        rs->full = 1;
        rs->worst = rs->d[rs->size - 1];
    }
}

int oracle_knn_hamming(const uint8_t *q, int nq, size_t q_stride, const uint8_t *t, int nt, size_t t_stride,
                       int nbytes, int k, int32_t *idx, int32_t *dist) {
    if (!q || !t || !idx || !dist || nq < 0 || nt < k || nbytes <= 0 || (k != 1 && k != 2)) return -1;
    lut_init();
    result_set rs;
    for (int qi = 0; qi < nq; ++qi) {
        const uint8_t *a = q + (size_t)qi * q_stride;
        rs_clear(&rs, k);
        for (int ti = 0; ti < nt; ++ti) {
            const uint8_t *b = t + (size_t)ti * t_stride;
            int d = 0;
            for (int j = 0; j < nbytes; ++j) d += g_lut[a[j] ^ b[j]];
            rs_add(&rs, (double)d, ti);
        }
        for (int j = 0; j < k; ++j) {
            idx[(size_t)qi * k + j] = rs.i[j];
            dist[(size_t)qi * k + j] = (int32_t)rs.d[j];
        }
    }
    return 0;
}

static float l2sq_f32(const float *a, const float *b, int size) {
    float result = 0.0f;
    float diff0, diff1, diff2, diff3;
    const float *last = a + size;
    const float *lastgroup = last - 3;
    while (a < lastgroup) {
        diff0 = a[0] - b[0];
        diff1 = a[1] - b[1];
        diff2 = a[2] - b[2];
        diff3 = a[3] - b[3];
        result += diff0 * diff0 + diff1 * diff1 + diff2 * diff2 + diff3 * diff3;
        a += 4;
        b += 4;
    }
    while (a < last) {
        diff0 = *a++ - *b++;
        result += diff0 * diff0;
    }
    return result;
}

int oracle_knn_l2sq_f32(const float *q, int nq, size_t q_stride, const float *t, int nt, size_t t_stride, int dim,
                        int k, int32_t *idx, float *dist) {
    if (!q || !t || !idx || !dist || nq < 0 || nt < k || dim <= 0 || (k != 1 && k != 2)) return -1;
    result_set rs;
    for (int qi = 0; qi < nq; ++qi) {
        const float *a = q + (size_t)qi * q_stride;
        rs_clear(&rs, k);
        for (int ti = 0; ti < nt; ++ti) {
            /* LinearIndex::findNeighbors calls distance_(data_row, query_vec, cols) */
            float d = l2sq_f32(t + (size_t)ti * t_stride, a, dim);
            rs_add(&rs, (double)d, ti);
        }
        for (int j = 0; j < k; ++j) {
            idx[(size_t)qi * k + j] = rs.i[j];
            dist[(size_t)qi * k + j] = (float)rs.d[j];
        }
    }
    return 0;
}

int oracle_ratio_filter_i32(const int32_t *idx, const int32_t *dist, int nq, int k, oracle_dmatch *out) {
    int n = 0;
    for (int qi = 0; qi < nq; ++qi) {
        if (k == 2) {
            /* matchers.cpp:605  if(dists[q][0] < (0.75f * dists[q][1]))  -- int vs float compare */
            if (!((float)dist[2 * qi] < (0.75f * (float)dist[2 * qi + 1]))) continue;
        }
        out[n].distance = (float)dist[(size_t)k * qi];
        out[n].queryIdx = qi;
        out[n].trainIdx = idx[(size_t)k * qi];
        out[n].imgIdx = -1;
        ++n;
    }
    return n;
}

int oracle_ratio_filter_f32(const int32_t *idx, const float *dist, int nq, int k, oracle_dmatch *out) {
    int n = 0;
    for (int qi = 0; qi < nq; ++qi) {
        if (k == 2) {
            /* matchers.cpp:681 -- applied to SQUARED distances */
            if (!(dist[2 * qi] < (0.75f * dist[2 * qi + 1]))) continue;
        }
        out[n].distance = dist[(size_t)k * qi];
        out[n].queryIdx = qi;
        out[n].trainIdx = idx[(size_t)k * qi];
        out[n].imgIdx = -1;
        ++n;
    }
    return n;
}

int oracle_get_matches_linear(int n_kp1, int n_kp2, const void *desc1, int rows1, const void *desc2, int rows2,
                              int cols, int desc_type, int ratio_test, oracle_dmatch *out, int *n_out) {
    *n_out = 0;
    if (n_kp1 < 15 || n_kp2 < 15) return -4;             /* matchers.cpp:123-127 */
    if (n_kp1 != rows1 || n_kp2 != rows2) return -1;     /* matchers.cpp:129-133 */
    if (desc_type != 0 && desc_type != 5) return -1;     /* matchers.cpp:540-547 */
    const int nn = ratio_test ? 2 : 1;                   /* matchers.cpp:529-538 */
    int32_t *idx = (int32_t *)malloc(sizeof(int32_t) * (size_t)rows1 * nn);
    int rc = 0;
    if (desc_type == 0) {
        int32_t *dist = (int32_t *)malloc(sizeof(int32_t) * (size_t)rows1 * nn);
        rc = oracle_knn_hamming((const uint8_t *)desc1, rows1, (size_t)cols, (const uint8_t *)desc2, rows2,
                                (size_t)cols, cols, nn, idx, dist);
        if (rc == 0) *n_out = oracle_ratio_filter_i32(idx, dist, rows1, nn, out);
        free(dist);
    } else {
        float *dist = (float *)malloc(sizeof(float) * (size_t)rows1 * nn);
        rc = oracle_knn_l2sq_f32((const float *)desc1, rows1, (size_t)cols, (const float *)desc2, rows2,
                                 (size_t)cols, cols, nn, idx, dist);
        if (rc == 0) *n_out = oracle_ratio_filter_f32(idx, dist, rows1, nn, out);
        free(dist);
    }
    free(idx);
    if (rc != 0) return -1;
    if (*n_out < 2) return -3;                           /* matchers.cpp:709-713 */
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------------
 * BRUTEFORCENMS: matchers.cpp:476-519 -> nmslibMatching<dist_t>(..., "seq_search", "bit_hamming"|"l2")
 * (M/include/nmslib/nmslib_matchers.h:159-424) on the vendored NMSLIB.  Restated from the vendored sources:
 *   - CV_8U rows are packed two bytes per int and handed to NMSLIB without the trailing length word, and
 *     SpaceBitHamming::HiddenDistance uses datalength/4 - 1 words (similarity_search/src/space/space_bit_hamming.cc:33-42):
 *     the LAST int is ignored, i.e. the last two bytes (one byte for odd widths) never take part -- ORB-256 is compared on
 *     240 bits;
 *   - CV_32F: SpaceLp<float>(2) = L2NormSIMD = sqrt(L2SqrSIMD) (src/distcomp_lp.cc:399-457): four lane accumulators over
 *     4-float chunks, res = ((t0+t1)+t2)+t3, scalar tail, then sqrtf -- TRUE distances, not squared;
 *   - K = 2 always; KNNQueue keeps a candidate only if dist < current worst (include/knnqueue.h:57-66), i.e. the two
 *     lexicographically smallest (dist, id) for sequentially allocated objects; results are sorted by distance with the
 *     heap's pop order kept on ties (larger id first), nmslib_matchers.h:360-385;
 *   - ratio test on the true distances d0 < 0.75f*d1 (:397-406); without ratio test the first sorted entry is emitted;
 *     matches sorted by queryIdx (:416-420); getMatches applies no minimum-match check on this branch.
 * The reference builds NMSLIB with -Ofast -march=native (similarity_search/CMakeLists.txt), so its float sums may be
 * contracted/reassociated on the build machine; this restates the source as written and is pinned against oracle/_ref
 * (built -O2 -msse4.2, no fast-math).
 * ---------------------------------------------------------------------------------------------------------------- */
#include <math.h>

static int nms_eff_bytes(int cols) { return (cols % 2 == 0) ? cols - 2 : cols - 1; }

static float nms_l2(const float *a, const float *b, int qty) {
    float t[4] = {0.f, 0.f, 0.f, 0.f};
    const int q4 = qty / 4;
    for (int c = 0; c < q4; ++c)
        for (int l = 0; l < 4; ++l) {
            const float d = a[4 * c + l] - b[4 * c + l];
            t[l] = t[l] + d * d;
        }
    float res = t[0] + t[1] + t[2] + t[3];
    for (int i = 4 * q4; i < qty; ++i) {
        const float d = a[i] - b[i];
        res += d * d;
    }
    return sqrtf(res);
}

/* desc_type 0 = CV_8U, 5 = CV_32F.  out must hold rows1 entries.  Returns the wrapper's code (0). */
int oracle_get_matches_bruteforce_nms(const void *desc1, int rows1, const void *desc2, int rows2, int cols, int desc_type,
                                      int ratio_test, oracle_dmatch *out, int *n_out) {
    *n_out = 0;
    if (desc_type != 0 && desc_type != 5) return -1;
    if (rows2 < 2) return -1;
    lut_init();
    const int eb = nms_eff_bytes(cols);
    int n = 0;
    for (int q = 0; q < rows1; ++q) {
        double d0 = 1e300, d1 = 1e300; /* two lexicographically smallest (dist, id) */
        int i0 = -1, i1 = -1;
        for (int t = 0; t < rows2; ++t) {
            double d;
            if (desc_type == 0) {
                const uint8_t *a = (const uint8_t *)desc1 + (size_t)q * cols, *b = (const uint8_t *)desc2 + (size_t)t * cols;
                int h = 0;
                for (int j = 0; j < eb; ++j) h += g_lut[a[j] ^ b[j]];
                d = (double)h;
            } else {
                d = (double)nms_l2((const float *)desc1 + (size_t)q * cols, (const float *)desc2 + (size_t)t * cols, cols);
            }
            if (i1 < 0 || d < d1) { /* queue not full, or strictly better than the current worst */
                if (i0 < 0 || d < d0) {
                    d1 = d0, i1 = i0;
                    d0 = d, i0 = t;
                } else {
                    d1 = d, i1 = t;
                }
            }
        }
        /* sorted by distance; on a tie the heap pops the larger id first and std::sort keeps that order */
        int first = i0;
        const double fd = d0;
        if (d0 == d1) first = i1;
        if (ratio_test && !((float)d0 < 0.75f * (float)d1)) continue;
        out[n].queryIdx = q;
        out[n].trainIdx = first;
        out[n].imgIdx = -1;
        out[n].distance = (float)fd;
        ++n;
    }
    *n_out = n;
    return 0;
}

/* ------------------------------------------------------------------------------------------------------------------
 * "CPU-best" tier of BASELINE.md section 3: same results as oracle_knn_hamming, but hardware popcount on 64-bit words
 * and OpenMP over the queries -- a fair upper bound for the host, timed beside the faithful single-thread LUT port.
 * nbytes must be a multiple of 8 here.  Returns the number of threads used, or -1.
 * ---------------------------------------------------------------------------------------------------------------- */
#ifdef _OPENMP
#include <omp.h>
#endif
int oracle_knn_hamming_fast(const uint8_t *q, int nq, size_t q_stride, const uint8_t *t, int nt, size_t t_stride, int nbytes,
                            int32_t *idx, int32_t *dist, int threads) {
    if (!q || !t || !idx || !dist || nt < 2 || nbytes <= 0 || nbytes % 8) return -1;
    const int nw = nbytes / 8;
    int used = 1;
#ifdef _OPENMP
    if (threads > 0) omp_set_num_threads(threads);
#pragma omp parallel
    {
#pragma omp single
        used = omp_get_num_threads();
    }
#pragma omp parallel for schedule(static)
#endif
    for (int qi = 0; qi < nq; ++qi) {
        uint64_t a[32];
        memcpy(a, q + (size_t)qi * q_stride, (size_t)nbytes);
        uint64_t k0 = ~0ull, k1 = ~0ull; /* (dist << 32 | idx): lexicographic (dist, idx) */
        for (int ti = 0; ti < nt; ++ti) {
            uint64_t b[32];
            memcpy(b, t + (size_t)ti * t_stride, (size_t)nbytes);
            unsigned d = 0;
            for (int w = 0; w < nw; ++w) d += (unsigned)__builtin_popcountll(a[w] ^ b[w]);
            const uint64_t key = ((uint64_t)d << 32) | (uint32_t)ti;
            if (key < k0) {
                k1 = k0;
                k0 = key;
            } else if (key < k1) {
                k1 = key;
            }
        }
        idx[2 * qi] = (int32_t)(k0 & 0xffffffffu);
        dist[2 * qi] = (int32_t)(k0 >> 32);
        idx[2 * qi + 1] = (int32_t)(k1 & 0xffffffffu);
        dist[2 * qi + 1] = (int32_t)(k1 >> 32);
    }
    return used;
}
