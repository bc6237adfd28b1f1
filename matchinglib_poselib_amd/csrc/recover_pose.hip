// recover_pose.hip -- cheirality check / pose recovery on gfx950.
//
// Replaces poselib::getPoseTriangPts (reference poselib/source/pose_estim.cpp:913-946) = recoverPose
// (poselib/source/five-point-nister/five-point.cpp:150-338) with t_only empty:
//   decomposeEssentialMat (:340-352): SVD(E), det fixes, R1 = U W Vt, R2 = U W^T Vt, t = U[:,2]   -> decompose_kernel
//   four cv::triangulatePoints calls against P0 = [I|0] (:200-268)                               -> triangulate_kernel
//   masks  z*w > 0,  (P*Q)z*w > 0,  z/w < dist,  AND with the incoming mask, countNonZero        -> triangulate_kernel
//   candidate choice by the if-chain (:299-336)                                                   -> host, from 4 counts
// One thread per (candidate, correspondence): the 4x4 DLT system [x*P(2,:)-P(0,:); y*P(2,:)-P(1,:)] of both views,
// its smallest right singular vector by a one-sided Jacobi SVD held entirely in registers (the algorithm of OpenCV's
// JacobiSVDImpl_, which the un-vendored cv::triangulatePoints ends in), then the three predicates.

#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

#include "mlpl_internal.h"

namespace mlpl {

namespace {

template <int N>
__device__ __forceinline__ void jacobi_right_vectors(double (&G)[N][N], double (&V)[N][N]) {
#pragma unroll
    for (int i = 0; i < N; ++i)
#pragma unroll
        for (int j = 0; j < N; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    const double eps = DBL_EPSILON * 2;
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < N - 1; ++p)
#pragma unroll
            for (int q = p + 1; q < N; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    alpha += G[i][p] * G[i][p];
                    beta += G[i][q] * G[i][q];
                    gamma += G[i][p] * G[i][q];
                }
                if (gamma * gamma <= (eps * eps) * (alpha * beta) || gamma == 0.0) continue;  // |gamma| <= eps sqrt(alpha beta)
                rotated = true;
                // A plane rotation stays orthogonal to rounding for ANY tangent t as long as c = (1 + t^2)^-1/2 is accurate and s = c t;
                // an inexact t only leaves the pair at ~1e-8 of its former inner product, which the next sweep removes, and the
                // iteration still ends on the test above.  So the tangent comes from v_rcp_f64 / v_sqrt_f64 (~2^-26) and only c is
                // Newton-refined: three IEEE divisions and three IEEE square roots (~150 instructions per pair) become ~25.
                const double zeta = (beta - alpha) * __builtin_amdgcn_rcp(2.0 * gamma);
                const double az = fabs(zeta);
                double t = __builtin_amdgcn_rcp(az + __builtin_amdgcn_sqrt(az * az + 1.0));
                t = zeta >= 0 ? t : -t;
                if (!(az < 1e150)) t = 0.5 / zeta;  // zeta^2 overflows: t = 1 / (2 zeta) to rounding
                const double x1 = t * t + 1.0;
                double c = __builtin_amdgcn_rsq(x1);
                c = c * (1.5 - 0.5 * x1 * c * c);
                c = c * (1.5 - 0.5 * x1 * c * c);
                const double s = c * t;
#pragma unroll
                for (int i = 0; i < N; ++i) {
                    const double gp = G[i][p], gq = G[i][q];
                    G[i][p] = c * gp - s * gq;
                    G[i][q] = s * gp + c * gq;
                    const double vp = V[i][p], vq = V[i][q];
                    V[i][p] = c * vp - s * vq;
                    V[i][q] = s * vp + c * vq;
                }
            }
        if (!rotated) break;
    }
}

// The 4 x 4 triangulation systems: the same one-sided Jacobi iteration with the six column pairs of a sweep taken as three ROUNDS of two
// disjoint pairs -- (0,1)(2,3), (0,2)(1,3), (0,3)(1,2) -- and no branch inside a round: a pair that is already orthogonal gets the identity
// rotation (c = 1, s = 0 leave its columns bit for bit).  The two rotations of a round touch different columns, so their dependent
// chains (three inner products, reciprocal, root, inverse root, two Newton steps) interleave in one lane; the cyclic order above left the
// kernel waiting 60 % of its wave cycles (SQ_WAIT_ANY / SQ_WAVE_CYCLES, profiles/r03_pmc_summary.json) with one chain per lane.
struct PlaneRot {
    double c, s;
    bool on;
};
__device__ __forceinline__ PlaneRot plane_rotation(double alpha, double beta, double gamma) {
    const double eps = DBL_EPSILON * 2;
    PlaneRot r;
    r.on = !(gamma * gamma <= (eps * eps) * (alpha * beta) || gamma == 0.0);
    const double zeta = (beta - alpha) * __builtin_amdgcn_rcp(2.0 * gamma);
    const double az = fabs(zeta);
    double t = __builtin_amdgcn_rcp(az + __builtin_amdgcn_sqrt(__builtin_fma(az, az, 1.0)));
    t = zeta >= 0 ? t : -t;
    if (!(az < 1e150)) t = 0.5 * __builtin_amdgcn_rcp(zeta);  // zeta^2 overflows: t = 1 / (2 zeta); any t of that size does (see above)
    const double x1 = __builtin_fma(t, t, 1.0);
    double c = __builtin_amdgcn_rsq(x1);
    c = c * __builtin_fma(-0.5 * x1, c * c, 1.5);
    c = c * __builtin_fma(-0.5 * x1, c * c, 1.5);
    r.c = r.on ? c : 1.0;
    r.s = r.on ? c * t : 0.0;
    return r;
}
template <int P, int Q>
__device__ __forceinline__ void pair_products(const double (&G)[4][4], double &alpha, double &beta, double &gamma) {
    alpha = beta = gamma = 0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        alpha = __builtin_fma(G[i][P], G[i][P], alpha);  // (the library is built with -ffp-contract=off: the fused forms are spelled out
        beta = __builtin_fma(G[i][Q], G[i][Q], beta);    //  where no bit pattern is pinned -- half the instructions of the iteration)
        gamma = __builtin_fma(G[i][P], G[i][Q], gamma);
    }
}
template <int P, int Q>
__device__ __forceinline__ void turn_columns(double (&G)[4][4], double (&V)[4][4], PlaneRot r) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const double gp = G[i][P], gq = G[i][Q];
        G[i][P] = __builtin_fma(r.c, gp, -(r.s * gq));
        G[i][Q] = __builtin_fma(r.s, gp, r.c * gq);
        const double vp = V[i][P], vq = V[i][Q];
        V[i][P] = __builtin_fma(r.c, vp, -(r.s * vq));
        V[i][Q] = __builtin_fma(r.s, vp, r.c * vq);
    }
}
template <int P0, int Q0, int P1, int Q1>
__device__ __forceinline__ bool jacobi4_round(double (&G)[4][4], double (&V)[4][4]) {
    double a0, b0, g0, a1, b1, g1;
    pair_products<P0, Q0>(G, a0, b0, g0);
    pair_products<P1, Q1>(G, a1, b1, g1);
    const PlaneRot r0 = plane_rotation(a0, b0, g0), r1 = plane_rotation(a1, b1, g1);
    turn_columns<P0, Q0>(G, V, r0);
    turn_columns<P1, Q1>(G, V, r1);
    return r0.on || r1.on;
}
__device__ __forceinline__ void jacobi4_right_vectors(double (&G)[4][4], double (&V)[4][4]) {
#pragma unroll
    for (int i = 0; i < 4; ++i)
#pragma unroll
        for (int j = 0; j < 4; ++j) V[i][j] = (i == j) ? 1.0 : 0.0;
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = jacobi4_round<0, 1, 2, 3>(G, V);
        rotated |= jacobi4_round<0, 2, 1, 3>(G, V);
        rotated |= jacobi4_round<0, 3, 1, 2>(G, V);
        if (!rotated) break;
    }
}

__device__ __forceinline__ double det3(const double *M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}

__device__ __forceinline__ void mat3_mul(const double *A, const double *B, double *C) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}

// out: P[4][12] candidate projection matrices, Rt[2][9] = R1,R2 and t[3] appended (42 + 18 + 3 doubles)
__global__ void decompose_kernel(const double *__restrict__ E, double *__restrict__ out) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    double G[3][3], V[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) G[i][j] = E[i * 3 + j];
    jacobi_right_vectors<3>(G, V);
    double w[3];
    for (int j = 0; j < 3; ++j) w[j] = sqrt(G[0][j] * G[0][j] + G[1][j] * G[1][j] + G[2][j] * G[2][j]);
    // descending order (selection sort on indices, first maximum wins on ties)
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a) {
        int best = a;
        for (int b = a + 1; b < 3; ++b)
            if (w[ord[b]] > w[ord[best]]) best = b;
        const int t = ord[a];
        ord[a] = ord[best];
        ord[best] = t;
    }
    double U[9], Vt[9], ws[3];
    for (int j = 0; j < 3; ++j) {
        ws[j] = w[ord[j]];
        for (int i = 0; i < 3; ++i) {
            U[i * 3 + j] = (ws[j] > 0) ? G[i][ord[j]] / ws[j] : 0.0;
            Vt[j * 3 + i] = V[i][ord[j]];
        }
    }
    if (!(ws[2] > 1e-12 * ws[0])) {  // rank-2 E: complete U with u1 x u2
        U[0 * 3 + 2] = U[1 * 3 + 0] * U[2 * 3 + 1] - U[2 * 3 + 0] * U[1 * 3 + 1];
        U[1 * 3 + 2] = U[2 * 3 + 0] * U[0 * 3 + 1] - U[0 * 3 + 0] * U[2 * 3 + 1];
        U[2 * 3 + 2] = U[0 * 3 + 0] * U[1 * 3 + 1] - U[1 * 3 + 0] * U[0 * 3 + 1];
    }
    if (det3(U) < 0)
        for (int i = 0; i < 9; ++i) U[i] = -U[i];
    if (det3(Vt) < 0)
        for (int i = 0; i < 9; ++i) Vt[i] = -Vt[i];
    const double W[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
    const double Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    double T[9], R1[9], R2[9];
    mat3_mul(U, W, T);
    mat3_mul(T, Vt, R1);
    mat3_mul(U, Wt, T);
    mat3_mul(T, Vt, R2);
    const double tv[3] = {U[2], U[5], U[8]};
    // P1=[R1|t] P2=[R2|t] P3=[R1|-t] P4=[R2|-t]   (five-point.cpp:185-193)
    for (int c = 0; c < 4; ++c) {
        const double *R = (c & 1) ? R2 : R1;
        const double sg = (c < 2) ? 1.0 : -1.0;
        for (int r = 0; r < 3; ++r) {
            for (int k = 0; k < 3; ++k) out[c * 12 + r * 4 + k] = R[r * 3 + k];
            out[c * 12 + r * 4 + 3] = sg * tv[r];
        }
    }
    for (int i = 0; i < 9; ++i) {
        out[48 + i] = R1[i];
        out[57 + i] = R2[i];
    }
    for (int i = 0; i < 3; ++i) out[66 + i] = tv[i];
}

__global__ __launch_bounds__(256) void triangulate_kernel(const double *__restrict__ P /*[4][12]*/, const double *__restrict__ p1,
                                                          const double *__restrict__ p2, int n, double dist,
                                                          const uint8_t *__restrict__ mask_in, double *__restrict__ Q /*[4][n][3]*/,
                                                          uint8_t *__restrict__ mask_out /*[4][n]*/, int32_t *__restrict__ counts) {
    __shared__ int wave_cnt[4];
    const int c = blockIdx.y;
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool good = false;
    if (i < n) {
        const double *Pc = P + c * 12;
        const double x1 = p1[2 * i], y1 = p1[2 * i + 1], x2 = p2[2 * i], y2 = p2[2 * i + 1];
        double G[4][4], V[4][4];
        // view 1: P0 = [I|0]
        const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            G[0][k] = x1 * P0[8 + k] - P0[k];
            G[1][k] = y1 * P0[8 + k] - P0[4 + k];
            G[2][k] = x2 * Pc[8 + k] - Pc[k];
            G[3][k] = y2 * Pc[8 + k] - Pc[4 + k];
        }
        jacobi4_right_vectors(G, V);
        double w[4];
#pragma unroll
        for (int j = 0; j < 4; ++j) w[j] = G[0][j] * G[0][j] + G[1][j] * G[1][j] + G[2][j] * G[2][j] + G[3][j] * G[3][j];
        int m = 0;
#pragma unroll
        for (int j = 1; j < 4; ++j)
            if (w[j] < w[m]) m = j;
        double X[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) X[k] = (m == 0) ? V[k][0] : (m == 1 ? V[k][1] : (m == 2 ? V[k][2] : V[k][3]));
        bool mk = (X[2] * X[3] > 0);
        const double qz = Pc[8] * X[0] + Pc[9] * X[1] + Pc[10] * X[2] + Pc[11] * X[3];
        mk = mk && (qz * X[3] > 0);
        const double qx = X[0] / X[3], qy = X[1] / X[3], qzz = X[2] / X[3];
        mk = mk && (qzz < dist);
        double *q = Q + ((size_t)c * n + i) * 3;
        q[0] = qx, q[1] = qy, q[2] = qzz;
        uint8_t mv = mk ? 255 : 0;
        if (mask_in) mv &= mask_in[i];
        mask_out[(size_t)c * n + i] = mv;
        good = (mv != 0);
    }
    const unsigned long long bal = __ballot(good);
    if ((threadIdx.x & 63) == 0) wave_cnt[threadIdx.x >> 6] = __popcll(bal);
    __syncthreads();
    if (threadIdx.x == 0) atomicAdd(&counts[c], wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3]);
}

// ---- batch of pairs (pair_batch_impl.h) -----------------------------------------------------------------------------------------------------
// decompose: block = pair, E read from a strided table; triangulate: blockIdx.z = pair, its coordinates at pair * pair_stride, its count
// counts[pair] (pairs flagged inactive: count forced to 0); select: the reference's if-chain (five-point.cpp:299-336) on the device, the
// chosen candidate's mask copied over the pair's mask.
__global__ void decompose_batch_kernel(const char *__restrict__ E_base, size_t E_stride, const int32_t *__restrict__ active,
                                       double *__restrict__ P_all /*[B][69]*/) {
    if (threadIdx.x != 0 || !active[blockIdx.x]) return;
    const double *E = reinterpret_cast<const double *>(E_base + (size_t)blockIdx.x * E_stride);
    double *out = P_all + (size_t)blockIdx.x * 69;
    double G[3][3], V[3][3];
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) G[i][j] = E[i * 3 + j];
    jacobi_right_vectors<3>(G, V);
    double w[3];
    for (int j = 0; j < 3; ++j) w[j] = sqrt(G[0][j] * G[0][j] + G[1][j] * G[1][j] + G[2][j] * G[2][j]);
    int ord[3] = {0, 1, 2};
    for (int a = 0; a < 2; ++a) {
        int best = a;
        for (int b = a + 1; b < 3; ++b)
            if (w[ord[b]] > w[ord[best]]) best = b;
        const int t = ord[a];
        ord[a] = ord[best];
        ord[best] = t;
    }
    double U[9], Vt[9], ws[3];
    for (int j = 0; j < 3; ++j) {
        ws[j] = w[ord[j]];
        for (int i = 0; i < 3; ++i) {
            U[i * 3 + j] = (ws[j] > 0) ? G[i][ord[j]] / ws[j] : 0.0;
            Vt[j * 3 + i] = V[i][ord[j]];
        }
    }
    if (!(ws[2] > 1e-12 * ws[0])) {
        U[0 * 3 + 2] = U[1 * 3 + 0] * U[2 * 3 + 1] - U[2 * 3 + 0] * U[1 * 3 + 1];
        U[1 * 3 + 2] = U[2 * 3 + 0] * U[0 * 3 + 1] - U[0 * 3 + 0] * U[2 * 3 + 1];
        U[2 * 3 + 2] = U[0 * 3 + 0] * U[1 * 3 + 1] - U[1 * 3 + 0] * U[0 * 3 + 1];
    }
    if (det3(U) < 0)
        for (int i = 0; i < 9; ++i) U[i] = -U[i];
    if (det3(Vt) < 0)
        for (int i = 0; i < 9; ++i) Vt[i] = -Vt[i];
    const double W[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
    const double Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    double T[9], R1[9], R2[9];
    mat3_mul(U, W, T);
    mat3_mul(T, Vt, R1);
    mat3_mul(U, Wt, T);
    mat3_mul(T, Vt, R2);
    const double tv[3] = {U[2], U[5], U[8]};
    for (int c = 0; c < 4; ++c) {
        const double *R = (c & 1) ? R2 : R1;
        const double sg = (c < 2) ? 1.0 : -1.0;
        for (int r = 0; r < 3; ++r) {
            for (int k = 0; k < 3; ++k) out[c * 12 + r * 4 + k] = R[r * 3 + k];
            out[c * 12 + r * 4 + 3] = sg * tv[r];
        }
    }
    for (int i = 0; i < 9; ++i) {
        out[48 + i] = R1[i];
        out[57 + i] = R2[i];
    }
    for (int i = 0; i < 3; ++i) out[66 + i] = tv[i];
}

constexpr int kTriPointsPerWave = 256;  // correspondences a wave of triangulate_batch_kernel compacts and walks
__global__ __launch_bounds__(256) void triangulate_batch_kernel(const double *__restrict__ P_all, const double *__restrict__ p1,
                                                                const double *__restrict__ p2, const int32_t *__restrict__ counts,
                                                                const int32_t *__restrict__ active, int pair_stride, double dist,
                                                                const uint8_t *__restrict__ mask_in /*[B][pair_stride]*/,
                                                                uint8_t *__restrict__ mask_out /*[B][4][pair_stride]*/,
                                                                int32_t *__restrict__ cand_counts /*[B][4]*/) {
    // A correspondence outside the incoming mask cannot pass whatever its 3-D point is (the batch form returns no points): no SVD for it.
    // About half of the correspondences are such (the RANSAC outliers), scattered over the lanes.  Every WAVE owns 256 consecutive
    // correspondences: it compacts the ones to triangulate into its own list (ballot ranks, no workgroup barrier anywhere) and walks the
    // list 64 at a time, so that all lanes are busy through the Jacobi sweeps but for the list's tail (as lane i = point i the kernel
    // ran 467 us per 128 pairs with half of every wave idle; with a workgroup-wide list and two barriers, 350).
    __shared__ int todo[4][kTriPointsPerWave];
    const int b = blockIdx.z, c = blockIdx.y;
    const int n = active[b] ? counts[b] : 0;
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
    const int base = (blockIdx.x * 4 + wv) * kTriPointsPerWave;
    if (base >= n) return;  // wave-uniform
    int total = 0;
#pragma unroll
    for (int k = 0; k < kTriPointsPerWave / 64; ++k) {
        const int i0 = base + k * 64 + lane;
        const bool want = i0 < n && (!mask_in || mask_in[(size_t)b * pair_stride + i0]);
        if (i0 < n && !want) mask_out[((size_t)b * 4 + c) * pair_stride + i0] = 0;
        const unsigned long long wb = __ballot(want);
        if (want) todo[wv][total + __popcll(wb & ((1ull << lane) - 1ull))] = i0;
        total += __popcll(wb);
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");  // the list is this wave's own: a wave-scope fence orders its LDS writes and reads
    __builtin_amdgcn_wave_barrier();
    const double *Pc = P_all + (size_t)b * 69 + c * 12;
    int n_good = 0;
    for (int at0 = 0; at0 < total; at0 += 64) {
        bool good = false;
        if (at0 + lane < total) {
            const int i = todo[wv][at0 + lane];
            const size_t at = ((size_t)b * pair_stride + i) * 2;
            const double x1 = p1[at], y1 = p1[at + 1], x2 = p2[at], y2 = p2[at + 1];
            double G[4][4], V[4][4];
            const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                G[0][k] = x1 * P0[8 + k] - P0[k];
                G[1][k] = y1 * P0[8 + k] - P0[4 + k];
                G[2][k] = x2 * Pc[8 + k] - Pc[k];
                G[3][k] = y2 * Pc[8 + k] - Pc[4 + k];
            }
            jacobi4_right_vectors(G, V);
            double w[4];
#pragma unroll
            for (int j = 0; j < 4; ++j) w[j] = G[0][j] * G[0][j] + G[1][j] * G[1][j] + G[2][j] * G[2][j] + G[3][j] * G[3][j];
            int m = 0;
#pragma unroll
            for (int j = 1; j < 4; ++j)
                if (w[j] < w[m]) m = j;
            double X[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) X[k] = (m == 0) ? V[k][0] : (m == 1 ? V[k][1] : (m == 2 ? V[k][2] : V[k][3]));
            bool mk = (X[2] * X[3] > 0);
            const double qz = Pc[8] * X[0] + Pc[9] * X[1] + Pc[10] * X[2] + Pc[11] * X[3];
            mk = mk && (qz * X[3] > 0);
            const double qzz = X[2] / X[3];
            mk = mk && (qzz < dist);
            uint8_t mv = mk ? 255 : 0;
            if (mask_in) mv &= mask_in[(size_t)b * pair_stride + i];
            mask_out[((size_t)b * 4 + c) * pair_stride + i] = mv;
            good = (mv != 0);
        }
        n_good += __popcll(__ballot(good));
    }
    if (lane == 0 && n_good) atomicAdd(&cand_counts[b * 4 + c], n_good);
}

__global__ __launch_bounds__(256) void select_pose_batch_kernel(const double *__restrict__ P_all, const int32_t *__restrict__ cand_counts,
                                                                const int32_t *__restrict__ counts, const int32_t *__restrict__ active,
                                                                int pair_stride, const uint8_t *__restrict__ cand_masks,
                                                                uint8_t *__restrict__ mask /*[B][pair_stride]*/,
                                                                PairPoseDev *__restrict__ out) {
    const int b = blockIdx.x;
    if (!active[b]) return;
    const int good1 = cand_counts[b * 4], good2 = cand_counts[b * 4 + 1], good3 = cand_counts[b * 4 + 2], good4 = cand_counts[b * 4 + 3];
    int pick, ret;
    if (good1 >= good2 && good1 >= good3 && good1 >= good4) {
        pick = 0, ret = good1;
    } else if (good2 && good2 >= good1 && good2 >= good3 && good2 >= good4) {
        pick = 1, ret = good2;
    } else if (good3 >= good1 && good3 >= good2 && good3 >= good4) {
        pick = 2, ret = good3;
    } else {
        ret = good4;
        pick = good4 ? 3 : -1;
    }
    const int n = counts[b];
    if (threadIdx.x == 0) {
        PairPoseDev &o = out[b];
        o.n_good = ret, o.pick = pick;
        const double *hP = P_all + (size_t)b * 69;
        if (pick >= 0) {
            for (int k = 0; k < 9; ++k) o.R[k] = hP[((pick & 1) ? 57 : 48) + k];
            const double sg = (pick < 2) ? 1.0 : -1.0;
            for (int k = 0; k < 3; ++k) o.t[k] = sg * hP[66 + k];
        } else {
            for (int k = 0; k < 9; ++k) o.R[k] = (k % 4 == 0) ? 1.0 : 0.0;
            o.t[0] = o.t[1] = o.t[2] = 0;
        }
    }
    for (int i = threadIdx.x; i < n; i += blockDim.x)
        mask[(size_t)b * pair_stride + i] = pick >= 0 ? cand_masks[((size_t)b * 4 + pick) * pair_stride + i] : 0;
}

}  // namespace

int launch_recover_pose_batch(const char *d_E_base, size_t E_stride, const double *d_p1, const double *d_p2, const int32_t *d_counts,
                              const int32_t *d_active, int B, int pair_stride, double dist, uint8_t *d_mask, double *d_P /*[B][69]*/,
                              uint8_t *d_cand_masks /*[B][4][pair_stride]*/, int32_t *d_cand_counts /*[B][4]*/, PairPoseDev *d_out,
                              hipStream_t s) {
    MLPL_HIP_TRY(hipMemsetAsync(d_cand_counts, 0, (size_t)B * 16, s));
    hipLaunchKernelGGL(decompose_batch_kernel, dim3(B), dim3(64), 0, s, d_E_base, E_stride, d_active, d_P);
    hipLaunchKernelGGL(triangulate_batch_kernel, dim3((pair_stride + 4 * kTriPointsPerWave - 1) / (4 * kTriPointsPerWave), 4, B), dim3(256), 0, s, (const double *)d_P, d_p1, d_p2, d_counts,
                       d_active, pair_stride, dist, (const uint8_t *)d_mask, d_cand_masks, d_cand_counts);
    hipLaunchKernelGGL(select_pose_batch_kernel, dim3(B), dim3(256), 0, s, (const double *)d_P, (const int32_t *)d_cand_counts, d_counts, d_active,
                       pair_stride, (const uint8_t *)d_cand_masks, d_mask, d_out);
    MLPL_HIP_TRY(hipGetLastError());
    return MLPL_OK;
}

}  // namespace mlpl

using namespace mlpl;

static int recover_pose_impl(mlpl_ctx *ctx, const double *E, const double *t_only, const double *p1, const double *p2, int n,
                             double dist, double *R, double *t, double *Q, uint8_t *mask_inout, bool dev = false,
                             hipStream_t stream = nullptr);

// Device-resident correspondences (d_p1, d_p2: n x 2 doubles; d_Q: n x 3 doubles or NULL; d_mask_inout: n bytes or NULL, all on
// the device); E, R, t are host.  One host hop (the four candidate counts decide which candidate's outputs are kept).
extern "C" int mlpl_recover_pose_dev(mlpl_ctx *ctx, const double E[9], const double *d_p1, const double *d_p2, int n, double dist,
                                     double R[9], double t[3], double *d_Q, uint8_t *d_mask_inout, void *stream) {
    if (!E || !ctx) {
        set_error("mlpl_recover_pose_dev: ctx and E are mandatory");
        return MLPL_E_BAD_INPUT;
    }
    return recover_pose_impl(ctx, E, nullptr, d_p1, d_p2, n, dist, R, t, d_Q, d_mask_inout, true, pick_stream(ctx, stream));
}

extern "C" int mlpl_recover_pose(mlpl_ctx *ctx, const double E[9], const double *p1, const double *p2, int n, double dist,
                                 double R[9], double t[3], double *Q, uint8_t *mask_inout) {
    if (!E) {
        set_error("mlpl_recover_pose: E is mandatory");
        return MLPL_E_BAD_INPUT;
    }
    return recover_pose_impl(ctx, E, nullptr, p1, p2, n, dist, R, t, Q, mask_inout);
}

extern "C" int mlpl_recover_pose_translation(mlpl_ctx *ctx, const double t_only[3], const double *p1, const double *p2, int n,
                                             double dist, double R[9], double t[3], double *Q, uint8_t *mask_inout) {
    if (!t_only) {
        set_error("mlpl_recover_pose_translation: t_only is mandatory");
        return MLPL_E_BAD_INPUT;
    }
    return recover_pose_impl(ctx, nullptr, t_only, p1, p2, n, dist, R, t, Q, mask_inout);
}

static int recover_pose_impl(mlpl_ctx *ctx, const double *E, const double *t_only, const double *p1, const double *p2, int n,
                             double dist, double *R, double *t, double *Q, uint8_t *mask_inout, bool dev, hipStream_t stream) {
    if (!ctx || !p1 || !p2 || !R || !t || (!Q && !dev) || n < 0) {
        set_error("mlpl_recover_pose: R, t and Q are mandatory outputs");  // pose_estim.cpp:925-926 returns -1
        return MLPL_E_BAD_INPUT;
    }
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = dev ? stream : ctx->stream;
    void *dp1 = const_cast<double *>(p1), *dp2 = const_cast<double *>(p2), *dQ, *dmask, *dsmall;
    int rc;
    const size_t nn = (size_t)std::max(n, 1);
    if (!dev) {
        if ((rc = ws_get(ctx, WS_AUX0, nn * 16, &dp1))) return rc;
        if ((rc = ws_get(ctx, WS_AUX1, nn * 16, &dp2))) return rc;
    }
    if ((rc = ws_get(ctx, WS_AUX5, nn * 4 * 3 * 8, &dQ))) return rc;
    if ((rc = ws_get(ctx, WS_AUX6, nn * 5, &dmask))) return rc;
    if ((rc = ws_get(ctx, WS_AUX2, 4096, &dsmall))) return rc;
    double *dE = (double *)dsmall;         // 9
    double *dP = dE + 16;                  // 69
    int32_t *dcnt = (int32_t *)(dE + 96);  // 4
    uint8_t *dmask_in = (uint8_t *)dmask + nn * 4;
    double hPt[69];
    if (E) {
        MLPL_HIP_TRY(hipMemcpyAsync(dE, E, 72, hipMemcpyHostToDevice, s));
    } else {
        // five-point.cpp:180-193 with t_only given: R1 = I, P1 = [I|t], P3 = [I|-t]; the R2 candidates do not exist
        // (their slots repeat P1/P3 and their counts are forced to 0 below)
        const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        for (int c = 0; c < 4; ++c)
            for (int r = 0; r < 3; ++r) {
                for (int k = 0; k < 3; ++k) hPt[c * 12 + r * 4 + k] = I3[r * 3 + k];
                hPt[c * 12 + r * 4 + 3] = (c < 2 ? 1.0 : -1.0) * t_only[r];
            }
        for (int i = 0; i < 9; ++i) hPt[48 + i] = hPt[57 + i] = I3[i];
        for (int i = 0; i < 3; ++i) hPt[66 + i] = t_only[i];
        MLPL_HIP_TRY(hipMemcpyAsync(dP, hPt, sizeof(hPt), hipMemcpyHostToDevice, s));
    }
    MLPL_HIP_TRY(hipMemsetAsync(dcnt, 0, 16, s));
    if (n > 0 && !dev) {
        MLPL_HIP_TRY(hipMemcpyAsync(dp1, p1, (size_t)n * 16, hipMemcpyHostToDevice, s));
        MLPL_HIP_TRY(hipMemcpyAsync(dp2, p2, (size_t)n * 16, hipMemcpyHostToDevice, s));
        if (mask_inout) MLPL_HIP_TRY(hipMemcpyAsync(dmask_in, mask_inout, (size_t)n, hipMemcpyHostToDevice, s));
    }
    if (n > 0 && dev && mask_inout) dmask_in = mask_inout;  // read in place; the chosen candidate's mask is copied back below
    if (E) hipLaunchKernelGGL(decompose_kernel, dim3(1), dim3(64), 0, s, (const double *)dE, dP);
    if (n > 0) {
        prof_mark(ctx, MLPL_PROF_RECOVER_POSE, 0, s);
        hipLaunchKernelGGL(triangulate_kernel, dim3((n + 255) / 256, 4), dim3(256), 0, s, (const double *)dP, (const double *)dp1,
                           (const double *)dp2, n, dist, mask_inout ? (const uint8_t *)dmask_in : (const uint8_t *)nullptr,
                           (double *)dQ, (uint8_t *)dmask, dcnt);
        prof_mark(ctx, MLPL_PROF_RECOVER_POSE, 1, s);
    }
    MLPL_HIP_TRY(hipGetLastError());
    int32_t cnt[4];
    double hP[69];
    MLPL_HIP_TRY(hipMemcpyAsync(cnt, dcnt, 16, hipMemcpyDeviceToHost, s));
    MLPL_HIP_TRY(hipMemcpyAsync(hP, dP, sizeof(hP), hipMemcpyDeviceToHost, s));
    MLPL_HIP_TRY(hipStreamSynchronize(s));
    const int good1 = cnt[0], good2 = E ? cnt[1] : 0, good3 = cnt[2], good4 = E ? cnt[3] : 0;
    // five-point.cpp:299-336
    int pick, ret;
    if (good1 >= good2 && good1 >= good3 && good1 >= good4) {
        pick = 0, ret = good1;
    } else if (good2 && good2 >= good1 && good2 >= good3 && good2 >= good4) {
        pick = 1, ret = good2;
    } else if (good3 >= good1 && good3 >= good2 && good3 >= good4) {
        pick = 2, ret = good3;
    } else {
        ret = good4;
        pick = good4 ? 3 : -1;
    }
    if (pick >= 0) {
        std::memcpy(R, hP + ((pick & 1) ? 57 : 48), 72);
        const double sg = (pick < 2) ? 1.0 : -1.0;
        for (int i = 0; i < 3; ++i) t[i] = sg * hP[66 + i];
        if (n > 0 && !dev) {
            MLPL_HIP_TRY(hipMemcpy(Q, (double *)dQ + (size_t)pick * n * 3, (size_t)n * 24, hipMemcpyDeviceToHost));
            if (mask_inout) MLPL_HIP_TRY(hipMemcpy(mask_inout, (uint8_t *)dmask + (size_t)pick * n, (size_t)n, hipMemcpyDeviceToHost));
        } else if (n > 0) {
            if (Q) MLPL_HIP_TRY(hipMemcpyAsync(Q, (double *)dQ + (size_t)pick * n * 3, (size_t)n * 24, hipMemcpyDeviceToDevice, s));
            if (mask_inout)
                MLPL_HIP_TRY(hipMemcpyAsync(mask_inout, (uint8_t *)dmask + (size_t)pick * n, (size_t)n, hipMemcpyDeviceToDevice, s));
        }
    } else {
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        std::memcpy(R, I, 72);
        t[0] = t[1] = t[2] = 0;
        if (!dev) {
            std::memset(Q, 0, (size_t)n * 24);
            if (mask_inout) std::memset(mask_inout, 0, (size_t)n);
        } else {
            if (Q) MLPL_HIP_TRY(hipMemsetAsync(Q, 0, (size_t)n * 24, s));
            if (mask_inout) MLPL_HIP_TRY(hipMemsetAsync(mask_inout, 0, (size_t)n, s));
        }
    }
    return ret;
}
