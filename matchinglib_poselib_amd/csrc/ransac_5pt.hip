// ransac_5pt.hip -- RANSAC / LMedS for the essential matrix on gfx950: one 5-point Nister solve per wavefront, grid-wide Sampson
// scoring, and a device-side replay of the reference's sequential best/niters rule.
//
// Replaces, under reference poselib/source/five-point-nister/ :
//   modelest.cpp:343-474  CvModelEstimator3::runRANSAC        (hyp_best_kernel + replay_kernel: prefix-max scan in iteration order)
//   modelest.cpp:483-564  CvModelEstimator3::runLMeDS         (median_kernel + lmeds_argmin_kernel)
//   modelest.cpp:567-650  getSubset / checkSubset             (sample table from the glibc rand() stream, host, pinned + mapped)
//   five-point.cpp:366-471 CvEMEstimator::run5Point           (solve5pt_kernel + roots_kernel)
//   five-point.cpp:476-503 computeReprojError3 + modelest.cpp:69-83 findInliers (score_models_kernel / score_models_block_kernel)
//
// solve5pt_kernel (64 threads = ONE wave per sample; cross-lane traffic through ~7 KiB of LDS, ordered by wave_sync(), no s_barrier):
//   1. the 5x9 epipolar matrix; its 4-dim null space by Householder QR of the 9x5 transpose (orthonormal basis, like the
//      reference's SVD basis; any orthonormal basis yields the same set of essential matrices);
//   2. the ten cubic constraints det(E)=0, E E^T E - 1/2 tr(E E^T) E = 0 for E = x E0 + y E1 + z E2 + E3: lane (i,j,k) of the
//      64 = 4^3 lanes evaluates the trilinear coefficient tensor, 20 lanes symmetrise it into the 10x20 matrix in the
//      reference's monomial order (five-point.cpp:813-823);
//   3. Gauss-Jordan with partial pivoting on the 10x20 system, 200 elements spread over the wave;
//   4. B(z) (3x13) and the degree-10 determinant polynomial as two convolution stages over the wave -> PolyRec in global memory.
// roots_kernel_t<polish> (SIX hypotheses per wave, one root per lane):
//   5. all complex roots simultaneously by Ehrlich-Aberth from cv::solvePoly's start values (1+i)^k (solvePoly itself iterates
//      Durand-Kerner; a converged simultaneous iteration delivers the same roots to rounding); a root is real iff
//      |imag| <= 1e-10 (five-point.cpp:438);
//   6. per real root: null vector of Bz, reject |xy1[2]| < 1e-10 (:457), E = x E0 + y E1 + z E2 + E3, Frobenius-normalised,
//      ballot-compacted into the per-hypothesis table and the dense model list.  polish = true (default): cross-product null
//      vector + Gauss-Newton polish of (x, y, z) on the ten constraints (polish_xyz: every model satisfies them to rounding);
//      polish = false: the 3x3 one-sided Jacobi SVD null vector, the plain root path of the CPU code.
// Scoring: fp64 Sampson error in the reference's operation order rounded to float exactly as the reference stores it,
//   `err <= thresh^2` counts and the 4-accumulator double sum of the float errors -- bit-identical to the CPU path for the same E (no
//   FMA contraction: this file is compiled with -ffp-contract=off like the reference's -msse4.2 build).  The RANSAC passes count
//   without dividing and on explicit FMAs inside a rigorous error band (sampson_inlier_fma; same counts), split the correspondences
//   of a model group over several workgroups (atomic counts), and compute error sums only for the models that can still win.
// The solver kernels must be launched with exactly 64 threads: wave_sync() orders LDS traffic inside ONE wave only.

#include <algorithm>
#include <cfloat>
#include <chrono>
#include <type_traits>
#include <climits>
#include <cmath>
#include <cstring>
#include <atomic>
#include <climits>
#include <condition_variable>
#include <ctime>
#include <memory>
#include <linux/futex.h>
#include <sys/mman.h>
#include <sys/syscall.h>
#include <ucontext.h>
#include <unistd.h>
#include <functional>
#include <mutex>
#include <new>
#include <string>
#include <thread>
#include <unordered_map>
#include <unordered_set>
#include <vector>

#include "mlpl_internal.h"
#include "usac_degen_math.h"

namespace mlpl {

namespace {

// ---------------------------------------------------------------------------------------------------------------
// helpers
// ---------------------------------------------------------------------------------------------------------------

// monomial order of the reference's coefficient matrix: exponents of (x, y, z); the 4th variable w = 1 takes the rest.
// kMonoVars[20][3] (variable indices of the three factors, 0=x 1=y 2=z 3=w, sorted):
//   {0, 0, 0}, {1, 1, 1}, {0, 0, 1}, {0, 1, 1}, {0, 0, 2}, {0, 0, 3}, {1, 1, 2}, {1, 1, 3}, {0, 1, 2}, {0, 1, 3},
//   {0, 2, 2}, {0, 2, 3}, {0, 3, 3}, {1, 2, 2}, {1, 2, 3}, {1, 3, 3}, {2, 2, 2}, {2, 2, 3}, {2, 3, 3}, {3, 3, 3}};

// the distinct orderings (a*16 + b*4 + c) of each monomial above, in the order (abc, acb, bac, bca, cab, cba) with repeats dropped
__constant__ int8_t kMonoNumPerms[20] = {1, 1, 3, 3, 3, 3, 3, 3, 6, 6, 3, 6, 3, 3, 6, 3, 1, 3, 3, 1};
__constant__ int8_t kMonoPerms[20][6] = {
    {0, 0, 0, 0, 0, 0}, {21, 0, 0, 0, 0, 0}, {1, 4, 16, 0, 0, 0}, {5, 17, 20, 0, 0, 0}, {2, 8, 32, 0, 0, 0}, {3, 12, 48, 0, 0, 0}, {22, 25, 37, 0, 0, 0}, {23, 29, 53, 0, 0, 0}, {6, 9, 18, 24, 33, 36}, {7, 13, 19, 28, 49, 52}, {10, 34, 40, 0, 0, 0}, {11, 14, 35, 44, 50, 56}, {15, 51, 60, 0, 0, 0}, {26, 38, 41, 0, 0, 0}, {27, 30, 39, 45, 54, 57}, {31, 55, 61, 0, 0, 0}, {42, 0, 0, 0, 0, 0}, {43, 46, 58, 0, 0, 0}, {47, 59, 62, 0, 0, 0}, {63, 0, 0, 0, 0, 0}};

struct SolveLds {
    double Q[5][9];     // epipolar rows; overwritten by the QR (as its transpose M[r][c] = Q[c][r])
    double V[5][9];     // Householder vectors
    double vn2[5];      // 2 / (their squared norms): the reflection factor (0 = no reflection)
    double EE[4][9];    // null-space basis
    double F[5][64];    // trilinear tensors, five constraint rows at a time
    double A[10][20];   // constraint matrix
    double b[3][13];    // B(z)
    double c[11];       // determinant polynomial, ascending
    double rr[10], ri[10];
};

__device__ __forceinline__ double wave_max(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmax(v, __shfl_xor(v, off));
    return v;
}

// Null vector of a (numerically) rank-2 3x3 matrix as the largest of the three cross products of its rows, scaled to unit length:
// ~60 instructions against the several thousand of the Jacobi SVD below.  Differs from the SVD's vector by O(sigma_3 / sigma_2); used
// where a Gauss-Newton polish on the original constraints follows.
__device__ __forceinline__ void null_vector_3x3_cross(const double *a, double *nv) {
    double c[3][3];
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const double *u = a + 3 * (t == 2 ? 1 : 0), *v = a + 3 * (t == 0 ? 1 : 2);  // row pairs (0,1), (0,2), (1,2)
        c[t][0] = u[1] * v[2] - u[2] * v[1];
        c[t][1] = u[2] * v[0] - u[0] * v[2];
        c[t][2] = u[0] * v[1] - u[1] * v[0];
    }
    double n2[3];
#pragma unroll
    for (int t = 0; t < 3; ++t) n2[t] = c[t][0] * c[t][0] + c[t][1] * c[t][1] + c[t][2] * c[t][2];
    const bool b1 = n2[1] > n2[0];
    double m2 = b1 ? n2[1] : n2[0];
    double v0 = b1 ? c[1][0] : c[0][0], v1 = b1 ? c[1][1] : c[0][1], v2 = b1 ? c[1][2] : c[0][2];
    const bool b2 = n2[2] > m2;
    m2 = b2 ? n2[2] : m2;
    v0 = b2 ? c[2][0] : v0, v1 = b2 ? c[2][1] : v1, v2 = b2 ? c[2][2] : v2;
    const double inv = m2 > 0 ? 1.0 / sqrt(m2) : 0.0;  // a zero matrix gives the zero vector: rejected by the caller's |v2| test
    nv[0] = v0 * inv, nv[1] = v1 * inv, nv[2] = v2 * inv;
}

// One-sided Jacobi SVD of a 3x3 (row-major a[9]); returns the right singular vector of the smallest singular value.
__device__ __forceinline__ void null_vector_3x3(const double *a, double *nv) {
    double G[3][3], V[3][3];
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            G[i][j] = a[i * 3 + j];
            V[i][j] = (i == j) ? 1.0 : 0.0;
        }
    const double eps = DBL_EPSILON * 2;
    for (int sweep = 0; sweep < 60; ++sweep) {
        bool rotated = false;
#pragma unroll
        for (int p = 0; p < 2; ++p)
#pragma unroll
            for (int q = p + 1; q < 3; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
                    alpha += G[i][p] * G[i][p];
                    beta += G[i][q] * G[i][q];
                    gamma += G[i][p] * G[i][q];
                }
                if (fabs(gamma) <= eps * sqrt(alpha * beta) || gamma == 0.0) continue;
                rotated = true;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
#pragma unroll
                for (int i = 0; i < 3; ++i) {
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
    double w[3];
#pragma unroll
    for (int j = 0; j < 3; ++j) w[j] = G[0][j] * G[0][j] + G[1][j] * G[1][j] + G[2][j] * G[2][j];
    int m = 0;
    if (w[1] < w[m]) m = 1;
    if (w[2] < w[m]) m = 2;
    nv[0] = (m == 0) ? V[0][0] : (m == 1 ? V[0][1] : V[0][2]);
    nv[1] = (m == 0) ? V[1][0] : (m == 1 ? V[1][1] : V[1][2]);
    nv[2] = (m == 0) ? V[2][0] : (m == 1 ? V[2][1] : V[2][2]);
}

// ---------------------------------------------------------------------------------------------------------------
// Steps 2..6 of the solver, shared by the minimal (5-point) kernel and the refit kernel: EE basis in LDS ->
// essential matrices.  Returns (per lane) whether this lane holds a valid model in Eout[9].
// ---------------------------------------------------------------------------------------------------------------
__device__ int g_dk_iters_dbg[16] = {0, 0, 0, 0};  // [sum, count, max, enabled, sample of the max, histogram of the sweep counts in
                                                   //  (<=8, <=12, <=16, <=24, <=32, <=64, <=128, <=256, <400, =400)] -- diagnostics only (tools/)

// Batched passes (pair_batch_impl.h: many image pairs per launch): one record per ACTIVE slot of a pass.  The hypothesis tables of slot a
// start at a * slot_stride; kernels that take `const PairSlot *ps` pick their per-pair arguments from ps[slot] (ps == nullptr: the
// single-pair form, arguments as passed).
struct PairSlot {
    const double4 *pts;  // packed correspondences of the pair (pack_points_kernel layout for ITS n)
    int32_t n;           // correspondences
    int32_t pair;        // index of the pair in the batch (replay state, mask)
    int32_t cnt;         // hypotheses of this pass
    int32_t iter_base;   // iterations of earlier passes
};

// What the per-hypothesis wave hands to the root-finding kernel: the degree-10 polynomial, B(z) and the null-space basis.
struct PolyRec {
    double c[11];
    double b[39];
    double EE[36];
    double ok;  // 0 = singular system, no models
    double pad;
};
static_assert(sizeof(PolyRec) == 88 * 8, "PolyRec layout");

// Steps 2..4 of the solver (one wave): EE basis in LDS -> PolyRec in global memory.
// The solver kernels run ONE wave per workgroup.  A wave's LDS instructions execute in program order, so a write by one lane is
// visible to a later read by another lane of the same wave without s_barrier; what is needed is only that the compiler keeps the
// order (and it waits for a read's data before using it anyway).  The ~100 __syncthreads() of a solve (s_barrier plus a full
// counter drain each) were most of its latency.
constexpr int kSolverThreads = 64;
__device__ __forceinline__ void wave_sync() {
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    __builtin_amdgcn_wave_barrier();
}

__device__ __forceinline__ void solve_from_basis(SolveLds &L, int lane, PolyRec *__restrict__ rec) {
    // ---- 2. trilinear coefficient tensors (lane = ordered index triple (i,j,k)), symmetrised into A ----
    {
        const int i = lane >> 4, j = (lane >> 2) & 3, k = lane & 3;
        const double *Ei = L.EE[i], *Ej = L.EE[j], *Ek = L.EE[k];
        double T[10];
        // det: rows 0,1,2 of E taken from basis i,j,k
        T[0] = Ei[0] * (Ej[4] * Ek[8] - Ej[5] * Ek[7]) - Ei[1] * (Ej[3] * Ek[8] - Ej[5] * Ek[6]) +
               Ei[2] * (Ej[3] * Ek[7] - Ej[4] * Ek[6]);
        // P = Ei * Ej^T, tr = trace(P);  C = P * Ek - 1/2 tr * Ek
        double P[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int b = 0; b < 3; ++b) P[r][b] = Ei[r * 3] * Ej[b * 3] + Ei[r * 3 + 1] * Ej[b * 3 + 1] + Ei[r * 3 + 2] * Ej[b * 3 + 2];
        const double htr = 0.5 * (P[0][0] + P[1][1] + P[2][2]);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c)
                T[1 + r * 3 + c] = P[r][0] * Ek[c] + P[r][1] * Ek[3 + c] + P[r][2] * Ek[6 + c] - htr * Ek[r * 3 + c];

        // the distinct orderings of this lane's monomial (lanes 0..19): index triples a*16 + b*4 + c, from a table
        int perm[6];
        int np = 0;
        if (lane < 20) {
            np = kMonoNumPerms[lane];
#pragma unroll
            for (int u = 0; u < 6; ++u) perm[u] = kMonoPerms[lane][u];
        }
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            wave_sync();
#pragma unroll
            for (int r = 0; r < 5; ++r) L.F[r][lane] = T[half * 5 + r];
            wave_sync();
            if (lane < 20) {
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    double sacc = 0;
#pragma unroll
                    for (int u = 0; u < 6; ++u)
                        if (u < np) sacc += L.F[r][perm[u]];
                    L.A[half * 5 + r][lane] = sacc;
                }
            }
        }
    }
    wave_sync();

    // ---- 3. Gauss-Jordan with partial pivoting: A <- [I | inv(A1) A2] ----
    bool singular = false;
    // Per column ONE read phase and ONE write phase.  The pivot column is read as a batch (ten loads in flight) and searched in
    // registers (first maximum, as a loop from `col` would find it); the row swap, the scaling of the pivot row and the elimination are
    // folded into the single pass that rewrites all 200 elements anyway: new A[r] comes from old A[src(r)] with src swapping col and
    // piv, and the scaled pivot-row element sc = A[piv][j] / pivot is formed in registers exactly as the stored one used to be.  Same
    // operations, same roundings; the separate swap / scale phases cost two more LDS round trips and four more orderings per column.
    for (int col = 0; col < 10; ++col) {
        double cv[10];
#pragma unroll
        for (int r = 0; r < 10; ++r) cv[r] = L.A[r][col];
        int piv = col;
        double pmax = -1.0, pvt = 0.0;
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            const double v = fabs(cv[r]);
            const bool take = (r == col) || (r > col && v > pmax);
            pmax = take ? v : pmax;
            pvt = take ? cv[r] : pvt;
            piv = take ? r : piv;
        }
        if (pmax < DBL_EPSILON * 1e-3) {
            singular = true;
            break;
        }
        const double inv = 1.0 / pvt;
        double nv[4];
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = lane + 64 * t;
            if (e < 200) {
                const int r = e / 20, j = e - r * 20;
                const int src = (r == col) ? piv : ((r == piv) ? col : r);
                const double sc = L.A[piv][j] * inv;
                nv[t] = (r == col) ? sc : (L.A[src][j] - L.A[src][col] * sc);
            }
        }
        wave_sync();
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            const int e = lane + 64 * t;
            if (e < 200) {
                const int r = e / 20, j = e - r * 20;
                L.A[r][j] = nv[t];
            }
        }
        wave_sync();
    }
    if (singular) {  // wave-uniform
        if (lane == 0) rec->ok = 0.0;
        return;
    }

    // ---- 4. B(z) rows and the determinant polynomial ----
    if (lane < 39) {
        const int i = lane / 13, j = lane - i * 13;
        const double *r1 = &L.A[2 * i + 4][10];
        const double *r2 = &L.A[2 * i + 5][10];
        double v1 = 0, v2 = 0;
        // row1: [1..3] <- r1[0..2], [5..7] <- r1[3..5], [9..12] <- r1[6..9];  row2: [0..2] <- r2[0..2], [4..6] <- r2[3..5], [8..11] <- r2[6..9]
        if (j >= 1 && j <= 3) v1 = r1[j - 1];
        else if (j >= 5 && j <= 7) v1 = r1[j - 2];
        else if (j >= 9) v1 = r1[j - 3];
        if (j <= 2) v2 = r2[j];
        else if (j >= 4 && j <= 6) v2 = r2[j - 1];
        else if (j >= 8 && j <= 11) v2 = r2[j - 2];
        L.b[i][j] = v1 - v2;
    }
    wave_sync();
    // det B(z) = sum over the 6 permutations of sgn * P[0][p0] * P[1][p1] * P[2][p2], with the entries of B as ascending polynomials
    // P[i][0] = b[i][3-k] (degree 3), P[i][1] = b[i][7-k] (degree 3), P[i][2] = b[i][12-k] (degree 4).  Two convolution stages spread over
    // the wave: lane (pi, m) forms coefficient m of P[0][p0] * P[1][p1] (48 lanes, <= 5 products each), then lane k adds up
    // sgn * (that product) * P[2][p2] for its coefficient k (11 lanes, <= 30 products each).  (The triple loop per output coefficient
    // this replaces ran ~25 index combinations x 6 permutations on 11 lanes -- the longest phase of the kernel.)
    {
        auto coef = [&](int row, int colm, int k) -> double {
            if (k < 0) return 0.0;
            if (colm == 0) return (k <= 3) ? L.b[row][3 - k] : 0.0;
            if (colm == 1) return (k <= 3) ? L.b[row][7 - k] : 0.0;
            return (k <= 4) ? L.b[row][12 - k] : 0.0;
        };
        // permutation pi = (p0, p1, p2): even ones first
        const int pi = lane >> 3, m = lane & 7;  // m = 0..7: degree of P0*P1 is at most 3 + 4
        const int p0 = (pi < 3) ? pi : pi - 3;
        const int p1 = (pi < 3) ? (pi + 1) % 3 : (pi - 3 + 2) % 3;
        if (lane < 48) {
            double q = 0;
#pragma unroll
            for (int i0 = 0; i0 <= 4; ++i0) q += coef(0, p0, i0) * coef(1, p1, m - i0);
            L.F[0][lane] = q;  // F is free again after phase 2
        }
        wave_sync();
        if (lane < 11) {
            double ck = 0;
#pragma unroll
            for (int pj = 0; pj < 6; ++pj) {
                const int q0 = (pj < 3) ? pj : pj - 3;
                const int q1 = (pj < 3) ? (pj + 1) % 3 : (pj - 3 + 2) % 3;
                const int q2 = 3 - q0 - q1;
                double sacc = 0;
#pragma unroll
                for (int i2 = 0; i2 <= 4; ++i2) {
                    const int mm = lane - i2;
                    if (mm >= 0 && mm <= 7) sacc += L.F[0][pj * 8 + mm] * coef(2, q2, i2);
                }
                ck += (pj < 3) ? sacc : -sacc;
            }
            L.c[lane] = ck;
        }
    }
    wave_sync();

    // ---- hand over to roots_kernel ----
    if (lane < 11) rec->c[lane] = L.c[lane];
    if (lane < 39) rec->b[lane] = L.b[lane / 13][lane % 13];
    if (lane < 36) rec->EE[lane] = L.EE[lane / 9][lane % 9];
    if (lane == 0) rec->ok = 1.0;
}

// Gauss-Newton polish of one solution (x, y, z) on the ten cubic constraints themselves (det E = 0, E E^T E - 1/2 tr(E E^T) E = 0 for
// E = x E0 + y E1 + z E2 + E3).  The elimination (10x10 Gauss-Jordan) and the degree-10 polynomial amplify rounding by the condition
// of the eliminated block, which depends on the null-space basis: on ~0.5 % of random 5-point samples the root path alone leaves E
// wrong by 1e-7..1e-5 (in the CPU path just as here, but on DIFFERENT samples, because its SVD basis differs from the Householder
// basis).  Two or three Newton steps on the constraints, which are well conditioned in E, take every accepted root to the solution of
// the polynomial system at rounding level (constraint residual <= 1e-13), so the models no longer depend on the basis.  A step is
// kept only while the residual norm decreases; a converged solution moves by ~1e-15.
__device__ __forceinline__ double constraint_eval(const double *EE, double x, double y, double z, double *step) {
    double E[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) E[k] = EE[k] * x + EE[9 + k] * y + EE[18 + k] * z + EE[27 + k];
    double G[9];  // E E^T
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) G[r * 3 + c] = E[r * 3] * E[c * 3] + E[r * 3 + 1] * E[c * 3 + 1] + E[r * 3 + 2] * E[c * 3 + 2];
    const double tr = G[0] + G[4] + G[8];
    double Cf[9];  // cofactors of E
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int i1 = (i + 1) % 3, i2 = (i + 2) % 3, j1 = (j + 1) % 3, j2 = (j + 2) % 3;
            Cf[i * 3 + j] = E[i1 * 3 + j1] * E[i2 * 3 + j2] - E[i1 * 3 + j2] * E[i2 * 3 + j1];
        }
    double F[10];
    F[0] = E[0] * Cf[0] + E[1] * Cf[1] + E[2] * Cf[2];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c)
            F[1 + r * 3 + c] = G[r * 3] * E[c] + G[r * 3 + 1] * E[3 + c] + G[r * 3 + 2] * E[6 + c] - 0.5 * tr * E[r * 3 + c];
    double f2 = 0;
#pragma unroll
    for (int k = 0; k < 10; ++k) f2 += F[k] * F[k];
    // Jacobian columns (directional derivatives along the basis matrices) -> normal equations
    double JtJ[6] = {0, 0, 0, 0, 0, 0}, JtF[3] = {0, 0, 0};
    double Jc[3][10];
#pragma unroll
    for (int v = 0; v < 3; ++v) {
        const double *D = EE + 9 * v;
        double dd = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) dd += Cf[k] * D[k];
        Jc[v][0] = dd;
        double DEt[9], EtE_col;  // D E^T
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) DEt[r * 3 + c] = D[r * 3] * E[c * 3] + D[r * 3 + 1] * E[c * 3 + 1] + D[r * 3 + 2] * E[c * 3 + 2];
        const double trd = DEt[0] + DEt[4] + DEt[8];  // tr(D E^T) = tr(E D^T)
        (void)EtE_col;
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) {
                // (D E^T) E + (E D^T) E + (E E^T) D - tr(E D^T) E - 1/2 tr(E E^T) D ;  E D^T = (D E^T)^T
                double a = 0;
#pragma unroll
                for (int k = 0; k < 3; ++k) a += (DEt[r * 3 + k] + DEt[k * 3 + r]) * E[k * 3 + c] + G[r * 3 + k] * D[k * 3 + c];
                Jc[v][1 + r * 3 + c] = a - trd * E[r * 3 + c] - 0.5 * tr * D[r * 3 + c];
            }
    }
#pragma unroll
    for (int k = 0; k < 10; ++k) {
        JtJ[0] += Jc[0][k] * Jc[0][k];
        JtJ[1] += Jc[0][k] * Jc[1][k];
        JtJ[2] += Jc[0][k] * Jc[2][k];
        JtJ[3] += Jc[1][k] * Jc[1][k];
        JtJ[4] += Jc[1][k] * Jc[2][k];
        JtJ[5] += Jc[2][k] * Jc[2][k];
        JtF[0] += Jc[0][k] * F[k];
        JtF[1] += Jc[1][k] * F[k];
        JtF[2] += Jc[2][k] * F[k];
    }
    // symmetric 3x3 solve by cofactors: [a b c; b d e; c e f]
    const double a = JtJ[0], b = JtJ[1], c = JtJ[2], d = JtJ[3], e = JtJ[4], f = JtJ[5];
    const double c00 = d * f - e * e, c01 = c * e - b * f, c02 = b * e - c * d;
    const double det = a * c00 + b * c01 + c * c02;
    if (det != 0 && det == det) {
        const double c11 = a * f - c * c, c12 = b * c - a * e, c22 = a * d - b * b;
        const double id = -1.0 / det;
        step[0] = (c00 * JtF[0] + c01 * JtF[1] + c02 * JtF[2]) * id;
        step[1] = (c01 * JtF[0] + c11 * JtF[1] + c12 * JtF[2]) * id;
        step[2] = (c02 * JtF[0] + c12 * JtF[1] + c22 * JtF[2]) * id;
    } else {
        step[0] = step[1] = step[2] = 0;
    }
    return f2;
}

// Squared residual of the ten cubic constraints only (no Jacobian): ~130 instructions against ~700 for constraint_eval.  *tr3 = |E|_F^6,
// the scale of f2 (the constraints are cubic in E).
__device__ __forceinline__ double constraint_residual(const double *EE, double x, double y, double z, double *tr3) {
    double E[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) E[k] = EE[k] * x + EE[9 + k] * y + EE[18 + k] * z + EE[27 + k];
    double G[9];  // E E^T
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) G[r * 3 + c] = E[r * 3] * E[c * 3] + E[r * 3 + 1] * E[c * 3 + 1] + E[r * 3 + 2] * E[c * 3 + 2];
    const double tr = G[0] + G[4] + G[8];
    const double det = E[0] * (E[4] * E[8] - E[5] * E[7]) - E[1] * (E[3] * E[8] - E[5] * E[6]) + E[2] * (E[3] * E[7] - E[4] * E[6]);
    double f2 = det * det;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) {
            const double F = G[r * 3] * E[c] + G[r * 3 + 1] * E[3 + c] + G[r * 3 + 2] * E[6 + c] - 0.5 * tr * E[r * 3 + c];
            f2 += F * F;
        }
    *tr3 = tr * tr * tr;
    return f2;
}

// Gauss-Newton on the ten cubic constraints in (x, y, z): a step is kept only while the residual falls.  One Jacobian evaluation plus one
// residual evaluation is the usual cost (the start is a root of the eliminated system: one step reaches rounding level, which the
// residual-only evaluation confirms); at most three steps.
__device__ __forceinline__ void polish_xyz(const double *EE, double &x, double &y, double &z) {
    double st[3];
    double pf = constraint_eval(EE, x, y, z, st);
    if (!(pf == pf)) return;
    for (int it = 0; it < 3; ++it) {
        if (!(st[0] == st[0] && st[1] == st[1] && st[2] == st[2])) return;
        const double nx = x + st[0], ny = y + st[1], nz = z + st[2];
        double tr3;
        const double f2 = constraint_residual(EE, nx, ny, nz, &tr3);
        if (!(f2 < pf)) return;  // no decrease (or NaN): keep the previous point
        x = nx, y = ny, z = nz, pf = f2;
        if (it == 2 || f2 <= 1e-30 * tr3) return;  // relative residual <= 1e-15: rounding level
        (void)constraint_eval(EE, x, y, z, st);
    }
}

// ---------------------------------------------------------------------------------------------------------------
// Steps 5..6: roots and models.  Durand-Kerner needs one lane per root, i.e. 10 lanes per hypothesis, so SIX hypotheses
// share a wave here (lane = 10*h + r) instead of idling 54 lanes of the solver wave.  All complex roots are found
// simultaneously from cv::solvePoly's start values (1+i)^k; the iteration is Ehrlich-Aberth (cubically convergent) rather
// than solvePoly's Durand-Kerner -- the roots a converged simultaneous iteration delivers are the same to rounding, only
// the sweep count differs (8 vs 27 on average); each hypothesis stops on its own once its corrections vanish to rounding;
// a root is real iff |imag| <= 1e-10 (five-point.cpp:438); per real root the null vector of Bz (3x3 one-sided Jacobi in
// registers), reject |xy1[2]| < 1e-10 (:457), E = x E0 + y E1 + z E2 + E3, Frobenius-normalised.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kHypPerWave = 6;
#ifndef MLPL_ROOTS_WAVES
#define MLPL_ROOTS_WAVES 3
#endif
#ifndef MLPL_SWEEP_CAP
#define MLPL_SWEEP_CAP 32
#endif
template <bool kPolish>  // compile-time: the polished instance does not carry the Jacobi SVD's registers
__device__ __forceinline__ void roots_body(const PolyRec *__restrict__ recs, int sample_offset, int n_samples,
                                           double *__restrict__ E_tab, int32_t *__restrict__ n_models,
                                           double *__restrict__ dense_E, int32_t *__restrict__ dense_id,
                                           int32_t *__restrict__ dense_total, int32_t *__restrict__ good_zero, int slot_stride, const int vbx) {
    constexpr bool polish = kPolish;
    __shared__ double R[kHypPerWave][88];
    __shared__ double rr[64], ri[64];
    if (blockDim.x != kSolverThreads) __builtin_trap();  // wave_sync() is a one-wave ordering
    const int lane = threadIdx.x;
    const int h = lane / 10, r = lane - h * 10;
    const int sample0 = sample_offset + vbx * kHypPerWave;
    if (slot_stride) {  // batched pass: a dense model list per slot (slot_stride is a multiple of kHypPerWave: a wave never straddles slots)
        const int a = sample0 / slot_stride;
        dense_total += a;
        dense_E += (size_t)a * slot_stride * 90;
        dense_id += (size_t)a * slot_stride * 10;
    }
    // cooperative load of the six records
    for (int i = lane; i < kHypPerWave * 88; i += 64) {
        const int hh = i / 88, k = i - hh * 88;
        const int smp = sample0 + hh;
        R[hh][k] = (smp < n_samples) ? reinterpret_cast<const double *>(recs + (smp - sample_offset))[k] : 0.0;
    }
    wave_sync();
    // the count table of the scoring pass is accumulated with atomics (correspondences split over four workgroups): start it at zero
    if (good_zero && h < kHypPerWave && sample0 + h < n_samples) good_zero[(size_t)(sample0 + h) * 10 + r] = 0;
    const bool lane_ok = (h < kHypPerWave) && (sample0 + h < n_samples) && (R[h < kHypPerWave ? h : 0][86] != 0.0);
    const int hs = h < kHypPerWave ? h : 0;
    const double *c = &R[hs][0];
    const double *b = &R[hs][11];
    const double *EE = &R[hs][50];
    int n = 10;
    for (; n > 1; n--)
        if (fabs(c[n]) > DBL_EPSILON) break;  // cv::solvePoly trims vanishing leading coefficients
    // Start values.  cv::solvePoly starts from (1+i)^k, |z| = 1 ... 22.6 whatever the polynomial; a converged simultaneous iteration
    // delivers the same ROOTS from any start, and nothing downstream depends on which lane holds which root (the models of a
    // hypothesis are told apart by inlier count and error sum, modelest.cpp:395-416; ARRSAC / USAC order them by E(0,0)).  So the starts are
    // put where the roots are: on two circles of radius rho / 2 and 2 rho around the origin, rho = |c_0 / c_n|^(1/n) the geometric mean
    // of the root moduli, alternating, at angles 2 pi k / 10 + 0.4 (not symmetric about the real axis: conjugate-symmetric starts of
    // a real polynomial can never separate a real pair).  Over 6000 polynomials of the C3 scene: 9.8 sweeps per hypothesis instead of
    // 13.5, 12.0 instead of 16.4 for the slowest of the six hypotheses of a wave.
    double pr, pim;
    {
        const double ca[10] = {0.9210609940028851, 0.5162596384230088, -0.08573535201472558, -0.6549823520202717, -0.9740483555854224, -0.9210609940028852, -0.5162596384230086, 0.0857353520147259, 0.6549823520202719, 0.9740483555854224};  // cos, sin of 2 pi k / 10 + 0.4
        const double sa[10] = {0.3894183423086505, 0.8564321255857607, 0.9963179459464289, 0.7556441745570417, 0.22634001188772324, -0.3894183423086503, -0.8564321255857609, -0.9963179459464289, -0.7556441745570415, -0.2263400118877229};
        double rho = 1.0;
        const double a0 = fabs(c[0]), an = fabs(c[n]);
        if (a0 > 0 && an > 0 && a0 <= DBL_MAX && an <= DBL_MAX) {
            // log2(a0 / an) from the exponents and a single-precision logarithm of the mantissa ratio; rho = 2^(that / n)
            const int e0 = __builtin_amdgcn_frexp_exp(a0), en = __builtin_amdgcn_frexp_exp(an);
            const float m0 = (float)__builtin_amdgcn_frexp_mant(a0), mn = (float)__builtin_amdgcn_frexp_mant(an);
            const float x = ((float)(e0 - en) + __log2f(m0 / mn)) / (float)n;
            const float xi = floorf(x);
            rho = ldexp((double)exp2f(x - xi), (int)xi);
        }
        rho *= (r & 1) ? 2.0 : 0.5;
        pr = rho * ca[r];
        pim = rho * sa[r];
    }
    const bool active = lane_ok && r < n;
    bool done = !lane_ok;  // identical for the 10 lanes of a hypothesis
    double cc[11];
#pragma unroll
    for (int k = 0; k < 11; ++k) cc[k] = c[k];
    int settle = 0, dk_sweeps = 0;
    for (int iter = 0; iter < MLPL_SWEEP_CAP; ++iter) {
        rr[lane] = pr;
        ri[lane] = pim;
        wave_sync();
        double dr = 0, di = 0;
        if (active && !done) {
            // Ehrlich-Aberth step: w = (p/p') / (1 - (p/p') * sum_{j != r} 1/(z - z_j));  p, p' by one Horner pass
            // g = g z + f, f = f z + c_j on fused multiply-adds (this file is built without contraction); the coefficients sit in
            // registers (cc, loaded once before the sweeps): read from LDS inside this loop every step waited for its own round trip
            double fr, fi = 0, gr = 0, gi = 0;  // f = p(z), g = p'(z)
            if (n == 10) {
                // The coefficients are real: synthetic division by the quadratic t^2 - s t + r with the roots z and conj z (s = 2 Re z,
                // r = |z|^2) costs two real FMAs per coefficient, b_k = c_k + s b_{k+1} - r b_{k+2}, and a second division of the quotient
                // gives the derivative: p(z) = (b_0 - x b_1) + i y b_1,  q(z) = (d_2 - x d_3) + i y d_3,  p'(z) = b_1 + 2 i y q(z).
                // Four FMAs per coefficient instead of the eight of a complex Horner pass for p and p' (a fifth of a sweep's instructions).
                const double sq = pr + pr, rq = __fma_rn(pr, pr, pim * pim);
                double b1 = 0.0, b0 = cc[10];   // b_{k+1}, b_k while walking down
                double d1 = 0.0, d0 = 0.0;      // the same for the quotient's coefficients b_2 .. b_10
#pragma unroll
                for (int j = 9; j >= 0; --j) {
                    // quotient recurrence consumes b_k for k >= 2: b0 holds b_{j+1} here
                    if (j >= 1) {
                        const double dn = __fma_rn(sq, d0, __fma_rn(-rq, d1, b0));
                        d1 = d0, d0 = dn;
                    }
                    const double bn = __fma_rn(sq, b0, __fma_rn(-rq, b1, cc[j]));
                    b1 = b0, b0 = bn;
                }
                // now b0 = b_0, b1 = b_1, d0 = d_2, d1 = d_3
                fr = __fma_rn(-pr, b1, b0);
                fi = pim * b1;
                const double qr = __fma_rn(-pr, d1, d0), qi = pim * d1;
                const double y2 = pim + pim;
                gr = __fma_rn(-y2, qi, b1);
                gi = y2 * qr;
            } else {  // vanishing leading coefficients were trimmed (cv::solvePoly does the same): rare
                fr = c[n];
                for (int j = n - 1; j >= 0; --j) {
                    const double t0 = __fma_rn(gr, pr, __fma_rn(-gi, pim, fr));
                    gi = __fma_rn(gr, pim, __fma_rn(gi, pr, fi));
                    gr = t0;
                    const double t1 = __fma_rn(fr, pr, __fma_rn(-fi, pim, c[j]));
                    fi = __fma_rn(fr, pim, fi * pr);
                    fr = t1;
                }
            }
            // The other roots of this hypothesis: all twenty LDS reads go out together, and the ten terms of the repulsion sum are
            // independent chains (as a loop over j < n with `continue` the terms ran one after the other, each behind its own LDS
            // round trip: two thirds of a sweep).  Terms beyond the degree, the root itself and coincident roots contribute zero.
            double zr[10], zi[10];
#pragma unroll
            for (int j = 0; j < 10; ++j) zr[j] = rr[h * 10 + j], zi[j] = ri[h * 10 + j];
            double sr = 0, si = 0;
#pragma unroll
            for (int j = 0; j < 10; ++j) {
                const double xr = pr - zr[j], xi = pim - zi[j];
                const double m2 = xr * xr + xi * xi;
                const bool use = j < n && j != r && m2 > 0;
                // Aberth's repulsion sum only steers the iteration (the fixed point is p(z) = 0 whatever S is: the correction is
                // q / (1 - q S) with q = p / p'), so the nine reciprocals per sweep are the bare v_rcp_f64 (~2^-26): one instruction
                // instead of the ~15 of an IEEE divide
                const double inv = __builtin_amdgcn_rcp(use ? m2 : 1.0);
                sr += use ? xr * inv : 0.0;
                si -= use ? xi * inv : 0.0;
            }
            const double g2 = gr * gr + gi * gi;
            if (g2 > 0) {
                // reciprocals by v_rcp_f64 + two Newton steps (to ~1 ulp): the correction vanishes with p(z) whatever their last bit is,
                // and an IEEE divide is a ~200-cycle dependent chain twice per sweep
                double ig = __builtin_amdgcn_rcp(g2);
                ig = __fma_rn(ig, __fma_rn(-g2, ig, 1.0), ig);
                ig = __fma_rn(ig, __fma_rn(-g2, ig, 1.0), ig);
                const double qr = (fr * gr + fi * gi) * ig, qi = (fi * gr - fr * gi) * ig;  // q = p/p'
                const double ur = 1.0 - (qr * sr - qi * si), ui = -(qr * si + qi * sr);     // 1 - q*S
                const double u2 = ur * ur + ui * ui;
                if (u2 > 0) {
                    double iu = __builtin_amdgcn_rcp(u2);
                    iu = __fma_rn(iu, __fma_rn(-u2, iu, 1.0), iu);
                    iu = __fma_rn(iu, __fma_rn(-u2, iu, 1.0), iu);
                    dr = (qr * ur + qi * ui) * iu;
                    di = (qi * ur - qr * ui) * iu;
                } else {
                    dr = qr, di = qi;
                }
            } else {
                dr = 1e-3 * (1.0 + fabs(pr)), di = 1e-3;  // stationary point of p: nudge off it
            }
            pr -= dr;
            pim -= di;
        }
        // convergence per hypothesis from two ballots (no LDS round trip, no sqrt / divide): a correction counts as large while
        // |d|^2 > 2e-18 (1 + |z|^2), i.e. |d| / (1 + |z|) above 1e-9 .. 1.4e-9 (never stricter than the 1e-9 of the sqrt form it replaces:
        // corrections at a double root stall around 1e-9 and a stricter test keeps such a wave sweeping for nothing)
        const double d2 = dr * dr + di * di, z2 = pr * pr + pim * pim;
        const bool moving = active && !done && d2 > 2e-18 * (1.0 + z2);
        const bool isnan = active && !done && !(d2 == d2 && z2 == z2);
        const unsigned long long hbits = ((1ull << 10) - 1ull) << (hs * 10);
        const bool any_moving = (__ballot(moving) & hbits) != 0, any_nan = (__ballot(isnan) & hbits) != 0;
        wave_sync();  // rr / ri are rewritten at the top of the next sweep
        if (!done) {
            dk_sweeps = iter + 1;
            // cubic convergence: once every correction is below 1e-9 (relative) one more sweep reaches rounding level
            if (any_nan) done = true;
            else if (!any_moving) done = (++settle >= 2);
            else settle = 0;
        }
        if (!__any(!done)) break;
    }
    if (g_dk_iters_dbg[3] && lane_ok && r == 0) {
        atomicAdd(&g_dk_iters_dbg[0], dk_sweeps);
        atomicAdd(&g_dk_iters_dbg[1], 1);
        if (atomicMax(&g_dk_iters_dbg[2], dk_sweeps) < dk_sweeps) g_dk_iters_dbg[4] = sample0 + h;
        const int edges[9] = {8, 12, 16, 24, 32, 64, 128, 256, 399};
        int bkt = 9;
        for (int e = 8; e >= 0; --e)
            if (dk_sweeps <= edges[e]) bkt = e;
        atomicAdd(&g_dk_iters_dbg[5 + bkt], 1);
    }

    // real roots -> essential matrices
    bool valid = false;
    double E[9];
    if (active && fabs(pim) <= 1e-10 && pr == pr) {
        const double z1 = pr, z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
        double bz[9];
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double *br = b + j * 13;
            bz[j * 3 + 0] = br[0] * z3 + br[1] * z2 + br[2] * z1 + br[3];
            bz[j * 3 + 1] = br[4] * z3 + br[5] * z2 + br[6] * z1 + br[7];
            bz[j * 3 + 2] = br[8] * z4 + br[9] * z3 + br[10] * z2 + br[11] * z1 + br[12];
        }
        double xy1[3];
        if constexpr (polish) null_vector_3x3_cross(bz, xy1);  // the polish step below puts (x, y, z) on the constraints whatever the last digits here
        else null_vector_3x3(bz, xy1);                         // plain root path: the SVD null vector, as the CPU path
        if (!(fabs(xy1[2]) < 1e-10)) {
            double x = xy1[0] / xy1[2], y = xy1[1] / xy1[2], zp = z1;
            if constexpr (polish) polish_xyz(EE, x, y, zp);
            double nrm = 0;
#pragma unroll
            for (int k = 0; k < 9; ++k) {
                E[k] = EE[k] * x + EE[9 + k] * y + EE[18 + k] * zp + EE[27 + k];
                nrm += E[k] * E[k];
            }
            nrm = sqrt(nrm);
#pragma unroll
            for (int k = 0; k < 9; ++k) E[k] /= nrm;
            valid = (nrm == nrm) && nrm > 0;
        }
    }
    // per-hypothesis compaction (slot = rank of this lane among the valid lanes of its hypothesis) + dense list
    const unsigned long long bal = __ballot(valid);
    const int total = __popcll(bal);
    const unsigned long long below = bal & ((1ull << lane) - 1ull);
    const unsigned long long hmask = (h < kHypPerWave) ? (((1ull << 10) - 1ull) << (h * 10)) : 0ull;
    const int slot = __popcll(below & hmask);
    const int cnt_h = __popcll(bal & hmask);
    int base = 0;
    if (dense_total) {
        if (lane == 0 && total > 0) base = atomicAdd(dense_total, total);
        base = __shfl(base, 0);
    }
    const int sample = sample0 + h;
    if (valid) {
        double *dst = E_tab + ((size_t)sample * 10 + slot) * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) dst[k] = E[k];
        if (dense_E) {
            const int pos = base + __popcll(below);
            double *dd = dense_E + (size_t)pos * 9;
#pragma unroll
            for (int k = 0; k < 9; ++k) dd[k] = E[k];
            dense_id[pos] = sample * 10 + slot;
        }
    }
    if (h < kHypPerWave && r == 0 && sample < n_samples) n_models[sample] = cnt_h;
}
template <bool kPolish>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(MLPL_ROOTS_WAVES, MLPL_ROOTS_WAVES))) void roots_kernel_t(const PolyRec *__restrict__ recs, int sample_offset, int n_samples,
                                                   double *__restrict__ E_tab, int32_t *__restrict__ n_models,
                                                   double *__restrict__ dense_E, int32_t *__restrict__ dense_id,
                                                   int32_t *__restrict__ dense_total, int32_t *__restrict__ good_zero, int slot_stride = 0) {
    roots_body<kPolish>(recs, sample_offset, n_samples, E_tab, n_models, dense_E, dense_id, dense_total, good_zero, slot_stride, blockIdx.x);
}

// launch helper: picks the instance by the context's solver_polish option
#define MLPL_LAUNCH_ROOTS(polish_flag, grid, stream, ...)                                                        \
    do {                                                                                                         \
        if (polish_flag) hipLaunchKernelGGL(roots_kernel_t<true>, grid, dim3(64), 0, stream, __VA_ARGS__);       \
        else hipLaunchKernelGGL(roots_kernel_t<false>, grid, dim3(64), 0, stream, __VA_ARGS__);                  \
    } while (0)

// Null-space basis of the five epipolar rows in L.Q (one wave): Householder QR of Q^T, then n_j = H_0 ... H_4 e_{5+j}.
__device__ __forceinline__ void householder_basis(SolveLds &L, int lane) {
    // ---- 1b. Householder QR of M = Q^T (9x5): M[r][c] = L.Q[c][r] ----
    for (int k = 0; k < 5; ++k) {
        double nrm2 = 0;
        for (int r = k; r < 9; ++r) nrm2 += L.Q[k][r] * L.Q[k][r];
        const double x0 = L.Q[k][k];
        const double alpha = (x0 >= 0 ? -1.0 : 1.0) * sqrt(nrm2);
        // v = x - alpha e_k ; |v|^2 = 2 (nrm2 - alpha x0)
        const double vn2 = 2.0 * (nrm2 - alpha * x0);
        const double hfac = vn2 > 0 ? 2.0 / vn2 : 0.0;
        wave_sync();
        if (lane < 9) L.V[k][lane] = (lane < k) ? 0.0 : ((lane == k) ? (x0 - alpha) : L.Q[k][lane]);
        if (lane == 0) L.vn2[k] = vn2 > 0 ? 2.0 / vn2 : 0.0;  // stored as the factor 2 / |v|^2 (0: no reflection)
        wave_sync();
        // apply H_k to the remaining columns c > k
        double upd = 0;
        bool doit = false;
        if (lane < 45) {
            const int c = lane / 9, r = lane - c * 9;
            doit = (c > k && r >= k && vn2 > 0);
            if (doit) {
                double dot = 0;
                for (int rr = k; rr < 9; ++rr) dot += L.V[k][rr] * L.Q[c][rr];
                upd = L.Q[c][r] - L.V[k][r] * dot * hfac;
            }
        }
        wave_sync();
        if (doit) L.Q[lane / 9][lane % 9] = upd;
        wave_sync();
    }
    // ---- 1c. null space: n_j = H_0 H_1 ... H_4 e_{5+j} ----
    if (lane < 36) {
        const int j = lane / 9, r = lane - j * 9;
        L.EE[j][r] = (r == 5 + j) ? 1.0 : 0.0;
    }
    wave_sync();
    for (int k = 4; k >= 0; --k) {
        double upd = 0;
        const int j = lane / 9, r = lane - j * 9;
        const double hfac = L.vn2[k];
        if (lane < 36) {
            double dot = 0;
            for (int rr = k; rr < 9; ++rr) dot += L.V[k][rr] * L.EE[j][rr];
            upd = L.EE[j][r] - L.V[k][r] * dot * hfac;
        }
        wave_sync();
        if (lane < 36) L.EE[j][r] = upd;
        wave_sync();
    }

}

__device__ __forceinline__ void solve5pt_body(const double *__restrict__ p1, const double *__restrict__ p2,
                                              const int32_t *__restrict__ samples, int sample_offset, int n_samples,
                                              PolyRec *__restrict__ recs /* indexed from sample_offset */,
                                              const PairSlot *__restrict__ ps, int slot_stride, const int vbx) {
    __shared__ SolveLds L;
    if (blockDim.x != kSolverThreads) __builtin_trap();  // wave_sync() is a one-wave ordering
    const int lane = threadIdx.x;
    const int sample = sample_offset + vbx;
    if (sample >= n_samples) return;
    if (ps) {  // batched pass: hypotheses beyond the slot's count are padding -- no models
        const int a = sample / slot_stride;
        if (sample - a * slot_stride >= ps[a].cnt) {
            if (lane == 0) recs[sample - sample_offset].ok = 0.0;
            return;
        }
    }

    // ---- 1a. epipolar rows Q[i] = [x1x2, y1x2, x2, x1y2, y1y2, y2, x1, y1, 1]  (five-point.cpp:375-383) ----
    if (lane < 5) {
        const int idx = samples[sample * 5 + lane];
        const double x1 = p1[2 * idx], y1 = p1[2 * idx + 1], x2 = p2[2 * idx], y2 = p2[2 * idx + 1];
        double *r = L.Q[lane];
        r[0] = x1 * x2, r[1] = y1 * x2, r[2] = x2, r[3] = x1 * y2, r[4] = y1 * y2, r[5] = y2, r[6] = x1, r[7] = y1, r[8] = 1.0;
    }
    wave_sync();
    householder_basis(L, lane);
    solve_from_basis(L, lane, recs + (sample - sample_offset));
}
__global__ __launch_bounds__(64) void solve5pt_kernel(const double *__restrict__ p1, const double *__restrict__ p2,
                                                      const int32_t *__restrict__ samples, int sample_offset, int n_samples,
                                                      PolyRec *__restrict__ recs /* indexed from sample_offset */,
                                                      const PairSlot *__restrict__ ps = nullptr, int slot_stride = 0) {
    solve5pt_body(p1, p2, samples, sample_offset, n_samples, recs, ps, slot_stride, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------
// solve5pt3_kernel: THREE hypotheses per wave, 20 lanes each (lanes 60..63 idle), the matrices held in REGISTERS.
// The one-hypothesis wave above keeps the 10x20 system in LDS and rewrites its 200 elements once per pivot column (ten LDS reads, an
// index divide and three more reads per element and column: ~2000 of its ~3500 wave instructions), with 20 of 64 lanes busy in most
// other phases.  Here lane (g, j) of hypothesis g owns COLUMN j of the system (ten doubles in registers): a pivot step is ten
// cross-lane broadcasts of the pivot column (ds_bpermute), the pivot search repeated by every lane in registers, one divide and ten
// multiply-subtract pairs -- no LDS traffic, no index arithmetic.  The Householder QR runs the same way (lane c < 5 owns epipolar row
// c; the reflection vector goes through LDS once per step), the null-space vectors are built by lanes j < 4 from the stored reflections.
// Only the 64-term trilinear tensor uses the whole wave, one hypothesis after the other, through the shared F buffer.
// Every floating-point operation, its operands and its order are those of solve5pt_kernel: the hand-over records are BIT-IDENTICAL
// (tests/test_gpu_solver_variants.py compares the two kernels through option solver_wave3).
// ---------------------------------------------------------------------------------------------------------------
constexpr int kSolveGroup = 20;       // lanes per hypothesis = columns of the constraint matrix
constexpr int kHypPerSolveWave = 3;
struct Solve3Lds {
    double V[kHypPerSolveWave][5][9];   // Householder vectors
    double vfac[kHypPerSolveWave][5];   // 2 / |v|^2 (0 = no reflection)
    double EE[kHypPerSolveWave][4][9];  // null-space bases
    double F[5][64];                    // trilinear tensor of ONE hypothesis, five constraint rows at a time
    double W[kHypPerSolveWave][160];    // [0,60) rows 4..9 x columns 10..19 of the eliminated system, [60,99) B(z), [100,148) P0 * P1
};

__device__ __forceinline__ void solve5pt3_body(const double *__restrict__ p1, const double *__restrict__ p2,
                                               const int32_t *__restrict__ samples, int sample_offset, int n_samples,
                                               PolyRec *__restrict__ recs /* indexed from sample_offset */,
                                               const PairSlot *__restrict__ ps, int slot_stride, const int vbx) {
    __shared__ Solve3Lds L;
    if (blockDim.x != kSolverThreads) __builtin_trap();  // wave_sync() is a one-wave ordering
    const int lane = threadIdx.x;
    const int gl = lane / kSolveGroup;
    const bool mine = gl < kHypPerSolveWave;
    const int g = mine ? gl : kHypPerSolveWave - 1;  // lanes 60..63 follow group 2 (reads only)
    const int j = lane - gl * kSolveGroup;
    const int sample = sample_offset + vbx * kHypPerSolveWave + g;
    bool live = mine && sample < n_samples;
    if (live && ps) {  // batched pass: hypotheses beyond the slot's count are padding -- no models
        const int a = sample / slot_stride;
        if (sample - a * slot_stride >= ps[a].cnt) {
            live = false;
            if (j == 0) recs[sample - sample_offset].ok = 0.0;
        }
    }
    const unsigned long long live_bits = __ballot(live);
    if (live_bits == 0) return;
    PolyRec *rec = recs + (live ? sample - sample_offset : 0);

    // ---- 1a. epipolar rows: lane (g, c < 5) holds row c = [x1x2, y1x2, x2, x1y2, y1y2, y2, x1, y1, 1]  (five-point.cpp:375-383) ----
    double q[9];
#pragma unroll
    for (int r = 0; r < 9; ++r) q[r] = 0.0;
    if (live && j < 5) {
        const int idx = samples[sample * 5 + j];
        const double x1 = p1[2 * idx], y1 = p1[2 * idx + 1], x2 = p2[2 * idx], y2 = p2[2 * idx + 1];
        q[0] = x1 * x2, q[1] = y1 * x2, q[2] = x2, q[3] = x1 * y2, q[4] = y1 * y2, q[5] = y2, q[6] = x1, q[7] = y1, q[8] = 1.0;
    }
    // ---- 1b. Householder QR of the 9x5 transpose: lane k forms reflection k, lanes c > k apply it to their rows ----
#pragma unroll
    for (int k = 0; k < 5; ++k) {
        if (live && j == k) {
            double nrm2 = 0;
#pragma unroll
            for (int r = k; r < 9; ++r) nrm2 += q[r] * q[r];
            const double x0 = q[k];
            const double alpha = (x0 >= 0 ? -1.0 : 1.0) * sqrt(nrm2);
            const double vn2 = 2.0 * (nrm2 - alpha * x0);  // |v|^2 for v = x - alpha e_k
#pragma unroll
            for (int r = 0; r < 9; ++r) L.V[g][k][r] = (r < k) ? 0.0 : ((r == k) ? (x0 - alpha) : q[r]);
            L.vfac[g][k] = vn2 > 0 ? 2.0 / vn2 : 0.0;
        }
        wave_sync();
        if (live && j > k && j < 5) {
            const double hfac = L.vfac[g][k];
            if (hfac != 0.0) {
                double v[9];
#pragma unroll
                for (int r = k; r < 9; ++r) v[r] = L.V[g][k][r];
                double dot = 0;
#pragma unroll
                for (int r = k; r < 9; ++r) dot += v[r] * q[r];
#pragma unroll
                for (int r = k; r < 9; ++r) q[r] = q[r] - v[r] * dot * hfac;
            }
        }
    }
    // ---- 1c. null space: lane (g, jn < 4) builds n_jn = H_0 H_1 ... H_4 e_{5+jn} ----
    if (live && j < 4) {
        double e[9];
#pragma unroll
        for (int r = 0; r < 9; ++r) e[r] = (r == 5 + j) ? 1.0 : 0.0;
#pragma unroll
        for (int k = 4; k >= 0; --k) {
            const double hfac = L.vfac[g][k];
            double v[9];
#pragma unroll
            for (int r = 0; r < 9; ++r) v[r] = L.V[g][k][r];
            double dot = 0;
#pragma unroll
            for (int r = k; r < 9; ++r) dot += v[r] * e[r];
#pragma unroll
            for (int r = 0; r < 9; ++r) e[r] = e[r] - v[r] * dot * hfac;
        }
#pragma unroll
        for (int r = 0; r < 9; ++r) {
            L.EE[g][j][r] = e[r];
            rec->EE[j * 9 + r] = e[r];
        }
    }
    wave_sync();

    // ---- 2. trilinear coefficient tensors, one hypothesis at a time over the whole wave (lane = index triple (i, j, k)); lane (g, m)
    //         adds up the orderings of monomial m: column m of the constraint matrix, a[0..9] ----
    double a[10];
#pragma unroll
    for (int r = 0; r < 10; ++r) a[r] = 0.0;
    int perm[6];
    const int np = (j < 20) ? kMonoNumPerms[j] : 0;
#pragma unroll
    for (int u = 0; u < 6; ++u) perm[u] = kMonoPerms[j][u];
    for (int t = 0; t < kHypPerSolveWave; ++t) {
        if (!((live_bits >> (t * kSolveGroup)) & 1ull)) continue;  // wave-uniform
        const int ti = lane >> 4, tj = (lane >> 2) & 3, tk = lane & 3;
        const double *Ei = L.EE[t][ti], *Ej = L.EE[t][tj], *Ek = L.EE[t][tk];
        double T[10];
        T[0] = Ei[0] * (Ej[4] * Ek[8] - Ej[5] * Ek[7]) - Ei[1] * (Ej[3] * Ek[8] - Ej[5] * Ek[6]) + Ei[2] * (Ej[3] * Ek[7] - Ej[4] * Ek[6]);
        double P[3][3];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int b = 0; b < 3; ++b) P[r][b] = Ei[r * 3] * Ej[b * 3] + Ei[r * 3 + 1] * Ej[b * 3 + 1] + Ei[r * 3 + 2] * Ej[b * 3 + 2];
        const double htr = 0.5 * (P[0][0] + P[1][1] + P[2][2]);
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) T[1 + r * 3 + c] = P[r][0] * Ek[c] + P[r][1] * Ek[3 + c] + P[r][2] * Ek[6 + c] - htr * Ek[r * 3 + c];
        const bool take = mine && gl == t;
#pragma unroll
        for (int half = 0; half < 2; ++half) {
            wave_sync();
#pragma unroll
            for (int r = 0; r < 5; ++r) L.F[r][lane] = T[half * 5 + r];
            wave_sync();
            if (take) {
#pragma unroll
                for (int r = 0; r < 5; ++r) {
                    double sacc = 0;
#pragma unroll
                    for (int u = 0; u < 6; ++u)
                        if (u < np) sacc += L.F[r][perm[u]];
                    a[half * 5 + r] = sacc;
                }
            }
        }
    }

    // ---- 3. Gauss-Jordan with partial pivoting on columns held in registers: A <- [I | inv(A1) A2] ----
    bool singular = false;
#pragma unroll
    for (int col = 0; col < 10; ++col) {
        const int src_lane = g * kSolveGroup + col;
        double cv[10];
#pragma unroll
        for (int r = 0; r < 10; ++r) cv[r] = __shfl(a[r], src_lane);
        int piv = col;
        double pmax = fabs(cv[col]), pvt = cv[col];
#pragma unroll
        for (int r = col + 1; r < 10; ++r) {
            const double v = fabs(cv[r]);
            const bool take = v > pmax;
            pmax = take ? v : pmax;
            pvt = take ? cv[r] : pvt;
            piv = take ? r : piv;
        }
        if (pmax < DBL_EPSILON * 1e-3) singular = true;  // (the one-hypothesis kernel stops here; the record is flagged below)
        const double inv = 1.0 / pvt;
        double a_piv = a[col];
#pragma unroll
        for (int r = col + 1; r < 10; ++r) a_piv = (piv == r) ? a[r] : a_piv;
        const double sc = a_piv * inv;
        const double t_col = a[col] - cv[col] * sc;  // what row `piv` receives: the old row `col`, eliminated
#pragma unroll
        for (int r = 0; r < 10; ++r) {
            if (r == col) continue;
            const double nv = a[r] - cv[r] * sc;
            a[r] = (r > col && piv == r) ? t_col : nv;
        }
        a[col] = sc;
    }

    // ---- 4. B(z) rows and the determinant polynomial (20 lanes per hypothesis, several rounds) ----
    double *W = L.W[g];
    if (mine && j >= 10) {
#pragma unroll
        for (int r = 4; r < 10; ++r) W[(r - 4) * 10 + (j - 10)] = a[r];
    }
    wave_sync();
#pragma unroll
    for (int t = 0; t < 2; ++t) {
        const int e = j + kSolveGroup * t;
        if (mine && e < 39) {
            const int i = e / 13, jj = e - i * 13;
            const double *r1 = &W[(2 * i) * 10];
            const double *r2 = &W[(2 * i + 1) * 10];
            double v1 = 0, v2 = 0;
            if (jj >= 1 && jj <= 3) v1 = r1[jj - 1];
            else if (jj >= 5 && jj <= 7) v1 = r1[jj - 2];
            else if (jj >= 9) v1 = r1[jj - 3];
            if (jj <= 2) v2 = r2[jj];
            else if (jj >= 4 && jj <= 6) v2 = r2[jj - 1];
            else if (jj >= 8 && jj <= 11) v2 = r2[jj - 2];
            const double bv = v1 - v2;
            W[60 + e] = bv;
            if (live) rec->b[e] = bv;
        }
    }
    wave_sync();
    {
        const double *bb = W + 60;  // b[row][k] = bb[row * 13 + k]
        auto coef = [&](int row, int colm, int k) -> double {
            if (k < 0) return 0.0;
            if (colm == 0) return (k <= 3) ? bb[row * 13 + 3 - k] : 0.0;
            if (colm == 1) return (k <= 3) ? bb[row * 13 + 7 - k] : 0.0;
            return (k <= 4) ? bb[row * 13 + 12 - k] : 0.0;
        };
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int e = j + kSolveGroup * t;
            if (mine && e < 48) {
                const int pi = e >> 3, m = e & 7;
                const int p0 = (pi < 3) ? pi : pi - 3;
                const int p1i = (pi < 3) ? (pi + 1) % 3 : (pi - 3 + 2) % 3;
                double qv = 0;
#pragma unroll
                for (int i0 = 0; i0 <= 4; ++i0) qv += coef(0, p0, i0) * coef(1, p1i, m - i0);
                W[100 + e] = qv;
            }
        }
        wave_sync();
        if (mine && j < 11) {
            double ck = 0;
#pragma unroll
            for (int pj = 0; pj < 6; ++pj) {
                const int q0 = (pj < 3) ? pj : pj - 3;
                const int q1 = (pj < 3) ? (pj + 1) % 3 : (pj - 3 + 2) % 3;
                const int q2 = 3 - q0 - q1;
                double sacc = 0;
#pragma unroll
                for (int i2 = 0; i2 <= 4; ++i2) {
                    const int mm = j - i2;
                    if (mm >= 0 && mm <= 7) sacc += W[100 + pj * 8 + mm] * coef(2, q2, i2);
                }
                ck += (pj < 3) ? sacc : -sacc;
            }
            if (live) rec->c[j] = ck;
        }
    }
    if (live && j == 0) rec->ok = singular ? 0.0 : 1.0;
}
__global__ __launch_bounds__(64) void solve5pt3_kernel(const double *__restrict__ p1, const double *__restrict__ p2,
                                                       const int32_t *__restrict__ samples, int sample_offset, int n_samples,
                                                       PolyRec *__restrict__ recs /* indexed from sample_offset */,
                                                       const PairSlot *__restrict__ ps = nullptr, int slot_stride = 0) {
    solve5pt3_body(p1, p2, samples, sample_offset, n_samples, recs, ps, slot_stride, blockIdx.x);
}

// the solver's first kernel: three hypotheses per wave (default) or the one-hypothesis wave (option solver_wave3 = 0)
static inline void launch_solve5pt(mlpl_ctx *ctx, int m, hipStream_t s, const double *p1, const double *p2, const int32_t *samples, int sample_offset,
                                   int n_samples, PolyRec *recs, const PairSlot *ps = nullptr, int slot_stride = 0) {
    if (ctx->opt_solver_wave3)
        hipLaunchKernelGGL(solve5pt3_kernel, dim3((m + kHypPerSolveWave - 1) / kHypPerSolveWave), dim3(kSolverThreads), 0, s, p1, p2, samples,
                           sample_offset, n_samples, recs, ps, slot_stride);
    else
        hipLaunchKernelGGL(solve5pt_kernel, dim3(m), dim3(kSolverThreads), 0, s, p1, p2, samples, sample_offset, n_samples, recs, ps, slot_stride);
}

// State of the device-side replay of runRANSAC (see replay_kernel); defined here because pack_points_kernel also initialises it.
struct ReplayState {
    int32_t maxGood;      // best inlier count so far (0 = none)
    int32_t niters;       // current iteration bound
    int32_t iter;         // iterations executed so far
    int32_t stop;         // 1 once iter >= niters
    double errminsum;     // error sum of the model held
    long long best;       // global index iteration*10 + slot of the model held, -1 = none
    double E[9];          // the model held
    int32_t refit_models; // models produced by the refit step (-1 = refit not run)
    int32_t refit_taken;  // slot of the refit model taken, -1 = none
    long long models_scored;  // essential matrices scored over all passes (statistics)
    // iteration bounds the replay evaluated ON THE DEVICE (no host table): (inlier count g, T(g)) per record-breaking count,
    // verified by the host against its libm after the call (ransac driver); t_count > kTUsedMax = list overflow
    int32_t t_count;
    int32_t t_pad;
    int32_t t_g[48];
    int32_t t_val[48];
};
constexpr int kTUsedMax = 48;


// ---------------------------------------------------------------------------------------------------------------
// Sampson scoring: one thread per model.
// ---------------------------------------------------------------------------------------------------------------
// Also kp[i] (a polynomial in |x1|+|y1|+1 and |x2|+|y2|+1), stored behind the points: the per-point factor of the error band of the
// fused-multiply-add fast path of the inlier predicate (sampson_inlier_fma).
// batch form (counts != nullptr): blockIdx.y = pair; its coordinates start at pair * pair_stride, its packed block at
// pts + pair * pack_stride (in double4 units), its count is counts[pair], its replay state st[pair]
__global__ void pack_points_kernel(const double *__restrict__ p1, const double *__restrict__ p2, int n, double4 *__restrict__ pts,
                                   ReplayState *__restrict__ st, int niters, int32_t *__restrict__ zero_ints, int n_zero,
                                   const int32_t *__restrict__ counts = nullptr, int pair_stride = 0, size_t pack_stride = 0) {
    if (counts) {
        const int b = blockIdx.y;
        n = counts[b];
        p1 += (size_t)b * pair_stride * 2, p2 += (size_t)b * pair_stride * 2;
        pts += (size_t)b * pack_stride;
        st += b;
        zero_ints = nullptr, n_zero = 0;
    }
    if (st && blockIdx.x == 0 && threadIdx.x < 64) {  // start state of the replay + zeroed counters: saves two stream operations per call
        if (threadIdx.x < n_zero) zero_ints[threadIdx.x] = 0;
        if (threadIdx.x == 0) {
            st->maxGood = 0, st->niters = niters, st->iter = 0, st->stop = 0;
            st->errminsum = DBL_MAX;
            st->best = -1;
            for (int k = 0; k < 9; ++k) st->E[k] = 0.0;
            st->refit_models = -1, st->refit_taken = -1;
            st->models_scored = 0;
            st->t_count = 0, st->t_pad = 0;
        }
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const double x1 = p1[2 * i], y1 = p1[2 * i + 1], x2 = p2[2 * i], y2 = p2[2 * i + 1];
        pts[i] = make_double4(x1, y1, x2, y2);
        const double P1 = fabs(x1) + fabs(y1) + 1.0, P2 = fabs(x2) + fabs(y2) + 1.0, pmag = P1 * P2;
        const double kp = 1.001 * pmag * (pmag + 1.0 + (P1 * P1 + P2 * P2));
        reinterpret_cast<double *>(pts + n)[i] = kp;  // kp of sampson_inlier_fma
        // single-precision copy for the fp32 pre-filter of the counting kernels (count_models_f32_kernel): the coordinates rounded to
        // float and the per-point factor KPs of its error band, rounded up; +inf switches the filter off for a point that is out of range
        float *f = reinterpret_cast<float *>(reinterpret_cast<double *>(pts + n) + n) + (size_t)i * 5;
        const double Pm = fmax(P1, P2);
        const double kps = 1.01 * (pmag * pmag + Pm * Pm + kp);
        f[0] = (float)x1, f[1] = (float)y1, f[2] = (float)x2, f[3] = (float)y2;
        f[4] = (pmag <= 0x1p40 && kps < 0x1p120) ? (float)kps * (1.0f + 0x1p-22f) : INFINITY;
    }
}

// computeReprojError3 (five-point.cpp:490-502): numerator and denominator of the Sampson error in the reference's operation
// order (k-ordered 3-term sums, no contraction); err = (float)(N / D).
__device__ __forceinline__ void sampson_nd(const double *e, double x1, double y1, double x2, double y2, double &N, double &D) {
    const double Ex1_0 = __dadd_rn(__dadd_rn(__dmul_rn(e[0], x1), __dmul_rn(e[1], y1)), e[2]);
    const double Ex1_1 = __dadd_rn(__dadd_rn(__dmul_rn(e[3], x1), __dmul_rn(e[4], y1)), e[5]);
    const double Ex1_2 = __dadd_rn(__dadd_rn(__dmul_rn(e[6], x1), __dmul_rn(e[7], y1)), e[8]);
    const double x2tEx1 = __dadd_rn(__dadd_rn(__dmul_rn(x2, Ex1_0), __dmul_rn(y2, Ex1_1)), Ex1_2);
    const double Etx2_0 = __dadd_rn(__dadd_rn(__dmul_rn(e[0], x2), __dmul_rn(e[3], y2)), e[6]);
    const double Etx2_1 = __dadd_rn(__dadd_rn(__dmul_rn(e[1], x2), __dmul_rn(e[4], y2)), e[7]);
    const double a = __dmul_rn(Ex1_0, Ex1_0), b = __dmul_rn(Ex1_1, Ex1_1), c = __dmul_rn(Etx2_0, Etx2_0), d = __dmul_rn(Etx2_1, Etx2_1);
    D = __dadd_rn(__dadd_rn(__dadd_rn(a, b), c), d);
    N = __dmul_rn(x2tEx1, x2tEx1);
}

__device__ __forceinline__ float sampson_err_f32(const double *e, double x1, double y1, double x2, double y2) {
    double N, D;
    sampson_nd(e, x1, y1, x2, y2, N, D);
    return (float)__ddiv_rn(N, D);
}

// findInliers' predicate  (double)(float)(N / D) <= thresh^2  (modelest.cpp:79-81) WITHOUT the division.
// Let f* be the largest float <= thresh^2 and qmax the largest DOUBLE whose float rounding is <= f* (the midpoint above f*, or its
// predecessor when ties-to-even would round the midpoint up): the predicate is  RN64(N / D) <= qmax,  i.e.  N / D < M  (or = M on an
// even tie) with M = qmax + half an ulp.  With p = RN64(qmax * D):  M * D lies in (p - ulp(p)/2, p + 1.6 ulp(p)), so
//     N < p             =>  inlier         (N <= pred(p) <= p - ulp(p)/2 < M * D)
//     N > p (1 + 2^-50) =>  outlier        (N > p + 4 ulp(p) > M * D)
// and only for N within four ulps of p -- probability ~2^-50 per correspondence -- the reference arithmetic itself decides.
// qmax comes from the host (inlier_bound()); qmax <= 0 means "no usable bound" (thresh^2 outside the float range, e.g. above FLT_MAX
// where every finite error passes) and, like D below 1e-200 (degenerate model or point), takes the reference arithmetic.
__device__ __forceinline__ bool sampson_inlier(double N, double D, double qmax, double thresh2) {
    const double p = __dmul_rn(qmax, D);
    const bool safe = D >= 1e-200 && qmax > 0;
    if (safe && N < p) return true;
    if (safe && N > __dmul_rn(p, 1.0 + 0x1p-50)) return false;
    return (double)(float)__ddiv_rn(N, D) <= thresh2;
}

// The same predicate on fused multiply-adds: ~28 instructions instead of ~41, exact by construction.
// N' = s'^2 and D' are evaluated with FMAs (any order).  Against the reference-order values: every product chain e_ij x2_i x1_j passes
// at most 4 (here) + 6 (reference) roundings, so |s' - s_ref| <= 10.1 u T with T = sum |e_ij x2_i x1_j| <= max|e| * pmag; likewise
// every component q of D (Ex1_0, Ex1_1, E^T x2_0, E^T x2_1) has |q' - q_ref| <= 5.1 u max|e| pmag.  With delta = 2^-49 max|e| pmag
// (= 16 u ...: slack for the rounding of pmag itself):
//     |N_ref - N'| <= delta (2 |s'| + delta) (1 + 2^-50) + 2^-51 N'
//     |p_ref - p'| <= qmax (4 delta sqrt(D') + 4 delta^2) + 11 u p' <= 2 delta (qmax + p') (1 + 2^-40) + 11 u p'      (2 sqrt(D) <= 1 + D)
// and sampson_inlier() decides `inlier` for N_ref < p_ref and `outlier` for N_ref > p_ref (1 + 2^-50).  Both follow from
//     |N' - p'| > h := delta (2 |s'| + 2 (qmax + p') + delta) (1 + 2^-40) + 2^-47 (N' + p')
// with the sign of N' - p'.  Otherwise (a correspondence within ~1e-12 of the threshold, a NaN, an unusable qmax) the reference arithmetic
// itself runs.  d0 = 2^-49 * max|e| * (1 + 2^-40), per model.
// The band is then loosened into ONE product of a per-model and a per-point constant (4 instructions fewer per evaluation): with
// P1 = |x1|+|y1|+1, P2 = |x2|+|y2|+1, pmag = P1 P2, psq = P1^2 + P2^2 and m = max|e| one has |s'| <= m pmag and D' <= 2 m^2 psq (all up to
// a factor 1 + 1e-15), hence  2|s'| + 2(qmax + p') + delta <= (2m + d0) pmag + 2 qmax + 4 qmax m^2 psq <= cmax (pmag + 1 + psq)  with
// cmax = max(2m + d0, 2 qmax, 4 qmax m^2), and  delta = d0 pmag.  So
//     h <= km * kp + 2^-46 max(N', p'),     km = 1.001 d0 cmax (model_band),   kp = 1.001 pmag (pmag + 1 + psq) (pack_points_kernel)
// (the factors 1.001 cover every rounding in forming km, kp and the bounds above).
__device__ __forceinline__ bool sampson_inlier_fma(const double *e, double km, double x1, double y1, double x2, double y2, double kp,
                                                   double qmax, double thresh2) {
    const double A = __fma_rn(e[0], x1, __fma_rn(e[1], y1, e[2]));
    const double B = __fma_rn(e[3], x1, __fma_rn(e[4], y1, e[5]));
    const double C = __fma_rn(e[6], x1, __fma_rn(e[7], y1, e[8]));
    const double sv = __fma_rn(x2, A, __fma_rn(y2, B, C));
    const double A2 = __fma_rn(e[0], x2, __fma_rn(e[3], y2, e[6]));
    const double B2 = __fma_rn(e[1], x2, __fma_rn(e[4], y2, e[7]));
    const double D = __fma_rn(A, A, __fma_rn(B, B, __fma_rn(A2, A2, B2 * B2)));
    const double N = sv * sv;
    const double p = qmax * D;
    const double h = __fma_rn(0x1p-46, fmax(N, p), km * kp);
    const double diff = N - p;
    if (fabs(diff) > h && qmax > 0) return diff < 0;  // false for NaNs: they take the reference path
    double Nr, Dr;
    sampson_nd(e, x1, y1, x2, y2, Nr, Dr);
    return sampson_inlier(Nr, Dr, qmax, thresh2);
}

__device__ __forceinline__ double model_band(const double *e, double qmax) {
    double m = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) m = fmax(m, fabs(e[k]));
    const double d0 = m * (0x1p-49 * (1.0 + 0x1p-40));
    const double cmax = fmax(fmax(2.0 * m + d0, 2.0 * qmax), 4.0 * qmax * m * m);
    // NaN entries must give a NaN band (every correspondence then takes the reference path): fmax drops NaNs, the product below keeps them
    return 1.001 * d0 * cmax + 0.0 * (e[0] + e[1] + e[2] + e[3] + e[4] + e[5] + e[6] + e[7] + e[8]);
}

// 4 lanes per model (lane j takes the correspondences i = j mod 4, in order), 64 models per 256-thread block; the
// correspondences go through LDS in tiles of 512.  The float errors are accumulated in double per lane and combined as
// (s0 + s2) + (s1 + s3): four interleaved accumulators, the shape of an SSE2 cv::sum over CV_32F.
// SUMS = false: inlier counts only (no division, no float rounding, no sum): the RANSAC passes, whose error sums are computed
// afterwards for the few models that can still win (candidate_kernel / esum_models_kernel).
// 128 models per 512-thread workgroup, 512-point tiles (20 KiB of LDS).  Every workgroup streams all n correspondences through its
// tile, so fewer, larger workgroups halve that traffic (measured at C3: 128 threads 1.12 ms, 256 threads 1.05 ms, 512 threads
// 1.02 ms per call; walking the correspondences wave-uniformly through the scalar data path instead of LDS: 1.28 ms).
constexpr int kScoreThreads = 512;
constexpr int kScoreModels = kScoreThreads / 4;
constexpr int kScoreTile = 512;
constexpr int kScoreThreadsSmall = 128;  // count-only passes of a few thousand models: 32 models per workgroup, one 256-point tile each
constexpr int kScoreTileSmall = 256;
template <bool SUMS, int kThreads = kScoreThreads, int kTile = kScoreTile>
__global__ __launch_bounds__(kThreads) void score_models_kernel(const double4 *__restrict__ pts, int n, const double *__restrict__ E_list,
                                                           const int32_t *__restrict__ ids, const int32_t *__restrict__ total_ptr,
                                                           int total_host, double thresh2, double qmax, int32_t *__restrict__ good,
                                                           double *__restrict__ esum) {
    __shared__ double4 tile[kTile];
    __shared__ double tile_mag[SUMS ? 1 : kTile];
    const double *__restrict__ pmag = reinterpret_cast<const double *>(pts + n);
    const int total = total_ptr ? *total_ptr : total_host;
    if (blockIdx.x * (kThreads / 4) >= total) return;  // block-uniform
    const int tid = threadIdx.x;
    const int j = tid & 3;
    const int m = blockIdx.x * (kThreads / 4) + (tid >> 2);
    const bool live = m < total;
    double e[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) e[k] = live ? E_list[(size_t)m * 9 + k] : 0.0;
    const double d0 = model_band(e, qmax);
    int cnt = 0;
    double s = 0.0;
    // gridDim.y > 1 (count-only form): the tiles are dealt round-robin to gridDim.y workgroups per model group, which add their
    // counts into a zeroed table -- units a fifth the size even out the load over the CUs without re-reading any correspondence
    for (int base = blockIdx.y * kTile; base < n; base += kTile * gridDim.y) {
        const int rows = min(kTile, n - base);
        __syncthreads();
        for (int i = tid; i < rows; i += kThreads) {
            tile[i] = pts[base + i];
            if constexpr (!SUMS) tile_mag[i] = pmag[base + i];
        }
        __syncthreads();
        if (live) {
#pragma unroll 4
            for (int i = j; i < rows; i += 4) {
                const double4 p = tile[i];
                if constexpr (SUMS) {
                    const float err = sampson_err_f32(e, p.x, p.y, p.z, p.w);
                    cnt += ((double)err <= thresh2) ? 1 : 0;
                    s = __dadd_rn(s, (double)err);
                } else {
                    cnt += sampson_inlier_fma(e, d0, p.x, p.y, p.z, p.w, tile_mag[i], qmax, thresh2) ? 1 : 0;
                }
            }
        }
    }
    // combine the 4 lanes of a model: counts add up, sums as (s0 + s2) + (s1 + s3)
    const int lane = tid & 63, gb = lane & ~3;
    const double s0 = __shfl(s, gb), s1 = __shfl(s, gb + 1), s2 = __shfl(s, gb + 2), s3 = __shfl(s, gb + 3);
    cnt += __shfl_xor(cnt, 1);
    cnt += __shfl_xor(cnt, 2);
    if (live && j == 0) {
        const int o = ids ? ids[m] : m;
        if (gridDim.y > 1) atomicAdd(&good[o], cnt);
        else good[o] = cnt;
        if constexpr (SUMS) esum[o] = __dadd_rn(__dadd_rn(s0, s2), __dadd_rn(s1, s3));
    }
}


// ---- count-only scoring with an fp32 pre-filter -------------------------------------------------------------------------------------
// The inlier predicate N <= qmax D is first evaluated in single precision on packed operations (v_pk_fma_f32: two correspondences per
// instruction) and accepted only outside a rigorous error band; the (few) evaluations inside the band -- and any point or model whose
// scale is out of range -- re-run the fp64 predicate (sampson_inlier_fma, itself exact by construction).  Same counts as every other
// scoring kernel.
//
// Error analysis (u = 2^-24; s, N, D, p the exact values on the double inputs; *32 the fp32 results from float-rounded inputs by FMAs):
//   |s32 - s| <= 8u T,  T = sum |e_ij x2_i x1_j| <= emax pmag   (three input roundings + four FMA roundings per term)   =: ds
//   |N32 - N| <= ds (2 |s32| + ds) + 1.01u N32 <= (2^-6 + 3u) N32 + 65 ds^2                       (2ab <= 2^-6 a^2 + 2^6 b^2)
//   every component q of D: |q32 - q| <= 5u emax Pmax =: dq;   |D32 - D| <= 4 dq sqrt(D32) + 4 dq^2 + 4.1u D32
//                                                                <= 15u D32 + 10.1u emax^2 Pmax^2      (2 sqrt(z) <= c + z/c, c = emax Pmax)
//   |p32 - p| <= 18u p32 + 10.2u qmax emax^2 Pmax^2                                               (p32 = fl(fl(qmax) D32))
// and the fp64 predicate is certain, with the sign of N - p, once |N - p| exceeds its own band km kp + 2^-46 max(N, p).  Hence
//   |N32 - p32| > H := KMs KPs + 2^-6 * 1.02 (N32 + p32),
//   KMs = 1.01 max(65 (8.01u emax)^2, 10.2u qmax emax^2, km)   per model,     KPs = 1.01 (pmag^2 + Pmax^2 + kp)   per point
// decides with the sign of N32 - p32.  Ranges: emax in [2^-40, 2^20], qmax in [2^-60, 1], pmag <= 2^40 keep every intermediate inside the
// normal float range or make its underflow error negligible against KMs KPs (>= 2^-116); outside them KMs / KPs is +inf, an overflow
// makes N32 + p32 infinite: H = +inf or NaN and the comparison fails -> fp64 path.
typedef float f32x2 __attribute__((ext_vector_type(2)));

// MPL = models per lane (option ransac_count_mpl, default 2 for the large passes).  Round 4 measured the kernel's hot loop in the ISA: per pair
// of evaluations 21 packed + 8 scalar-lane vector instructions (116 issue cycles per wave on a SIMD), i.e. the C3 counting pass is at ~0.67
// of the VECTOR ISSUE rate with 39 algorithmic FLOP per evaluation priced at 0.41 of the fp32 FLOP peak -- the gap between the two figures
// is instructions that are not multiply-adds (comparisons, counts, the band), not waiting.  MPL = 2 (a lane applies every pair of
// correspondences it reads from LDS to two models: half the LDS traffic per evaluation) first needed 138 VGPRs, halved the waves per SIMD
// and was slower (0.34 against 0.27 ms at C3); with the fp64 copy of the model no longer held across the loop (the rare undecided
// evaluation reads it again) it needs 126, keeps four waves per SIMD and is 1-4 % faster (C3 0.283 against 0.285-0.309 ms, C5 counting
// 3.22-3.27 against 3.36 ms per 512 pairs; tools/count_mpl_ab.py, same counts) -- the LDS reads were not what the loop waits for.
// DEFER (option ransac_count_defer, round 5; default).  Round 4 found that a build WITHOUT the fp64 path (wrong counts) runs the C3 pass in 0.25
// instead of 0.33 ms although only ~3e-5 of the evaluations take that path: what it costs is its presence in the loop (registers, code, the
// waits around it), not its executions.  Here the loop holds no fp64 code at all: an evaluation the band does not decide is QUEUED in LDS
// as (model of the workgroup, correspondence) -- one LDS atomic and one LDS store on the undecided lanes -- and the queue is decided after
// the tile loop by all threads of the workgroup, one entry per thread: the same predicate on the same operands, its verdict added to the
// model's count through an LDS counter.  Same counts by construction (a sum of the same 0 / 1 verdicts).  A queue that overflows
// (kDeferCap entries; a band that is infinite -- out-of-range scales -- queues everything) makes the workgroup recount its share in fp64
// afterwards, outside the loop as well.
constexpr int kDeferCap = 2048;
template <int kThreads, int kTile, int MPL = 1, bool DEFER = false, int WPE = 0>   // WPE: waves per SIMD the register budget is cut for (0 = by shape)
__global__ __launch_bounds__(kThreads, (WPE ? WPE : (DEFER ? (kThreads == 256 ? 5 : 4) : 1))) void count_models_f32_kernel(const double4 *__restrict__ pts, int n, const double *__restrict__ E_list,
                                                                    const int32_t *__restrict__ ids, const int32_t *__restrict__ total_ptr,
                                                                    int total_host, double thresh2, double qmax, int32_t *__restrict__ good,
                                                                    const PairSlot *__restrict__ ps = nullptr, int slot_stride = 0) {
    if (ps) {  // batched pass: blockIdx.z = slot
        const int a = blockIdx.z;
        pts = ps[a].pts, n = ps[a].n;
        E_list += (size_t)a * slot_stride * 90, ids += (size_t)a * slot_stride * 10, total_ptr += a;
    }
    // pair layout: the two consecutive correspondences of a lane class (j = i & 3) sit side by side, so a lane reads its pair's
    // (x1a,x1b, y1a,y1b, x2a,x2b, y2a,y2b) with two ds_read_b128 and (KPs_a, KPs_b) with one ds_read_b64
    __shared__ __attribute__((aligned(16))) float tile_xy[(kTile / 2) * 8];
    __shared__ __attribute__((aligned(8))) float tile_k[(kTile / 2) * 2];
    constexpr int kPerBlock = (kThreads / 4) * MPL;
    __shared__ int tile_kmax_bits;                        // DEFER: the tile's largest per-point band constant (non-negative floats order as their bit patterns)
    __shared__ uint32_t dq[DEFER ? kDeferCap : 1];        // (model of this workgroup) << 23 | correspondence
    __shared__ int dq_n[2];                               // entries pushed (may exceed the capacity: overflow) | unused
    __shared__ int dq_cnt[DEFER ? kPerBlock : 1];         // inliers the deferred evaluations add to the workgroup's models
    const double *__restrict__ kp64 = reinterpret_cast<const double *>(pts + n);
    const float *__restrict__ rec = reinterpret_cast<const float *>(kp64 + n);
    const int total = total_ptr ? *total_ptr : total_host;
    if (blockIdx.x * kPerBlock >= total) return;  // block-uniform
    const int tid = threadIdx.x;
    if constexpr (DEFER) {
        for (int i = tid; i < kPerBlock; i += kThreads) dq_cnt[i] = 0;
        if (tid < 2) dq_n[tid] = 0;   // (visible behind the first barrier of the tile loop, before anyone pushes)
    }
    const int j = tid & 3;
    const int m0 = blockIdx.x * kPerBlock + (tid >> 2) * MPL;
    const double u = 0x1p-24;
    bool live[MPL];
    f32x2 E[MPL][9], KM[MPL];
    int cnt[MPL];
#pragma unroll
    for (int mi = 0; mi < MPL; ++mi) {
        live[mi] = m0 + mi < total;
        double e[9];  // (not kept: the fp64 predicate of the rare undecided evaluation reads the model again -- eighteen registers per model)
#pragma unroll
        for (int k = 0; k < 9; ++k) e[k] = live[mi] ? E_list[(size_t)(m0 + mi) * 9 + k] : 0.0;
        const double km = model_band(e, qmax);
        double emax = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) emax = fmax(emax, fabs(e[k]));
        const double kms64 = 1.01 * fmax(fmax(65.0 * (8.01 * u * emax) * (8.01 * u * emax), 10.2 * u * qmax * emax * emax), km);
        const bool in_range = emax >= 0x1p-40 && emax <= 0x1p20 && qmax >= 0x1p-60 && qmax <= 1.0 && km == km;
        const float kms = in_range ? (float)kms64 * (1.0f + 0x1p-22f) : INFINITY;
#pragma unroll
        for (int k = 0; k < 9; ++k) E[mi][k] = f32x2{(float)e[k], (float)e[k]};
        KM[mi] = f32x2{kms, kms};
        cnt[mi] = 0;
    }
    auto exact = [&](int mi, int i) {  // sampson_inlier_fma on correspondence i for model mi
        // (Measured: forming the addresses below only here -- an asm barrier on i keeps them out of the loop, five vector instructions of
        // its fifty-eight -- changes nothing: 0.284 ms either way.  Like the halved LDS reads of MPL = 2, which gave 1-4 %: the loop is
        // not bound by the number of its vector instructions.)
        double e[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) e[k] = E_list[(size_t)(m0 + mi) * 9 + k];
        const double4 p = pts[i];
        return sampson_inlier_fma(e, model_band(e, qmax), p.x, p.y, p.z, p.w, kp64[i], qmax, thresh2) ? 1 : 0;
    };
    const f32x2 Q = {(float)qmax, (float)qmax}, C6 = {0x1p-6f * 1.02f, 0x1p-6f * 1.02f};
    // deferred evaluations: the push (lane-divergent, a few LDS instructions; the verdict comes later through dq_cnt)
    const int ml0 = (tid >> 2) * MPL;  // first model of this lane within the workgroup
    // `prov` = what the loop has already counted for this evaluation (DEFER: the sign bit of its fp32 difference, see below; 0 for the
    // unpaired correspondence): the queue's verdict replaces it, i.e. adds (verdict - prov)
    auto undecided = [&](int mi, int i, uint32_t prov = 0u) -> int {
        if constexpr (DEFER) {
            // (a queue that has overflowed stays overflowed and is not counted further: with an infinite band EVERY evaluation comes here,
            // up to 256 models x 2^23 correspondences per workgroup -- the counter must not wrap)
            if (*reinterpret_cast<volatile int *>(&dq_n[0]) <= kDeferCap) {
                const int k = atomicAdd(&dq_n[0], 1);
                if (k < kDeferCap) dq[k] = (prov << 31) | ((uint32_t)(ml0 + mi) << 23) | (uint32_t)i;   // (n < 2^23, kPerBlock <= 256: the launcher's conditions for this instance)
            }
            return 0;
        } else {
            return exact(mi, i);
        }
    };
    static_assert(!DEFER || kPerBlock <= 256, "queue entry: 8 bits of model, 23 of correspondence, 1 provisional verdict");
    uint32_t sg[MPL];   // DEFER: sign bits of the fp32 differences of the last <= 32 evaluations per model (the newest in bit 0)
#pragma unroll
    for (int mi = 0; mi < MPL; ++mi) sg[mi] = 0u;
    // (Measured and not kept, round 4: fetching the NEXT tile into registers while the current one is evaluated -- no change: with two
    // workgroups per CU the other workgroup's evaluation already covers a tile's load latency.)
    for (int base = blockIdx.y * kTile; base < n; base += kTile * gridDim.y) {
        const int rows = min(kTile, n - base);
        __syncthreads();
        if constexpr (DEFER) {
            if (tid == 0) tile_kmax_bits = 0;
            __syncthreads();
        }
        float kmax_mine = 0.f;
        for (int i = tid; i < rows; i += kThreads) {
            const float *r = rec + (size_t)(base + i) * 5;
            const int k = i >> 2, slot = ((k >> 1) * 4 + (i & 3)), half = k & 1;
            float *xy = tile_xy + slot * 8 + half;
            xy[0] = r[0], xy[2] = r[1], xy[4] = r[2], xy[6] = r[3];
            if constexpr (DEFER) kmax_mine = (r[4] > kmax_mine || !(r[4] == r[4])) ? r[4] : kmax_mine;   // a NaN constant must win (it disables the fast path)
            else tile_k[slot * 2 + half] = r[4];
        }
        if constexpr (DEFER) {
            // One band constant per TILE instead of one per correspondence (round 5): H needs an upper bound of KMs * KPs, and KPs = 1.01 (pmag^2 +
            // Pmax^2 + kp) varies by at most a factor ~2 over the correspondences (Pmax is the global maximum), so the tile's largest serves all
            // of them -- one packed multiply and one LDS read less per pair of evaluations; the band grows only in its absolute term.
            int kb = __float_as_int(kmax_mine);
            kb = (kmax_mine == kmax_mine) ? kb : 0x7FC00000;   // NaN orders above +inf as a bit pattern
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) kb = max(kb, __shfl_xor(kb, off));
            if ((tid & 63) == 0) atomicMax(&tile_kmax_bits, kb);
        }
        __syncthreads();
        if (live[0]) {
            const int nk = (rows - j + 3) >> 2;  // correspondences of this lane's class in the tile: i = 4k + j, k < nk
            const int npairs = nk >> 1;
            f32x2 KMKP[MPL];   // DEFER: the absolute term of the band for this tile, per model
            if constexpr (DEFER) {
                const float kt = __int_as_float(tile_kmax_bits);
#pragma unroll
                for (int mi = 0; mi < MPL; ++mi) KMKP[mi] = KM[mi] * f32x2{kt, kt} * f32x2{1.0f + 0x1p-22f, 1.0f + 0x1p-22f};   // (rounded up)
            }
            // the fp32 difference and band of one pair of correspondences (packed) under model mi
            auto diff_band = [&](const int mi, const float4 &v0, const float4 &v1, const f32x2 KP, f32x2 &diff, f32x2 &H) {
                const f32x2 X1 = {v0.x, v0.y}, Y1 = {v0.z, v0.w}, X2 = {v1.x, v1.y}, Y2 = {v1.z, v1.w};
                const f32x2 *Em = E[mi];
                const f32x2 A = __builtin_elementwise_fma(Em[0], X1, __builtin_elementwise_fma(Em[1], Y1, Em[2]));
                const f32x2 B = __builtin_elementwise_fma(Em[3], X1, __builtin_elementwise_fma(Em[4], Y1, Em[5]));
                const f32x2 C = __builtin_elementwise_fma(Em[6], X1, __builtin_elementwise_fma(Em[7], Y1, Em[8]));
                const f32x2 S = __builtin_elementwise_fma(X2, A, __builtin_elementwise_fma(Y2, B, C));
                const f32x2 A2 = __builtin_elementwise_fma(Em[0], X2, __builtin_elementwise_fma(Em[3], Y2, Em[6]));
                const f32x2 B2 = __builtin_elementwise_fma(Em[1], X2, __builtin_elementwise_fma(Em[4], Y2, Em[7]));
                const f32x2 D = __builtin_elementwise_fma(A, A, __builtin_elementwise_fma(B, B, __builtin_elementwise_fma(A2, A2, B2 * B2)));
                const f32x2 N = S * S;
                // N32 -+ fl(qmax) D32 with ONE rounding each (the analysis above rounds the product first: at most an ulp of either
                // quantity more, far inside the 1.02 / 1.01 slack of H)
                diff = __builtin_elementwise_fma(-Q, D, N);
                if constexpr (DEFER) H = __builtin_elementwise_fma(C6, __builtin_elementwise_fma(Q, D, N), KMKP[mi]);
                else H = __builtin_elementwise_fma(C6, __builtin_elementwise_fma(Q, D, N), KM[mi] * KP);
            };
            // DEFER (round 6): NP pairs x MPL models = NP * MPL independent chains are computed FIRST and ONE rare branch follows them (a branch
            // between the chains kept the compiler from interleaving them: every dependent packed op waited out its predecessor).  The
            // non-multiply-add instructions of an evaluation are ONE funnel shift and ONE comparison (were two comparisons and an
            // add-with-carry): decided <=> |diff| > H, and a decided evaluation is an inlier exactly when diff is negative -- the SIGN BIT of
            // diff is shifted into sg[mi] (v_alignbit_b32) for every evaluation and the bits are counted 32 at a time (v_bcnt) by the caller;
            // an undecided evaluation is queued together with the bit that was counted for it, and the queue adds (verdict - bit).
            // Same counts by construction.
            auto eval_group = [&](const int q, auto np_tag) {
                constexpr int NP = decltype(np_tag)::value;
                f32x2 df[NP][MPL], Hh[NP][MPL];
#pragma unroll
                for (int p = 0; p < NP; ++p) {
                    const int slot = (q + p) * 4 + j;
                    const float4 v0 = *reinterpret_cast<const float4 *>(tile_xy + slot * 8);
                    const float4 v1 = *reinterpret_cast<const float4 *>(tile_xy + slot * 8 + 4);
#pragma unroll
                    for (int mi = 0; mi < MPL; ++mi) diff_band(mi, v0, v1, f32x2{0.f, 0.f}, df[p][mi], Hh[p][mi]);
                }
                bool any_undecided = false;
#pragma unroll
                for (int p = 0; p < NP; ++p)
#pragma unroll
                    for (int mi = 0; mi < MPL; ++mi) {
                        sg[mi] = __builtin_amdgcn_alignbit(sg[mi], __float_as_uint(df[p][mi].x), 31);
                        sg[mi] = __builtin_amdgcn_alignbit(sg[mi], __float_as_uint(df[p][mi].y), 31);
                        // (every comparison is false for a NaN / inf band: undecided)
                        const bool c0 = __builtin_fabsf(df[p][mi].x) > Hh[p][mi].x, c1 = __builtin_fabsf(df[p][mi].y) > Hh[p][mi].y;
                        any_undecided |= live[mi] && !(c0 && c1);
                    }
                if (__builtin_expect(any_undecided, 0)) {  // inside the band (or out of range): the fp64 predicate decides, later
#pragma unroll
                    for (int p = 0; p < NP; ++p)
#pragma unroll
                        for (int mi = 0; mi < MPL; ++mi) {
                            if (!live[mi]) continue;
                            const int i0 = base + 4 * (2 * (q + p)) + j, i1 = i0 + 4;
                            if (!(__builtin_fabsf(df[p][mi].x) > Hh[p][mi].x)) undecided(mi, i0, __float_as_uint(df[p][mi].x) >> 31);
                            if (!(__builtin_fabsf(df[p][mi].y) > Hh[p][mi].y)) undecided(mi, i1, __float_as_uint(df[p][mi].y) >> 31);
                        }
                }
            };
            auto eval_pair = [&](const int q) {   // (!DEFER: the forms of rounds 3-4, for A/B)
                const int slot = q * 4 + j;
                const float4 v0 = *reinterpret_cast<const float4 *>(tile_xy + slot * 8);
                const float4 v1 = *reinterpret_cast<const float4 *>(tile_xy + slot * 8 + 4);
                const float2 kk = *reinterpret_cast<const float2 *>(tile_k + slot * 2);
                const f32x2 KP = f32x2{kk.x, kk.y};
#pragma unroll
                for (int mi = 0; mi < MPL; ++mi) {
                    f32x2 diff, H;
                    diff_band(mi, v0, v1, KP, diff, H);
                    // decided inlier: diff < -H; decided outlier: diff > H (H >= 0; every comparison is false for a NaN / inf band)
                    const bool in0 = diff.x < -H.x, in1 = diff.y < -H.y;
                    const bool c0 = in0 || diff.x > H.x, c1 = in1 || diff.y > H.y;
                    cnt[mi] += in0 ? 1 : 0;
                    cnt[mi] += in1 ? 1 : 0;
                    if (__builtin_expect(live[mi] && !(c0 && c1), 0)) {  // inside the band (or out of range): the fp64 predicate decides
                        const int i0 = base + 4 * (2 * q) + j, i1 = i0 + 4;
                        if (!c0) cnt[mi] += undecided(mi, i0);
                        if (!c1) cnt[mi] += undecided(mi, i1);
                    }
                }
            };
            // two pairs per trip, by hand: the loop body holds wave-wide (convergent) operations, which the unroller will not leave a remainder for.
            // (Measured and not kept, round 5: fetching the NEXT two pairs' operands from LDS before the current two are evaluated -- 126 registers
            // instead of 102, C3 scoring pass 0.278 against 0.263 ms, C5 counting 2.95 against 2.77 ms: gpurun_out/r5/count_defer_ab4.log.)
            if constexpr (DEFER) {
                // blocks of 16 pairs = 32 evaluations per model: sg[mi] then holds exactly the block's sign bits (older ones have left through
                // bit 31; a shorter last block is masked)
                for (int qb = 0; qb < npairs; qb += 16) {
                    const int qe = min(qb + 16, npairs);
                    int q = qb;
                    for (; q + 2 <= qe; q += 2) eval_group(q, std::integral_constant<int, 2>{});
                    if (q < qe) eval_group(q, std::integral_constant<int, 1>{});
                    const int r = 2 * (qe - qb);
                    const uint32_t keep = r >= 32 ? 0xFFFFFFFFu : ((1u << r) - 1u);
#pragma unroll
                    for (int mi = 0; mi < MPL; ++mi) cnt[mi] += __popc(sg[mi] & keep);
                }
            } else {
#pragma unroll 2
                for (int q = 0; q < npairs; ++q) eval_pair(q);
            }
            if (nk & 1) {  // the unpaired last correspondence of the class: fp64
                const int i0 = base + 4 * (nk - 1) + j;
#pragma unroll
                for (int mi = 0; mi < MPL; ++mi)
                    if (live[mi]) cnt[mi] += undecided(mi, i0);
            }
        }
    }
    if constexpr (DEFER) {
        __syncthreads();
        const int pushed = dq_n[0];
        if (pushed <= kDeferCap) {  // the queue: one entry per thread, the fp64 predicate on the same operands
            for (int k = tid; k < pushed; k += kThreads) {
                const uint32_t e = dq[k];
                const int ml = (int)((e >> 23) & 0xFFu), i = (int)(e & 0x7FFFFFu), prov = (int)(e >> 31);
                double ee[9];
#pragma unroll
                for (int q = 0; q < 9; ++q) ee[q] = E_list[(size_t)(blockIdx.x * kPerBlock + ml) * 9 + q];
                const double4 p = pts[i];
                const int fix = (sampson_inlier_fma(ee, model_band(ee, qmax), p.x, p.y, p.z, p.w, kp64[i], qmax, thresh2) ? 1 : 0) - prov;
                if (fix) atomicAdd(&dq_cnt[ml], fix);
            }
        } else {  // overflow (workgroup-uniform): this workgroup's share once more, every evaluation by the fp64 predicate
#pragma unroll
            for (int mi = 0; mi < MPL; ++mi) cnt[mi] = 0;
            for (int base = blockIdx.y * kTile; base < n; base += kTile * gridDim.y) {
                const int rows = min(kTile, n - base);
                for (int i = base + j; i < base + rows; i += 4)
#pragma unroll
                    for (int mi = 0; mi < MPL; ++mi)
                        if (live[mi]) cnt[mi] += exact(mi, i);
            }
        }
        __syncthreads();
    }
#pragma unroll
    for (int mi = 0; mi < MPL; ++mi) {
        int c = cnt[mi];
        c += __shfl_xor(c, 1);
        c += __shfl_xor(c, 2);
        if (live[mi] && j == 0) {
            if constexpr (DEFER) c += dq_cnt[ml0 + mi];
            const int o = ids ? ids[m0 + mi] : m0 + mi;
            if (gridDim.y > 1) atomicAdd(&good[o], c);
            else good[o] = c;
        }
    }
}


// Scoring for FEW models (an adaptive RANSAC pass of <= ~1000 hypotheses, the refit's <= 10 models): one 256-thread block per
// model.  score_models_kernel gives a model 4 lanes that walk all n correspondences, which is the right shape for 10^5 models
// but leaves the chip idle and takes n/4 serial error evaluations (~55 instructions each) when there are a few hundred.  Here
// the errors of a model are evaluated by all 256 threads into LDS; only the additions stay serial, because the error sum is
// DEFINED by its order (four interleaved double accumulators, element i -> accumulator i mod 4, combined as (s0+s2)+(s1+s3)):
// lane 0 reads four consecutive errors per ds_read_b128 and keeps the four chains in flight.  Same counts, same sums, bit for bit.
constexpr int kScoreBlockMaxN = 16384;  // floats of (dynamic) LDS per block: 4 n bytes, up to 64 KiB
// COUNT: write inlier counts; SUMS: write error sums (needs the n-float LDS buffer).  `pick` != NULL: the models to process are
// E_list[pick[c]] for c < total (candidate ids; outputs indexed by the id), else E_list[c] (outputs through `ids`).
template <bool COUNT, bool SUMS>
__global__ __launch_bounds__((SUMS && !COUNT) ? 1024 : 256) void score_models_block_kernel(const double4 *__restrict__ pts, int n, const double *__restrict__ E_list,
                                                                 const int32_t *__restrict__ ids, const int32_t *__restrict__ total_ptr,
                                                                 int total_host, double thresh2, double qmax, int32_t *__restrict__ good,
                                                                 double *__restrict__ esum, const int32_t *__restrict__ pick,
                                                                 const PairSlot *__restrict__ ps = nullptr, int slot_stride = 0) {
    if (ps) {  // batched pass: blockIdx.y = slot; `pick` ids and the outputs are local to the slot
        const int a = blockIdx.y;
        pts = ps[a].pts, n = ps[a].n;
        E_list += (size_t)a * slot_stride * 90, total_ptr += a, esum += (size_t)a * slot_stride * 10, pick += (size_t)a * slot_stride * 10;
        if (good) good += (size_t)a * slot_stride * 10;
    }
    extern __shared__ __attribute__((aligned(16))) float errs[];  // n floats when SUMS
    __shared__ int wave_cnt[4];  // (the counting instances run 256 threads; the sums-only one 1024: its errors are a latency-bound
                                 // gather of n points, and four times the loads in flight shorten it from 5.8 to ~2.5 us at n = 4000)
    const int total = total_ptr ? *total_ptr : total_host;
    const int tid = threadIdx.x;
    for (int c = blockIdx.x; c < total; c += gridDim.x) {  // block-uniform loop
        const int m = pick ? pick[c] : c;
        const int o = pick ? m : (ids ? ids[m] : m);
        double e[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) e[k] = E_list[(size_t)m * 9 + k];
        const double d0 = model_band(e, qmax);
        const double *__restrict__ pmag = reinterpret_cast<const double *>(pts + n);
        int cnt = 0;
        for (int i = tid; i < n; i += (int)blockDim.x) {
            const double4 p = pts[i];
            if constexpr (SUMS) {
                const float err = sampson_err_f32(e, p.x, p.y, p.z, p.w);
                errs[i] = err;
                if constexpr (COUNT) cnt += ((double)err <= thresh2) ? 1 : 0;
            } else {
                cnt += sampson_inlier_fma(e, d0, p.x, p.y, p.z, p.w, pmag[i], qmax, thresh2) ? 1 : 0;
            }
        }
        if constexpr (COUNT) {
#pragma unroll
            for (int d = 32; d > 0; d >>= 1) cnt += __shfl_xor(cnt, d);
            if ((tid & 63) == 0) wave_cnt[tid >> 6] = cnt;
        }
        __syncthreads();
        if constexpr (COUNT) {
            if (tid == 0) good[o] = wave_cnt[0] + wave_cnt[1] + wave_cnt[2] + wave_cnt[3];
        }
        if constexpr (SUMS) {
            if (tid < 4) {  // lane j = accumulator j: elements j, j + 4, ... in order; eight reads in flight per chain step
                double sacc = 0;
                int i = tid;
                for (; i + 28 < n; i += 32) {
                    const float v0 = errs[i], v1 = errs[i + 4], v2 = errs[i + 8], v3 = errs[i + 12], v4 = errs[i + 16], v5 = errs[i + 20],
                                v6 = errs[i + 24], v7 = errs[i + 28];
                    sacc = __dadd_rn(sacc, (double)v0);
                    sacc = __dadd_rn(sacc, (double)v1);
                    sacc = __dadd_rn(sacc, (double)v2);
                    sacc = __dadd_rn(sacc, (double)v3);
                    sacc = __dadd_rn(sacc, (double)v4);
                    sacc = __dadd_rn(sacc, (double)v5);
                    sacc = __dadd_rn(sacc, (double)v6);
                    sacc = __dadd_rn(sacc, (double)v7);
                }
                for (; i < n; i += 4) sacc = __dadd_rn(sacc, (double)errs[i]);
                const double s0 = __shfl(sacc, 0), s1 = __shfl(sacc, 1), s2 = __shfl(sacc, 2), s3 = __shfl(sacc, 3);
                if (tid == 0) esum[o] = __dadd_rn(__dadd_rn(s0, s2), __dadd_rn(s1, s3));
            }
        }
        __syncthreads();  // errs / wave_cnt are reused by the next model of this block
    }
}

// Which models can still win (modelest.cpp:377-416)?  The replay takes a hypothesis only if its best count beats, or ties, the
// best count of everything before it (a tie is decided by the error sums), so error sums are needed only for the models that
// hold their hypothesis' top count where that count is >= 5 and >= the running maximum before the hypothesis (incl. the count
// carried over from earlier passes).  One block: max-scan over the hypotheses in iteration order, candidates appended to a list.
__global__ void hyp_max_kernel(const int32_t *__restrict__ n_models, const int32_t *__restrict__ good, int cnt,
                               int32_t *__restrict__ hmax) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    const int nm = n_models[i];
    int h = 0;
    for (int m = 0; m < nm; ++m) h = max(h, good[(size_t)i * 10 + m]);
    hmax[i] = h;
}

// Scans over the hypotheses of a pass in iteration order, one 1024-thread block.  Wave w owns the contiguous range
// [w*R*64, (w+1)*R*64), R = ceil(cnt / 1024), and walks it in rows of 64 (lane = position in the row): loads are coalesced and
// independent of the scan state, row scans are wave shuffles, and the 16 waves meet only once per phase to exchange one value each
// (the chunk-per-step form of round 1 paid a global-load latency and up to three block barriers per 1024 hypotheses).
constexpr int kScanRows = 32;  // rows of 64 per wave: a pass holds at most 16 * 32 * 64 = 32768 hypotheses (the driver's pass size cap).
                               // A thread keeps its whole column in registers: every load of a scan is issued before the first use -- the
                               // tables were written by kernels on other XCDs, a dependent batch of loads costs ~2 us each time
constexpr int kMaxScanEvents = 1024;  // record / tie events of one pass that are expanded in parallel after the scan (one thread each)

template <bool kMax>
__device__ __forceinline__ int wave_inclusive_scan(int v, int lane) {
#pragma unroll
    for (int off = 1; off < 64; off <<= 1) {
        const int o = __shfl_up(v, off);
        if (lane >= off) v = kMax ? max(v, o) : min(v, o);
    }
    return v;
}
template <bool kMax>
__device__ __forceinline__ int wave_reduce(int v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) {
        const int o = __shfl_xor(v, off);
        v = kMax ? max(v, o) : min(v, o);
    }
    return v;
}
// Every wave contributes `v` (wave-uniform); returns the combination of `carry` and the contributions of the waves before this one.
template <bool kMax>
__device__ __forceinline__ int waves_exclusive_scan_16(int v, int carry, int *wave_s) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    __syncthreads();  // wave_s may still be read from an earlier exchange
    if (lane == 0) wave_s[wave] = v;
    __syncthreads();
    int pre = carry;
#pragma unroll
    for (int w = 0; w < 16; ++w) {
        const int c = wave_s[w];
        if (w < wave) pre = kMax ? max(pre, c) : min(pre, c);
    }
    return pre;
}

__global__ __launch_bounds__(1024) void candidate_kernel(const int32_t *__restrict__ n_models, const int32_t *__restrict__ good,
                                                         const int32_t *__restrict__ hmax_in, int cnt, int carried_best,
                                                         int32_t *__restrict__ cand, int32_t *__restrict__ cand_count, int ev_cap,
                                                         const PairSlot *__restrict__ ps = nullptr, int slot_stride = 0,
                                                         const ReplayState *__restrict__ st_all = nullptr) {
    if (ps) {  // batched pass: blockIdx.x = slot; the candidate ids are local to the slot
        const int a = blockIdx.x;
        n_models += (size_t)a * slot_stride, good += (size_t)a * slot_stride * 10, hmax_in += (size_t)a * slot_stride;
        cand += (size_t)a * slot_stride * 10, cand_count += a;
        cnt = ps[a].cnt, carried_best = st_all[ps[a].pair].maxGood;
    }
    __shared__ int wave_s[16];
    __shared__ int ev_pos[kMaxScanEvents], ev_val[kMaxScanEvents], ev_n;  // ev_cap <= kMaxScanEvents entries are used (tests shrink it)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) {
        *cand_count = 0;  // single block: reset here instead of a memset launch (ordered by the barriers below)
        ev_n = 0;
    }
    const int R = (cnt + 1023) / 1024;
    const int w0 = wave * R * 64;
    auto hyp_max = [&](int i) {
        if (i >= cnt) return 0;
        if (hmax_in) return hmax_in[i];
        int h = 0;  // few hypotheses: no separate hyp_max pass
        const int nm0 = n_models[i];
        for (int m = 0; m < nm0; ++m) h = max(h, good[(size_t)i * 10 + m]);
        return h;
    };
    int hv[kScanRows];
#pragma unroll
    for (int r = 0; r < kScanRows; ++r) hv[r] = (r < R) ? hyp_max(w0 + r * 64 + lane) : 0;
    int local = 0;
#pragma unroll
    for (int r = 0; r < kScanRows; ++r) local = max(local, hv[r]);
    int carry = waves_exclusive_scan_16<true>(wave_reduce<true>(local), carried_best, wave_s);  // best count before this wave's range
#pragma unroll
    for (int r = 0; r < kScanRows; ++r) {
        if (r < R) {  // wave-uniform (no break: the loop has to unroll fully for hv[] to stay in registers)
        const int i = w0 + r * 64 + lane;
        const int hmax = hv[r];
        const int inc = wave_inclusive_scan<true>(hmax, lane);
        const int up = __shfl_up(inc, 1);
        const int before = lane ? max(carry, up) : carry;  // best count of everything before hypothesis i
        if (i < cnt && hmax >= 5 && hmax >= before) {
            // a hypothesis that holds or ties the running best: a handful per pass.  They are only noted here and expanded below by
            // one thread each -- inline, their dependent loads ran one after the other inside the wave that met most of them
            const int e = atomicAdd(&ev_n, 1);
            if (e < ev_cap) {
                ev_pos[e] = i;
                ev_val[e] = hmax;
            } else {  // list full (never seen: it takes that many successive ties/records): expand in place
                const int nm = n_models[i];
                for (int m = 0; m < nm; ++m)
                    if (good[(size_t)i * 10 + m] == hmax) cand[atomicAdd(cand_count, 1)] = i * 10 + m;
            }
        }
        carry = max(carry, __shfl(inc, 63));
        }
    }
    __syncthreads();
    if (tid < min(ev_n, ev_cap)) {
        const int i = ev_pos[tid], hmax = ev_val[tid];
        const int nm = n_models[i];
        for (int m = 0; m < nm; ++m)
            if (good[(size_t)i * 10 + m] == hmax) cand[atomicAdd(cand_count, 1)] = i * 10 + m;
    }
}

__global__ void inlier_mask_kernel(const double4 *__restrict__ pts, int n, const double *__restrict__ E, double thresh2,
                                   uint8_t *__restrict__ mask, const PairSlot *__restrict__ ps = nullptr,
                                   const ReplayState *__restrict__ st_all = nullptr, int mask_stride = 0) {
    if (ps) {  // batch: blockIdx.y = slot
        const PairSlot P = ps[blockIdx.y];
        pts = P.pts, n = P.n, E = st_all[P.pair].E, mask += (size_t)P.pair * mask_stride;
    }
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double e[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) e[k] = E[k];
    const double4 p = pts[i];
    const float err = sampson_err_f32(e, p.x, p.y, p.z, p.w);
    mask[i] = ((double)err <= thresh2) ? 1 : 0;
}

// ---------------------------------------------------------------------------------------------------------------
// LMedS (CvModelEstimator3::runLMeDS, modelest.cpp:483-564): per-model median of the Sampson errors by an exact
// 4-pass byte radix select over the float bit patterns (the reference sorts them as ints, :541), one block per model.
// The errors are recomputed in every pass (55 fp64 instructions) instead of being stored: a model's error vector would
// be 4n bytes of HBM round trip per pass, the recompute is free next to it.
// ---------------------------------------------------------------------------------------------------------------
struct LmedsState {
    double minMedian;
    long long best;  // iteration*10 + slot, -1 = none
    double E[9];
    int32_t inliers;
    int32_t pad;
};

__device__ __forceinline__ uint32_t lmeds_key(float err) { return __float_as_uint(err) ^ 0x80000000u; }  // signed-int order

// CACHE: the model's n error keys are computed once into (dynamic) LDS and the five passes read them from there; without it
// (n above kScoreBlockMaxN) every pass recomputes the errors.
template <bool CACHE>
__global__ __launch_bounds__(256) void median_kernel(const double4 *__restrict__ pts, int n, const double *__restrict__ dense_E,
                                                     const int32_t *__restrict__ ids, const int32_t *__restrict__ total_ptr,
                                                     int total_host, double *__restrict__ median_out) {
    extern __shared__ uint32_t keys[];  // n keys when CACHE
    __shared__ uint32_t hist[256];
    __shared__ uint32_t s_bin, s_below;
    __shared__ uint32_t s_cnt_lt, s_max_lt;
    const int m = blockIdx.x;
    if (m >= (total_ptr ? *total_ptr : total_host)) return;  // block-uniform
    const int tid = threadIdx.x;
    double e[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) e[k] = dense_E[(size_t)m * 9 + k];
    uint32_t k = (uint32_t)(n / 2);  // upper middle (the median itself for odd n)
    uint32_t prefix = 0, care = 0;
    auto key_of = [&](int i) -> uint32_t {
        if constexpr (CACHE) return keys[i];
        const double4 p = pts[i];
        return lmeds_key(sampson_err_f32(e, p.x, p.y, p.z, p.w));
    };
    if constexpr (CACHE) {
        for (int i = tid; i < n; i += 256) {
            const double4 p = pts[i];
            keys[i] = lmeds_key(sampson_err_f32(e, p.x, p.y, p.z, p.w));
        }
    }
    for (int pass = 3; pass >= 0; --pass) {
        hist[tid] = 0;
        __syncthreads();
        const int sh = pass * 8;
        for (int i = tid; i < n; i += 256) {
            const uint32_t v = key_of(i);
            if ((v & care) == prefix) atomicAdd(&hist[(v >> sh) & 255u], 1u);
        }
        __syncthreads();
        if (tid < 64) {
            // lane l owns bins 4l..4l+3; wave-wide inclusive scan of the lane sums, then the crossing bin
            const uint32_t h0 = hist[4 * tid], h1 = hist[4 * tid + 1], h2 = hist[4 * tid + 2], h3 = hist[4 * tid + 3];
            const uint32_t mine = h0 + h1 + h2 + h3;
            uint32_t incl = mine;
#pragma unroll
            for (int d = 1; d < 64; d <<= 1) {
                const uint32_t o = __shfl_up(incl, d);
                if (tid >= d) incl += o;
            }
            const uint32_t excl = incl - mine;
            if (excl <= k && k < incl) {  // exactly one lane
                uint32_t c = excl, b = 4 * tid;
                if (k >= c + h0) {
                    c += h0;
                    ++b;
                    if (k >= c + h1) {
                        c += h1;
                        ++b;
                        if (k >= c + h2) {
                            c += h2;
                            ++b;
                        }
                    }
                }
                s_bin = b;
                s_below = c;
            }
        }
        __syncthreads();
        prefix |= s_bin << sh;
        care |= 255u << sh;
        k -= s_below;
        __syncthreads();
    }
    const uint32_t key_hi = prefix;
    float f_hi = __uint_as_float(key_hi ^ 0x80000000u);
    double med;
    if (n & 1) {
        med = (double)f_hi;
    } else {
        // lower middle: the largest key below key_hi if exactly n/2 keys are below it, else key_hi again (duplicates)
        if (tid == 0) {
            s_cnt_lt = 0;
            s_max_lt = 0;
        }
        __syncthreads();
        uint32_t c = 0, mx = 0;
        for (int i = tid; i < n; i += 256) {
            const uint32_t v = key_of(i);
            if (v < key_hi) {
                ++c;
                mx = max(mx, v);
            }
        }
        atomicAdd(&s_cnt_lt, c);
        atomicMax(&s_max_lt, mx);
        __syncthreads();
        const float f_lo = (s_cnt_lt == (uint32_t)(n / 2)) ? __uint_as_float(s_max_lt ^ 0x80000000u) : f_hi;
        med = __dmul_rn((double)__fadd_rn(f_lo, f_hi), 0.5);  // modelest.cpp:544: float sum, then * 0.5 in double
    }
    if (tid == 0) median_out[ids ? ids[m] : m] = med;
}

// `if (median < minMedian)` in iteration then slot order (modelest.cpp:546-551) == the first minimum under (median, id).
__global__ __launch_bounds__(1024) void lmeds_argmin_kernel(const int32_t *__restrict__ n_models, const double *__restrict__ medians,
                                                            const double *__restrict__ E_tab, int cnt, LmedsState *__restrict__ st) {
    __shared__ double s_med[1024];
    __shared__ int s_id[1024];
    const int tid = threadIdx.x;
    double bm = DBL_MAX;
    int bi = -1;
    for (int id = tid; id < cnt * 10; id += 1024) {
        if ((id % 10) >= n_models[id / 10]) continue;
        const double v = medians[id];
        if (v < bm) {  // ids rise within a thread: ties keep the earlier one
            bm = v;
            bi = id;
        }
    }
    s_med[tid] = bm;
    s_id[tid] = bi;
    __syncthreads();
    for (int w = 512; w > 0; w >>= 1) {
        if (tid < w) {
            const double om = s_med[tid + w];
            const int oi = s_id[tid + w];
            if (oi >= 0 && (s_id[tid] < 0 || om < s_med[tid] || (om == s_med[tid] && oi < s_id[tid]))) {
                s_med[tid] = om;
                s_id[tid] = oi;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        st->minMedian = s_med[0];
        st->best = s_id[0];
        st->inliers = 0;
        if (s_id[0] >= 0)
            for (int k = 0; k < 9; ++k) st->E[k] = E_tab[(size_t)s_id[0] * 9 + k];
    }
}

__global__ void inlier_mask_count_kernel(const double4 *__restrict__ pts, int n, const double *__restrict__ E, double thresh2,
                                         uint8_t *__restrict__ mask, int32_t *__restrict__ count) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    bool in = false;
    if (i < n) {
        double e[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) e[k] = E[k];
        const double4 p = pts[i];
        in = (double)sampson_err_f32(e, p.x, p.y, p.z, p.w) <= thresh2;
        mask[i] = in ? 1 : 0;
    }
    const unsigned long long b = __ballot(in);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(count, __popcll(b));
}

// ---------------------------------------------------------------------------------------------------------------
// Refit on all inliers (the reference's `lesqu`, modelest.cpp:420-464): Gram matrix of the masked epipolar rows,
// its 4 smallest eigenvectors (cyclic Jacobi, one wave), then the same solver tail.
// ---------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256) void gram_kernel(const double4 *__restrict__ pts, const uint8_t *__restrict__ mask, int n,
                                                   double *__restrict__ gram_part /* [gridDim.x][45] */) {
    __shared__ double red[4][45];
    double acc[45];
#pragma unroll
    for (int k = 0; k < 45; ++k) acc[k] = 0;
    for (int i = blockIdx.x * blockDim.x + threadIdx.x; i < n; i += gridDim.x * blockDim.x) {
        if (!mask[i]) continue;
        const double4 p = pts[i];
        const double q[9] = {p.x * p.z, p.y * p.z, p.z, p.x * p.w, p.y * p.w, p.w, p.x, p.y, 1.0};
        int t = 0;
#pragma unroll
        for (int a = 0; a < 9; ++a)
#pragma unroll
            for (int b = a; b < 9; ++b) acc[t++] += q[a] * q[b];
    }
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int k = 0; k < 45; ++k) {
        double v = acc[k];
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < 45)
        gram_part[(size_t)blockIdx.x * 45 + threadIdx.x] =
            red[0][threadIdx.x] + red[1][threadIdx.x] + red[2][threadIdx.x] + red[3][threadIdx.x];
}

constexpr double kJacobiTol = 1e-32;  // on the squared off-diagonal mass relative to the squared diagonal (1e-28 is no faster: measured)

struct Jacobi9Lds {
    double G[9][9], Gn[9][9], Vv[9][9], Vn[9][9];  // (G, Gn) and (Vv, Vn) adjacent: the iteration indexes them as [2][9][9]
    int partner[10], ring[10];
    double cself[10], cpart[10];
};

// Eigen-decomposition of the symmetric 9x9 J.G by one wave: on return J.G is diagonal (the eigenvalues, unordered) and the columns
// of J.Vv are the eigenvectors.  J.Vv must hold the identity on entry.
__device__ __forceinline__ int jacobi9_wave(Jacobi9Lds &J, int lane) {  // returns the sweeps taken
    // Jacobi eigenvalue iteration on the symmetric 9x9 in the parallel (round-robin) ordering: the 9 indices plus one idle
    // slot form 5 disjoint pairs per round, 9 rounds visit all 36 pairs once (= one sweep).  Lanes 0..4 compute the rotations
    // of a round, then all lanes apply J^T G J and V J element-wise (disjoint rotations commute).
    // The matrices ping-pong between (G, Vv) and (Gn, Vn): a round reads one pair and writes the other.  Slot k of the ring holds, in
    // round r, index 1 + (k - 1 - r) mod 9 (slot 0 keeps index 0; index 9 is the idle slot).
    double(*GG)[9][9] = reinterpret_cast<double(*)[9][9]>(&J.G[0][0]);    // GG[0] = G, GG[1] = Gn
    double(*VV)[9][9] = reinterpret_cast<double(*)[9][9]>(&J.Vv[0][0]);   // VV[0] = Vv, VV[1] = Vn
    int cur = 0, sweeps = 0;
    // the (at most two) matrix elements this lane owns
    const int e0 = lane, e1 = lane + 64;
    const int a0 = e0 / 9, b0 = e0 - a0 * 9, a1 = e1 / 9, b1 = e1 - a1 * 9;
    const bool has1 = e1 < 81;
#ifndef MLPL_JACOBI_MAX_SWEEPS
#define MLPL_JACOBI_MAX_SWEEPS 60
#endif
    for (int sweep = 0; sweep < MLPL_JACOBI_MAX_SWEEPS; ++sweep) {
        // off-diagonal mass against the diagonal (wave reduction; every lane gets the totals)
        double off = 0, diag = 0;
        {
            const double v0 = GG[cur][a0][b0];
            if (a0 == b0) diag += v0 * v0;
            else if (a0 < b0) off += v0 * v0;
            if (has1) {
                const double v1 = GG[cur][a1][b1];
                if (a1 == b1) diag += v1 * v1;
                else if (a1 < b1) off += v1 * v1;
            }
        }
#pragma unroll
        for (int d = 32; d > 0; d >>= 1) {
            off += __shfl_xor(off, d);
            diag += __shfl_xor(diag, d);
        }
        if (off <= kJacobiTol * diag) break;  // wave-uniform
        ++sweeps;
        for (int round = 0; round < 9; ++round) {
            const double(*Gc)[9] = GG[cur];
            const double(*Vc)[9] = VV[cur];
            double(*Gx)[9] = GG[cur ^ 1];
            double(*Vx)[9] = VV[cur ^ 1];
            if (lane < 5) {
                const int p0 = lane == 0 ? 0 : 1 + (lane - 1 - round + 9) % 9, q0 = 1 + (9 - lane - 1 - round + 9) % 9;
                const int p = min(p0, q0), q = max(p0, q0);
                double c = 1.0, sn = 0.0;
                if (q < 9) {
                    const double apq = Gc[p][q];
                    if (apq != 0.0) {
                        // The rotation is orthogonal to rounding for ANY tangent tt as long as c = (1 + tt^2)^-1/2 is accurate and s = tt c;
                        // an inexact tt only leaves a'_pq at ~1e-8 a_pq instead of 0, which the next sweep removes.  So the tangent comes
                        // from the hardware approximations (v_rcp_f64 / v_sqrt_f64, ~2^-27) and only c gets a Newton step: two IEEE
                        // divisions and two IEEE square roots (~500 dependent cycles per round) become ~150.
                        const double theta = (Gc[q][q] - Gc[p][p]) * __builtin_amdgcn_rcp(2.0 * apq);
                        const double at = fabs(theta);
                        double tt = __builtin_amdgcn_rcp(at + __builtin_amdgcn_sqrt(at * at + 1.0));
                        tt = theta >= 0 ? tt : -tt;
                        if (!(at < 1e150)) tt = 0.5 / theta;  // theta^2 overflows: t = 1 / (2 theta) to rounding
                        const double x = tt * tt + 1.0;
                        const double y0 = __builtin_amdgcn_rsq(x);
                        c = y0 * (1.5 - 0.5 * x * y0 * y0);
                        c = c * (1.5 - 0.5 * x * c * c);
                        sn = tt * c;
                    }
                }
                // column p' = c col_p - s col_q ; column q' = s col_p + c col_q
                J.partner[p] = q;
                J.cself[p] = c;
                J.cpart[p] = -sn;
                J.partner[q] = p;
                J.cself[q] = c;
                J.cpart[q] = sn;
            }
            wave_sync();
#pragma unroll
            for (int which = 0; which < 2; ++which) {
                if (which == 1 && !has1) break;
                const int a = which ? a1 : a0, b = which ? b1 : b0;
                const int pa = J.partner[a], pb = J.partner[b];
                const double ca = J.cself[a], ka = J.cpart[a], cb = J.cself[b], kb = J.cpart[b];
                // an index paired with the idle slot keeps c = 1, k = 0; its partner index (9) is never read with weight
                const double gab = Gc[a][b];
                const double gapb = (pb < 9) ? Gc[a][pb] : 0.0;
                const double gpab = (pa < 9) ? Gc[pa][b] : 0.0;
                const double gpapb = (pa < 9 && pb < 9) ? Gc[pa][pb] : 0.0;
                Gx[a][b] = ca * cb * gab + ca * kb * gapb + ka * cb * gpab + ka * kb * gpapb;
                const double vab = Vc[a][b];
                const double vapb = (pb < 9) ? Vc[a][pb] : 0.0;
                Vx[a][b] = cb * vab + kb * vapb;
            }
            wave_sync();
            cur ^= 1;
        }
    }
    if (cur) {  // an odd number of rounds: the result sits in the second pair
        J.G[a0][b0] = J.Gn[a0][b0];
        J.Vv[a0][b0] = J.Vn[a0][b0];
        if (has1) {
            J.G[a1][b1] = J.Gn[a1][b1];
            J.Vv[a1][b1] = J.Vn[a1][b1];
        }
        wave_sync();
    }
    return sweeps;
}

// ---- the eigenvector of the SMALLEST eigenvalue of a symmetric positive semi-definite 9 x 9, by inverse iteration ------------------------
// The re-weighted 8-point fits (USAC's REF_WEIGHTS refits, robustEssentialRefine's rounds) need one eigenvector, not nine, and come in
// chains whose previous member is an excellent start.  With A = G + delta I (delta = 2^-44 trace G: A is safely positive definite, same
// eigenvectors) a Cholesky factor and a pair of triangular solves per step multiply the component along the smallest eigenvector by
// (lambda_2 + delta) / (lambda_1 + delta) -- typically 10^2 ... 10^6 -- against all others: 3-6 steps from a warm start, ~0.4 us each,
// where the Jacobi sweeps of jacobi9_wave take 25-40 us per decomposition.  Layout: lane i < 9 holds row i of A, of L and column i of L
// in registers; a value crosses lanes by v_readlane (no LDS, no barrier); every loop is unrolled, every index static.
// Returns the steps taken, or 0 when it has not converged in kInvIterMax steps (lambda_2 close to lambda_1: the caller falls back to
// jacobi9_wave, as it does when the factorisation meets a non-positive pivot) -- so near-degenerate systems keep their old path.
// Gp: the upper triangle of G, packed by rows (45 values, LDS or global), each divided by `divisor`; start: 9 values or null (cold start).
// x_out (9 values) and *lambda_out (x^T G x) are written by lanes 0..8 / lane 0; the caller orders them with wave_sync().
constexpr int kInvIterMax = 40;
__device__ __forceinline__ double bcast_f64(double v, int src) {
    const int lo = __builtin_amdgcn_readlane(__double2loint(v), src), hi = __builtin_amdgcn_readlane(__double2hiint(v), src);
    return __hiloint2double(hi, lo);
}
__device__ __forceinline__ int smallest_eigvec9_wave(const double *Gp, double divisor, const double *start, double *x_out, double *lambda_out, int lane) {
    const int i = lane < 9 ? lane : 8;  // (the other lanes shadow lane 8: uniform control flow, nothing stored)
    double g[9], a[9], l[9], c[9];
    double trace = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const int lo = i < k ? i : k, hi = i < k ? k : i;
        g[k] = Gp[lo * 9 - lo * (lo - 1) / 2 + (hi - lo)] / divisor;
        l[k] = 0, c[k] = 0;
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        double d = 0;
#pragma unroll
        for (int m = 0; m < 9; ++m) d = (i == m) ? g[m] : d;  // own diagonal entry
        trace += bcast_f64(d, k);
    }
    if (!(trace > 0) || !(trace < 1e300)) return 0;
    const double delta = trace * 0x1p-44;
#pragma unroll
    for (int k = 0; k < 9; ++k) a[k] = g[k] + ((i == k) ? delta : 0.0);
    // Cholesky, column by column
    double inv_d = 0;
#pragma unroll
    for (int j = 0; j < 9; ++j) {
        double sdiag = a[j];
#pragma unroll
        for (int k = 0; k < j; ++k) sdiag -= l[k] * l[k];
        const double sj = bcast_f64(sdiag, j);
        if (!(sj > 0)) return 0;  // wave-uniform
        const double dj = sqrt(sj), inv = 1.0 / dj;
        double t = a[j];
#pragma unroll
        for (int k = 0; k < j; ++k) t -= l[k] * bcast_f64(l[k], j);
        const double lij = t * inv;
        l[j] = (i > j) ? lij : ((i == j) ? dj : 0.0);
        inv_d = (i == j) ? inv : inv_d;
#pragma unroll
        for (int m = j + 1; m < 9; ++m) {
            const double v = bcast_f64(lij, m);
            c[m] = (i == j) ? v : c[m];
        }
    }
    double x = start ? start[i] : (1.0 / 3.0);
    {   // (a start that is not a unit vector: normalise; a zero / non-finite one: cold start)
        double n2 = 0;
#pragma unroll
        for (int m = 0; m < 9; ++m) n2 += bcast_f64(x * x, m);
        x = (n2 > 0x1p-200 && n2 < 0x1p200) ? x / sqrt(n2) : (1.0 / 3.0);
    }
    double d_prev = 0;
    int it = 0;
    bool done = false;
    for (; it < kInvIterMax && !done;) {
        ++it;
        double acc = 0, y = 0, z = 0;
#pragma unroll
        for (int j = 0; j < 9; ++j) {  // L y = x
            const double yj = bcast_f64((x - acc) * inv_d, j);
            y = (i == j) ? yj : y;
            acc += (i > j) ? l[j] * yj : 0.0;
        }
        acc = 0;
#pragma unroll
        for (int j = 8; j >= 0; --j) {  // L^T z = y
            const double zj = bcast_f64((y - acc) * inv_d, j);
            z = (i == j) ? zj : z;
            acc += (i < j) ? c[j] * zj : 0.0;
        }
        double n2 = 0;
#pragma unroll
        for (int m = 0; m < 9; ++m) n2 += bcast_f64(z * z, m);
        if (!(n2 > 0) || !(n2 < 1e300)) return 0;
        const double xn = z / sqrt(n2);
        double d = 0;
#pragma unroll
        for (int m = 0; m < 9; ++m) d = fmax(d, bcast_f64(fabs(xn - x), m));
        x = xn;
        // geometric convergence with ratio q = d / d_prev: what is still to come is ~ d q / (1 - q)
        if (d <= 0x1p-51)
            done = true;  // at rounding level
        else if (it > 1 && d < d_prev) {
            const double q = d / d_prev;
            done = d * q / (1.0 - q) <= 0x1p-52;
        }
        d_prev = d;
    }
    if (!done) return 0;
    // Rayleigh quotient with G itself
    double gx = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) gx += g[k] * bcast_f64(x, k);
    double lam = 0;
#pragma unroll
    for (int m = 0; m < 9; ++m) lam += bcast_f64(x * gx, m);
    // Settling says that the iterates stopped moving, not where: a start without a component along the smallest eigenvector --
    // a cold start against a vector whose entries sum to zero, a warm start after a large change of the weights -- can sit on the
    // second-smallest one until rounding brings the component back.  Two certificates, or the caller's Jacobi path:
    // (i) (lam, x) is an eigenpair of G to rounding: |G x - lam x| <= 2^-40 trace;
    // (ii) lam is THE smallest eigenvalue and separated: G - (lam + 2^-40 trace) I has exactly one negative pivot in its L D L^T
    //     factorisation (Sylvester's law of inertia; a zero or non-finite pivot refuses).
    const double mu = trace * 0x1p-40;
    double rmax = 0;
#pragma unroll
    for (int m = 0; m < 9; ++m) rmax = fmax(rmax, bcast_f64(fabs(gx - lam * x), m));
    if (!(rmax <= mu)) return 0;
    {
        double b[9], ll[9], dk[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) b[k] = g[k] - ((i == k) ? lam + mu : 0.0), ll[k] = 0, dk[k] = 0;
        int neg = 0;
#pragma unroll
        for (int j = 0; j < 9; ++j) {
            double sdiag = b[j];
#pragma unroll
            for (int k = 0; k < j; ++k) sdiag -= ll[k] * ll[k] * dk[k];
            const double dj = bcast_f64(sdiag, j);
            if (!(dj != 0.0) || !(fabs(dj) < 1e300)) return 0;  // wave-uniform
            neg += dj < 0 ? 1 : 0;
            dk[j] = dj;
            double t = b[j];
#pragma unroll
            for (int k = 0; k < j; ++k) t -= ll[k] * bcast_f64(ll[k], j) * dk[k];
            ll[j] = (i > j) ? t / dj : 0.0;
        }
        if (neg != 1) return 0;
    }
    if (lane < 9) x_out[lane] = x;
    if (lane == 0) *lambda_out = lam;
    return it;
}

// tests: smallest_eigvec9_wave and jacobi9_wave on a batch of symmetric 9 x 9 matrices, one wave per matrix.  G: [count][81] row-major;
// start: [count][9] or null; out: [count][12] = {steps (0 = not settled), lambda, 0, x[9]}; jac: [count][10] = {smallest eigenvalue, its vector}
__global__ __launch_bounds__(64) void eig9_debug_kernel(const double *__restrict__ G, const double *__restrict__ start, int count, double *__restrict__ out,
                                                        double *__restrict__ jac) {
    __shared__ double gp[45], x[9], lam, st[9];
    __shared__ Jacobi9Lds J;
    const int b = blockIdx.x, lane = threadIdx.x;
    if (b >= count) return;
    if (lane < 45) {
        int a = 0, rem = lane;
        while (rem >= 9 - a) rem -= 9 - a, ++a;
        gp[lane] = G[(size_t)b * 81 + a * 9 + (a + rem)];
    }
    if (lane < 9 && start) st[lane] = start[(size_t)b * 9 + lane];
    for (int e = lane; e < 81; e += 64) {
        J.G[e / 9][e % 9] = G[(size_t)b * 81 + e];
        J.Vv[e / 9][e % 9] = (e / 9 == e % 9) ? 1.0 : 0.0;
    }
    if (lane == 0) lam = 0;
    wave_sync();
    const int steps = smallest_eigvec9_wave(gp, 1.0, start ? st : nullptr, x, &lam, lane);
    wave_sync();
    if (lane == 0) out[(size_t)b * 12] = steps, out[(size_t)b * 12 + 1] = lam, out[(size_t)b * 12 + 2] = 0;
    if (lane < 9) out[(size_t)b * 12 + 3 + lane] = steps > 0 ? x[lane] : 0.0;
    jacobi9_wave(J, lane);
    if (lane == 0) {
        int m = 0;
        for (int a = 1; a < 9; ++a)
            if (J.G[a][a] < J.G[m][m]) m = a;
        jac[(size_t)b * 10] = J.G[m][m];
        for (int k = 0; k < 9; ++k) jac[(size_t)b * 10 + 1 + k] = J.Vv[k][m];
    }
}

// Indices of the nine eigenvalues in descending order, the first of equal values first (what a selection sort from the top gives).  The
// diagonal is read once (nine loads in flight) and every index gets its position by counting in registers; as a selection sort over LDS
// the 36 comparisons were 72 dependent round trips (~4 us on one lane).
__device__ __forceinline__ void order_desc9(const Jacobi9Lds &J, int *order) {
    double d[9];
#pragma unroll
    for (int a = 0; a < 9; ++a) {
        d[a] = J.G[a][a];
        order[a] = a;  // (NaNs leave positions unassigned: keep every entry a valid index)
    }
#pragma unroll
    for (int a = 0; a < 9; ++a) {
        int pos = 0;
#pragma unroll
        for (int b = 0; b < 9; ++b) pos += ((d[b] > d[a]) || (d[b] == d[a] && b < a)) ? 1 : 0;
        order[pos] = a;
    }
}

// One workgroup (one wave) per system: blockIdx.x selects the system's parts (part_stride doubles apart) and its record.  `gate` (optional):
// an int per system, gate_stride bytes apart -- a system whose gate is 0 is not solved and its record says "no models" (USAC's local
// optimisation: chains whose step has no fit, usac_impl.h).  `warm` (optional): 82 doubles per system, warm_stride bytes apart; [0] != 0
// says that [1..81] hold the eigenvectors V_prev of the system's previous fit: consecutive fits of a chain see nearly the same Gram
// matrix, so the iteration starts from V_prev^T G V_prev (nearly diagonal) and accumulates onto V_prev -- the same decomposition to the
// same tolerance in 2-4 sweeps instead of 7-8 (a basis vector may come out with the other sign; the solutions do not depend on it).  The
// eigenvectors of this fit are left there for the next.
__device__ __forceinline__ void refit_solve_body(const double *__restrict__ gram_part, int nparts, PolyRec *__restrict__ rec, size_t part_stride,
                                                 const char *__restrict__ gate, size_t gate_stride, const int vbx, char *warm_base = nullptr,
                                                 size_t warm_stride = 0) {
    __shared__ SolveLds L;
    __shared__ Jacobi9Lds J;
    __shared__ double gsum[45];
    if (blockDim.x != kSolverThreads) __builtin_trap();  // wave_sync() is a one-wave ordering
    const int lane = threadIdx.x;
    gram_part += (size_t)vbx * part_stride;
    rec += vbx;
    if (gate && *reinterpret_cast<const int32_t *>(gate + (size_t)vbx * gate_stride) == 0) {
        if (lane == 0) rec->ok = 0.0;
        return;
    }
    double *warm = warm_base ? reinterpret_cast<double *>(warm_base + (size_t)vbx * warm_stride) : nullptr;
    const bool is_warm = warm && warm[0] != 0.0;  // wave-uniform
    if (lane < 45) {  // fixed summation order over the blocks: run-to-run deterministic
        double sacc = 0;
        for (int pblk = 0; pblk < nparts; ++pblk) sacc += gram_part[(size_t)pblk * 45 + lane];
        gsum[lane] = sacc;
    }
    if (is_warm)
        for (int e = lane; e < 81; e += 64) J.Vv[e / 9][e % 9] = warm[1 + e];
    wave_sync();
    if (!is_warm) {
        if (lane == 0) {
            int t = 0;
            for (int a = 0; a < 9; ++a)
                for (int b = a; b < 9; ++b) {
                    J.G[a][b] = gsum[t];
                    J.G[b][a] = gsum[t];
                    ++t;
                }
        }
        for (int e = lane; e < 81; e += 64) J.Vv[e / 9][e % 9] = (e / 9 == e % 9) ? 1.0 : 0.0;
    } else {
        // Gn = G V_prev (G read from the packed upper triangle), then G = V_prev^T Gn, upper triangle mirrored
        for (int e = lane; e < 81; e += 64) {
            const int a = e / 9, b = e - a * 9;
            double acc = 0;
            for (int k = 0; k < 9; ++k) {
                const int lo = a < k ? a : k, hi = a < k ? k : a;
                acc += gsum[lo * 9 - lo * (lo - 1) / 2 + (hi - lo)] * J.Vv[k][b];
            }
            J.Gn[a][b] = acc;
        }
        wave_sync();
        for (int e = lane; e < 81; e += 64) {
            const int a = e / 9, b = e - a * 9;
            if (a > b) continue;
            double acc = 0;
            for (int k = 0; k < 9; ++k) acc += J.Vv[k][a] * J.Gn[k][b];
            J.G[a][b] = acc;
            J.G[b][a] = acc;
        }
    }
    wave_sync();
    jacobi9_wave(J, lane);
    if (warm) {
        for (int e = lane; e < 81; e += 64) warm[1 + e] = J.Vv[e / 9][e % 9];
        if (lane == 0) warm[0] = 1.0;
    }
    if (lane == 0) {
        // order eigenvalues descending; EE = eigenvectors of the 4 smallest, in descending order (five-point.cpp:388)
        int order[9];
        order_desc9(J, order);
        for (int j = 0; j < 4; ++j)
            for (int r = 0; r < 9; ++r) L.EE[j][r] = J.Vv[r][order[5 + j]];
    }
    wave_sync();
    solve_from_basis(L, lane, rec);
}
__global__ __launch_bounds__(64) void refit_solve_kernel(const double *__restrict__ gram_part, int nparts,
                                                         PolyRec *__restrict__ rec, size_t part_stride = 0,
                                                         const char *__restrict__ gate = nullptr, size_t gate_stride = 0) {
    refit_solve_body(gram_part, nparts, rec, part_stride, gate, gate_stride, blockIdx.x);
}

// ---------------------------------------------------------------------------------------------------------------
// Device-side replay of CvModelEstimator3::runRANSAC's sequential update rule (modelest.cpp:377-416) over the score
// tables of one chunk of hypotheses.  The rule is a running arg-max under the total order (good desc, error sum asc,
// earlier first) restricted to good >= 5, and `niters` after an iteration is min(niters, T[best good so far]) where
// T[g] = cvRANSACUpdateNumIters1(confidence, (n-g)/n, 5, +inf) is tabulated on the HOST (glibc log/pow, so the values are
// the CPU path's).  So the loop's stopping point and winner are a prefix-max scan, a find-first and an arg-max reduction.
// ---------------------------------------------------------------------------------------------------------------

// cvRANSACUpdateNumIters1 (modelest.cpp:86-109) with max_iters = "infinity", evaluated on the device.  The host recomputes
// every value the replay used with glibc and falls back to a host-built table if one differs (device log/pow are not glibc's,
// so a quotient within an ulp-scale distance of x.5 could round the other way; that has probability ~1e-8 per value).
// `log_num` = log(max(1 - p, DBL_MIN)) comes from the host (glibc's value, constant for the call); (1 - ep)^5 is four multiplications
// here, not pow(): the device's pow() walks lookup tables in memory and a record-breaking count waited microseconds for it.  The
// host's re-evaluation with the reference formula (update_num_iters) decides whether a device value is accepted.
__device__ __forceinline__ int dev_num_iters(double log_num, int n, int g) {
    double ep = (double)(n - g) / n;
    ep = fmin(fmax(ep, 0.), 1.);
    const double x = 1. - ep, x2 = x * x;
    double denom = 1. - x2 * x2 * x;
    if (denom < DBL_MIN) return 0;
    denom = log(denom);
    return (denom >= 0 || -log_num >= (double)INT32_MAX * (-denom)) ? INT32_MAX : (int)round(log_num / denom);
}

// Per-hypothesis arg-max under (good desc, error sum asc, slot asc): the only model of a hypothesis that can ever be taken.
__global__ void hyp_best_kernel(const int32_t *__restrict__ n_models, const int32_t *__restrict__ good,
                                const double *__restrict__ esum, int cnt, int32_t *__restrict__ hgood, double *__restrict__ hsum,
                                int32_t *__restrict__ hslot) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    const int nm = n_models[i];
    int bg = 0, bs = -1;
    double be = 0;
    for (int m = 0; m < nm; ++m) {
        const int g = good[(size_t)i * 10 + m];
        const double e = esum[(size_t)i * 10 + m];
        if (bs < 0 || g > bg || (g == bg && e < be)) {
            bg = g;
            be = e;
            bs = m;
        }
    }
    hgood[i] = bg;
    hsum[i] = be;
    hslot[i] = bs;
}

__global__ __launch_bounds__(1024) void replay_kernel(const int32_t *__restrict__ hgood, const double *__restrict__ hsum,
                                                      const int32_t *__restrict__ hslot, const double *__restrict__ E_tab, int cnt,
                                                      const int32_t *__restrict__ Ttab, int npts, long long base_index,
                                                      const int32_t *__restrict__ dense_total, ReplayState *__restrict__ st,
                                                      double log_num, int ev_cap, const PairSlot *__restrict__ ps = nullptr,
                                                      int slot_stride = 0) {
    if (ps) {  // batched pass: blockIdx.x = slot
        const int a = blockIdx.x;
        hgood += (size_t)a * slot_stride, hsum += (size_t)a * slot_stride, hslot += (size_t)a * slot_stride;
        E_tab += (size_t)a * slot_stride * 90, dense_total += a;
        cnt = ps[a].cnt, npts = ps[a].n, base_index = (long long)ps[a].iter_base * 10, st += ps[a].pair;
    }
    __shared__ int wave_s[16];
    __shared__ int stop_idx;
    __shared__ int ev_pos[kMaxScanEvents], ev_val[kMaxScanEvents], ev_n;
    __shared__ int s_best_good[1024];
    __shared__ double s_best_sum[1024];
    __shared__ int s_best_idx[1024];
    const int tid = threadIdx.x;
    const int maxGood0 = st->maxGood, niters0 = st->niters, iter0 = st->iter;
    if (tid == 0) {
        stop_idx = cnt;  // "no stop inside this chunk"
        ev_n = 0;
    }

    // pass 1: running best count (prefix max in iteration order), niters after each iteration, first iteration after which the loop
    // ends.  Wave-owned contiguous ranges, see above.
    const int lane = tid & 63, wave = tid >> 6;
    const int R = (cnt + 1023) / 1024;
    const int w0 = wave * R * 64;
    int hv[kScanRows];  // this thread's column of the pass: hypothesis w0 + r * 64 + lane in row r
#pragma unroll
    for (int r = 0; r < kScanRows; ++r) {
        const int i = w0 + r * 64 + lane;
        hv[r] = (r < R && i < cnt) ? hgood[i] : 0;
    }
    int local = 0;
#pragma unroll
    for (int r = 0; r < kScanRows; ++r) local = max(local, hv[r]);
    const int wpre = waves_exclusive_scan_16<true>(wave_reduce<true>(local), maxGood0, wave_s);  // running count before this wave's range
    // The bound after iteration i is min(niters0, T(running_i)).  T falls as the count rises and the running count never falls, so
    // T(running_i) = min over the record-breaking iterations j <= i of T(running_j): evaluate T only where the running count changes
    // (a handful of iterations per pass -- log/pow in fp64 for every lane doubled this kernel's time).  The count carried in from
    // earlier passes is already inside niters0.
    auto bound_at = [&](int running) {
        const int g = min(running, npts);
        return Ttab ? Ttab[g] : dev_num_iters(log_num, npts, g);
    };
    // Sweep 1 only NOTES the record-breaking iterations (position, count); their bounds are then evaluated by one thread each (the fp64
    // pow + log of a bound is a few microseconds -- inline, the records of the first wave's range ran one after the other).
    {
        int carry = wpre;
#pragma unroll
        for (int r = 0; r < kScanRows; ++r) {
            if (r < R) {  // wave-uniform
            const int i = w0 + r * 64 + lane;
            const int inc = wave_inclusive_scan<true>(hv[r], lane);
            const int running = max(carry, inc);
            const int up = __shfl_up(inc, 1);
            const int prev_running = lane ? max(carry, up) : carry;  // running count before this iteration
            if (i < cnt && running >= 5 && running > prev_running) {
                const int e = atomicAdd(&ev_n, 1);
                if (e < ev_cap) {
                    ev_pos[e] = i;
                    ev_val[e] = running;
                }
            }
            carry = max(carry, __shfl(inc, 63));
            }
        }
    }
    __syncthreads();
    const bool ev_overflow = ev_n > ev_cap;  // more records than the list holds (adversarial input): serial fallback below
    const int nev = ev_overflow ? 0 : ev_n;
    if (ev_overflow && tid == 0) {
        int running = maxGood0, Tmin = INT32_MAX;
        for (int i = 0; i < cnt; ++i) {
            const int h = hgood[i];
            if (h > running) {
                running = h;
                if (running >= 5) {
                    const int Tv = bound_at(running);
                    if (!Ttab) {
                        const int slot = atomicAdd(&st->t_count, 1);
                        if (slot < kTUsedMax) {
                            st->t_g[slot] = min(running, npts);
                            st->t_val[slot] = Tv;
                        }
                    }
                    Tmin = min(Tmin, Tv);
                }
            }
            if (iter0 + i + 1 >= min(niters0, Tmin)) {
                stop_idx = i;
                break;
            }
        }
    }
    if (tid < nev) {
        const int running = ev_val[tid];
        const int Tv = bound_at(running);
        if (!Ttab) {  // remember the bound used for this count: the host re-evaluates it with its libm
            const int slot = atomicAdd(&st->t_count, 1);
            if (slot < kTUsedMax) {
                st->t_g[slot] = min(running, npts);
                st->t_val[slot] = Tv;
            }
        }
        ev_val[tid] = Tv;  // the list now holds (position, bound)
    }
    __syncthreads();
    // bounds of the events inside / before this wave's range
    int local_T = INT32_MAX, before_T = INT32_MAX;
    const int w1 = w0 + R * 64;
    for (int e = lane; e < nev; e += 64) {
        const int p = ev_pos[e], Tv = ev_val[e];
        if (p < w0) before_T = min(before_T, Tv);
        else if (p < w1) local_T = min(local_T, Tv);
    }
    const int wave_T = wave_reduce<false>(local_T);
    const int tpre = wave_reduce<false>(before_T);  // min bound over the events before this wave's range
    // Can the loop end inside this wave's range at all?  The bound only falls, so the smallest bound of the range is the one after its
    // last event: no stop unless the last iteration of the range reaches it (wave-uniform; all but one wave skip the second sweep).
    if (!ev_overflow && w0 < cnt && iter0 + min(cnt, w1) >= min(niters0, min(tpre, wave_T))) {
        int carry_T = tpre;
        for (int r = 0; r < R; ++r) {
            const int i = w0 + r * 64 + lane;
            int Tv = INT32_MAX;  // the bound of an event AT this iteration, from the list
            for (int e = 0; e < nev; ++e)
                if (ev_pos[e] == i) Tv = ev_val[e];
            Tv = wave_inclusive_scan<false>(Tv, lane);
            const bool stop_here = i < cnt && iter0 + i + 1 >= min(niters0, min(carry_T, Tv));
            if (__ballot(stop_here)) {  // wave-uniform: the first stopping lane of the first such row
                if (stop_here) atomicMin(&stop_idx, i);
                break;
            }
            carry_T = min(carry_T, __shfl(Tv, 63));
        }
    }
    __syncthreads();
    const int processed = min(cnt, stop_idx + 1);

    // pass 2: arg-max over the processed iterations (good desc, sum asc, earlier first), only counts >= 5 qualify
    int bg = 0, bi = -1;
    double bs = 0;
#pragma unroll
    for (int half = 0; half < 2; ++half) {  // two batches of 16 rows: the register budget of a 1024-thread block is 128
        double ev[kScanRows / 2];
#pragma unroll
        for (int q = 0; q < kScanRows / 2; ++q) {  // the error sums of this thread's column, the loads of a batch in flight together
            const int r = half * (kScanRows / 2) + q, i = w0 + r * 64 + lane;
            ev[q] = (r < R && i < processed && hv[r] >= 5) ? hsum[i] : 0.0;
        }
#pragma unroll
        for (int q = 0; q < kScanRows / 2; ++q) {
            const int r = half * (kScanRows / 2) + q, i = w0 + r * 64 + lane, g = hv[r];
            const double e = ev[q];
            if (r < R && i < processed && g >= 5 && (bi < 0 || g > bg || (g == bg && e < bs))) {  // i increases with r: strict comparisons keep the earlier one
                bg = g;
                bs = e;
                bi = i;
            }
        }
    }
    s_best_good[tid] = bg;
    s_best_sum[tid] = bs;
    s_best_idx[tid] = bi;
    __syncthreads();
    for (int off = 512; off > 0; off >>= 1) {
        if (tid < off) {
            const int g2 = s_best_good[tid + off], i2 = s_best_idx[tid + off];
            const double e2 = s_best_sum[tid + off];
            const int g1 = s_best_good[tid], i1b = s_best_idx[tid];
            const double e1 = s_best_sum[tid];
            // exact ties go to the earlier iteration, as in the sequential loop
            const bool take2 = (i2 >= 0) && (i1b < 0 || g2 > g1 || (g2 == g1 && (e2 < e1 || (e2 == e1 && i2 < i1b))));
            if (take2) {
                s_best_good[tid] = g2;
                s_best_sum[tid] = e2;
                s_best_idx[tid] = i2;
            }
        }
        __syncthreads();
    }
    if (tid == 0) {
        const int g = s_best_good[0], hyp = s_best_idx[0];
        const double e = s_best_sum[0];
        int newMax = maxGood0;
        if (hyp >= 0 && (g > max(maxGood0, 4) || (g == max(maxGood0, 5) && st->errminsum > e))) {
            // (for g == maxGood0 the carried model is earlier and keeps the tie unless the sum is strictly smaller)
            const int idx = hyp * 10 + hslot[hyp];
            newMax = g;
            st->errminsum = e;
            st->best = base_index + idx;
            for (int k = 0; k < 9; ++k) st->E[k] = E_tab[(size_t)idx * 9 + k];
        }
        st->maxGood = newMax;
        // newMax is a count pass 1 (or an earlier chunk) has already evaluated and recorded
        const int nit = (newMax >= 5) ? min(niters0, Ttab ? Ttab[min(newMax, npts)] : dev_num_iters(log_num, npts, min(newMax, npts)))
                                      : niters0;
        st->niters = nit;
        st->iter = iter0 + processed;
        st->stop = (iter0 + processed >= nit) ? 1 : 0;
        st->models_scored += *dense_total;
    }
}

// modelest.cpp:444-463: keep a refit model if it has more inliers, or as many and a smaller error sum.
__global__ void refit_decide_kernel(const int32_t *__restrict__ n_models, const int32_t *__restrict__ good,
                                    const double *__restrict__ esum, const double *__restrict__ E_tab, ReplayState *__restrict__ st) {
    if (threadIdx.x != 0 || blockIdx.x != 0) return;
    const int nm = n_models[0];
    st->refit_models = nm;
    st->refit_taken = -1;
    for (int m = 0; m < nm; ++m) {
        const int g = good[m];
        const double e = esum[m];
        if (g > max(st->maxGood, 4)) {
            st->maxGood = g;
            st->errminsum = e;
            st->refit_taken = m;
        } else if (g == st->maxGood && st->errminsum > e) {
            st->errminsum = e;
            st->refit_taken = m;
        }
    }
    if (st->refit_taken >= 0)
        for (int k = 0; k < 9; ++k) st->E[k] = E_tab[(size_t)st->refit_taken * 9 + k];
}

// ---------------------------------------------------------------------------------------------------------------
// host side
// ---------------------------------------------------------------------------------------------------------------

// glibc srand()/rand() (TYPE_3 additive feedback), so that a seed reproduces the reference's sample stream
// glibc srand()/rand() (TYPE_3 additive feedback generator: r[k+3] += r[k] over a ring of 31 words, output r >> 1, the first 310
// outputs discarded), produced 31 values at a time: one unrolled pass over the ring has no index wrap-around and three independent
// dependency chains.  After seeding the generator is block-aligned (310 = 10 * 31), so block k holds outputs 31 k .. 31 k + 30.
struct GlibcRand {
    uint32_t r[31];
    int32_t out[31];
    int pos;
    void refill() {
#pragma unroll
        for (int k = 0; k < 28; ++k) {
            r[k + 3] += r[k];
            out[k] = (int32_t)(r[k + 3] >> 1);
        }
        for (int k = 28; k < 31; ++k) {
            r[k - 28] += r[k];
            out[k] = (int32_t)(r[k - 28] >> 1);
        }
        pos = 0;
    }
    void seed(unsigned s) {
        if (s == 0) s = 1;
        int32_t t[31];
        t[0] = (int32_t)s;
        for (int i = 1; i < 31; ++i) {
            const long hi = t[i - 1] / 127773, lo = t[i - 1] % 127773;
            long w = 16807 * lo - 2836 * hi;
            if (w < 0) w += 2147483647;
            t[i] = (int32_t)w;
        }
        for (int i = 0; i < 31; ++i) r[i] = (uint32_t)t[i];
        for (int i = 0; i < 10; ++i) refill();
        pos = 31;
    }
    int next() {
        if (pos == 31) refill();
        return out[pos++];
    }
};

// getSubset (modelest.cpp:567-610): 5 distinct indices, duplicates redrawn.  checkSubset (:613-650) returns
// `i >= i1` with i0 == i1, i.e. true for every input, so no geometric rejection ever happens in the reference.
// v % n for 0 <= v < 2^31 without a division (Lemire's fastmod, exact for 32-bit operands): M = floor((2^64 - 1) / n) + 1.
struct FastMod {
    uint64_t M;
    uint32_t n;
    explicit FastMod(int n_) : M(UINT64_C(0xFFFFFFFFFFFFFFFF) / (uint32_t)n_ + 1), n((uint32_t)n_) {}
    int operator()(int v) const { return (int)(((unsigned __int128)(M * (uint32_t)v) * n) >> 64); }
};

// The raw rand() stream of a seed does not depend on the data, and the reference seeds with time(NULL): every call within the same
// second -- and every call of a run with a fixed seed -- walks the same stream.  The context therefore keeps the stream of the last
// seed (up to kRandCacheMax values, extended on demand); a call then only reduces it modulo ITS n and parses the samples.
constexpr size_t kRandCacheMax = size_t(1) << 22;  // 16 MiB
struct RandCache {
    unsigned seed = 0;
    bool valid = false;
    GlibcRand gen;             // positioned behind the last cached value
    std::vector<int32_t> raw;  // outputs 0 .. raw.size() - 1 of srand(seed)
};

struct RandCursor {
    RandCache *c;
    size_t pos = 0;
    bool live = false;  // beyond the cached prefix: a private generator continues
    GlibcRand tail;
    size_t cap;
    RandCursor(RandCache *cache, unsigned seed, size_t cap_) : c(cache), cap(cap_ ? cap_ : kRandCacheMax) {
        if (!c->valid || c->seed != seed || c->raw.size() > cap) {
            c->seed = seed;
            c->valid = true;
            c->gen.seed(seed);
            c->raw.clear();
        }
    }
    // pointer to `count` (<= 31) consecutive cached values at the cursor, or nullptr when they are not all inside the cache
    const int32_t *peek(size_t count) {
        if (live) return nullptr;
        while (c->raw.size() < pos + count) {
            if (c->raw.size() + 31 > cap) return nullptr;
            c->gen.refill();
            c->raw.insert(c->raw.end(), c->gen.out, c->gen.out + 31);
            c->gen.pos = 31;
        }
        return c->raw.data() + pos;
    }
    int next() {
        if (!live) {
            if (const int32_t *p = peek(1)) {
                ++pos;
                return *p;
            }
            // the cache is full and the cursor is at its end: continue with a private copy of the generator
            tail = c->gen;
            live = true;
        }
        return tail.next();
    }
};

void draw_sample(RandCursor &g, const FastMod &mod, int32_t *idx) {
    if (const int32_t *o = g.peek(5)) {  // the common case in one go: five independent reductions, no duplicate among them
        const int v0 = mod(o[0]), v1 = mod(o[1]), v2 = mod(o[2]), v3 = mod(o[3]), v4 = mod(o[4]);
        if (v0 != v1 && v0 != v2 && v0 != v3 && v0 != v4 && v1 != v2 && v1 != v3 && v1 != v4 && v2 != v3 && v2 != v4 && v3 != v4) {
            idx[0] = v0, idx[1] = v1, idx[2] = v2, idx[3] = v3, idx[4] = v4;
            g.pos += 5;
            return;
        }
    }
    for (int i = 0; i < 5;) {  // the reference's loop: one draw per pick, duplicates redrawn
        const int v = mod(g.next());
        bool dup = false;
        for (int j = 0; j < i; ++j) dup = dup || (idx[j] == v);
        if (dup) continue;
        idx[i++] = v;
    }
}


// ---------------------------------------------------------------------------------------------------------------
// getSubset (modelest.cpp:567-610) for ONE pair and a large pass, on the device.  The stream position of sample i + 1 depends on the
// redraws of sample i, but redraws are rare (a sample repeats an index with probability ~10 / n) and between two of them the sample
// starts are an arithmetic progression of step 5.  Three launches:
//   draw_scan_kernel   every stream position p of the pass's window in parallel: the five values at p reduced modulo n; where two of
//                      them coincide, p is an EVENT CANDIDATE -- a sample starting there redraws -- and the thread replays the reference's
//                      draw-by-draw loop from p (draws consumed, the five indices) into a candidate list;
//   draw_chain_kernel  one wave walks from event to event: the next event after position p is the smallest candidate q >= p with
//                      q = p (mod 5); the samples before it form a segment (first sample, first position), the event sample takes the
//                      candidate's indices, the walk continues behind the draws it consumed.  ~40 steps for 20 000 samples at n = 5000;
//   draw_fill_kernel   every sample in parallel: its segment by binary search, its five values at first position + 5 (i - first sample),
//                      written unless they repeat (then it is an event sample, already written).
// The raw stream of the seed is cached on the device (uploaded when the seed changes).  Anything that does not fit -- window or lists too
// small (tiny n: every second sample redraws), stream longer than the cache -- sets an overflow flag and the call is redone with the host
// drawing the table.  The host (9 ns per sample, overlapped slice by slice with the solver) had become the solver phase's critical path
// once the elimination kernel took three hypotheses per wave: 275 us for 20 000 hypotheses of which the kernels need 140.
// ---------------------------------------------------------------------------------------------------------------
constexpr int kDrawCandCap = 4096;
constexpr int kDrawSegCap = 1024;
struct DrawCand {
    int32_t p, c;       // window-relative position, draws consumed by the sample starting there (-1: the stream ran out)
    int32_t idx[5];
    int32_t pad;
};
struct DrawCtl {
    int32_t pos;        // stream position at the start of the next pass
    int32_t pos_prev;   // ... of the pass being filled
    int32_t overflow;
    int32_t ncand, nseg;
    int32_t pad[3];
    int32_t seg_first[kDrawSegCap], seg_pos[kDrawSegCap];
};

__global__ __launch_bounds__(256) void draw_scan_kernel(const int32_t *__restrict__ raw, int raw_len, int n, int window, DrawCtl *__restrict__ ctl,
                                                        DrawCand *__restrict__ cand) {
    const int w = blockIdx.x * blockDim.x + threadIdx.x;
    if (w >= window) return;
    const int p = ctl->pos + w;
    if (p + 5 > raw_len) return;  // (a sample that would have to start here makes the chain kernel report the overflow)
    const uint32_t un = (uint32_t)n;
    int v[5];
#pragma unroll
    for (int k = 0; k < 5; ++k) v[k] = (int)((uint32_t)raw[p + k] % un);
    const bool dup = v[0] == v[1] || v[0] == v[2] || v[0] == v[3] || v[0] == v[4] || v[1] == v[2] || v[1] == v[3] || v[1] == v[4] || v[2] == v[3] ||
                     v[2] == v[4] || v[3] == v[4];
    if (!dup) return;
    int idx[5] = {0, 0, 0, 0, 0};
    int c = 0, q = p;
    bool out = false;
    while (c < 5) {  // the reference's loop: one draw per pick, repeats redrawn
        if (q >= raw_len) {
            out = true;
            break;
        }
        const int x = (int)((uint32_t)raw[q++] % un);
        bool rep = false;
#pragma unroll
        for (int k = 0; k < 5; ++k) rep = rep || (k < c && idx[k] == x);
        if (rep) continue;
#pragma unroll
        for (int k = 0; k < 5; ++k) idx[k] = (k == c) ? x : idx[k];
        ++c;
    }
    const int slot = atomicAdd(&ctl->ncand, 1);
    if (slot < kDrawCandCap) {
        DrawCand d;
        d.p = w, d.c = out ? -1 : q - p, d.pad = 0;
#pragma unroll
        for (int k = 0; k < 5; ++k) d.idx[k] = idx[k];
        cand[slot] = d;
    }
}

// Minimum over the wave on data-parallel-primitive moves (row shifts inside the rows of 16 lanes, then the two row broadcasts of gfx9):
// six dependent VALU instructions, against six dependent LDS-crossbar round trips of a __shfl_xor butterfly (~700 cycles, which made a
// chain step 640 ns).  Lanes without a source keep INT_MAX.
__device__ __forceinline__ int wave_min_i32(int x) {
    x = min(x, __builtin_amdgcn_update_dpp(INT_MAX, x, 0x111, 0xf, 0xf, false));  // row_shr:1
    x = min(x, __builtin_amdgcn_update_dpp(INT_MAX, x, 0x112, 0xf, 0xf, false));  // row_shr:2
    x = min(x, __builtin_amdgcn_update_dpp(INT_MAX, x, 0x114, 0xf, 0xf, false));  // row_shr:4
    x = min(x, __builtin_amdgcn_update_dpp(INT_MAX, x, 0x118, 0xf, 0xf, false));  // row_shr:8: lane 15 of a row holds the row's minimum
    x = min(x, __builtin_amdgcn_update_dpp(INT_MAX, x, 0x142, 0xa, 0xf, false));  // row_bcast:15 into rows 1 and 3
    x = min(x, __builtin_amdgcn_update_dpp(INT_MAX, x, 0x143, 0xc, 0xf, false));  // row_bcast:31 into rows 2 and 3
    return __builtin_amdgcn_readlane(x, 63);
}

__global__ __launch_bounds__(64) void draw_chain_kernel(int raw_len, int cnt, int window, DrawCtl *__restrict__ ctl, const DrawCand *__restrict__ cand,
                                                        int32_t *__restrict__ out) {
    __shared__ int cp[kDrawCandCap], cc[kDrawCandCap];
    __shared__ int ev_i[kDrawSegCap], ev_k[kDrawSegCap];  // event samples met on the way: (sample, candidate), written after the walk
    __shared__ int sg_first[kDrawSegCap], sg_pos[kDrawSegCap];  // the segments, copied out after the walk (a global store per step stalled it)
    const int lane = threadIdx.x;
    // (this kernel is a chain of memory round trips around a short walk: every load that can be issued early is)
    const int nc = ctl->ncand, ovf_in = ctl->overflow, pos_in = ctl->pos;
    int2 first4[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) first4[u] = *reinterpret_cast<const int2 *>(&cand[min(lane + 64 * u, kDrawCandCap - 1)]);  // (p, c) of a candidate
    bool overflow = ovf_in != 0 || nc > kDrawCandCap;
    const int ncl = min(nc, kDrawCandCap);
#pragma unroll
    for (int u = 0; u < 4; ++u)
        if (lane + 64 * u < ncl) cp[lane + 64 * u] = first4[u].x, cc[lane + 64 * u] = first4[u].y;
    for (int k = lane + 256; k < ncl; k += 64) cp[k] = cand[k].p, cc[k] = cand[k].c;
    __syncthreads();
    int p = 0, i = 0, nseg = 0, p_end = 0, nev = 0;
    // up to 256 candidates (the usual case: ~10 / n of the window's positions) stay in registers, four per lane: a step of the walk is then
    // a handful of compares and the wave minimum, no LDS round trip
    const bool in_regs = ncl <= 256;
    int rp[4], rcn[4], rcl[4];  // position, draws consumed, position mod 5 (a progression's events share its residue)
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int k = lane + 64 * u;
        rp[u] = (in_regs && k < ncl) ? cp[k] : INT_MAX;
        rcn[u] = (in_regs && k < ncl) ? cc[k] : -1;
        rcl[u] = (in_regs && k < ncl) ? cp[k] % 5 : -1;
    }
    int cev = -1;  // draws consumed by the event sample found in this step
    while (!overflow) {
        // the next event: the smallest candidate position q >= p on this progression
        int best = INT_MAX, bk = -1, bc = -1;
        if (in_regs) {
            const int pcl = p % 5;
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                const int q = rp[u];
                const bool take = rcl[u] == pcl && q >= p && q < best;
                best = take ? q : best, bk = take ? lane + 64 * u : bk, bc = take ? rcn[u] : bc;
            }
        } else {
            for (int k = lane; k < ncl; k += 64) {
                const int q = cp[k];
                if (q >= p && (q - p) % 5 == 0 && q < best) best = q, bk = k, bc = cc[k];
            }
        }
        {
            const int gbest = wave_min_i32(best);
            const unsigned long long who = __ballot(best == gbest && gbest != INT_MAX);
            const int leader = who ? __ffsll((long long)who) - 1 : 0;
            bk = who ? __builtin_amdgcn_readlane(bk, leader) : -1;
            cev = who ? __builtin_amdgcn_readlane(bc, leader) : -1;
            best = gbest;
        }
        const long long ie = (best == INT_MAX) ? (long long)cnt : (long long)i + (best - p) / 5;
        if (nseg >= kDrawSegCap) {
            overflow = true;
            break;
        }
        if (lane == 0) sg_first[nseg] = i, sg_pos[nseg] = p;
        ++nseg;
        if (ie >= cnt) {  // the rest of the pass lies on this progression
            p_end = p + 5 * (cnt - i);
            break;
        }
        const int c = cev;
        if (c < 0) {
            overflow = true;
            break;
        }
        if (lane == 0) ev_i[nseg - 1] = (int)ie, ev_k[nseg - 1] = bk;  // (one event per segment closed)
        ++nev;
        p = best + c, i = (int)ie + 1, p_end = p;
        if (i >= cnt) break;
    }
    if (!overflow && (p_end > window || pos_in + p_end > raw_len)) overflow = true;
    __syncthreads();
    for (int k = lane; k < nseg; k += 64) ctl->seg_first[k] = sg_first[k], ctl->seg_pos[k] = sg_pos[k];
    // the event samples, all lanes (inside the walk each one cost the wave a memory round trip: 32 us for ~45 events)
    for (int e = lane; e < nev && !overflow; e += 64) {
        const DrawCand d = cand[ev_k[e]];
#pragma unroll
        for (int k = 0; k < 5; ++k) out[(size_t)ev_i[e] * 5 + k] = d.idx[k];
    }
    if (lane == 0) {
        ctl->nseg = nseg;
        ctl->pos_prev = pos_in;
        ctl->pos = pos_in + p_end;
        ctl->ncand = 0;  // for the next pass
        if (overflow) ctl->overflow = 1;
    }
}

__global__ __launch_bounds__(256) void draw_fill_kernel(const int32_t *__restrict__ raw, int raw_len, int n, int cnt, const DrawCtl *__restrict__ ctl,
                                                        int32_t *__restrict__ out) {
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= cnt) return;
    int v[5] = {0, 1, 2, 3, 4};  // what an overflowed pass leaves: valid indices (the call is redone)
    bool write = true;
    if (!ctl->overflow) {
        int lo = 0, hi = ctl->nseg - 1;  // the last segment whose first sample is <= i
        while (lo < hi) {
            const int mid = (lo + hi + 1) >> 1;
            if (ctl->seg_first[mid] <= i) lo = mid;
            else hi = mid - 1;
        }
        const long long p = (long long)ctl->pos_prev + ctl->seg_pos[lo] + 5ll * (i - ctl->seg_first[lo]);
        if (p + 5 <= raw_len) {
            const uint32_t un = (uint32_t)n;
#pragma unroll
            for (int k = 0; k < 5; ++k) v[k] = (int)((uint32_t)raw[p + k] % un);
            write = !(v[0] == v[1] || v[0] == v[2] || v[0] == v[3] || v[0] == v[4] || v[1] == v[2] || v[1] == v[3] || v[1] == v[4] || v[2] == v[3] ||
                      v[2] == v[4] || v[3] == v[4]);  // (a repeat: an event sample, written by the chain kernel)
        }
    }
    if (write) {
#pragma unroll
        for (int k = 0; k < 5; ++k) out[(size_t)i * 5 + k] = v[k];
    }
}

// cvRANSACUpdateNumIters1 (modelest.cpp:86-109)
int update_num_iters(double p, double ep, int model_points, int max_iters) {
    p = std::max(p, 0.);
    p = std::min(p, 1.);
    ep = std::max(ep, 0.);
    ep = std::min(ep, 1.);
    double num = std::max(1. - p, DBL_MIN);
    double denom = 1. - std::pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = std::log(num);
    denom = std::log(denom);
    return denom >= 0 || -num >= (double)max_iters * (-denom) ? max_iters : (int)std::round(num / denom);
}

// qmax of sampson_inlier(): the largest double whose float rounding is <= thresh^2.
double inlier_bound(double thresh2) {
    float f = (float)thresh2;
    if ((double)f > thresh2) f = std::nextafterf(f, 0.0f);  // f* = largest float <= thresh^2
    if (!(f >= 0.0f) || !std::isfinite(f)) return -1.0;     // nothing (or everything) passes: the kernels' slow path handles it
    const float fn = std::nextafterf(f, INFINITY);
    if (!std::isfinite(fn)) return -1.0;
    const double mid = 0.5 * ((double)f + (double)fn);      // exact: two adjacent floats
    uint32_t bits;
    std::memcpy(&bits, &f, 4);
    return (bits & 1u) ? std::nextafter(mid, 0.0) : mid;     // ties-to-even sends the midpoint to f* only if f* is even
}

}  // namespace

// Launchers used by the C ABI ------------------------------------------------------------------------------------

// Sampson scoring of up to `max_models` models (dense list, live count on the device or the host): few models -> one block per
// model, many -> 4 lanes per model (see the two kernels).  Identical results either way.
constexpr int kScoreBlockMaxModels = 24576;
static void launch_score(hipStream_t s, const double4 *pts, int n, const double *E_list, const int32_t *ids, const int32_t *total_ptr,
                         int total_host, int max_models, double thresh2, int32_t *good, double *esum, bool sums = true,
                         double qmax = -1.0, int point_splits = 1, bool f32_filter = false, int mpl = 1, bool defer = false, int count_threads = 512, int count_wpe = 5) {
    if (max_models <= 0) return;
    const size_t lds = (size_t)((n + 3) / 4 * 4) * sizeof(float);
    const bool block = n <= kScoreBlockMaxN && max_models <= kScoreBlockMaxModels;
    if (sums) {
        if (block)
            hipLaunchKernelGGL((score_models_block_kernel<true, true>), dim3(max_models), dim3(256), lds, s, pts, n, E_list, ids, total_ptr,
                               total_host, thresh2, qmax, good, esum, (const int32_t *)nullptr);
        else
            hipLaunchKernelGGL(score_models_kernel<true>, dim3((max_models + kScoreModels - 1) / kScoreModels), dim3(kScoreThreads), 0, s, pts, n, E_list, ids, total_ptr,
                               total_host, thresh2, qmax, good, esum);
    } else if (point_splits == 0) {  // one workgroup per model, no accumulation: callers that do not zero the table
        hipLaunchKernelGGL((score_models_block_kernel<true, false>), dim3(max_models), dim3(256), 0, s, pts, n, E_list, ids, total_ptr,
                           total_host, thresh2, qmax, good, esum, (const int32_t *)nullptr);
    } else if (max_models <= kScoreBlockMaxModels) {
        const dim3 grid((max_models + kScoreThreadsSmall / 4 - 1) / (kScoreThreadsSmall / 4), point_splits);
        if (f32_filter)
            hipLaunchKernelGGL((count_models_f32_kernel<kScoreThreadsSmall, kScoreTileSmall>), grid, dim3(kScoreThreadsSmall), 0, s, pts, n, E_list, ids,
                               total_ptr, total_host, thresh2, qmax, good);
        else
            hipLaunchKernelGGL((score_models_kernel<false, kScoreThreadsSmall, kScoreTileSmall>), grid, dim3(kScoreThreadsSmall), 0, s, pts, n, E_list,
                               ids, total_ptr, total_host, thresh2, qmax, good, esum);
    } else {
        const dim3 grid((max_models + kScoreModels - 1) / kScoreModels, point_splits);
        if (f32_filter && mpl == 2 && defer && n < (1 << 23) && count_threads == 256 && count_wpe == 6)   // A/B: 80 VGPRs, six 4-wave workgroups per CU
            hipLaunchKernelGGL((count_models_f32_kernel<256, kScoreTile, 2, true, 6>), dim3((max_models + 127) / 128, point_splits),
                               dim3(256), 0, s, pts, n, E_list, ids, total_ptr, total_host, thresh2, qmax, good);
        else if (f32_filter && mpl == 2 && defer && n < (1 << 23) && count_threads == 256)   // round 6: 4-wave workgroups at 96 VGPRs, five per CU
            hipLaunchKernelGGL((count_models_f32_kernel<256, kScoreTile, 2, true>), dim3((max_models + 127) / 128, point_splits),
                               dim3(256), 0, s, pts, n, E_list, ids, total_ptr, total_host, thresh2, qmax, good);
        else if (f32_filter && mpl == 2 && defer && n < (1 << 23))
            hipLaunchKernelGGL((count_models_f32_kernel<kScoreThreads, kScoreTile, 2, true>), dim3((max_models + 2 * kScoreModels - 1) / (2 * kScoreModels), point_splits),
                               dim3(kScoreThreads), 0, s, pts, n, E_list, ids, total_ptr, total_host, thresh2, qmax, good);
        else if (f32_filter && mpl == 2)
            hipLaunchKernelGGL((count_models_f32_kernel<kScoreThreads, kScoreTile, 2>), dim3((max_models + 2 * kScoreModels - 1) / (2 * kScoreModels), point_splits),
                               dim3(kScoreThreads), 0, s, pts, n, E_list, ids, total_ptr, total_host, thresh2, qmax, good);
        else if (f32_filter)
            hipLaunchKernelGGL((count_models_f32_kernel<kScoreThreads, kScoreTile>), grid, dim3(kScoreThreads), 0, s, pts, n, E_list, ids, total_ptr,
                               total_host, thresh2, qmax, good);
        else
            hipLaunchKernelGGL((score_models_kernel<false, kScoreThreads>), grid, dim3(kScoreThreads), 0, s, pts, n, E_list, ids, total_ptr,
                               total_host, thresh2, qmax, good, esum);
    }
}

// Workgroups per model group of the count-only scoring pass (see score_models_kernel); with more than one the counts are ACCUMULATED
// into a table the caller has zeroed (roots_kernel does that for the RANSAC pass).  launch_score(point_splits = 0) is the form for
// callers with an unzeroed table: one workgroup per model.
static int score_point_splits(int n, int max_models, bool sums, int tiles_per_wg = 2) {
    if (sums) return 1;
    const int ntiles = (n + kScoreTile - 1) / kScoreTile;
    if (max_models <= kScoreBlockMaxModels)  // small passes: one 256-point tile per workgroup (occupancy is what hides the LDS latency)
        return std::max(1, std::min(32, (n + kScoreTileSmall - 1) / kScoreTileSmall));
    return tiles_per_wg == 1 ? std::max(1, std::min(16, ntiles)) : std::max(1, std::min(8, ntiles / 2));
}

static int pack_points(mlpl_ctx *ctx, const double *d_p1, const double *d_p2, int n, double4 **d_pts, hipStream_t s,
                       ReplayState *d_st = nullptr, int niters = 0, int32_t *zero_ints = nullptr, int n_zero = 0) {
    void *buf = nullptr;
    int rc = ws_get(ctx, WS_AUX3, (size_t)n * (sizeof(double4) + sizeof(double) + 5 * sizeof(float)), &buf);  // points, kp[n], float records
    if (rc) return rc;
    hipLaunchKernelGGL(pack_points_kernel, dim3((n + 255) / 256), dim3(256), 0, s, d_p1, d_p2, n, (double4 *)buf, d_st, niters, zero_ints,
                       n_zero);
    *d_pts = (double4 *)buf;
    return MLPL_OK;
}

struct RansacBuffers {
    int32_t *samples;   // [chunk][5]
    double *E_tab;      // [chunk][10][9]
    int32_t *n_models;  // [chunk]
    double *dense_E;    // [chunk*10][9]
    int32_t *dense_id;  // [chunk*10]
    int32_t *good;      // [chunk*10]
    double *esum;       // [chunk*10]
    int32_t *total;     // [1]
    int32_t *hgood;     // [chunk]  per-hypothesis best count
    int32_t *hslot;     // [chunk]
    double *hsum;       // [chunk]
    PolyRec *recs;      // [chunk]  solver -> roots hand-over
    int32_t *cand;      // [chunk*10] ids of the models whose error sum is needed (lazy sums)
    int32_t *cand_count;  // [1]
};

static int alloc_ransac(mlpl_ctx *ctx, int chunk, RansacBuffers &B) {
    void *p;
    int rc;
    if ((rc = ws_get(ctx, WS_AUX4, (size_t)chunk * 5 * 4, &p))) return rc;
    B.samples = (int32_t *)p;
    if ((rc = ws_get(ctx, WS_AUX5, (size_t)chunk * 90 * 8, &p))) return rc;
    B.E_tab = (double *)p;
    if ((rc = ws_get(ctx, WS_AUX6, (size_t)chunk * 90 * 8, &p))) return rc;
    B.dense_E = (double *)p;
    if ((rc = ws_get(ctx, WS_PARTIAL, (size_t)chunk * sizeof(PolyRec), &p))) return rc;
    B.recs = (PolyRec *)p;
    // one slot for the small integer/double tables: [n_models | dense_id | good | total] + esum
    const size_t ints = (size_t)chunk * (1 + 10 + 10 + 2 + 10) + 32;
    if ((rc = ws_get(ctx, WS_AUX7, ints * 4 + (size_t)chunk * 11 * 8 + 64, &p))) return rc;
    B.esum = (double *)p;
    B.hsum = B.esum + (size_t)chunk * 10;
    B.n_models = (int32_t *)(B.hsum + chunk);
    B.dense_id = B.n_models + chunk;
    B.good = B.dense_id + (size_t)chunk * 10;
    B.hgood = B.good + (size_t)chunk * 10;
    B.hslot = B.hgood + chunk;
    B.total = B.hslot + chunk;
    B.cand_count = B.total + 8;
    B.cand = B.total + 16;
    return MLPL_OK;
}

#include "batch_hub.h"
#include "hub_kernels.h"
void hub_streams_free(void *p) {
    HubStreams *h = static_cast<HubStreams *>(p);
    if (!h) return;
    for (HubLane &L : h->lane) {
        for (int i = 0; i < kHubMaxGroups; ++i) {
            if (L.ev[i]) (void)hipEventDestroy(L.ev[i]);
            if (i && L.aux[i]) (void)hipStreamDestroy(L.aux[i]);
        }
        if (L.own) (void)hipStreamDestroy(L.own);
        if (L.done) (void)hipEventDestroy(L.done);
        if (L.items_host) (void)hipHostFree(L.items_host);
        if (L.items_dev) (void)hipFree(L.items_dev);
    }
    if (h->copy) (void)hipStreamDestroy(h->copy);
    if (h->feed) (void)hipStreamDestroy(h->feed);
    if (h->feed_start) (void)hipEventDestroy(h->feed_start);
    delete h;
}
#include "arrsac_impl.h"
#include "usac_impl.h"
#include "pair_batch_impl.h"
#include "pair_batch_usac.h"

void free_rand_cache(void *p) { delete static_cast<RandCache *>(p); }

}  // namespace mlpl

using namespace mlpl;

extern "C" {

// The run scheduler of the launch hub by itself (no GPU, no context): n fibers on `workers` threads, each passing `rounds` times through
// the hub's hand-over -- announce, block until the generation advances -- against a stand-in hub on the calling thread that releases a
// round once every fiber has announced.  Returns the number of hand-overs completed (n * rounds when the scheduler works), -1 on bad
// arguments.  tests/test_fiber_scheduler.py runs it without a GPU, also with far more fibers than workers and than host cores.
long long mlpl_debug_fiber_selftest(int n, int workers, int rounds) {
    if (n < 1 || n > 4096 || workers < 1 || workers > 256 || rounds < 1 || rounds > 100000) return -1;
    HubThreads pool;
    std::atomic<int> pending{n};
    std::atomic<uint32_t> gen{0}, hub_word{0};
    std::atomic<long long> done{0};
    std::vector<int> order_seen((size_t)n, 0);
    pool.start(n, [&](int k) {
        volatile char pad[4096];  // (every fiber really has its own stack)
        pad[0] = (char)k;
        for (int r = 0; r < rounds; ++r) {
            const uint32_t g = gen.load(std::memory_order_acquire);
            if (pending.fetch_sub(1, std::memory_order_acq_rel) == 1) {
                hub_word.fetch_add(1, std::memory_order_release);
                futex_wake_u32(&hub_word, 1);
            }
            hub_block_until_changed(&gen, g);
            if (gen.load(std::memory_order_acquire) != g + 1) return;  // released exactly once per round
            ++order_seen[(size_t)k];
            done.fetch_add(1, std::memory_order_relaxed);
        }
        (void)pad[0];
    }, workers);
    for (int r = 0; r < rounds; ++r) {
        for (;;) {
            const uint32_t w = hub_word.load(std::memory_order_acquire);
            if (pending.load(std::memory_order_acquire) == 0) break;
            futex_wait_u32(&hub_word, w);
        }
        pending.store(n, std::memory_order_release);
        gen.fetch_add(1, std::memory_order_release);
        futex_wake_u32(&gen, INT_MAX);
    }
    pool.wait();
    for (int k = 0; k < n; ++k)
        if (order_seen[(size_t)k] != rounds) return -2;
    return done.load();
}

int mlpl_ransac_last_stats(mlpl_ctx *ctx, long long stats[2]) {
    if (!ctx || !stats) return MLPL_E_BAD_INPUT;
    stats[0] = ctx->last_ransac_iters;
    stats[1] = ctx->last_ransac_models;
    return MLPL_OK;
}

int mlpl_debug_ransac_draw(mlpl_ctx *ctx, long long out[2]) {
    if (!ctx || !out) return MLPL_E_BAD_INPUT;
    out[0] = ctx->ransac_draw_fallbacks;
    out[1] = ctx->last_ransac_dev_draw;
    return MLPL_OK;
}

int mlpl_debug_eig9(mlpl_ctx *ctx, const double *G, const double *start, int count, double *out12, double *jacobi10) {
    if (!ctx || !G || !out12 || !jacobi10 || count < 1 || count > (1 << 20)) {
        set_error("mlpl_debug_eig9: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    void *dG, *dS, *dO, *dJ;
    int rc;
    if ((rc = ws_get(ctx, WS_AUX0, (size_t)count * 81 * 8, &dG))) return rc;
    if ((rc = ws_get(ctx, WS_AUX1, (size_t)count * 9 * 8, &dS))) return rc;
    if ((rc = ws_get(ctx, WS_AUX5, (size_t)count * 12 * 8, &dO))) return rc;
    if ((rc = ws_get(ctx, WS_AUX6, (size_t)count * 10 * 8, &dJ))) return rc;
    MLPL_HIP_TRY(hipMemcpyAsync(dG, G, (size_t)count * 81 * 8, hipMemcpyHostToDevice, s));
    if (start) MLPL_HIP_TRY(hipMemcpyAsync(dS, start, (size_t)count * 9 * 8, hipMemcpyHostToDevice, s));
    hipLaunchKernelGGL(eig9_debug_kernel, dim3(count), dim3(64), 0, s, (const double *)dG, start ? (const double *)dS : nullptr, count, (double *)dO, (double *)dJ);
    MLPL_HIP_TRY(hipGetLastError());
    MLPL_HIP_TRY(hipMemcpyAsync(out12, dO, (size_t)count * 12 * 8, hipMemcpyDeviceToHost, s));
    MLPL_HIP_TRY(hipMemcpyAsync(jacobi10, dJ, (size_t)count * 10 * 8, hipMemcpyDeviceToHost, s));
    MLPL_HIP_TRY(hipStreamSynchronize(s));
    return MLPL_OK;
}

int mlpl_debug_dk_stats(mlpl_ctx *ctx, int enable, int stats[16]) {
    if (!ctx) return MLPL_E_BAD_INPUT;
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    MLPL_HIP_TRY(hipDeviceSynchronize());
    int h[16] = {0};
    MLPL_HIP_TRY(hipMemcpyFromSymbol(h, HIP_SYMBOL(g_dk_iters_dbg), sizeof(h)));
    if (stats) std::memcpy(stats, h, sizeof(h));
    const int z[16] = {0, 0, 0, enable ? 1 : 0};
    MLPL_HIP_TRY(hipMemcpyToSymbol(HIP_SYMBOL(g_dk_iters_dbg), z, sizeof(z)));
    return MLPL_OK;
}

int mlpl_solve_5pt(mlpl_ctx *ctx, const double *p1, const double *p2, int n, const int32_t *samples, int n_samples,
                   double *E_out, int32_t *n_models) {
    if (!ctx || !p1 || !p2 || !samples || !E_out || !n_models || n < 5 || n_samples < 0) {
        set_error("mlpl_solve_5pt: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    for (long long i = 0; i < (long long)n_samples * 5; ++i)
        if (samples[i] < 0 || samples[i] >= n) {
            set_error("mlpl_solve_5pt: sample index out of range");
            return MLPL_E_BAD_INPUT;
        }
    if (n_samples == 0) return MLPL_OK;
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    void *dp1, *dp2;
    int rc;
    if ((rc = ws_get(ctx, WS_AUX0, (size_t)n * 16, &dp1))) return rc;
    if ((rc = ws_get(ctx, WS_AUX1, (size_t)n * 16, &dp2))) return rc;
    RansacBuffers B;
    if ((rc = alloc_ransac(ctx, n_samples, B))) return rc;
    MLPL_HIP_TRY(hipMemcpyAsync(dp1, p1, (size_t)n * 16, hipMemcpyHostToDevice, s));
    MLPL_HIP_TRY(hipMemcpyAsync(dp2, p2, (size_t)n * 16, hipMemcpyHostToDevice, s));
    MLPL_HIP_TRY(hipMemcpyAsync(B.samples, samples, (size_t)n_samples * 20, hipMemcpyHostToDevice, s));
    MLPL_HIP_TRY(hipMemsetAsync(B.E_tab, 0, (size_t)n_samples * 720, s));
    prof_mark(ctx, MLPL_PROF_SOLVE_5PT, 0, s);
    launch_solve5pt(ctx, n_samples, s, (const double *)dp1, (const double *)dp2, B.samples, 0, n_samples, B.recs);
    MLPL_LAUNCH_ROOTS(ctx->opt_solver_polish, dim3((n_samples + kHypPerWave - 1) / kHypPerWave), s, (const PolyRec *)B.recs, 0, n_samples,
                      B.E_tab, B.n_models, (double *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr);
    prof_mark(ctx, MLPL_PROF_SOLVE_5PT, 1, s);
    MLPL_HIP_TRY(hipGetLastError());
    MLPL_HIP_TRY(hipMemcpyAsync(E_out, B.E_tab, (size_t)n_samples * 720, hipMemcpyDeviceToHost, s));
    MLPL_HIP_TRY(hipMemcpyAsync(n_models, B.n_models, (size_t)n_samples * 4, hipMemcpyDeviceToHost, s));
    MLPL_HIP_TRY(hipStreamSynchronize(s));
    return MLPL_OK;
}

// err_sum == NULL: count-only (division-free) kernels with the squared threshold given as is; shape 1 / 2 forces the
// 4-lanes-per-model / block-per-model kernel (tests), 0 = automatic.
static int score_models_impl(mlpl_ctx *ctx, const double *p1, const double *p2, int n, const double *E, int n_models, double thresh2,
                             int32_t *count, double *err_sum, int shape) {
    if (n_models == 0) return MLPL_OK;
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    void *dp1, *dp2, *dE, *dgood, *dsum;
    int rc;
    if ((rc = ws_get(ctx, WS_AUX0, (size_t)n * 16, &dp1))) return rc;
    if ((rc = ws_get(ctx, WS_AUX1, (size_t)n * 16, &dp2))) return rc;
    if ((rc = ws_get(ctx, WS_AUX5, (size_t)n_models * 72, &dE))) return rc;
    if ((rc = ws_get(ctx, WS_AUX6, (size_t)n_models * 8, &dsum))) return rc;
    if ((rc = ws_get(ctx, WS_AUX7, (size_t)n_models * 4, &dgood))) return rc;
    MLPL_HIP_TRY(hipMemcpyAsync(dp1, p1, (size_t)n * 16, hipMemcpyHostToDevice, s));
    MLPL_HIP_TRY(hipMemcpyAsync(dp2, p2, (size_t)n * 16, hipMemcpyHostToDevice, s));
    MLPL_HIP_TRY(hipMemcpyAsync(dE, E, (size_t)n_models * 72, hipMemcpyHostToDevice, s));
    double4 *pts;
    if ((rc = pack_points(ctx, (const double *)dp1, (const double *)dp2, n, &pts, s))) return rc;
    prof_mark(ctx, MLPL_PROF_SCORE, 0, s);
    const double qmax = inlier_bound(thresh2);
    if (shape == 1 && !err_sum && ctx->opt_ransac_f32_filter && ctx->opt_ransac_count_mpl == 2 && ctx->opt_ransac_count_defer && n < (1 << 23) &&
        ctx->opt_ransac_count_threads == 256)
        hipLaunchKernelGGL((count_models_f32_kernel<256, kScoreTile, 2, true>), dim3((n_models + 127) / 128), dim3(256), 0, s,
                           (const double4 *)pts, n, (const double *)dE, (const int32_t *)nullptr, (const int32_t *)nullptr, n_models, thresh2, qmax,
                           (int32_t *)dgood);
    else if (shape == 1 && !err_sum && ctx->opt_ransac_f32_filter && ctx->opt_ransac_count_mpl == 2 && ctx->opt_ransac_count_defer && n < (1 << 23))
        hipLaunchKernelGGL((count_models_f32_kernel<kScoreThreads, kScoreTile, 2, true>), dim3((n_models + 2 * kScoreModels - 1) / (2 * kScoreModels)), dim3(kScoreThreads), 0, s,
                           (const double4 *)pts, n, (const double *)dE, (const int32_t *)nullptr, (const int32_t *)nullptr, n_models, thresh2, qmax,
                           (int32_t *)dgood);
    else if (shape == 1 && !err_sum && ctx->opt_ransac_f32_filter && ctx->opt_ransac_count_mpl == 2)
        hipLaunchKernelGGL((count_models_f32_kernel<kScoreThreads, kScoreTile, 2>), dim3((n_models + 2 * kScoreModels - 1) / (2 * kScoreModels)), dim3(kScoreThreads), 0, s,
                           (const double4 *)pts, n, (const double *)dE, (const int32_t *)nullptr, (const int32_t *)nullptr, n_models, thresh2, qmax,
                           (int32_t *)dgood);
    else if (shape == 1 && !err_sum && ctx->opt_ransac_f32_filter)
        hipLaunchKernelGGL((count_models_f32_kernel<kScoreThreads, kScoreTile>), dim3((n_models + kScoreModels - 1) / kScoreModels), dim3(kScoreThreads), 0, s,
                           (const double4 *)pts, n, (const double *)dE, (const int32_t *)nullptr, (const int32_t *)nullptr, n_models, thresh2, qmax,
                           (int32_t *)dgood);
    else if (shape == 1 && !err_sum)
        hipLaunchKernelGGL(score_models_kernel<false>, dim3((n_models + kScoreModels - 1) / kScoreModels), dim3(kScoreThreads), 0, s, (const double4 *)pts, n,
                           (const double *)dE, (const int32_t *)nullptr, (const int32_t *)nullptr, n_models, thresh2, qmax,
                           (int32_t *)dgood, (double *)dsum);
    else if (shape == 2 && !err_sum)
        hipLaunchKernelGGL((score_models_block_kernel<true, false>), dim3(std::min(n_models, 4096)), dim3(256), 0, s, (const double4 *)pts,
                           n, (const double *)dE, (const int32_t *)nullptr, (const int32_t *)nullptr, n_models, thresh2, qmax,
                           (int32_t *)dgood, (double *)dsum, (const int32_t *)nullptr);
    else
        launch_score(s, (const double4 *)pts, n, (const double *)dE, nullptr, nullptr, n_models, n_models, thresh2, (int32_t *)dgood,
                     (double *)dsum, err_sum != nullptr, qmax);
    prof_mark(ctx, MLPL_PROF_SCORE, 1, s);
    MLPL_HIP_TRY(hipGetLastError());
    MLPL_HIP_TRY(hipMemcpyAsync(count, dgood, (size_t)n_models * 4, hipMemcpyDeviceToHost, s));
    if (err_sum) MLPL_HIP_TRY(hipMemcpyAsync(err_sum, dsum, (size_t)n_models * 8, hipMemcpyDeviceToHost, s));
    MLPL_HIP_TRY(hipStreamSynchronize(s));
    return MLPL_OK;
}

int mlpl_score_models(mlpl_ctx *ctx, const double *p1, const double *p2, int n, const double *E, int n_models, double thresh,
                      int32_t *count, double *err_sum) {
    if (!ctx || !p1 || !p2 || !E || !count || !err_sum || n < 1 || n_models < 0) {
        set_error("mlpl_score_models: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    return score_models_impl(ctx, p1, p2, n, E, n_models, thresh * thresh, count, err_sum, 0);
}

int mlpl_count_models(mlpl_ctx *ctx, const double *p1, const double *p2, int n, const double *E, int n_models, double thresh2, int shape,
                      int32_t *count) {
    if (!ctx || !p1 || !p2 || !E || !count || n < 1 || n_models < 0 || shape < 0 || shape > 2) {
        set_error("mlpl_count_models: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    return score_models_impl(ctx, p1, p2, n, E, n_models, thresh2, count, nullptr, shape);
}

int mlpl_ransac_essential_dev(mlpl_ctx *ctx, const double *d_p1, const double *d_p2, int n, double thresh, double confidence,
                              int max_iters, int refit, uint32_t seed, double E[9], uint8_t *d_mask, int *n_inliers,
                              int *iters_used, void *stream) {
    if (!ctx || !d_p1 || !d_p2 || !E || !d_mask || n < 6 || max_iters < 1 || !(thresh > 0)) {
        set_error("mlpl_ransac_essential: bad arguments (n=%d max_iters=%d); n must exceed the 5 model points", n, max_iters);
        return MLPL_E_BAD_INPUT;
    }
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = pick_stream(ctx, stream);
    if (n_inliers) *n_inliers = 0;
    if (iters_used) *iters_used = 0;
    const double thresh2 = thresh * thresh;  // modelest.cpp:79
    // inlier counts without the division, error sums only where a tie has to be broken (sampson_inlier / candidate_kernel)
    const double qmax = inlier_bound(thresh2);
    const bool lazy = ctx->opt_ransac_lazy_sums != 0 && qmax > 0 && n <= kScoreBlockMaxN;

    double4 *pts;
    int rc;

    const int kChunk = ctx->opt_ransac_chunk > 0 ? ctx->opt_ransac_chunk : 32768;
    const int chunk_cap = std::min(max_iters, kChunk);
    RansacBuffers B;
    if ((rc = alloc_ransac(ctx, chunk_cap, B))) return rc;

    // small device block: replay state | niters table | refit scratch
    const int gblocks = std::max(1, std::min(256, (n + 255) / 256));
    void *dsm;
    const size_t off_T = 1024, off_refit = off_T + ((size_t)(n + 1) * 4 + 255) / 256 * 256;
    if ((rc = ws_get(ctx, WS_AUX2, off_refit + (144 + (size_t)gblocks * 45) * 8, &dsm))) return rc;
    ReplayState *d_st = (ReplayState *)dsm;
    int32_t *d_T = (int32_t *)((char *)dsm + off_T);
    double *d_rf = (double *)((char *)dsm + off_refit);

    // pinned staging: samples up, state down
    void *pin;
    if ((rc = pinned_get(ctx, (size_t)chunk_cap * 20 + (size_t)(n + 1) * 4 + 2048, &pin))) return rc;
    static_assert(sizeof(ReplayState) <= 1008, "staging layout");
    ReplayState *h_st = (ReplayState *)pin;
    int32_t *h_ovf = (int32_t *)((char *)pin + 1008);  // overflow flag of the device-side sampling, read back with the final state
    int32_t *h_T = (int32_t *)((char *)pin + 1024);
    int32_t *h_samples = h_T + (n + 1);
    int32_t *d_samples_mapped = nullptr;  // device view of the pinned sample table
    MLPL_HIP_TRY(hipHostGetDevicePointer((void **)&d_samples_mapped, h_samples, 0));

    // T[g] = cvRANSACUpdateNumIters1(confidence, (n-g)/n, 5, "infinity")  (modelest.cpp:86-109).  The replay evaluates it on the
    // device for the handful of counts it meets and the host checks exactly those values against its libm after the call; the
    // full host table (n + 1 log/pow pairs, 0.3 ms at n = 8192 -- more than the rest of an adaptive call, and rebuilt whenever
    // n changes, i.e. for every image pair) is only built when that check fails.
    const bool use_table = ctx->ransac_force_table != 0 || ctx->opt_ransac_host_table != 0;
    if (use_table) {
        if (ctx->ransac_T_n != n || ctx->ransac_T_conf != confidence || !ctx->ransac_T_host) {
            delete[] ctx->ransac_T_host;
            ctx->ransac_T_host = new int32_t[(size_t)n + 1];
            for (int g = 0; g <= n; ++g) ctx->ransac_T_host[g] = update_num_iters(confidence, (double)(n - g) / n, 5, INT32_MAX);
            ctx->ransac_T_n = n;
            ctx->ransac_T_conf = confidence;
        }
        std::memcpy(h_T, ctx->ransac_T_host, (size_t)(n + 1) * 4);
    } else {
        d_T = nullptr;
    }
    ReplayState init;
    std::memset(&init, 0, sizeof(init));
    init.niters = max_iters;
    init.errminsum = DBL_MAX;
    init.best = -1;
    init.refit_models = -1;
    init.refit_taken = -1;
    // the packing kernel also writes the start state of the replay and zeroes the dense-list counter of the first pass
    if ((rc = pack_points(ctx, d_p1, d_p2, n, &pts, s, d_st, max_iters, B.total, 1))) return rc;
    if (use_table) MLPL_HIP_TRY(hipMemcpyAsync(d_T, h_T, (size_t)(n + 1) * 4, hipMemcpyHostToDevice, s));

    if (!ctx->rand_cache) ctx->rand_cache = new RandCache();
    RandCursor rng(static_cast<RandCache *>(ctx->rand_cache), seed, (size_t)ctx->opt_rand_cache_max);
    const FastMod fmod_n(n);
    const double log_num = std::log(std::max(1. - std::min(std::max(confidence, 0.), 1.), DBL_MIN));  // update_num_iters' numerator
    // Device-side sampling for large passes (see draw_scan_kernel): the raw stream of the seed on the device, a control block behind it.
    // (eligible: enough hypotheses to be worth three launches, and few enough samples that redraw -- ~ 56 / n of the window's positions --
    // for the candidate list)
    bool dev_draw = ctx->opt_ransac_device_draw != 0 && !ctx->ransac_force_host_draw && n >= 64 && chunk_cap >= 4096 &&
                    (double)chunk_cap * 60.0 / n < 0.75 * kDrawCandCap;
    const int32_t *d_raw = nullptr;
    DrawCtl *d_ctl = nullptr;
    DrawCand *d_cand = nullptr;
    int raw_len = 0;
    const double redraw = std::min(5.0, 40.0 / n);  // window margin per sample: ~3x the expected redraws (10 / n repeats x ~1.1 draws)
    if (dev_draw) {
        const size_t need = (size_t)((double)max_iters * (5.0 + redraw)) + 4096;
        RandCache *rc_ = static_cast<RandCache *>(ctx->rand_cache);
        if (need > kRandCacheMax || need > (size_t)(ctx->opt_rand_cache_max > 0 ? ctx->opt_rand_cache_max : kRandCacheMax)) dev_draw = false;
        else {
            while (rc_->raw.size() < need) {
                rc_->gen.refill();
                rc_->raw.insert(rc_->raw.end(), rc_->gen.out, rc_->gen.out + 31);
                rc_->gen.pos = 31;
            }
            void *blk = nullptr;
            const size_t raw_bytes = (kRandCacheMax * 4 + 255) / 256 * 256;
            if ((rc = ws_get(ctx, WS_RAND_RAW, raw_bytes + sizeof(DrawCtl) + (size_t)kDrawCandCap * sizeof(DrawCand) + 256, &blk))) return rc;
            if (ctx->rand_dev_ptr != blk || ctx->rand_dev_seed != seed) ctx->rand_dev_ptr = blk, ctx->rand_dev_seed = seed, ctx->rand_dev_len = 0;
            if (ctx->rand_dev_len < need) {  // (pageable source: the copy stages through the runtime; only when the seed changes or the run is longer)
                MLPL_HIP_TRY(hipMemcpyAsync((int32_t *)blk + ctx->rand_dev_len, rc_->raw.data() + ctx->rand_dev_len, (need - ctx->rand_dev_len) * 4,
                                            hipMemcpyHostToDevice, s));
                ctx->rand_dev_len = need;
            }
            d_raw = (const int32_t *)blk;
            raw_len = (int)ctx->rand_dev_len;
            d_ctl = (DrawCtl *)((char *)blk + raw_bytes);
            d_cand = (DrawCand *)(d_ctl + 1);
            MLPL_HIP_TRY(hipMemsetAsync(d_ctl, 0, 32, s));  // position 0, no overflow, empty lists
        }
    }
    ReplayState cur = init;
    for (int base = 0; base < max_iters; base += chunk_cap) {
        const int cnt = std::min(chunk_cap, std::min(max_iters, cur.niters) - base);
        if (cnt <= 0) break;
        if (base > 0) {
            MLPL_HIP_TRY(hipStreamSynchronize(s));  // the pinned sample buffer is being reused
            MLPL_HIP_TRY(hipMemsetAsync(B.total, 0, 4, s));  // (the first pass's counter was zeroed by pack_points_kernel)
        }
        // The glibc-stream sample table is drawn on the host into pinned, device-mapped memory that the solver reads in place
        // (20 bytes per wave over PCIe, no staging copy in the stream).  The host draws ~9 ns per sample, the device solves 13-25 ns
        // per sample: a large pass runs in three slices (1024, up to 8192, the rest) so that the device starts after ~10 us of drawing
        // and always finds the next slice ready; up to 4096 hypotheses are one slice (two small solver launches would cost more than
        // the wait).
        const int point_splits = score_point_splits(n, cnt * 10, !lazy, ctx->opt_ransac_count_tiles);
        const int ev_cap = ctx->opt_ransac_event_cap > 0 ? std::min(ctx->opt_ransac_event_cap, kMaxScanEvents) : kMaxScanEvents;
        // Large passes: the root kernel of a slice goes to the helper stream, so it runs beside the elimination kernel of the next
        // slice (both are latency-bound at these sizes and leave most issue slots idle; unlike the counting kernel, which saturates
        // the vector units and gains nothing from company -- DESIGN section 5).
        // (samples drawn on the device: nothing paces the solver, one elimination and one root launch -- 52 + 56 us for 20 000 hypotheses --
        //  beat the four overlapped slice pairs, whose small launches are latency-bound: 177 us)
        // Round 5 (option ransac_dev_split, per mille of the pass in its FIRST of two slices; 0 = one slice, the default): with the solver
        // polish the root kernel of 20 000 hypotheses is 3334 waves for 3072 slots (89 us against the elimination's 59), which suggested two
        // slices again -- the second slice's elimination beside the first slice's roots, no root launch overflowing the chip.  Measured
        // (tools/c3_opt_ab.py, same process, alternating, identical results): one slice 0.483 ms per C3 call, 40 / 50 / 60 / 70 % in the
        // first slice 0.508 / 0.500 / 0.500 / 0.511 ms.  Not enabled.
        const int dev_split = (dev_draw && cnt > 8192 && ctx->opt_ransac_overlap != 0) ? ctx->opt_ransac_dev_split : 0;
        const bool overlap = cnt > 4096 && ctx->opt_ransac_overlap != 0 && (!dev_draw || dev_split > 0);
        const int32_t *d_samples = d_samples_mapped;
        if (dev_draw) {
            const int window = (int)std::min<long long>((long long)((double)cnt * (5.0 + redraw)) + 256, INT32_MAX / 2);
            hipLaunchKernelGGL(draw_scan_kernel, dim3((window + 255) / 256), dim3(256), 0, s, d_raw, raw_len, n, window, d_ctl, d_cand);
            hipLaunchKernelGGL(draw_chain_kernel, dim3(1), dim3(64), 0, s, raw_len, cnt, window, d_ctl, (const DrawCand *)d_cand, B.samples);
            hipLaunchKernelGGL(draw_fill_kernel, dim3((cnt + 255) / 256), dim3(256), 0, s, d_raw, raw_len, n, cnt, (const DrawCtl *)d_ctl, B.samples);
            d_samples = B.samples;
        }
        int slice_no = 0;
        for (int off = 0; off < cnt; ++slice_no) {
            // slices of a large pass: 1024 first (the device starts at once), a small last one (its root kernel is the exposed tail
            // of the solver chain: ~50 us for 2048 hypotheses, ~90 us for 12000), the rest in two halves
            int m = cnt;
            if (cnt > 4096 && !dev_draw) {
                const int last = std::min(2048, cnt / 8), mid = cnt - 1024 - last;
                m = slice_no == 0 ? 1024 : slice_no == 1 ? mid / 2 : slice_no == 2 ? mid - mid / 2 : last;
            } else if (dev_split > 0) {
                const int first = std::max(6, std::min(cnt - 6, (int)((long long)cnt * dev_split / 1000) / 6 * 6));
                m = slice_no == 0 ? first : cnt - first;
            }
            if (!dev_draw)
                for (int i = off; i < off + m; ++i) draw_sample(rng, fmod_n, &h_samples[(size_t)i * 5]);
            prof_mark(ctx, MLPL_PROF_SOLVE_5PT, 0, s);
            // (the hand-over records are indexed from the slice's first sample: each slice gets its own part of the buffer, the root
            // kernel of slice i and the elimination kernel of slice i+1 run side by side)
            launch_solve5pt(ctx, m, s, d_p1, d_p2, d_samples, off, off + m, B.recs + off);
            hipStream_t sr = overlap ? ctx->aux_stream[slice_no & 1] : s;  // two helper streams: consecutive root kernels overlap too
            if (overlap) {  // everything the root kernel reads is complete once this event fires (the first one also covers the setup)
                MLPL_HIP_TRY(hipEventRecord(ctx->aux_ev[slice_no], s));
                MLPL_HIP_TRY(hipStreamWaitEvent(sr, ctx->aux_ev[slice_no], 0));
            }
            MLPL_LAUNCH_ROOTS(ctx->opt_solver_polish, dim3((m + kHypPerWave - 1) / kHypPerWave), sr, (const PolyRec *)(B.recs + off), off,
                              off + m, B.E_tab, B.n_models, B.dense_E, B.dense_id, B.total, point_splits > 1 ? B.good : (int32_t *)nullptr);
            prof_mark(ctx, MLPL_PROF_SOLVE_5PT, 1, s);
            off += m;
        }
        if (overlap) {  // join: the counting pass needs every root kernel
            for (int h = 0; h < 2; ++h) {
                MLPL_HIP_TRY(hipEventRecord(ctx->aux_ev[6 + h], ctx->aux_stream[h]));
                MLPL_HIP_TRY(hipStreamWaitEvent(s, ctx->aux_ev[6 + h], 0));
            }
        }
        prof_mark(ctx, MLPL_PROF_SCORE, 0, s);
        prof_mark(ctx, MLPL_PROF_COUNT, 0, s);
        launch_score(s, (const double4 *)pts, n, (const double *)B.dense_E, (const int32_t *)B.dense_id, (const int32_t *)B.total, 0,
                     cnt * 10, thresh2, B.good, B.esum, !lazy, qmax, point_splits, ctx->opt_ransac_f32_filter != 0, ctx->opt_ransac_count_mpl,
                     ctx->opt_ransac_count_defer != 0, ctx->opt_ransac_count_threads, ctx->opt_ransac_count_wpe);
        prof_mark(ctx, MLPL_PROF_COUNT, 1, s);
        if (lazy) {
            // error sums only for the models that can still win (ties on the inlier count are decided by them)
            const bool sep = cnt > 2048;  // many hypotheses: per-hypothesis maxima in a grid-wide pass first
            if (sep)
                hipLaunchKernelGGL(hyp_max_kernel, dim3((cnt + 255) / 256), dim3(256), 0, s, (const int32_t *)B.n_models,
                                   (const int32_t *)B.good, cnt, B.hgood);
            hipLaunchKernelGGL(candidate_kernel, dim3(1), dim3(1024), 0, s, (const int32_t *)B.n_models, (const int32_t *)B.good,
                               sep ? (const int32_t *)B.hgood : (const int32_t *)nullptr, cnt, cur.maxGood, B.cand, B.cand_count, ev_cap);
            hipLaunchKernelGGL((score_models_block_kernel<false, true>), dim3(256), dim3(1024), (size_t)((n + 3) / 4 * 4) * sizeof(float), s,
                               (const double4 *)pts, n, (const double *)B.E_tab, (const int32_t *)nullptr, (const int32_t *)B.cand_count, 0,
                               thresh2, qmax, (int32_t *)nullptr, B.esum, (const int32_t *)B.cand);
        }
        prof_mark(ctx, MLPL_PROF_SCORE, 1, s);
        hipLaunchKernelGGL(hyp_best_kernel, dim3((cnt + 255) / 256), dim3(256), 0, s, (const int32_t *)B.n_models,
                           (const int32_t *)B.good, (const double *)B.esum, cnt, B.hgood, B.hsum, B.hslot);
        hipLaunchKernelGGL(replay_kernel, dim3(1), dim3(1024), 0, s, (const int32_t *)B.hgood, (const double *)B.hsum,
                           (const int32_t *)B.hslot, (const double *)B.E_tab, cnt, (const int32_t *)d_T, n, (long long)base * 10,
                           (const int32_t *)B.total, d_st, log_num, ev_cap);
        MLPL_HIP_TRY(hipGetLastError());
        if (base + chunk_cap < max_iters) {  // more chunks may follow: the host needs niters / stop to size the next one
            MLPL_HIP_TRY(hipMemcpyAsync(h_st, d_st, sizeof(ReplayState), hipMemcpyDeviceToHost, s));
            MLPL_HIP_TRY(hipStreamSynchronize(s));
            cur = *h_st;
            if (cur.stop) break;
        }
    }

    // mask of the model held (no-op result when none was found; checked below)
    hipLaunchKernelGGL(inlier_mask_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const double4 *)pts, n,
                       (const double *)d_st->E, thresh2, d_mask);
    if (refit) {
        // modelest.cpp:420-464: solve on all inliers, keep a refit model if it scores better
        double *d_Etab = d_rf;                       // 90 doubles
        int32_t *d_nm = (int32_t *)(d_rf + 96);      // 1 int
        int32_t *d_good = d_nm + 4;                  // 10 ints
        double *d_es = d_rf + 112;                   // 10 doubles
        double *d_gram = d_rf + 144;                 // [gblocks][45]
        hipLaunchKernelGGL(gram_kernel, dim3(gblocks), dim3(256), 0, s, (const double4 *)pts, (const uint8_t *)d_mask, n, d_gram);
        hipLaunchKernelGGL(refit_solve_kernel, dim3(1), dim3(64), 0, s, (const double *)d_gram, gblocks, B.recs);
        MLPL_LAUNCH_ROOTS(ctx->opt_solver_polish, dim3(1), s, (const PolyRec *)B.recs, 0, 1, d_Etab, d_nm, (double *)nullptr,
                          (int32_t *)nullptr, (int32_t *)nullptr, (int32_t *)nullptr);
        launch_score(s, (const double4 *)pts, n, (const double *)d_Etab, nullptr, (const int32_t *)d_nm, 0, 10, thresh2, d_good, d_es);
        hipLaunchKernelGGL(refit_decide_kernel, dim3(1), dim3(64), 0, s, (const int32_t *)d_nm, (const int32_t *)d_good,
                           (const double *)d_es, (const double *)d_Etab, d_st);
        // re-evaluating the mask with the (possibly unchanged) model held is idempotent
        hipLaunchKernelGGL(inlier_mask_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const double4 *)pts, n,
                           (const double *)d_st->E, thresh2, d_mask);
    }
    MLPL_HIP_TRY(hipGetLastError());
    MLPL_HIP_TRY(hipMemcpyAsync(h_st, d_st, sizeof(ReplayState), hipMemcpyDeviceToHost, s));
    *h_ovf = 0;
    if (dev_draw) MLPL_HIP_TRY(hipMemcpyAsync(h_ovf, &d_ctl->overflow, 4, hipMemcpyDeviceToHost, s));
    MLPL_HIP_TRY(hipStreamSynchronize(s));  // the single host hop of the call (per 32768-iteration chunk)
    const ReplayState fin = *h_st;
    if (dev_draw) {  // the device-side sampling ran out of window / list / stream somewhere: redo with the host drawing the table
        if (*h_ovf) {
            ctx->ransac_force_host_draw = 1;
            const int rc2 = mlpl_ransac_essential_dev(ctx, d_p1, d_p2, n, thresh, confidence, max_iters, refit, seed, E, d_mask, n_inliers,
                                                      iters_used, stream);
            ctx->ransac_force_host_draw = 0;
            ctx->ransac_draw_fallbacks++;
            return rc2;
        }
    }
    if (!use_table) {
        // every iteration bound the device used must be what the CPU path's libm gives; otherwise redo the call on a host table
        bool ok = fin.t_count <= kTUsedMax;
        for (int i = 0; ok && i < fin.t_count; ++i)
            ok = fin.t_val[i] == update_num_iters(confidence, (double)(n - fin.t_g[i]) / n, 5, INT32_MAX);
        if (!ok) {
            ctx->ransac_force_table = 1;
            const int rc2 = mlpl_ransac_essential_dev(ctx, d_p1, d_p2, n, thresh, confidence, max_iters, refit, seed, E, d_mask, n_inliers,
                                                      iters_used, stream);
            ctx->ransac_force_table = 0;
            ctx->ransac_table_fallbacks++;
            return rc2;
        }
    }
    ctx->last_ransac_models = fin.models_scored;
    ctx->last_ransac_iters = fin.iter;
    ctx->last_ransac_dev_draw = dev_draw ? 1 : 0;
    if (iters_used) *iters_used = fin.iter;
    if (fin.maxGood <= 0) {
        set_error("mlpl_ransac_essential: no model found");
        return MLPL_E_FAILED;
    }
    if (refit && fin.refit_models <= 0) {
        // the reference returns `result` (still false) when the refit kernel yields no model: modelest.cpp:442-443
        set_error("mlpl_ransac_essential: refit produced no model (reference returns false here)");
        return MLPL_E_FAILED;
    }
    std::memcpy(E, fin.E, 72);
    if (n_inliers) *n_inliers = fin.maxGood;
    return MLPL_OK;
}

int mlpl_ransac_essential(mlpl_ctx *ctx, const double *p1, const double *p2, int n, double thresh, double confidence,
                          int max_iters, int refit, uint32_t seed, double E[9], uint8_t *mask, int *n_inliers,
                          int *iters_used) {
    if (!ctx || !p1 || !p2 || !E || !mask || n < 6) {
        set_error("mlpl_ransac_essential: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    void *dp1, *dp2, *dmask;
    int rc;
    if ((rc = ws_get(ctx, WS_AUX0, (size_t)n * 16, &dp1))) return rc;
    if ((rc = ws_get(ctx, WS_AUX1, (size_t)n * 16, &dp2))) return rc;
    if ((rc = ws_get(ctx, WS_MATCH, (size_t)n, &dmask))) return rc;
    MLPL_HIP_TRY(hipMemcpyAsync(dp1, p1, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    MLPL_HIP_TRY(hipMemcpyAsync(dp2, p2, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    rc = mlpl_ransac_essential_dev(ctx, (const double *)dp1, (const double *)dp2, n, thresh, confidence, max_iters, refit,
                                   seed, E, (uint8_t *)dmask, n_inliers, iters_used, ctx->stream);
    if (rc) return rc;
    MLPL_HIP_TRY(hipMemcpy(mask, dmask, (size_t)n, hipMemcpyDeviceToHost));
    return MLPL_OK;
}

int mlpl_median_models(mlpl_ctx *ctx, const double *p1, const double *p2, int n, const double *E, int n_models, double *median) {
    if (!ctx || !p1 || !p2 || !E || !median || n < 1 || n_models < 0) {
        set_error("mlpl_median_models: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    if (n_models == 0) return MLPL_OK;
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    void *dp1, *dp2, *dE, *dmed;
    int rc;
    if ((rc = ws_get(ctx, WS_AUX0, (size_t)n * 16, &dp1))) return rc;
    if ((rc = ws_get(ctx, WS_AUX1, (size_t)n * 16, &dp2))) return rc;
    if ((rc = ws_get(ctx, WS_AUX5, (size_t)n_models * 72, &dE))) return rc;
    if ((rc = ws_get(ctx, WS_AUX6, (size_t)n_models * 8, &dmed))) return rc;
    MLPL_HIP_TRY(hipMemcpyAsync(dp1, p1, (size_t)n * 16, hipMemcpyHostToDevice, s));
    MLPL_HIP_TRY(hipMemcpyAsync(dp2, p2, (size_t)n * 16, hipMemcpyHostToDevice, s));
    MLPL_HIP_TRY(hipMemcpyAsync(dE, E, (size_t)n_models * 72, hipMemcpyHostToDevice, s));
    double4 *pts;
    if ((rc = pack_points(ctx, (const double *)dp1, (const double *)dp2, n, &pts, s))) return rc;
    if (n <= kScoreBlockMaxN)
        hipLaunchKernelGGL(median_kernel<true>, dim3(n_models), dim3(256), (size_t)n * 4, s, (const double4 *)pts, n, (const double *)dE,
                           (const int32_t *)nullptr, (const int32_t *)nullptr, n_models, (double *)dmed);
    else
        hipLaunchKernelGGL(median_kernel<false>, dim3(n_models), dim3(256), 0, s, (const double4 *)pts, n, (const double *)dE,
                           (const int32_t *)nullptr, (const int32_t *)nullptr, n_models, (double *)dmed);
    MLPL_HIP_TRY(hipGetLastError());
    MLPL_HIP_TRY(hipMemcpyAsync(median, dmed, (size_t)n_models * 8, hipMemcpyDeviceToHost, s));
    MLPL_HIP_TRY(hipStreamSynchronize(s));
    return MLPL_OK;
}

int mlpl_lmeds_essential_dev(mlpl_ctx *ctx, const double *d_p1, const double *d_p2, int n, double confidence, int max_iters,
                             uint32_t seed, double E[9], uint8_t *d_mask, int *n_inliers, double *min_median, void *stream) {
    if (!ctx || !d_p1 || !d_p2 || !E || !d_mask || n < 6 || max_iters < 1 || !(confidence > 0 && confidence < 1)) {
        set_error("mlpl_lmeds_essential: bad arguments (n=%d max_iters=%d); n must exceed the 5 model points", n, max_iters);
        return MLPL_E_BAD_INPUT;
    }
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = pick_stream(ctx, stream);
    if (n_inliers) *n_inliers = 0;
    if (min_median) *min_median = DBL_MAX;

    // modelest.cpp:506-508 (host libm, as the reference)
    const double outlierRatio = 0.45;
    int niters = (int)std::round(std::log(1. - confidence) / std::log(1. - std::pow(1. - outlierRatio, 5.0)));
    niters = std::min(std::max(niters, 3), max_iters);

    double4 *pts;
    int rc = pack_points(ctx, d_p1, d_p2, n, &pts, s);
    if (rc) return rc;
    RansacBuffers B;
    if ((rc = alloc_ransac(ctx, niters, B))) return rc;
    void *dsm;
    if ((rc = ws_get(ctx, WS_AUX2, 4096, &dsm))) return rc;
    LmedsState *d_st = (LmedsState *)dsm;
    void *pin;
    if ((rc = pinned_get(ctx, (size_t)niters * 20 + 1024, &pin))) return rc;
    LmedsState *h_st = (LmedsState *)pin;
    int32_t *h_samples = (int32_t *)((char *)pin + 512);
    int32_t *d_samples_mapped = nullptr;
    MLPL_HIP_TRY(hipHostGetDevicePointer((void **)&d_samples_mapped, h_samples, 0));

    if (!ctx->rand_cache) ctx->rand_cache = new RandCache();
    RandCursor rng(static_cast<RandCache *>(ctx->rand_cache), seed, (size_t)ctx->opt_rand_cache_max);
    const FastMod fmod_n(n);
    for (int i = 0; i < niters; ++i) draw_sample(rng, fmod_n, &h_samples[(size_t)i * 5]);
    MLPL_HIP_TRY(hipMemsetAsync(B.total, 0, 4, s));
    prof_mark(ctx, MLPL_PROF_SOLVE_5PT, 0, s);
    launch_solve5pt(ctx, niters, s, d_p1, d_p2, (const int32_t *)d_samples_mapped, 0, niters, B.recs);
    MLPL_LAUNCH_ROOTS(ctx->opt_solver_polish, dim3((niters + kHypPerWave - 1) / kHypPerWave), s, (const PolyRec *)B.recs, 0, niters, B.E_tab,
                      B.n_models, B.dense_E, B.dense_id, B.total, (int32_t *)nullptr);
    prof_mark(ctx, MLPL_PROF_SOLVE_5PT, 1, s);
    prof_mark(ctx, MLPL_PROF_SCORE, 0, s);
    if (n <= kScoreBlockMaxN)
        hipLaunchKernelGGL(median_kernel<true>, dim3(niters * 10), dim3(256), (size_t)n * 4, s, (const double4 *)pts, n,
                           (const double *)B.dense_E, (const int32_t *)B.dense_id, (const int32_t *)B.total, 0, B.esum);
    else
        hipLaunchKernelGGL(median_kernel<false>, dim3(niters * 10), dim3(256), 0, s, (const double4 *)pts, n, (const double *)B.dense_E,
                           (const int32_t *)B.dense_id, (const int32_t *)B.total, 0, B.esum);
    prof_mark(ctx, MLPL_PROF_SCORE, 1, s);
    hipLaunchKernelGGL(lmeds_argmin_kernel, dim3(1), dim3(1024), 0, s, (const int32_t *)B.n_models, (const double *)B.esum,
                       (const double *)B.E_tab, niters, d_st);
    MLPL_HIP_TRY(hipGetLastError());
    MLPL_HIP_TRY(hipMemcpyAsync(h_st, d_st, sizeof(LmedsState), hipMemcpyDeviceToHost, s));
    MLPL_HIP_TRY(hipStreamSynchronize(s));
    ctx->last_ransac_iters = niters;
    if (h_st->best < 0 || !(h_st->minMedian < DBL_MAX)) {
        set_error("mlpl_lmeds_essential: no model found");
        return MLPL_E_FAILED;
    }
    const double minMedian = h_st->minMedian;
    // modelest.cpp:555-559
    double sigma = 2.5 * 1.4826 * (1 + 5. / (n - 5)) * std::sqrt(minMedian);
    sigma = std::max(sigma, 0.001);
    hipLaunchKernelGGL(inlier_mask_count_kernel, dim3((n + 255) / 256), dim3(256), 0, s, (const double4 *)pts, n,
                       (const double *)d_st->E, sigma * sigma, d_mask, &d_st->inliers);
    MLPL_HIP_TRY(hipGetLastError());
    std::memcpy(E, h_st->E, 72);
    if (min_median) *min_median = minMedian;
    MLPL_HIP_TRY(hipMemcpyAsync(h_st, d_st, sizeof(LmedsState), hipMemcpyDeviceToHost, s));
    MLPL_HIP_TRY(hipStreamSynchronize(s));
    if (n_inliers) *n_inliers = h_st->inliers;
    if (h_st->inliers < 5) {  // modelest.cpp:561: result = count >= modelPoints
        set_error("mlpl_lmeds_essential: fewer than 5 inliers under the LMedS sigma");
        return MLPL_E_FAILED;
    }
    return MLPL_OK;
}

int mlpl_lmeds_essential(mlpl_ctx *ctx, const double *p1, const double *p2, int n, double confidence, int max_iters,
                         uint32_t seed, double E[9], uint8_t *mask, int *n_inliers, double *min_median) {
    if (!ctx || !p1 || !p2 || !E || !mask || n < 6) {
        set_error("mlpl_lmeds_essential: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    void *dp1, *dp2, *dmask;
    int rc;
    if ((rc = ws_get(ctx, WS_AUX0, (size_t)n * 16, &dp1))) return rc;
    if ((rc = ws_get(ctx, WS_AUX1, (size_t)n * 16, &dp2))) return rc;
    if ((rc = ws_get(ctx, WS_MATCH, (size_t)n, &dmask))) return rc;
    MLPL_HIP_TRY(hipMemcpyAsync(dp1, p1, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    MLPL_HIP_TRY(hipMemcpyAsync(dp2, p2, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    rc = mlpl_lmeds_essential_dev(ctx, (const double *)dp1, (const double *)dp2, n, confidence, max_iters, seed, E,
                                  (uint8_t *)dmask, n_inliers, min_median, ctx->stream);
    // the mask is meaningful whenever a model was found, also when the inlier count fails the final test
    if (rc && rc != MLPL_E_FAILED) return rc;
    MLPL_HIP_TRY(hipMemcpy(mask, dmask, (size_t)n, hipMemcpyDeviceToHost));
    return rc;
}

int mlpl_arrsac_essential_dev(mlpl_ctx *ctx, const double *d_p1, const double *d_p2, int n, double thresh, int refine,
                              uint64_t rng_state[2], double E[9], uint8_t *d_mask, int *n_inliers, void *stream) {
    if (!ctx || !d_p1 || !d_p2 || !E || !d_mask || !rng_state || n < 6 || !(thresh > 0)) {
        set_error("mlpl_arrsac_essential_dev: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    return arrsac_essential_dev(ctx, d_p1, d_p2, n, thresh, refine, rng_state, E, d_mask, n_inliers, pick_stream(ctx, stream));
}

int mlpl_arrsac_essential(mlpl_ctx *ctx, const double *p1, const double *p2, int n, double thresh, int refine, uint64_t rng_state[2],
                          double E[9], uint8_t *mask, int *n_inliers) {
    if (!ctx || !p1 || !p2 || !E || !mask || !rng_state || n < 6 || !(thresh > 0)) {
        set_error("mlpl_arrsac_essential: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    void *dp1, *dp2, *dmask;
    int rc;
    if ((rc = ws_get(ctx, WS_AUX0, (size_t)n * 16, &dp1))) return rc;
    if ((rc = ws_get(ctx, WS_AUX1, (size_t)n * 16, &dp2))) return rc;
    if ((rc = ws_get(ctx, WS_MATCH, (size_t)n, &dmask))) return rc;
    MLPL_HIP_TRY(hipMemcpyAsync(dp1, p1, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    MLPL_HIP_TRY(hipMemcpyAsync(dp2, p2, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    rc = arrsac_essential_dev(ctx, (const double *)dp1, (const double *)dp2, n, thresh, refine, rng_state, E, (uint8_t *)dmask, n_inliers,
                              ctx->stream);
    // the mask is meaningful whenever a hypothesis was found, also when its inlier count fails the final test
    if (rc && !(rc == MLPL_E_FAILED && n_inliers && *n_inliers > 0)) return rc;
    MLPL_HIP_TRY(hipMemcpy(mask, dmask, (size_t)n, hipMemcpyDeviceToHost));
    return rc;
}

int mlpl_arrsac_essential_batch_dev(mlpl_ctx *ctx, int n_problems, const double *d_p1, const double *d_p2, int stride, const int32_t *counts, double thresh,
                                    int refine, uint64_t *rng_states, double *E, uint8_t *d_masks, int32_t *n_inliers, int32_t *status, void *stream) {
    if (!ctx || n_problems < 0 || !d_p1 || !d_p2 || stride < 1 || !counts || !rng_states || !E || !d_masks || !status || !(thresh > 0)) {
        set_error("mlpl_arrsac_essential_batch_dev: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    for (int b = 0; b < n_problems; ++b)
        if (counts[b] < 6 || counts[b] > stride) {
            set_error("mlpl_arrsac_essential_batch_dev: problem %d has %d correspondences (6 ... stride %d)", b, counts[b], stride);
            return MLPL_E_BAD_INPUT;
        }
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    try {
        return arrsac_essential_batch_dev(ctx, n_problems, d_p1, d_p2, stride, counts, thresh, refine, rng_states, E, d_masks, n_inliers, status,
                                          pick_stream(ctx, stream));
    } catch (const std::bad_alloc &) {
        set_error("mlpl_arrsac_essential_batch_dev: out of host memory");
        return MLPL_E_NOMEM;
    }
}

int mlpl_arrsac_sample_models(mlpl_ctx *ctx, const double *p1, const double *p2, int n, const int32_t *idx, int m, int kind, double thresh,
                              double *E_out, int32_t *n_models, uint8_t *valid) {
    if (!ctx || !p1 || !p2 || !idx || !E_out || !n_models || !valid || n < 5 || m < 5 || m > kArrMaxSample || (kind != 0 && kind != 1) ||
        (kind == 0 && m > 7) || (kind == 1 && m < 8) || !(thresh > 0)) {
        set_error("mlpl_arrsac_sample_models: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    for (int i = 0; i < m; ++i)
        if (idx[i] < 0 || idx[i] >= n) {
            set_error("mlpl_arrsac_sample_models: sample index out of range");
            return MLPL_E_BAD_INPUT;
        }
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    void *dp1, *dp2;
    int rc;
    if ((rc = ws_get(ctx, WS_AUX0, (size_t)n * 16, &dp1))) return rc;
    if ((rc = ws_get(ctx, WS_AUX1, (size_t)n * 16, &dp2))) return rc;
    MLPL_HIP_TRY(hipMemcpyAsync(dp1, p1, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    MLPL_HIP_TRY(hipMemcpyAsync(dp2, p2, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    return arrsac_sample_models(ctx, (const double *)dp1, (const double *)dp2, n, idx, m, kind, thresh, E_out, n_models, valid, ctx->stream);
}

int mlpl_robust_essential_refine(mlpl_ctx *ctx, const double *p1, const double *p2, int n, const uint8_t *mask, const double E_init[9],
                                 double th, double E_refined[9], int info[2]) {
    if (!ctx || !p1 || !p2 || !E_init || !E_refined || n < 1 || !(th > 0)) {
        set_error("mlpl_robust_essential_refine: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = ctx->stream;
    void *dp1, *dp2, *dmask, *dE;
    int rc;
    if ((rc = ws_get(ctx, WS_AUX0, (size_t)n * 16, &dp1))) return rc;
    if ((rc = ws_get(ctx, WS_AUX1, (size_t)n * 16, &dp2))) return rc;
    if ((rc = ws_get(ctx, WS_MATCH, (size_t)n, &dmask))) return rc;
    if ((rc = ws_get(ctx, WS_AUX7, 256, &dE))) return rc;
    MLPL_HIP_TRY(hipMemcpyAsync(dp1, p1, (size_t)n * 16, hipMemcpyHostToDevice, s));
    MLPL_HIP_TRY(hipMemcpyAsync(dp2, p2, (size_t)n * 16, hipMemcpyHostToDevice, s));
    if (mask) MLPL_HIP_TRY(hipMemcpyAsync(dmask, mask, (size_t)n, hipMemcpyHostToDevice, s));
    else MLPL_HIP_TRY(hipMemsetAsync(dmask, 1, (size_t)n, s));
    MLPL_HIP_TRY(hipMemcpyAsync(dE, E_init, 72, hipMemcpyHostToDevice, s));
    double4 *pts = nullptr;
    if ((rc = pack_points(ctx, (const double *)dp1, (const double *)dp2, n, &pts, s))) return rc;
    double *d_E = (double *)dE;
    int32_t *d_info = (int32_t *)(d_E + 18);
    {
        Launcher L;
        L.s = s;
        ArrRefineArgs ra{{1, 1}, (const double4 *)pts, (const uint8_t *)dmask, n, (const double *)d_E, th, d_E + 9, d_info,
                         (ctx->opt_arrsac_refine_warm_start ? 1 : 0) | (ctx->opt_eig_inverse_iteration ? 2 : 0)};
        L.launch(HK_ARR_REFINE, ra);
    }
    MLPL_HIP_TRY(hipGetLastError());
    double h[20];
    MLPL_HIP_TRY(hipMemcpyAsync(h, d_E + 9, 88, hipMemcpyDeviceToHost, s));
    MLPL_HIP_TRY(hipStreamSynchronize(s));
    std::memcpy(E_refined, h, 72);
    if (info) std::memcpy(info, h + 9, 8);
    return MLPL_OK;
}

int mlpl_pair_pose_batch_dev(mlpl_ctx *ctx, int n_pairs, const uint8_t *d_q, int nq, const uint8_t *d_t, int nt, int nbytes, const float *d_kp1,
                             const float *d_kp2, const double K0[4], const double K1[4], double thresh, int max_iters, double confidence,
                             int refit, const uint32_t *seeds, double dist, mlpl_pair_result *out, mlpl_dmatch *d_matches_out, void *stream) {
    if (!ctx || !d_q || !d_t || !d_kp1 || !d_kp2 || !K0 || !K1 || !out || !seeds || n_pairs < 1 || nq < 1 || nt < 2 || nbytes < 1 || max_iters < 1 ||
        !(thresh > 0)) {
        set_error("mlpl_pair_pose_batch_dev: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = pick_stream(ctx, stream);
    const int per = ctx->opt_pair_batch > 0 ? ctx->opt_pair_batch : kBatchPairsPerCall;
    std::memset(ctx->last_batch_stats, 0, sizeof(ctx->last_batch_stats));
    if (refit) {  // the refit step is not batched: the single-pair pipeline, pair by pair (matches are not exported on this path)
        if (d_matches_out) {
            set_error("mlpl_pair_pose_batch_dev: d_matches_out needs refit = 0");
            return MLPL_E_UNSUPPORTED;
        }
        for (int b = 0; b < n_pairs; ++b) {
            const int rc = mlpl_pair_pose_dev(ctx, d_q + (size_t)b * nq * nbytes, nq, d_t + (size_t)b * nt * nbytes, nt, nbytes, d_kp1 + (size_t)b * nq * 2,
                                              d_kp2 + (size_t)b * nt * 2, K0, K1, thresh, max_iters, confidence, refit, seeds[b], dist, &out[b], stream);
            if (rc) return rc;
        }
        return MLPL_OK;
    }
    // internal batches of `per` pairs, one after the other.  (Measured and not kept: alternate batches on two host threads with a
    // context and stream each, to cover one batch's host hops with the other's kernels -- 15.7 -> 15.4 ms per 512 pairs: the device is
    // busy already.)
    long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int at = 0; at < n_pairs; at += per) {
        const int B = std::min(per, n_pairs - at);
        const int rc = pair_pose_batch_dev(ctx, B, d_q + (size_t)at * nq * nbytes, nq, d_t + (size_t)at * nt * nbytes, nt, nbytes,
                                           d_kp1 + (size_t)at * nq * 2, d_kp2 + (size_t)at * nt * 2, K0, K1, thresh, max_iters, confidence, seeds + at,
                                           dist, out + at, d_matches_out ? d_matches_out + (size_t)at * nq : nullptr, s);
        if (rc) return rc;
        for (int i = 0; i < 8; ++i) acc[i] += ctx->last_batch_stats[i];
    }
    std::memcpy(ctx->last_batch_stats, acc, sizeof(acc));
    return MLPL_OK;
}

// Two (or more) batched calls in flight on one GPU, inside the library: lane l = (ctxs[l], streams[l]) takes the l-th contiguous share of
// the batch on a thread of its own (lane 0: the calling thread), so that one lane's host hops fall into the other's kernels.  The Python
// form of this (batch.BatchLanes: a ThreadPoolExecutor) left 2-3 ms of thread hand-off and interpreter-lock waits in one step of ~25
// (tools/lanes_tail_probe.py); here the caller makes ONE call and sleeps until every lane has returned.  Records do not depend on the
// lane count: every pair's result is a function of its inputs and its seed.
int mlpl_pair_pose_batch_lanes_dev(mlpl_ctx *const *ctxs, void *const *streams, int lanes, int n_pairs, const uint8_t *d_q, int nq, const uint8_t *d_t, int nt,
                                   int nbytes, const float *d_kp1, const float *d_kp2, const double K0[4], const double K1[4], double thresh, int max_iters,
                                   double confidence, const uint32_t *seeds, double dist, mlpl_pair_result *out, mlpl_dmatch *d_matches_out,
                                   double *lane_span_ms) {
    if (!ctxs || !streams || lanes < 1 || lanes > 8 || n_pairs < 1 || !seeds || !out) {
        set_error("mlpl_pair_pose_batch_lanes_dev: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    for (int l = 0; l < lanes; ++l) {
        if (!ctxs[l] || ctxs[l]->device != ctxs[0]->device) {
            set_error("mlpl_pair_pose_batch_lanes_dev: every lane needs a context of the same device");
            return MLPL_E_BAD_INPUT;
        }
        for (int m = 0; m < l; ++m)
            if (ctxs[m] == ctxs[l]) {  // a context serves one call at a time (its workspaces and pinned blocks are the call's)
                set_error("mlpl_pair_pose_batch_lanes_dev: lanes %d and %d share a context", m, l);
                return MLPL_E_BAD_INPUT;
            }
    }
    const int L = std::min(lanes, n_pairs);
    const auto t0 = std::chrono::steady_clock::now();
    std::vector<int> rcs((size_t)L, 0);
    std::vector<std::string> msgs((size_t)L);
    auto run_lane = [&](int l) {
        const int b0 = (int)((long long)n_pairs * l / L), b1 = (int)((long long)n_pairs * (l + 1) / L);
        const auto ts = std::chrono::steady_clock::now();
        try {
            rcs[l] = b1 > b0 ? mlpl_pair_pose_batch_dev(ctxs[l], b1 - b0, d_q + (size_t)b0 * nq * nbytes, nq, d_t + (size_t)b0 * nt * nbytes, nt, nbytes,
                                                        d_kp1 + (size_t)b0 * nq * 2, d_kp2 + (size_t)b0 * nt * 2, K0, K1, thresh, max_iters, confidence, 0, seeds + b0,
                                                        dist, out + b0, d_matches_out ? d_matches_out + (size_t)b0 * nq : nullptr, streams[l])
                             : MLPL_OK;
            if (rcs[l]) msgs[l] = mlpl_last_error();  // (the error text is thread-local)
            else if (hipStreamSynchronize(reinterpret_cast<hipStream_t>(streams[l])) != hipSuccess) rcs[l] = MLPL_E_HIP, msgs[l] = "lane stream synchronisation failed";
        } catch (...) {  // (nothing may unwind off a lane's thread)
            rcs[l] = MLPL_E_INTERNAL;
            try { msgs[l] = "mlpl_pair_pose_batch_lanes_dev: a lane ended with a C++ exception"; } catch (...) {}
        }
        if (lane_span_ms) {
            lane_span_ms[2 * l] = std::chrono::duration<double, std::milli>(ts - t0).count();
            lane_span_ms[2 * l + 1] = std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t0).count();
        }
    };
    {
        std::vector<std::thread> others;
        struct Join {  // whatever happens below, no joinable thread is destroyed
            std::vector<std::thread> &t;
            ~Join() {
                for (auto &x : t)
                    if (x.joinable()) x.join();
            }
        } join{others};
        try {
            others.reserve((size_t)L);
            for (int l = 1; l < L; ++l) others.emplace_back(run_lane, l);
        } catch (const std::exception &e) {  // a thread could not be started: the lanes that did start finish their shares, the call fails
            set_error("mlpl_pair_pose_batch_lanes_dev: %s", e.what());
            return MLPL_E_INTERNAL;
        }
        run_lane(0);
    }
    for (int l = 0; l < L; ++l)
        if (rcs[l]) {
            set_error("%s", msgs[l].c_str());
            return rcs[l];
        }
    return MLPL_OK;
}

int mlpl_sorted_match_idx(const mlpl_dmatch *matches, int n, uint32_t *sorted_idx) {
    if ((!matches || !sorted_idx) && n > 0) return MLPL_E_BAD_INPUT;
    if (n < 0) return MLPL_E_BAD_INPUT;
    try {
        sorted_match_idx(matches, n, sorted_idx);
    } catch (const std::bad_alloc &) {
        return MLPL_E_NOMEM;
    }
    return MLPL_OK;
}

static int usac_check_params(const mlpl_usac_params *P, int n, const char *who);
int mlpl_pair_pose_batch_usac_dev(mlpl_ctx *ctx, int n_pairs, const uint8_t *d_q, int nq, const uint8_t *d_t, int nt, int nbytes, const float *d_kp1,
                                  const float *d_kp2, const double K0[4], const double K1[4], const mlpl_usac_params *usac, int prosac,
                                  const uint32_t *seeds, double dist, mlpl_pair_result *out, mlpl_dmatch *d_matches_out, void *stream) {
    if (!ctx || !d_q || !d_t || !d_kp1 || !d_kp2 || !K0 || !K1 || !out || !seeds || !usac || n_pairs < 1 || nq < 1 || nt < 2 || nbytes < 1) {
        set_error("mlpl_pair_pose_batch_usac_dev: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    int rc;
    mlpl_usac_params chk = *usac;
    chk.sorted_idx = nullptr;
    if ((rc = usac_check_params(&chk, 0, "mlpl_pair_pose_batch_usac_dev"))) return rc;
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = pick_stream(ctx, stream);
    const int per = ctx->opt_pair_batch_seq > 0 ? ctx->opt_pair_batch_seq : kSeqBatchPairsPerCall;
    try {
        for (int at = 0; at < n_pairs; at += per) {
            const int B = std::min(per, n_pairs - at);
            rc = pair_pose_batch_usac_dev(ctx, B, d_q + (size_t)at * nq * nbytes, nq, d_t + (size_t)at * nt * nbytes, nt, nbytes, d_kp1 + (size_t)at * nq * 2,
                                          d_kp2 + (size_t)at * nt * 2, K0, K1, &chk, prosac, seeds + at, dist, out + at,
                                          d_matches_out ? d_matches_out + (size_t)at * nq : nullptr, s);
            if (rc) return rc;
        }
    } catch (const std::bad_alloc &) {
        set_error("mlpl_pair_pose_batch_usac_dev: out of host memory");
        return MLPL_E_NOMEM;
    }
    return MLPL_OK;
}

int mlpl_pair_pose_batch_arrsac_dev(mlpl_ctx *ctx, int n_pairs, const uint8_t *d_q, int nq, const uint8_t *d_t, int nt, int nbytes, const float *d_kp1,
                                    const float *d_kp2, const double K0[4], const double K1[4], double thresh, int refine, uint64_t *rng_states, double dist,
                                    mlpl_pair_result *out, mlpl_dmatch *d_matches_out, void *stream) {
    if (!ctx || !d_q || !d_t || !d_kp1 || !d_kp2 || !K0 || !K1 || !out || !rng_states || n_pairs < 1 || nq < 1 || nt < 2 || nbytes < 1 || !(thresh > 0)) {
        set_error("mlpl_pair_pose_batch_arrsac_dev: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = pick_stream(ctx, stream);
    const int per = ctx->opt_pair_batch_seq > 0 ? ctx->opt_pair_batch_seq : kSeqBatchPairsPerCall;
    try {
        for (int at = 0; at < n_pairs; at += per) {
            const int B = std::min(per, n_pairs - at);
            const int rc = pair_pose_batch_usac_dev(ctx, B, d_q + (size_t)at * nq * nbytes, nq, d_t + (size_t)at * nt * nbytes, nt, nbytes,
                                                    d_kp1 + (size_t)at * nq * 2, d_kp2 + (size_t)at * nt * 2, K0, K1, nullptr, 0, nullptr, dist, out + at,
                                                    d_matches_out ? d_matches_out + (size_t)at * nq : nullptr, s, thresh, refine, rng_states + 2 * (size_t)at);
            if (rc) return rc;
        }
    } catch (const std::bad_alloc &) {
        set_error("mlpl_pair_pose_batch_arrsac_dev: out of host memory");
        return MLPL_E_NOMEM;
    }
    return MLPL_OK;
}

int mlpl_ransac_essential_batch_dev(mlpl_ctx *ctx, int n_problems, const double *d_p1, const double *d_p2, int stride, const int32_t *counts, double thresh,
                                    int max_iters, double confidence, const uint32_t *seeds, int recover_pose, double dist, mlpl_pair_result *out,
                                    uint8_t *d_masks, void *stream) {
    if (!ctx || !d_p1 || !d_p2 || !counts || !seeds || !out || n_problems < 1 || stride < 1 || max_iters < 1 || !(thresh > 0)) {
        set_error("mlpl_ransac_essential_batch_dev: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    for (int b = 0; b < n_problems; ++b)
        if (counts[b] < 0 || counts[b] > stride) {
            set_error("mlpl_ransac_essential_batch_dev: counts[%d] = %d outside [0, stride = %d]", b, counts[b], stride);
            return MLPL_E_BAD_INPUT;
        }
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = pick_stream(ctx, stream);
    const int per = ctx->opt_pair_batch > 0 ? ctx->opt_pair_batch : kBatchPairsPerCall;
    std::memset(ctx->last_batch_stats, 0, sizeof(ctx->last_batch_stats));
    long long acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    const double K1_[4] = {1, 1, 0, 0};
    for (int at = 0; at < n_problems; at += per) {
        const int B = std::min(per, n_problems - at);
        const int rc = pair_pose_batch_dev(ctx, B, nullptr, stride, nullptr, stride, 0, nullptr, nullptr, K1_, K1_, thresh, max_iters, confidence, seeds + at, dist,
                                           out + at, nullptr, s, d_p1 + (size_t)at * stride * 2, d_p2 + (size_t)at * stride * 2, counts + at, recover_pose ? 1 : 0,
                                           d_masks ? d_masks + (size_t)at * stride : nullptr);
        if (rc) return rc;
        for (int i = 0; i < 8; ++i) acc[i] += ctx->last_batch_stats[i];
    }
    std::memcpy(ctx->last_batch_stats, acc, sizeof(acc));
    return MLPL_OK;
}

int mlpl_pair_batch_last_stats(mlpl_ctx *ctx, long long stats[8]) {
    if (!ctx || !stats) return MLPL_E_BAD_INPUT;
    std::memcpy(stats, ctx->last_batch_stats, sizeof(ctx->last_batch_stats));
    return MLPL_OK;
}

void mlpl_usac_default_params(mlpl_usac_params *p, double th) {
    if (!p) return;
    p->th = th, p->conf = 0.99, p->max_hyp = 50000, p->estimator = 0, p->refine = 0, p->seed = 1u;
    p->prosac_beta = 0.09, p->sprt_delta = 0.05, p->sprt_epsilon = 0.15, p->sprt_mS = 8.5, p->sprt_tM = 2314.0, p->sorted_idx = nullptr;
    p->check_degeneracy = 0, p->reserved = 0, p->th_pixels = 0.8, p->focal_length = 800.0;
}

static int usac_check_params(const mlpl_usac_params *P, int n, const char *who) {
    if (!P || n < 0 || !(P->th > 0) || !(P->conf >= 0 && P->conf <= 1) || P->max_hyp < 1 || !(P->sprt_delta > 0 && P->sprt_delta < 1) ||
        !(P->sprt_epsilon > 0 && P->sprt_epsilon < 1) || !(P->sprt_mS > 0) || !(P->sprt_tM > 0)) {
        set_error("%s: bad arguments", who);
        return MLPL_E_BAD_INPUT;
    }
    if ((P->check_degeneracy & ~3) || (P->check_degeneracy == 2) || (P->check_degeneracy && !(P->th_pixels > 0 && P->focal_length > 0))) {
        set_error("%s: bad degeneracy-test parameters", who);
        return MLPL_E_BAD_INPUT;
    }
    if ((P->estimator != 0 && P->estimator != 2) || !(P->refine == 0 || (P->refine >= 4 && P->refine <= 7))) {
        set_error("%s: estimator %d / refinement %d not built (POSE_NISTER, POSE_STEWENIUS with REF_WEIGHTS, REF_STEWENIUS(_WEIGHTS), REF_NISTER(_WEIGHTS) are)",
                  who, P->estimator, P->refine);
        return MLPL_E_UNSUPPORTED;
    }
    if (P->refine != 0 && (P->check_degeneracy & 2)) {  // the reference tests after local optimisations only with the 8-point refinements
        set_error("%s: check_degeneracy = 3 goes with the 8-point refinement only (usac_estimations.cpp:368-375)", who);
        return MLPL_E_BAD_INPUT;
    }
    if (P->sorted_idx)
        for (int i = 0; i < n; ++i)
            if (P->sorted_idx[i] >= (uint32_t)n) {
                set_error("%s: sorted index out of range", who);
                return MLPL_E_BAD_INPUT;
            }
    return MLPL_OK;
}

int mlpl_usac_essential_dev(mlpl_ctx *ctx, const double *d_p1, const double *d_p2, int n, const mlpl_usac_params *params, double E[9],
                            uint8_t *d_mask, double results[12], void *stream) {
    if (!ctx || !d_p1 || !d_p2 || !E) {
        set_error("mlpl_usac_essential_dev: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    int rc;
    if ((rc = usac_check_params(params, n, "mlpl_usac_essential_dev"))) return rc;
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    try {  // the host side of a run allocates (sample cache, bit rows): nothing may unwind through the C boundary
        return usac_essential_dev(ctx, d_p1, d_p2, n, params, E, d_mask, results, pick_stream(ctx, stream));
    } catch (const std::bad_alloc &) {
        set_error("mlpl_usac_essential_dev: out of host memory");
        return MLPL_E_NOMEM;
    }
}

int mlpl_usac_last_degeneracy(mlpl_ctx *ctx, double info[16], uint8_t *flags_rot, uint8_t *flags_nomot, int n) {
    if (!ctx || !info) {
        set_error("mlpl_usac_last_degeneracy: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    std::memcpy(info, ctx->last_usac_degen, sizeof(ctx->last_usac_degen));
    if (flags_rot || flags_nomot) {
        if (ctx->last_usac_degen[0] == 0.0 || !ctx->last_usac_flags || n != ctx->last_usac_flags_n) {
            set_error("mlpl_usac_last_degeneracy: the last call ran no degeneracy tests on %d correspondences", n);
            return MLPL_E_BAD_INPUT;
        }
        if (flags_rot) std::memcpy(flags_rot, ctx->last_usac_flags, (size_t)n);
        if (flags_nomot) std::memcpy(flags_nomot, ctx->last_usac_flags + n, (size_t)n);
    }
    return MLPL_OK;
}

int mlpl_usac_essential(mlpl_ctx *ctx, const double *p1, const double *p2, int n, const mlpl_usac_params *params, double E[9],
                        uint8_t *mask, double results[12]) {
    if (!ctx || !p1 || !p2 || !E) {
        set_error("mlpl_usac_essential: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    int rc;
    if ((rc = usac_check_params(params, n, "mlpl_usac_essential"))) return rc;
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    void *dp1, *dp2;
    const size_t pb = (size_t)std::max(n, 1) * 16;
    if ((rc = ws_get(ctx, WS_AUX0, pb, &dp1)) || (rc = ws_get(ctx, WS_AUX1, pb, &dp2))) return rc;
    MLPL_HIP_TRY(hipMemcpyAsync(dp1, p1, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    MLPL_HIP_TRY(hipMemcpyAsync(dp2, p2, (size_t)n * 16, hipMemcpyHostToDevice, ctx->stream));
    // the sequential part runs on the host and reads the correspondences there: it takes the caller's copies and returns the mask directly
    try {
        return usac_essential_dev(ctx, (const double *)dp1, (const double *)dp2, n, params, E, nullptr, results, ctx->stream, p1, p2, mask);
    } catch (const std::bad_alloc &) {
        set_error("mlpl_usac_essential: out of host memory");
        return MLPL_E_NOMEM;
    }
}

int mlpl_usac_essential_batch_dev(mlpl_ctx *ctx, int n_problems, const double *d_p1, const double *d_p2, int stride, const int32_t *counts,
                                  const mlpl_usac_params *params, double *E, uint8_t *d_masks, double *results, int32_t *status, double *degen,
                                  double *trace, int trace_cap, int32_t *trace_lens, void *stream) {
    if (!ctx || n_problems < 0 || !d_p1 || !d_p2 || stride < 1 || !counts || !params || !E || !results || !status) {
        set_error("mlpl_usac_essential_batch_dev: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    int rc;
    for (int b = 0; b < n_problems; ++b) {
        if (counts[b] < 0 || counts[b] > stride) {
            set_error("mlpl_usac_essential_batch_dev: problem %d has %d correspondences (stride %d)", b, counts[b], stride);
            return MLPL_E_BAD_INPUT;
        }
        if ((rc = usac_check_params(&params[b], counts[b], "mlpl_usac_essential_batch_dev"))) return rc;
    }
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    UsacBatchTrace tr{trace, trace_cap, trace_lens};
    if (trace && trace_lens)
        for (int b = 0; b < n_problems; ++b) trace_lens[b] = 0;
    try {
        return usac_essential_batch_dev(ctx, n_problems, d_p1, d_p2, stride, counts, params, E, d_masks, results, status, degen, trace ? &tr : nullptr,
                                        pick_stream(ctx, stream));
    } catch (const std::bad_alloc &) {
        set_error("mlpl_usac_essential_batch_dev: out of host memory");
        return MLPL_E_NOMEM;
    }
}

int mlpl_usac_last_stats(mlpl_ctx *ctx, long long stats[8]) {
    if (!ctx || !stats) return MLPL_E_BAD_INPUT;
    std::memcpy(stats, ctx->last_usac_stats, sizeof(ctx->last_usac_stats));
    return MLPL_OK;
}

int mlpl_debug_usac_trace(mlpl_ctx *ctx, double *buf, int cap_records) {
    if (!ctx) return MLPL_E_BAD_INPUT;
    const int len = ctx->usac_trace_len;
    ctx->usac_trace = buf, ctx->usac_trace_cap = buf ? cap_records : 0, ctx->usac_trace_len = 0;
    return len;
}

int mlpl_debug_arrsac_trace(mlpl_ctx *ctx, int32_t *buf, int cap) {
    if (!ctx) return MLPL_E_BAD_INPUT;
    const int len = ctx->arrsac_trace_len;
    ctx->arrsac_trace = buf, ctx->arrsac_trace_cap = buf ? cap : 0, ctx->arrsac_trace_len = 0;
    return len;
}

int mlpl_arrsac_last_stats(mlpl_ctx *ctx, long long stats[12]) {
    if (!ctx || !stats) return MLPL_E_BAD_INPUT;
    std::memcpy(stats, ctx->last_arrsac_stats, sizeof(ctx->last_arrsac_stats));
    return MLPL_OK;
}

}  // extern "C"
