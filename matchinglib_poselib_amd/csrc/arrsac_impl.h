// arrsac_impl.h -- ARRSAC (estimateEssentialMat's default method) on the MI355X.  Included by ransac_5pt.hip inside namespace mlpl: it
// reuses that file's solver tail (solve_from_basis, roots_kernel_t), its Jacobi 9x9 and the reference-arithmetic Sampson error.
//
// Reference: CvModelEstimator3::runARRSAC (poselib/source/five-point-nister/modelest.cpp:197-341) around theia::Arrsac
// (poselib/include/arrsac/arrsac.h:236-547) with EssentialMatEstimatorTheia (modelest.cpp:111-195), the PROSAC and uniform samplers
// (include/arrsac/prosac_sampler.h:85-157, random_sampler.h:57-78), the sequential probability ratio test
// (include/arrsac/sequential_probability_ratio.h:88-126, source/arrsac/sequential_probability_ratio.cc:38-62), CvEMEstimator::ValidModel
// (five-point.cpp:534-601) and, for `refine`, robustEssentialRefine (poselib/source/pose_estim.cpp:337-792).
//
// How a sequential estimator runs on a GPU.  The reference forms ONE hypothesis at a time; which sample comes next depends on the outcome
// of the previous test only through rare events (an accepted hypothesis with a new best support switches the sampler to an inner RANSAC on
// its inliers for 20 turns and shrinks the hypothesis budget).  Between events the sample sequence is a pure function of the two cv::RNG
// streams.  So the host runs the reference's control flow literally, and whenever it needs the models of a sample it has not seen, it first
// plays the samplers FORWARD from copies of their state under the assumption "no event", and sends the next up to 128..512 samples to the
// device as one batch:
//   arrsac_sample_kernel   one wave per sample: 5 points -> Householder null space; 6..7 points -> 4 smallest eigenvectors of the Gram
//                          matrix (the reference runs its 5-point kernel on them, cv::SVD of a 6x9 / 7x9 system); 8..14 points ->
//                          cv::findFundamentalMat(FM_8POINT) restated (float32 inputs, normalisation, 9x9 eigenvector, rank 2)
//   roots_kernel_t         the degree-10 polynomial's roots -> up to 10 essential matrices per sample (as RANSAC)
//   arrsac_check_kernel    one wave per model: ValidModel (Eigen's two-sided Jacobi SVD restated, V's column signs included) and the
//                          inlier bit of the model for the first 1024 correspondences (float Sampson error < thresh^2, the only
//                          correspondences ARRSAC ever tests: 100 in the first stage, < 900 in the preemptive stage)
// Results land in a cache keyed by the sample (kind, ordered indices).  The SPRT walks, the sigma / epsilon / threshold updates, the
// hypothesis bookkeeping and the preemptive stage's std::sort run on the host over those bit rows -- they are the control plane, a few
// thousand scalar operations.  A wrong guess costs a discarded batch, never a wrong result: the cache is keyed by content and the random
// streams the host advances are the real ones.  Final mask: inlier_mask_count_kernel; refinement: arrsac_refine_kernel (one workgroup,
// iteratively re-weighted 9x9 eigenproblem, no host hop inside).

namespace {

constexpr int kArrSmpStride = 16;     // int32 per sample: [0] size m, [1..14] indices, [15] kind (0 = 5-point solver, 1 = 8-point fit)
constexpr int kArrMaxSample = 14;
constexpr int kArrFlagPoints = 1024;  // correspondences a hypothesis can ever be tested on
constexpr int kArrFlagWords = kArrFlagPoints / 64;
constexpr int kArrHeadWords = 2;      // the first stage tests a hypothesis on the first 100 correspondences only
constexpr int kArrPoolSamples = 8192; // samples whose models the device keeps per call (speculation included)
constexpr int kArrBatchCap = 512;

// ---- Eigen::JacobiSVD<Matrix3d>(M, ComputeFullU | ComputeFullV) restated: two-sided Jacobi in Eigen's pair order, left rotation =
// symmetrising rotation * right^T, signs of the singular values absorbed by U, descending sort by column swaps.  V's columns carry
// Eigen's signs (ValidModel's epipole is V.col(2)). ----
struct JRot {
    double c, s;
};
template <int P, int Q>
__device__ __forceinline__ void jrot_rows(double *M, JRot j) {
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double x = M[P * 3 + i], y = M[Q * 3 + i];
        M[P * 3 + i] = j.c * x + j.s * y;
        M[Q * 3 + i] = -j.s * x + j.c * y;
    }
}
template <int P, int Q>
__device__ __forceinline__ void jrot_cols(double *M, JRot j) {  // applyOnTheRight(p, q, j): the columns turn with j^T
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double x = M[i * 3 + P], y = M[i * 3 + Q];
        M[i * 3 + P] = j.c * x - j.s * y;
        M[i * 3 + Q] = j.s * x + j.c * y;
    }
}
template <int P, int Q>
__device__ __forceinline__ bool jsvd_pair(double *W, double *U, double *V, double &max_diag) {
    const double threshold = fmax(DBL_MIN, 2.0 * DBL_EPSILON * max_diag);
    if (!(fabs(W[P * 3 + Q]) > threshold || fabs(W[Q * 3 + P]) > threshold)) return false;
    double m00 = W[P * 3 + P], m01 = W[P * 3 + Q], m10 = W[Q * 3 + P], m11 = W[Q * 3 + Q];
    JRot rot1;
    const double t = m00 + m11, d = m10 - m01;
    if (fabs(d) < DBL_MIN) {
        rot1 = {1.0, 0.0};
    } else {
        // (c, s) = (u, 1) / sqrt(1 + u^2): orthogonal to rounding for any u as long as the inverse root is accurate -- u = t / d from
        // v_rcp_f64 (~2^-26), the inverse root from v_rsq_f64 with two Newton steps; an inexact u leaves the 2x2 block slightly
        // unsymmetric, which the next sweep removes (the iteration ends on its own threshold test)
        double u = t * __builtin_amdgcn_rcp(d);
        if (!(fabs(u) < 1e150)) u = t / d;
        const double x = 1.0 + u * u;
        double it = __builtin_amdgcn_rsq(x);
        it = it * (1.5 - 0.5 * x * it * it);
        it = it * (1.5 - 0.5 * x * it * it);
        rot1 = {u * it, it};
    }
    {
        const double a0 = rot1.c * m00 + rot1.s * m10, a1 = rot1.c * m01 + rot1.s * m11;
        const double b1 = -rot1.s * m01 + rot1.c * m11;
        m00 = a0, m01 = a1, m11 = b1;
    }
    JRot jr = {1.0, 0.0};
    const double deno = 2.0 * fabs(m01);
    if (!(deno < DBL_MIN)) {
        double tau = (m00 - m11) * __builtin_amdgcn_rcp(deno);
        if (!(fabs(tau) < 1e150)) tau = (m00 - m11) / deno;
        const double w = __builtin_amdgcn_sqrt(tau * tau + 1.0);
        const double tt = tau > 0 ? __builtin_amdgcn_rcp(tau + w) : __builtin_amdgcn_rcp(tau - w);
        const double sign_t = tt > 0 ? 1.0 : -1.0;
        const double x2 = tt * tt + 1.0;
        double nn = __builtin_amdgcn_rsq(x2);
        nn = nn * (1.5 - 0.5 * x2 * nn * nn);
        nn = nn * (1.5 - 0.5 * x2 * nn * nn);
        jr = {nn, -sign_t * (m01 / fabs(m01)) * fabs(tt) * nn};
    }
    const JRot jl = {rot1.c * jr.c + rot1.s * jr.s, -rot1.c * jr.s + rot1.s * jr.c};  // rot1 * jr^T
    jrot_rows<P, Q>(W, jl);
    jrot_cols<P, Q>(U, JRot{jl.c, -jl.s});
    jrot_cols<P, Q>(W, jr);
    jrot_cols<P, Q>(V, jr);
    max_diag = fmax(max_diag, fmax(fabs(W[P * 3 + P]), fabs(W[Q * 3 + Q])));
    return true;
}
template <int A, int B>
__device__ __forceinline__ void jswap_cols(double *sv, double *U, double *V) {
    double t = sv[A];
    sv[A] = sv[B], sv[B] = t;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        t = U[r * 3 + A], U[r * 3 + A] = U[r * 3 + B], U[r * 3 + B] = t;
        t = V[r * 3 + A], V[r * 3 + A] = V[r * 3 + B], V[r * 3 + B] = t;
    }
}
__device__ void svd3_eigen(const double *Min, double *sv, double *U, double *V) {
    double W[9];
    double scale = 0;
#pragma unroll
    for (int i = 0; i < 9; ++i) scale = fmax(scale, fabs(Min[i]));
    if (scale == 0) scale = 1;
#pragma unroll
    for (int i = 0; i < 9; ++i) {
        W[i] = Min[i] / scale;
        U[i] = V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    }
    double max_diag = fmax(fabs(W[0]), fmax(fabs(W[4]), fabs(W[8])));
    for (int guard = 0; guard < 100; ++guard) {
        bool any = jsvd_pair<1, 0>(W, U, V, max_diag);
        any |= jsvd_pair<2, 0>(W, U, V, max_diag);
        any |= jsvd_pair<2, 1>(W, U, V, max_diag);
        if (!any) break;
    }
#pragma unroll
    for (int i = 0; i < 3; ++i) {
        const double a = W[i * 4];
        sv[i] = fabs(a) * scale;
        if (a < 0) {
#pragma unroll
            for (int r = 0; r < 3; ++r) U[r * 3 + i] = -U[r * 3 + i];
        }
    }
    // descending order, first maximum wins (Eigen: swap with the largest of the tail, stop at a zero)
    if (sv[1] > sv[0] && sv[1] >= sv[2]) jswap_cols<0, 1>(sv, U, V);
    else if (sv[2] > sv[0] && sv[2] > sv[1]) jswap_cols<0, 2>(sv, U, V);
    if (sv[0] != 0 && sv[2] > sv[1]) jswap_cols<1, 2>(sv, U, V);
}

__device__ __forceinline__ bool arr_is_zero(double d) { return d < 1e-3 && d > -1e-3; }  // five-point.hpp:76-81

// CvEMEstimator::ValidModel (five-point.cpp:534-601) by one wave.  The reference walks the m sample correspondences with E; at the first
// one that violates the oriented epipolar constraint it starts over with -E and counts the violations; V(-M) == V(M) for the
// decomposition above (the sign goes into U), so the epipole e2 is the same in both passes and the products only change sign.  Hence:
//   P_i = correspondence i violates under E,  Q_i = under -E;   valid = (no P_i) or (#Q_i / m < 0.4).
// Lane 0 decomposes E^T (singular-value tests, epipole); lane i < m evaluates P_i and Q_i; two ballots finish.  `sh` = 4 doubles of LDS.
__device__ __forceinline__ bool valid_model_wave(double x1, double y1, double x2, double y2, int m, const double *E, double *sh, int lane) {
    if (lane == 0) {
        double Et[9], sv[3], U[9], V[9];
#pragma unroll
        for (int r = 0; r < 3; ++r)
#pragma unroll
            for (int c = 0; c < 3; ++c) Et[r * 3 + c] = E[c * 3 + r];
        svd3_eigen(Et, sv, U, V);
        const bool ok = !(sv[0] / sv[1] > 1.2) && arr_is_zero(0.01 * sv[2] / sv[1]);
        sh[0] = V[2], sh[1] = V[5], sh[2] = V[8], sh[3] = ok ? 1.0 : 0.0;
    }
    wave_sync();
    const double e2[3] = {sh[0], sh[1], sh[2]};
    const bool sv_ok = sh[3] != 0.0;
    bool P = false, Q = false;
    if (lane < m) {
        const double l1[3] = {e2[1] - e2[2] * y2, e2[2] * x2 - e2[0], e2[0] * y2 - e2[1] * x2};
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double l2 = E[j * 3] * x1 + E[j * 3 + 1] * y1 + E[j * 3 + 2];
            const bool skip = arr_is_zero(0.1 * l1[j]) || arr_is_zero(0.1 * l2);
            const double pr = l1[j] * l2;
            P = P || (!skip && pr < 0);
            Q = Q || (!skip && pr > 0);  // l1 * (-l2) < 0
        }
    }
    const bool anyP = __ballot(P) != 0;
    const int fail = __popcll(__ballot(Q));
    return sv_ok && (!anyP || !((float)fail / (float)m >= 0.4f));
}

__device__ __forceinline__ double arr_row_elem(const double *p, int a) {  // epipolar row [x1x2, y1x2, x2, x1y2, y1y2, y2, x1, y1, 1]
    const double x1 = p[0], y1 = p[1], x2 = p[2], y2 = p[3];
    switch (a) {
        case 0: return x1 * x2;
        case 1: return y1 * x2;
        case 2: return x2;
        case 3: return x1 * y2;
        case 4: return y1 * y2;
        case 5: return y2;
        case 6: return x1;
        case 7: return y1;
        default: return 1.0;
    }
}

struct ArrSampleArgs {
    KHdr hdr;
    const double * p1;
    const double * p2;
    const int32_t * smp;
    int n_samples;
    PolyRec * recs;
    double * direct_E;
    int32_t * direct_ok;
    int inverse_iteration;  // option eig_inverse_iteration
};
__device__ __forceinline__ void arrsac_sample_body(const ArrSampleArgs &A, const int vbx, const int vby) {
    const double *__restrict__ p1 = A.p1;
    const double *__restrict__ p2 = A.p2;
    const int32_t *__restrict__ smp = A.smp;
    const int n_samples = A.n_samples;
    PolyRec *__restrict__ recs = A.recs;
    double *__restrict__ direct_E = A.direct_E;
    int32_t *__restrict__ direct_ok = A.direct_ok;

    __shared__ SolveLds L;
    __shared__ Jacobi9Lds J;
    __shared__ double xn[kArrMaxSample][4];
    __shared__ double nrm[6];
    __shared__ double s_gp[45], s_xv[9], s_xlam;  // smallest_eigvec9_wave (8-point fits)
    __shared__ int s_ok;
    if (blockDim.x != kSolverThreads) __builtin_trap();
    const int lane = threadIdx.x, b = vbx;
    if (b >= n_samples) return;
    const int32_t *sm = smp + (size_t)b * kArrSmpStride;
    const int m = sm[0], kind = sm[15];
    if (lane == 0) direct_ok[b] = 0;
    if (kind == 0 && m == 5) {
        if (lane < 5) {
            const int idx = sm[1 + lane];
            const double x1 = p1[2 * idx], y1 = p1[2 * idx + 1], x2 = p2[2 * idx], y2 = p2[2 * idx + 1];
            double *r = L.Q[lane];
            r[0] = x1 * x2, r[1] = y1 * x2, r[2] = x2, r[3] = x1 * y2, r[4] = y1 * y2, r[5] = y2, r[6] = x1, r[7] = y1, r[8] = 1.0;
        }
        wave_sync();
        householder_basis(L, lane);
        solve_from_basis(L, lane, recs + b);
        return;
    }
    if (lane < m) {
        const int idx = sm[1 + lane];
        double x1 = p1[2 * idx], y1 = p1[2 * idx + 1], x2 = p2[2 * idx], y2 = p2[2 * idx + 1];
        if (kind == 1) x1 = (double)(float)x1, y1 = (double)(float)y1, x2 = (double)(float)x2, y2 = (double)(float)y2;  // convertTo(CV_32F)
        xn[lane][0] = x1, xn[lane][1] = y1, xn[lane][2] = x2, xn[lane][3] = y2;
    }
    wave_sync();
    if (kind == 1) {
        if (lane == 0) {  // run8Point's normalisation (OpenCV calib3d fundam.cpp)
            double c1x = 0, c1y = 0, c2x = 0, c2y = 0, s1 = 0, s2 = 0;
            for (int i = 0; i < m; ++i) c1x += xn[i][0], c1y += xn[i][1], c2x += xn[i][2], c2y += xn[i][3];
            const double t = 1. / m;
            c1x *= t, c1y *= t, c2x *= t, c2y *= t;
            for (int i = 0; i < m; ++i) {
                const double dx1 = xn[i][0] - c1x, dy1 = xn[i][1] - c1y, dx2 = xn[i][2] - c2x, dy2 = xn[i][3] - c2y;
                s1 += sqrt(dx1 * dx1 + dy1 * dy1);
                s2 += sqrt(dx2 * dx2 + dy2 * dy2);
            }
            s1 *= t, s2 *= t;
            s_ok = !(s1 < (double)FLT_EPSILON || s2 < (double)FLT_EPSILON);
            s1 = sqrt(2.) / s1, s2 = sqrt(2.) / s2;
            nrm[0] = c1x, nrm[1] = c1y, nrm[2] = s1, nrm[3] = c2x, nrm[4] = c2y, nrm[5] = s2;
        }
        wave_sync();
        for (int k = lane; k < 88; k += 64) reinterpret_cast<double *>(recs + b)[k] = 0.0;  // no polynomial: the root kernel skips it
        if (!s_ok) return;
        if (lane < m) {
            xn[lane][0] = (xn[lane][0] - nrm[0]) * nrm[2], xn[lane][1] = (xn[lane][1] - nrm[1]) * nrm[2];
            xn[lane][2] = (xn[lane][2] - nrm[3]) * nrm[5], xn[lane][3] = (xn[lane][3] - nrm[4]) * nrm[5];
        }
        wave_sync();
    }
    for (int e = lane; e < 81; e += 64) {
        const int a = e / 9, c = e - a * 9;
        double acc = 0;
        for (int i = 0; i < m; ++i) acc += arr_row_elem(xn[i], a) * arr_row_elem(xn[i], c);
        J.G[a][c] = acc;
        J.Vv[a][c] = (a == c) ? 1.0 : 0.0;
    }
    wave_sync();
    // The 8-point fit needs ONE eigenvector (the smallest eigenvalue's): inverse iteration first (smallest_eigvec9_wave, ransac_5pt.hip:
    // ~3 us where the Jacobi decomposition of these fourteen-point systems takes ~50); a system it does not settle on -- among them every
    // one whose second-smallest eigenvalue is below DBL_EPSILON, the reference's rank test: convergence needs that eigenvalue well above
    // the shift 2^-44 trace -- takes the Jacobi path with the test as before.
    int steps = 0;
    if (kind == 1 && A.inverse_iteration) {
        if (lane < 45) {
            int a = 0, rem = lane;
            while (rem >= 9 - a) rem -= 9 - a, ++a;
            s_gp[lane] = J.G[a][a + rem];
        }
        wave_sync();
        steps = smallest_eigvec9_wave(s_gp, 1.0, nullptr, s_xv, &s_xlam, lane);  // wave-uniform
        wave_sync();
    }
    if (steps == 0) jacobi9_wave(J, lane);
    if (kind == 0) {
        if (lane == 0) {
            int order[9];
            order_desc9(J, order);
            for (int j = 0; j < 4; ++j)
                for (int r = 0; r < 9; ++r) L.EE[j][r] = J.Vv[r][order[5 + j]];
        }
        wave_sync();
        solve_from_basis(L, lane, recs + b);
        return;
    }
    if (lane == 0) {
        int order[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
        if (steps == 0) order_desc9(J, order);
        if (steps > 0 || !(fabs(J.G[order[7]][order[7]]) < DBL_EPSILON)) {
            double F0[9], sv[3], U[9], V[9];
            for (int k = 0; k < 9; ++k) F0[k] = steps > 0 ? s_xv[k] : J.Vv[k][order[8]];
            svd3_eigen(F0, sv, U, V);
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) F0[r * 3 + c] = U[r * 3] * sv[0] * V[c * 3] + U[r * 3 + 1] * sv[1] * V[c * 3 + 1];
            // F = T2^T F0 T1 with T = [s 0 -s cx; 0 s -s cy; 0 0 1]
            const double s1 = nrm[2], a1 = -nrm[2] * nrm[0], b1 = -nrm[2] * nrm[1];
            const double s2 = nrm[5], a2 = -nrm[5] * nrm[3], b2 = -nrm[5] * nrm[4];
            double G2[9];  // T2^T F0
            for (int c = 0; c < 3; ++c) {
                G2[c] = s2 * F0[c];
                G2[3 + c] = s2 * F0[3 + c];
                G2[6 + c] = a2 * F0[c] + b2 * F0[3 + c] + F0[6 + c];
            }
            double F[9];
            for (int r = 0; r < 3; ++r) {
                F[r * 3] = G2[r * 3] * s1;
                F[r * 3 + 1] = G2[r * 3 + 1] * s1;
                F[r * 3 + 2] = G2[r * 3] * a1 + G2[r * 3 + 1] * b1 + G2[r * 3 + 2];
            }
            if (fabs(F[8]) > (double)FLT_EPSILON) {
                const double inv = 1. / F[8];
                for (int k = 0; k < 9; ++k) F[k] *= inv;
            }
            for (int k = 0; k < 9; ++k) direct_E[(size_t)b * 9 + k] = F[k];
            direct_ok[b] = 1;
        }
    }
}
MLPL_HUB_KERNEL(HK_ARR_SAMPLE, ArrSampleArgs, arrsac_sample_body, 64);

// ValidModel for ONE model on ONE lane: valid_model_wave's arithmetic, the sample's points in a loop instead of across lanes (the same
// operations per point, so the same verdict).  The 3 x 3 SVD is sequential code; one wave per model left 63 lanes idle for ~15 us, which a
// single run hides behind nothing anyway but a batch of 128 runs pays 164 000 times per round (0.7 ms of a 1 ms round).
__device__ __forceinline__ bool valid_model_lane(const double *E, const int32_t *sm, int m, const double *__restrict__ p1, const double *__restrict__ p2) {
    double Et[9], sv[3], U[9], V[9];
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int c = 0; c < 3; ++c) Et[r * 3 + c] = E[c * 3 + r];
    svd3_eigen(Et, sv, U, V);
    const bool sv_ok = !(sv[0] / sv[1] > 1.2) && arr_is_zero(0.01 * sv[2] / sv[1]);
    const double e2[3] = {V[2], V[5], V[8]};
    bool anyP = false;
    int fail = 0;
    for (int t = 0; t < m; ++t) {
        const int idx = sm[1 + t];
        const double x1 = p1[2 * idx], y1 = p1[2 * idx + 1], x2 = p2[2 * idx], y2 = p2[2 * idx + 1];
        const double l1[3] = {e2[1] - e2[2] * y2, e2[2] * x2 - e2[0], e2[0] * y2 - e2[1] * x2};
        bool P = false, Q = false;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const double l2 = E[j * 3] * x1 + E[j * 3 + 1] * y1 + E[j * 3 + 2];
            const bool skip = arr_is_zero(0.1 * l1[j]) || arr_is_zero(0.1 * l2);
            const double pr = l1[j] * l2;
            P = P || (!skip && pr < 0);
            Q = Q || (!skip && pr > 0);  // l1 * (-l2) < 0
        }
        anyP = anyP || P;
        fail += Q ? 1 : 0;
    }
    return sv_ok && (!anyP || !((float)fail / (float)m >= 0.4f));
}

// Two launches per batch of samples: arrsac_valid (one LANE per model slot: sign convention, ValidModel) and arrsac_check (one WAVE per
// valid model: its inlier bits).
struct ArrCheckArgs {
    KHdr hdr;
    const double4 * pts;
    int flag_points;
    const double * p1;
    const double * p2;
    const int32_t * smp;
    int n_samples;
    double * E_tab;
    const int32_t * n_models;
    const double * direct_E;
    const int32_t * direct_ok;
    double thresh2;
    int32_t * out_nm;
    int32_t * out_valid;
    double * out_e00;
    unsigned long long * out_head;
    unsigned long long * flag_rows;
    int32_t * d_valid;  // [n_samples][10], device: arrsac_valid -> arrsac_check
};
__device__ __forceinline__ void arrsac_valid_body(const ArrCheckArgs &A, const int vbx, const int vby) {
    const int g = vbx * 64 + (int)threadIdx.x;  // model slot
    const int b = g / 10, slot = g - b * 10;
    if (b >= A.n_samples) return;
    const int32_t *sm = A.smp + (size_t)b * kArrSmpStride;
    const int m = sm[0], kind = sm[15];
    const int nm = kind ? A.direct_ok[b] : A.n_models[b];
    if (slot == 0) A.out_nm[b] = nm;
    if (slot >= nm) {
        A.out_valid[b * 10 + slot] = 0;
        A.d_valid[b * 10 + slot] = 0;
        return;
    }
    double e[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) e[k] = kind ? A.direct_E[(size_t)b * 9 + k] : A.E_tab[((size_t)b * 10 + slot) * 9 + k];
    if (!kind) {
        // sign convention of a 5-point solution: its element of largest magnitude is positive.  (The reference's sign comes from the
        // last null vector cv::SVD happens to return -- rounding noise decides it -- and ValidModel is not sign-symmetric.)
        int at = 0;
#pragma unroll
        for (int k = 1; k < 9; ++k)
            if (fabs(e[k]) > fabs(e[at])) at = k;
        double pick = e[0];
#pragma unroll
        for (int k = 1; k < 9; ++k) pick = (at == k) ? e[k] : pick;
        if (pick < 0) {
#pragma unroll
            for (int k = 0; k < 9; ++k) e[k] = -e[k];
        }
    }
#pragma unroll
    for (int k = 0; k < 9; ++k) A.E_tab[((size_t)b * 10 + slot) * 9 + k] = e[k];
    const bool valid = valid_model_lane(e, sm, m, A.p1, A.p2);
    A.out_valid[b * 10 + slot] = valid ? 1 : 0;
    A.d_valid[b * 10 + slot] = valid ? 1 : 0;
    if (valid) A.out_e00[b * 10 + slot] = e[0];
}
MLPL_HUB_KERNEL(HK_ARR_VALID, ArrCheckArgs, arrsac_valid_body, 64);

__device__ __forceinline__ void arrsac_check_body(const ArrCheckArgs &A, const int vbx, const int vby) {
    const double4 *__restrict__ pts = A.pts;
    const int flag_points = A.flag_points;
    const double thresh2 = A.thresh2;
    unsigned long long *__restrict__ out_head = A.out_head;
    unsigned long long *__restrict__ flag_rows = A.flag_rows;
    const int lane = threadIdx.x;
    const int b = vbx / 10, slot = vbx - b * 10;
    if (b >= A.n_samples) return;
    if (!A.d_valid[b * 10 + slot]) return;  // wave-uniform
    double e[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) e[k] = A.E_tab[((size_t)b * 10 + slot) * 9 + k];
    // the sixteen correspondences of this lane: all loads go out together (one after the other, each of the sixteen rounds waited
    // for its own: 4.4 us of a 13 us wave)
    double4 pp[kArrFlagWords];
#pragma unroll
    for (int w = 0; w < kArrFlagWords; ++w) pp[w] = pts[min(w * 64 + lane, flag_points - 1)];
#pragma unroll
    for (int w = 0; w < kArrFlagWords; ++w) {
        const int i = w * 64 + lane;
        const double4 p = pp[w];
        const bool in = i < flag_points && (double)sampson_err_f32(e, p.x, p.y, p.z, p.w) < thresh2;  // Estimator::Error(...) < error_thresh, strict
        const unsigned long long bal = __ballot(in);
        if (lane == 0) {
            flag_rows[((size_t)b * 10 + slot) * kArrFlagWords + w] = bal;  // the whole row stays on the device (arrsac_gather_kernel)
            if (w < kArrHeadWords) out_head[((size_t)b * 10 + slot) * kArrHeadWords + w] = bal;
        }
    }
}
MLPL_HUB_KERNEL(HK_ARR_CHECK, ArrCheckArgs, arrsac_check_body, 64);

// Rows of the device-side model pool the host asks for (accepted hypotheses; models of the preemptive stage's generation branch):
// out[i] = { E[9], bits[16] } of pool row rows[i].
struct ArrGatherArgs {
    KHdr hdr;
    const int32_t * rows;
    int n_rows;
    const double * E_pool;
    const unsigned long long * F_pool;
    double * out;
};
__device__ __forceinline__ void arrsac_gather_body(const ArrGatherArgs &A, const int vbx, const int vby) {
    const int32_t *__restrict__ rows = A.rows;
    const int n_rows = A.n_rows;
    const double *__restrict__ E_pool = A.E_pool;
    const unsigned long long *__restrict__ F_pool = A.F_pool;
    double *__restrict__ out = A.out;

    const int i = vbx, t = threadIdx.x;
    if (i >= n_rows || t >= 9 + kArrFlagWords) return;
    const size_t row = (size_t)rows[i];
    double *o = out + (size_t)i * (9 + kArrFlagWords);
    if (t < 9) o[t] = E_pool[row * 9 + t];
    else reinterpret_cast<unsigned long long *>(o)[t] = F_pool[row * kArrFlagWords + (t - 9)];
}
MLPL_HUB_KERNEL(HK_ARR_GATHER, ArrGatherArgs, arrsac_gather_body, 32);

// Inlier bits of pool models over ALL n correspondences (the preemptive stage beyond the first kArrFlagPoints: reached when the
// hypothesis set keeps growing instead of halving, i.e. at inlier ratios above ~0.9): out[i] = words_full words for pool row rows[i].
struct ArrExtendArgs {
    KHdr hdr;
    const int32_t * rows;
    int n_rows;
    const double4 * pts;
    int n;
    int words_full;
    const double * E_pool;
    double thresh2;
    unsigned long long * out;
};
__device__ __forceinline__ void arrsac_extend_body(const ArrExtendArgs &A, const int vbx, const int vby) {
    const int32_t *__restrict__ rows = A.rows;
    const int n_rows = A.n_rows;
    const double4 *__restrict__ pts = A.pts;
    const int n = A.n;
    const int words_full = A.words_full;
    const double *__restrict__ E_pool = A.E_pool;
    const double thresh2 = A.thresh2;
    unsigned long long *__restrict__ out = A.out;

    const int i = vbx, lane = threadIdx.x;
    if (i >= n_rows) return;
    double e[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) e[k] = E_pool[(size_t)rows[i] * 9 + k];
    for (int w = 0; w < words_full; ++w) {
        const int j = w * 64 + lane;
        bool in = false;
        if (j < n) {
            const double4 p = pts[j];
            in = (double)sampson_err_f32(e, p.x, p.y, p.z, p.w) < thresh2;
        }
        const unsigned long long bal = __ballot(in);
        if (lane == 0) out[(size_t)i * words_full + w] = bal;
    }
}
MLPL_HUB_KERNEL(HK_ARR_EXTEND, ArrExtendArgs, arrsac_extend_body, 64);

// robustEssentialRefine(inliers, E_init, th, iters = 0, makeClosestE = true) (pose_estim.cpp:337-792; model 0, no normalisation): up to
// 50 rounds of { pseudo-Huber weights on the Sampson distance under the current matrix (pose_helper.cpp:115-143, BA_driver.cpp:2639-2648),
// weighted 9x9 normal matrix, eigenvector of its smallest eigenvalue, closest essential matrix (getClosestE, pose_helper.cpp:152-177) },
// stopped by the reference's tests on the residual.  One workgroup; the normal matrix is summed unnormalised and divided by the weight
// norm afterwards (the reference scales every row first).  info = {rounds, status: 0 converged/exhausted, 1 stopped on an invalid matrix,
// 2 rejected: fewer than 50 points, 3 rejected: rank-deficient system (E_init is returned by both)}.
// 512 threads: the 46 running sums of a thread need ~130 registers; at 1024 threads per workgroup (four waves per SIMD, 128 registers
// each) the compiler spilled 213 of them and every round spent 60 us in scratch traffic.
constexpr int kArrRefineThreads = 512;
struct ArrRefineArgs {
    KHdr hdr;
    const double4 * pts;
    const uint8_t * mask;
    int n;
    const double * E_init;
    double th;
    double * E_out;
    int32_t * info;
    int warm_start;
};
__device__ __forceinline__ void arrsac_refine_body(const ArrRefineArgs &A, const int vbx, const int vby) {
    const double4 *__restrict__ pts = A.pts;
    const uint8_t *__restrict__ mask = A.mask;
    const int n = A.n;
    const double *__restrict__ E_init = A.E_init;
    const double th = A.th;
    double *__restrict__ E_out = A.E_out;
    int32_t *__restrict__ info = A.info;

    __shared__ double red[kArrRefineThreads / 64][46];
    __shared__ double stage[16][kArrRefineThreads];
    __shared__ Jacobi9Lds J;
    __shared__ double F3[9];
    __shared__ double s_err_old;
    __shared__ double s_xv[9], s_xprev[9], s_xlambda;  // smallest_eigvec9_wave: this round's vector, the previous round's (the next start)
    __shared__ int s_x_have, s_jv_have;
    __shared__ int s_stop, s_cnt;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    if (tid == 0) s_cnt = 0, s_stop = 0, s_err_old = 1e12, s_x_have = 0, s_jv_have = 0;
    if (tid < 9) F3[tid] = E_init[tid];
    __syncthreads();
    {
        int c = 0;
        for (int i = tid; i < n; i += kArrRefineThreads) c += mask[i] ? 1 : 0;
        for (int off = 32; off > 0; off >>= 1) c += __shfl_xor(c, off);
        if (lane == 0 && c) atomicAdd(&s_cnt, c);
    }
    __syncthreads();
    const int npoints = s_cnt;
    if (npoints < 50) {
        if (tid < 9) E_out[tid] = E_init[tid];
        if (tid == 0) info[0] = 0, info[1] = 2;
        return;
    }
    const double min_diff = th / 10, min_err = th * th / 100 * npoints;
    int j = 0;
    for (; j < 50; ++j) {
        double e[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) e[k] = F3[k];
        double acc[46];
#pragma unroll
        for (int k = 0; k < 46; ++k) acc[k] = 0;
        for (int i = tid; i < n; i += kArrRefineThreads) {
            if (!mask[i]) continue;
            const double4 p = pts[i];
            const double x0 = p.x, y0 = p.y, x1 = p.z, y1 = p.w;
            // SampsonL1: num = x2^T E x1, denom1 = 1 / (sqrt(|E x1|_xy^2 + |E^T x2|_xy^2) + 1e-8)
            const double g0 = x1 * e[0] + y1 * e[3] + e[6], g1 = x1 * e[1] + y1 * e[4] + e[7], g2 = x1 * e[2] + y1 * e[5] + e[8];
            const double num = g0 * x0 + g1 * y0 + g2;
            const double h0 = e[0] * x0 + e[1] * y0 + e[2], h1 = e[3] * x0 + e[4] * y0 + e[5];
            const double denom1 = 1 / (sqrt(h0 * h0 + h1 * h1 + g0 * g0 + g1 * g1) + 1e-8);
            const double d_abs = fabs(num * denom1) + 1e-12;
            const double qq = d_abs / th;
            const double w = sqrt(2 * th * th * (sqrt(1 + qq * qq) - 1)) / d_abs;  // costPseudoHuber
            const double f = denom1 * w, f2 = f * f;
            const double r[9] = {x1 * x0, x1 * y0, x1, y1 * x0, y1 * y0, y1, x0, y0, 1.0};
            int t = 0;
#pragma unroll
            for (int a = 0; a < 9; ++a)
#pragma unroll
                for (int c = a; c < 9; ++c) acc[t++] += f2 * r[a] * r[c];
            acc[45] += f2;
        }
        // block-wide sums: three slices of sixteen staged value-major in LDS, thirty-two threads per value add sixteen entries each and
        // finish inside their half wave (six butterfly steps per value were 552 ds_bpermute per wave and round)
#pragma unroll
        for (int c0 = 0; c0 < 48; c0 += 16) {
#pragma unroll
            for (int k = 0; k < 16; ++k)
                if (c0 + k < 46) stage[k][tid] = acc[c0 + k];
            __syncthreads();
            const int v = tid >> 5, part = tid & 31;
            double sacc = 0;
            if (c0 + v < 46) {
#pragma unroll
                for (int q = 0; q < kArrRefineThreads / 32; ++q) sacc += stage[v][part + 32 * q];
            }
#pragma unroll
            for (int off = 16; off > 0; off >>= 1) sacc += __shfl_xor(sacc, off);
            if (c0 + v < 46 && part == 0) red[0][c0 + v] = sacc;
            __syncthreads();
        }
        if (wave == 0) {
            const double wn2 = red[0][45];
            // From the second round on the iteration starts from the previous round's eigenvectors (J.Vv): consecutive rounds re-weight the
            // same correspondences with a slightly different matrix, V_prev^T G V_prev is nearly diagonal and two or three sweeps finish
            // where a cold start needs eight (as usac_fit_from_cov does for the local optimisation's refits; ~35 -> ~12 us per round).
            // First the inverse iteration for the one eigenvector the round needs (smallest_eigvec9_wave, ransac_5pt.hip; from the previous
            // round's vector: ~2 us instead of ~35); a system it does not settle on takes the Jacobi path below.  (Its "second smallest
            // eigenvalue below DBL_EPSILON" failure cannot pass unnoticed here: the iteration converges only when that eigenvalue is well
            // above the shift 2^-44 trace, and the trace of the normalised matrix is at least 1.)
            int steps = 0;
            if (A.warm_start & 2) steps = smallest_eigvec9_wave(red[0], wn2, (j > 0 && s_x_have) ? s_xprev : nullptr, s_xv, &s_xlambda, lane);
            if (steps == 0) {
                const bool jwarm = j > 0 && (A.warm_start & 1) && s_jv_have;
                if (!jwarm) {
                    if (lane == 0) {
                        int t = 0;
                        for (int a = 0; a < 9; ++a)
                            for (int c = a; c < 9; ++c) {
                                const double v = red[0][t++] / wn2;
                                J.G[a][c] = v;
                                J.G[c][a] = v;
                            }
                    }
                    for (int ee = lane; ee < 81; ee += 64) J.Vv[ee / 9][ee % 9] = (ee / 9 == ee % 9) ? 1.0 : 0.0;
                } else {
                    for (int ee = lane; ee < 81; ee += 64) {  // Gn = G V_prev (G from the packed upper triangle, scaled)
                        const int a = ee / 9, b = ee - a * 9;
                        double sacc = 0;
                        for (int k = 0; k < 9; ++k) {
                            const int lo = a < k ? a : k, hi = a < k ? k : a;
                            sacc += (red[0][lo * 9 - lo * (lo - 1) / 2 + (hi - lo)] / wn2) * J.Vv[k][b];
                        }
                        J.Gn[a][b] = sacc;
                    }
                    wave_sync();
                    for (int ee = lane; ee < 81; ee += 64) {  // G = V_prev^T Gn, upper triangle mirrored
                        const int a = ee / 9, b = ee - a * 9;
                        if (a > b) continue;
                        double sacc = 0;
                        for (int k = 0; k < 9; ++k) sacc += J.Vv[k][a] * J.Gn[k][b];
                        J.G[a][b] = sacc;
                        J.G[b][a] = sacc;
                    }
                }
                wave_sync();
                jacobi9_wave(J, lane);
            }
            wave_sync();
            if (lane == 0) {
                double F2[9], err = 0;
                bool rank_deficient = false;
                if (steps > 0) {
                    for (int k = 0; k < 9; ++k) F2[k] = s_xv[k];
                    err = s_xlambda;
                } else {
                    int order[9];
                    order_desc9(J, order);
                    rank_deficient = fabs(J.G[order[7]][order[7]]) < DBL_EPSILON;
                    for (int k = 0; k < 9; ++k) F2[k] = J.Vv[k][order[8]];
                    err = J.G[order[8]][order[8]];  // |A1 lastCol|^2 = the eigenvalue
                    s_jv_have = 1;
                }
                for (int k = 0; k < 9; ++k) s_xprev[k] = F2[k];
                s_x_have = 1;
                if (rank_deficient) {
                    s_stop = 3;  // "Refinement failed!": the initial matrix is returned
                } else {
                    double sv[3], U[9], V[9];
                    svd3_eigen(F2, sv, U, V);
                    if (!arr_is_zero(sv[2]) || sv[0] / sv[1] > 1.5 || sv[0] / sv[1] < 0.66 || !(sv[1] > 0)) {
                        s_stop = 2;  // "taking last valid E"
                    } else {
                        for (int r = 0; r < 3; ++r)
                            for (int c = 0; c < 3; ++c) F3[r * 3 + c] = U[r * 3] * sv[0] * V[c * 3] + U[r * 3 + 1] * sv[1] * V[c * 3 + 1];
                        const double diff = fabs(s_err_old - err);
                        if (j > 1 && (diff < min_diff || err < min_err)) s_stop = 1;
                        s_err_old = err;
                    }
                }
            }
        }
        __syncthreads();
        if (s_stop) break;
    }
    if (tid < 9) E_out[tid] = (s_stop == 3) ? E_init[tid] : F3[tid];
    // info[0] = the reference's loop counter when the loop ended (the index of the round a stopping test fired in, i.e. one LESS than the
    // rounds executed then; 50 when exhausted); info[1]: 3 = the system of the first round was rank deficient ("Refinement failed!" in the
    // reference), as distinct from 2 = fewer than 50 correspondences ("too less points", written above)
    if (tid == 0) info[0] = j, info[1] = (s_stop == 3) ? 3 : (s_stop == 2 ? 1 : 0);
}
MLPL_HUB_KERNEL(HK_ARR_REFINE, ArrRefineArgs, arrsac_refine_body, kArrRefineThreads);

// ------------------------------------------------------------------------------------------------------------------------------
// host side
// ------------------------------------------------------------------------------------------------------------------------------
struct CvRng {  // cv::RNG (OpenCV core): multiply-with-carry
    uint64_t state;
    unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

struct ArrModelHost {
    int row;                       // row of the device-side pool (E, all flag words)
    int full;                      // index into ArrsacRun::full_rows once the whole row was fetched, else -1
    int ext = -1;                  // index into ArrsacRun::ext_rows (bits of ALL correspondences) once computed, else -1
    uint64_t head[kArrHeadWords];  // inlier bits of the first 128 correspondences
};
struct ArrFullRow {
    double E[9];
    uint64_t bits[kArrFlagWords];
};
// A sample as the device sees it: [0] size m, [1..14] ordered indices (0 beyond m), [15] kind -- also the cache key (no heap traffic).
struct ArrKey {
    int32_t v[kArrSmpStride];
    int size() const { return v[0]; }
    int kind() const { return v[15]; }
    void reset(int kind) {
        std::memset(v, 0, sizeof(v));
        v[15] = kind;
    }
    void push(int idx) { v[1 + v[0]++] = idx; }
    bool has(int idx) const {
        for (int i = 0; i < v[0]; ++i)
            if (v[1 + i] == idx) return true;
        return false;
    }
    bool operator==(const ArrKey &o) const { return std::memcmp(v, o.v, sizeof(v)) == 0; }
};
struct ArrKeyHash {
    size_t operator()(const ArrKey &k) const {
        uint64_t h = 1469598103934665603ull;
        for (int i = 0; i < kArrSmpStride; ++i) h = (h ^ (uint32_t)k.v[i]) * 1099511628211ull;
        return (size_t)h;
    }
};
struct ArrIds {  // ids of a sample's VALID models in the order of the convention
    int n;
    int id[10];
    size_t size() const { return (size_t)n; }
    bool empty() const { return n == 0; }
    int operator[](size_t i) const { return id[i]; }
    const int *begin() const { return id; }
    const int *end() const { return id + n; }
};

struct ArrBufs {  // a run's buffers when the caller provides them (batched runs: one slice per run)
    char *dev = nullptr, *pin = nullptr, *pin_dev = nullptr;
};

struct ArrsacRun {
    mlpl_ctx *ctx;
    hipStream_t s;
    Launcher L;                    // launches now (a run alone) or records for the hub (a run of a batch, batch_hub.h)
    const ArrBufs *bufs = nullptr;
    int32_t *trace_buf = nullptr;  // turn records of the first stage (this run's buffer)
    int trace_cap = 0, trace_len = 0;
    const double *d_p1, *d_p2;
    const double4 *pts;
    int n, flag_points;
    double thresh2;
    // device / pinned buffers of a batch
    int32_t *d_smp = nullptr;
    PolyRec *d_recs = nullptr;
    double *d_direct = nullptr;
    int32_t *d_direct_ok = nullptr, *d_nm5 = nullptr, *d_valid = nullptr;
    char *h_out = nullptr, *h_out_dev = nullptr;
    int32_t *h_smp = nullptr;
    double *d_final = nullptr;     // [0..8] best model, [9..17] refined, then 4 ints: inlier count, refinement info
    double *h_final = nullptr, *h_final_dev = nullptr;
    // results
    std::vector<ArrModelHost> pool;
    std::vector<ArrFullRow> full_rows;
    std::vector<std::vector<uint64_t>> ext_rows;
    double *d_Epool = nullptr;
    unsigned long long *d_Fpool = nullptr;
    int pool_samples = 0;  // samples whose rows are in use
    std::unordered_map<ArrKey, ArrIds, ArrKeyHash> cache;  // sample -> its valid models
    long long stats[12] = {0};                 // [8] batches, [9] samples sent, [10] samples used

    // Arrsac(5, thr^2, 500, 100, 14, 8) (modelest.cpp:270) and its members (arrsac.h:102-118)
    static constexpr int kMinSample = 5, kMaxHyps = 500, kBlock = 100, kNonMin = 14, kNonMinMin = 8, kMaxInner = 20;
    double sigma = 0.05, epsilon = 0.1;
    static constexpr double kConf = 0.95, kTimeRatio = 250.0;
    int verified_accum = 0, num_rejected = 0;
    double rejected_accum = 0.0;
    CvRng prosac_rng, random_rng;

    static size_t out_bytes(int B) {
        const size_t batch = (size_t)B * 4 + (size_t)B * 40 + (size_t)B * 80 + (size_t)B * 10 * kArrHeadWords * 8 + 64;
        return std::max(batch, (size_t)B * 10 * sizeof(ArrFullRow) + (size_t)B * 40);  // the same blocks carry gathered rows
    }
    struct Layout {
        size_t d_pts, d_recs, d_direct, d_valid, d_final, d_Epool, d_Fpool, dev_total;
        size_t p_out, p_smp, p_final, pin_total;
    };
    static Layout layout(int n) {
        Layout Y;
        auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        size_t o = 0;
        Y.d_pts = o, o = up(o + (size_t)std::max(n, 1) * sizeof(double4));
        Y.d_recs = o, o = up(o + (size_t)kArrBatchCap * sizeof(PolyRec));
        Y.d_direct = o, o = up(o + (size_t)kArrBatchCap * (72 + 8));
        Y.d_valid = o, o = up(o + (size_t)kArrBatchCap * 10 * 4);
        Y.d_final = o, o = up(o + 256);
        Y.d_Epool = o, o = up(o + (size_t)kArrPoolSamples * 720);
        Y.d_Fpool = o, o = up(o + (size_t)kArrPoolSamples * 10 * kArrFlagWords * 8);
        Y.dev_total = o + 256;
        o = 0;
        Y.p_out = o, o = up(o + std::max(out_bytes(kArrBatchCap), (size_t)((std::max(n, 1) + 63) / 64) * 8 + 8192));
        Y.p_smp = o, o = up(o + (size_t)kArrBatchCap * kArrSmpStride * 4);
        Y.p_final = o, o = up(o + 256);
        Y.pin_total = o + 256;
        return Y;
    }

    // buffers + the packed correspondences (a run alone takes the blocks from the context)
    int alloc() {
        const Layout Y = layout(n);
        char *dev, *pin, *pin_dev;
        if (bufs) {
            dev = bufs->dev, pin = bufs->pin, pin_dev = bufs->pin_dev;
        } else {
            void *p;
            int rc;
            if ((rc = ws_get(ctx, WS_ARR_E, Y.dev_total, &p))) return rc;
            dev = (char *)p;
            if ((rc = pinned_get(ctx, Y.pin_total, &p))) return rc;
            pin = (char *)p;
            void *alias = nullptr;
            MLPL_HIP_TRY(hipHostGetDevicePointer(&alias, pin, 0));
            pin_dev = (char *)alias;
        }
        double4 *d_pts = (double4 *)(dev + Y.d_pts);
        d_recs = (PolyRec *)(dev + Y.d_recs);
        d_direct = (double *)(dev + Y.d_direct);
        d_direct_ok = (int32_t *)(d_direct + (size_t)kArrBatchCap * 9);
        d_nm5 = d_direct_ok + kArrBatchCap;
        d_valid = (int32_t *)(dev + Y.d_valid);
        d_final = (double *)(dev + Y.d_final);
        d_Epool = (double *)(dev + Y.d_Epool);
        d_Fpool = (unsigned long long *)(dev + Y.d_Fpool);
        // the sample table and the small per-model results live in the pinned (mapped) host block: the kernels read / write them directly
        h_out = pin + Y.p_out, h_out_dev = pin_dev + Y.p_out;
        h_smp = (int32_t *)(pin + Y.p_smp), d_smp = (int32_t *)(pin_dev + Y.p_smp);
        h_final = (double *)(pin + Y.p_final), h_final_dev = (double *)(pin_dev + Y.p_final);
        PackPtsArgs pa{{(n + 255) / 256, 1}, d_p1, d_p2, n, d_pts};
        L.launch(HK_PACK_POINTS, pa);
        pts = d_pts;
        return MLPL_OK;
    }

    // one device batch: every key not yet cached gets its entry
    int run_batch(const std::vector<ArrKey> &keys) {
        const int B = (int)keys.size();
        if (B == 0) return MLPL_OK;
        for (int b = 0; b < B; ++b) std::memcpy(h_smp + (size_t)b * kArrSmpStride, keys[b].v, sizeof(keys[b].v));
        if (pool_samples + B > kArrPoolSamples) {
            set_error("mlpl_arrsac_essential: more than %d samples solved in one call (internal capacity)", kArrPoolSamples);
            return MLPL_E_INTERNAL;
        }
        // layout of the result block: out_nm[B] | out_valid[B*10] | E00[B*10] | head[B*10*2]; models and whole flag rows go to the pool
        const size_t off_valid = (size_t)B * 4, off_e00 = ((size_t)B * 44 + 7) & ~(size_t)7, off_head = off_e00 + (size_t)B * 80;
        const size_t total = off_head + (size_t)B * 10 * kArrHeadWords * 8;
        // the small per-model results are written by the check kernel straight into the pinned (mapped) host block: no copy command
        char *o_base = h_out_dev;
        int32_t *o_nm = (int32_t *)o_base, *o_valid = (int32_t *)(o_base + off_valid);
        double *o_e00 = (double *)(o_base + off_e00);
        unsigned long long *o_head = (unsigned long long *)(o_base + off_head);
        double *o_E = d_Epool + (size_t)pool_samples * 90;
        unsigned long long *o_rows = d_Fpool + (size_t)pool_samples * 10 * kArrFlagWords;
        ArrSampleArgs sa{{B, 1}, d_p1, d_p2, (const int32_t *)d_smp, B, d_recs, d_direct, d_direct_ok, ctx->opt_eig_inverse_iteration};
        L.launch(HK_ARR_SAMPLE, sa);
        RootsArgs ra{{(B + kHypPerWave - 1) / kHypPerWave, 1}, (const PolyRec *)d_recs, B, o_E, d_nm5};
        L.launch(ctx->opt_solver_polish ? HK_ROOTS_POLISH : HK_ROOTS_PLAIN, ra);
        ArrCheckArgs ca{{B * 10, 1}, pts, flag_points, d_p1, d_p2, (const int32_t *)d_smp, B, o_E, (const int32_t *)d_nm5, (const double *)d_direct,
                        (const int32_t *)d_direct_ok, thresh2, o_nm, o_valid, o_e00, o_head, o_rows, d_valid};
        ArrCheckArgs va = ca;
        va.hdr.gx = (B * 10 + 63) / 64;
        L.launch(HK_ARR_VALID, va);
        L.launch(HK_ARR_CHECK, ca);
        (void)total;
        int rcw;
        if ((rcw = L.sync())) return rcw;  // kernel completion makes its system-scope writes visible
        const int32_t *h_nm = (const int32_t *)h_out, *h_valid = (const int32_t *)(h_out + off_valid);
        const double *h_e00 = (const double *)(h_out + off_e00);
        const uint64_t *h_head = (const uint64_t *)(h_out + off_head);
        for (int b = 0; b < B; ++b) {
            ArrIds ids;
            ids.n = 0;
            // order of a sample's solutions: ascending E(0,0) under the sign convention of arrsac_check_kernel.  (The reference's order is
            // cv::solvePoly's root order for a polynomial written in cv::SVD's null-space basis, which rounding noise decides.)
            int order[10], nv = 0;
            const int nm = std::min(h_nm[b], 10);
            for (int i = 0; i < nm; ++i)
                if (h_valid[b * 10 + i]) order[nv++] = i;  // an invalid model's E(0,0) is not written; the order among valid ones is what counts
            std::stable_sort(order, order + nv, [&](int x, int y) { return h_e00[b * 10 + x] < h_e00[b * 10 + y]; });
            for (int oi = 0; oi < nv; ++oi) {
                const int slot = order[oi];
                ArrModelHost mh;
                mh.row = (pool_samples + b) * 10 + slot;
                mh.full = -1;
                std::memcpy(mh.head, h_head + ((size_t)b * 10 + slot) * kArrHeadWords, sizeof(mh.head));
                ids.id[ids.n++] = (int)pool.size();
                pool.push_back(mh);
            }
            cache.emplace(keys[b], ids);
        }
        pool_samples += B;
        stats[8]++;
        stats[9] += B;
        return MLPL_OK;
    }

    // Fetches the whole pool rows (model + all flag words) of the given models, once each: one gather kernel, one host hop.
    int ensure_full(const std::vector<int> &ids) {
        std::vector<int> need;
        for (int id : ids)
            if (pool[id].full == -1) {
                pool[id].full = -2;  // queued
                need.push_back(id);
            }
        const size_t per_full = (out_bytes(kArrBatchCap) - 4096) / (sizeof(ArrFullRow) + 4);
        for (size_t at = 0; at < need.size(); at += per_full) {
            const int cnt = (int)std::min(need.size() - at, per_full);
            // the row list goes out and the gathered rows come back through the pinned (mapped) block: no copy commands
            const size_t off_g = ((size_t)cnt * 4 + 255) & ~(size_t)255;
            int32_t *h_rows = (int32_t *)h_out;
            for (int i = 0; i < cnt; ++i) h_rows[i] = pool[need[at + i]].row;
            ArrGatherArgs ga{{cnt, 1}, (const int32_t *)h_out_dev, cnt, (const double *)d_Epool, (const unsigned long long *)d_Fpool, (double *)(h_out_dev + off_g)};
            L.launch(HK_ARR_GATHER, ga);
            int rcw;
            if ((rcw = L.sync())) return rcw;
            const ArrFullRow *g = (const ArrFullRow *)(h_out + off_g);
            for (int i = 0; i < cnt; ++i) {
                pool[need[at + i]].full = (int)full_rows.size();
                full_rows.push_back(g[i]);
            }
        }
        return MLPL_OK;
    }

    // Bits over all n correspondences for the given models, once each (the preemptive stage past the first kArrFlagPoints).
    int ensure_ext(const std::vector<int> &ids) {
        std::vector<int> need;
        for (int id : ids)
            if (pool[id].ext == -1) {
                pool[id].ext = -2;
                need.push_back(id);
            }
        if (need.empty()) return MLPL_OK;
        const int wf = (n + 63) / 64;
        const size_t cap = std::max(out_bytes(kArrBatchCap), (size_t)wf * 8 + 8192);  // = the pinned block's size (layout)
        const size_t per = std::max<size_t>(1, (cap - 4096) / ((size_t)wf * 8 + 4));
        if ((size_t)wf * 8 + 4096 > cap) {
            set_error("mlpl_arrsac_essential: %d correspondences exceed the staging block of the preemptive stage (internal capacity)", n);
            return MLPL_E_INTERNAL;
        }
        for (size_t at = 0; at < need.size(); at += per) {
            const int cnt = (int)std::min(need.size() - at, per);
            const size_t off_g = ((size_t)cnt * 4 + 255) & ~(size_t)255;
            int32_t *h_rows = (int32_t *)h_out;
            for (int i = 0; i < cnt; ++i) h_rows[i] = pool[need[at + i]].row;
            ArrExtendArgs ea{{cnt, 1}, (const int32_t *)h_out_dev, cnt, pts, n, wf, (const double *)d_Epool, thresh2, (unsigned long long *)(h_out_dev + off_g)};
            L.launch(HK_ARR_EXTEND, ea);
            int rcw;
            if ((rcw = L.sync())) return rcw;
            const uint64_t *g = (const uint64_t *)(h_out + off_g);
            for (int i = 0; i < cnt; ++i) {
                pool[need[at + i]].ext = (int)ext_rows.size();
                ext_rows.emplace_back(g + (size_t)i * wf, g + (size_t)(i + 1) * wf);
            }
        }
        return MLPL_OK;
    }

    bool bit(int id, int i) const {
        const ArrModelHost &m = pool[id];
        if (i < kArrHeadWords * 64) return (m.head[i >> 6] >> (i & 63)) & 1;
        if (i >= flag_points) return (ext_rows[m.ext][i >> 6] >> (i & 63)) & 1;     // callers compute the row first (ensure_ext)
        return (full_rows[m.full].bits[i >> 6] >> (i & 63)) & 1;                    // callers fetch the row first (ensure_full)
    }

    // ProsacSampler::Sample for sample number k over the first `count` correspondences (prosac_sampler.h:85-157).  The growth function
    // (subset size n and T'_n after k turns, Eq. 3-5 of the PROSAC paper as the reference evaluates them) does not depend on the stream:
    // one pass tabulates it for every k the call can reach.
    std::vector<int> prosac_nn;
    std::vector<double> prosac_tnp;
    void prosac_table(int count) {
        prosac_nn.assign(kMaxHyps + 2, kMinSample);
        prosac_tnp.assign(kMaxHyps + 2, 1.0);
        double t_n = 200000;
        int nn = kMinSample;
        for (int i = 0; i < kMinSample; i++) t_n *= static_cast<double>(nn - i) / (double)((size_t)count - i);
        double t_n_prime = 1.0;
        for (int t = 1; t <= kMaxHyps + 1; t++) {
            if (t > t_n_prime && nn < count) {
                const double t_n_plus1 = (t_n * ((double)nn + 1.0)) / ((double)nn + 1.0 - (double)kMinSample);
                t_n_prime += std::ceil(t_n_plus1 - t_n);
                t_n = t_n_plus1;
                nn++;
            }
            prosac_nn[t] = nn, prosac_tnp[t] = t_n_prime;
        }
    }
    void prosac_sample(CvRng &rng, int k, ArrKey &key) const {
        const int nn = prosac_nn[k];
        const double t_n_prime = prosac_tnp[k];
        key.reset(0);
        const bool all_random = t_n_prime < k;
        const int picks = all_random ? kMinSample : kMinSample - 1, range = all_random ? nn : nn - 1;
        for (int i = 0; i < picks; i++) {
            int r;
            do r = rng.uniform(0, range);
            while (key.has(r));
            key.push(r);
        }
        if (!all_random) key.push(nn - 1);
    }
    // RandomSampler::Sample (random_sampler.h:57-78): `size` distinct positions of the universe
    static void random_sample(CvRng &rng, const std::vector<int> &universe, int size, int kind, ArrKey &key) {
        key.reset(kind);
        int used[kArrMaxSample];
        for (int i = 0; i < size; i++) {
            int r;
            do r = rng.uniform(0, (int)universe.size());
            while (std::find(used, used + i, r) != used + i);
            used[i] = r;
            key.push(universe[r]);
        }
    }
    static int to_int_x86(double v) { return (v > -2147483649.0 && v < 2147483648.0) ? (int)v : INT_MIN; }  // cvttsd2si
    static int hyps_needed(double eps, double power) {
        return std::min(kMaxHyps, to_int_x86(std::ceil(std::log(1.0 - kConf) / std::log(1.0 - std::pow(eps, power)))));
    }
    static double sprt_threshold(double sg, double ep, int nmv) {  // sequential_probability_ratio.cc:38-62
        const double c = (1.0 - sg) * std::log((1.0 - sg) / (1.0 - ep)) + sg * std::log(sg / ep);
        const double a_0 = kTimeRatio * c / static_cast<double>(nmv) + 1.0;
        double th = a_0;
        for (int i = 0; i < 1000; i++) {
            const double nt = a_0 + std::log(th);
            const double step = std::fabs(nt - th);
            th = nt;
            if (step < 1e-4) break;
        }
        return th;
    }
    // SequentialProbabilityRatioTest over the first `count` correspondences, on the model's bit row
    bool sprt(int id, int count, double dt, double *ratio, int *num_inl) const {
        *num_inl = 0;
        double lr = 1.0;
        const double up = sigma / epsilon, down = (1.0 - sigma) / (1.0 - epsilon);
        for (int i = 0; i < count; ++i) {
            if (bit(id, i)) {
                lr *= up;
                *num_inl += 1;
            } else {
                lr *= down;
            }
            if (lr > dt) {
                *ratio = static_cast<double>(*num_inl) / static_cast<double>(i + 1);
                return false;
            }
        }
        *ratio = static_cast<double>(*num_inl) / static_cast<double>(count);
        return true;
    }
    void rejected(double ratio) {
        rejected_accum += ratio;
        num_rejected++;
        const double st = rejected_accum / static_cast<double>(num_rejected);
        if (st > 0) sigma = st;
    }

    struct Scored {
        int id;
        double score;
    };
    // diagnostics (mlpl_debug_arrsac_trace): 20 ints per turn of the first stage
    void trace_turn(int k, int inner_turn, const ArrKey &key, int nvalid, const int *res) {
        if (!trace_buf || trace_len + 20 > trace_cap) return;
        int32_t *r = trace_buf + trace_len;
        r[0] = k, r[1] = inner_turn, r[2] = key.size();
        for (int i = 0; i < 5; ++i) r[3 + i] = key.v[1 + i];
        r[8] = nvalid;
        for (int i = 0; i < 10; ++i) r[9 + i] = i < nvalid ? res[i] : -1;
        r[19] = (key.size() > 5 ? key.v[6] : 0) + 100 * (key.size() > 6 ? key.v[7] : 0);  // indices of the first stage are < 100
        trace_len += 20;
    }

    // Arrsac::GenerateInitialHypothesisSet (arrsac.h:236-372)
    int initial_set(int count, std::vector<Scored> &accepted, int *rc_out) {
        int k = 1, k2 = 0, m_prime = kMaxHyps, inner_its = 0, max_num_inliers = 0, random_size = kNonMin;
        bool inner = false;
        std::vector<int> data;
        ArrKey key;
        prosac_table(count);
        while (k <= m_prime) {
            // the sample of this turn, from the real streams
            if (!inner) {
                prosac_sample(prosac_rng, k, key);
                stats[2]++;
            } else {
                const int kind = (random_size == kMinSample || random_size < kNonMinMin) ? 0 : 1;
                random_sample(random_rng, data, random_size, kind, key);
                stats[3]++;
            }
            auto it = cache.find(key);
            if (it == cache.end()) {
                // play the samplers forward under "no event" and solve what is coming in one batch
                std::vector<ArrKey> batch(1, key);
                std::unordered_set<ArrKey, ArrKeyHash> in_batch;
                in_batch.insert(key);
                CvRng pr = prosac_rng, rr = random_rng;
                int kk = k + 1, its = inner_its;
                bool in = inner;
                if (in && ++its == kMaxInner) its = 0, in = false;
                // speculation depth: 128 samples while the device pool is at most half full, 16 afterwards.  An event (at most one per value of
                // the best support, < 100) costs one batch, so the pool of 8192 samples cannot overflow in this stage: 4096 / 128 = 32 deep
                // batches, then 256 shallow ones.
                const int want = pool_samples <= kArrPoolSamples / 2 ? 128 : 16;
                ArrKey nk;
                while (kk <= m_prime && (int)batch.size() < want) {
                    if (!in) {
                        prosac_sample(pr, kk, nk);
                    } else {
                        const int kind = (random_size == kMinSample || random_size < kNonMinMin) ? 0 : 1;
                        random_sample(rr, data, random_size, kind, nk);
                        if (++its == kMaxInner) its = 0, in = false;
                    }
                    kk++;
                    if (cache.find(nk) == cache.end() && in_batch.insert(nk).second) batch.push_back(nk);
                }
                if ((*rc_out = run_batch(batch))) return 0;
                it = cache.find(key);
            }
            stats[10]++;
            const ArrIds &hyps = it->second;
            const int inner_turn = inner ? 1 : 0;
            if (inner) {
                inner_its++;
                if (inner_its == kMaxInner) inner_its = 0, inner = false;
            }
            if (hyps.empty()) {
                trace_turn(k, inner_turn, key, 0, nullptr);
                k2++, k++;
                continue;
            }
            verified_accum += (int)hyps.size();
            const double dt = sprt_threshold(sigma, epsilon, verified_accum / (k - k2));
            int res[10];
            for (size_t j = 0; j < hyps.size(); j++) {
                int num_inl;
                double ratio;
                const bool ok = sprt(hyps[j], count, dt, &ratio, &num_inl);
                if (j < 10) res[j] = (ok ? 1000 : 0) + num_inl;
                if (!ok) {
                    rejected(ratio);
                } else if (num_inl > max_num_inliers) {
                    max_num_inliers = num_inl;
                    accepted.push_back(Scored{hyps[j], (double)num_inl});
                    if (num_inl > kMinSample) {
                        inner = true;
                        inner_its = 0;
                        random_size = std::max(std::min(kNonMin, (int)std::floor((float)max_num_inliers / 2.0f)), kMinSample);
                        data.clear();
                        for (int i = 0; i < count; i++)
                            if (bit(hyps[j], i)) data.push_back(i);
                        epsilon = ratio == 1.0 ? 0.9999 : ratio;
                        m_prime = hyps_needed(epsilon, (double)kMinSample);
                        stats[4]++;
                    }
                } else {
                    accepted.push_back(Scored{hyps[j], (double)num_inl});
                }
            }
            trace_turn(k, inner_turn, key, (int)std::min<size_t>(hyps.size(), 10), res);
            k++;
        }
        if (accepted.empty()) return 0;
        return k - k2 - 1;
    }

    // Arrsac::Estimate (arrsac.h:375-547); returns the pool id of the best model or -1
    int estimate(int *rc_out) {
        const int sub_block = (int)std::floor((float)kBlock / 5.0);
        const int kill_thresh = (int)std::floor(7.0 * (float)sub_block / 12.0);
        std::vector<Scored> hyps;
        int k = initial_set(std::min(n, kBlock), hyps, rc_out);
        stats[0] = k, stats[1] = (long long)hyps.size();
        if (*rc_out || k == 0) return -1;
        if (n <= kBlock) {
            double hi = 0.0;
            int at = 0;
            for (size_t i = 0; i < hyps.size(); i++)
                if (hyps[i].score > hi) hi = hyps[i].score, at = (int)i;
            return hyps[at].id;
        }
        auto cmp = [](Scored a, Scored b) { return a.score > b.score; };  // CompareScoredData (arrsac.h:82-84); std::sort as the reference
        std::vector<int> all(n);
        for (int i = 0; i < n; ++i) all[i] = i;
        int nh = (int)hyps.size();
        {  // the preemptive stage walks the accepted hypotheses' bits beyond the first block: fetch their rows (one host hop)
            std::vector<int> ids;
            for (const Scored &h : hyps) ids.push_back(h.id);
            if ((*rc_out = ensure_full(ids))) return -1;
        }
        int i = kBlock;
        for (; i < n; i++) {
            if (i >= flag_points) {
                // Past the correspondences every model was tested on.  Reached when the hypothesis set GROWS (generation branch, inlier
                // ratios above ~0.9: k climbs towards 500 and nothing halves): the survivors' bits over all n are computed on demand.
                std::vector<int> ids;
                for (const Scored &h : hyps)
                    if (pool[h.id].ext < 0) ids.push_back(h.id);
                if (!ids.empty() && (*rc_out = ensure_ext(ids))) return -1;
            }
            if ((i + 1) % kBlock == 0) {
                std::sort(hyps.begin(), hyps.end(), cmp);
                double max_inliers = hyps[0].score;
                epsilon = max_inliers / static_cast<double>(i + 1);
                if (epsilon == 1.0) epsilon = 0.9999;
                int temp_max = hyps_needed(epsilon, (double)(i + 1));
                if (temp_max > k) {
                    int k2 = 0;
                    ArrKey key;
                    for (int j = 0; j < (long long)temp_max - k; j++) {
                        random_sample(random_rng, all, kMinSample, 0, key);
                        stats[5]++;
                        auto it = cache.find(key);
                        if (it == cache.end()) {  // the uniform sampler's future does not depend on any outcome
                            std::vector<ArrKey> batch(1, key);
                            std::unordered_set<ArrKey, ArrKeyHash> in_batch;
                            in_batch.insert(key);
                            CvRng rr = random_rng;
                            ArrKey nk;
                            for (int jj = j + 1, kk = k + 1; jj < (long long)temp_max - kk && (int)batch.size() < kArrBatchCap; ++jj, ++kk) {
                                random_sample(rr, all, kMinSample, 0, nk);
                                if (cache.find(nk) == cache.end() && in_batch.insert(nk).second) batch.push_back(nk);
                            }
                            if ((*rc_out = run_batch(batch))) return -1;
                            // these are tested on i + 1 > 128 correspondences: their whole rows are needed
                            std::vector<int> ids;
                            for (const ArrKey &bk : batch)
                                for (int id : cache.find(bk)->second) ids.push_back(id);
                            if ((*rc_out = ensure_full(ids))) return -1;
                            if (i + 1 > flag_points && (*rc_out = ensure_ext(ids))) return -1;
                            it = cache.find(key);
                        }
                        stats[10]++;
                        const ArrIds &est = it->second;
                        if (est.empty()) {
                            k2++;
                            continue;
                        }
                        {  // a sample met before (first stage, earlier block) may hold only its leading bits: complete the rows (no-ops otherwise)
                            std::vector<int> ids(est.begin(), est.end());
                            if ((*rc_out = ensure_full(ids))) return -1;
                            if (i + 1 > flag_points && (*rc_out = ensure_ext(ids))) return -1;
                        }
                        verified_accum += (int)est.size();
                        const double dt = sprt_threshold(sigma, epsilon, verified_accum / (k + j + 1 - k2));
                        for (size_t m = 0; m < est.size(); m++) {
                            int num_inl;
                            double ratio;
                            const bool ok = sprt(est[m], i + 1, dt, &ratio, &num_inl);
                            if (!ok) {
                                rejected(ratio);
                            } else if (num_inl > (int)max_inliers) {
                                hyps.push_back(Scored{est[m], (double)num_inl});
                                max_inliers = static_cast<double>(num_inl);
                                epsilon = ratio == 1.0 ? 0.9999 : ratio;
                                temp_max = hyps_needed(epsilon, (double)(i + 1));
                                if (temp_max <= (k + j + 1)) break;
                            } else {
                                hyps.push_back(Scored{est[m], (double)num_inl});
                            }
                        }
                        k++;
                    }
                    nh = (int)hyps.size();
                } else {
                    const int n1 = std::max(1, (int)std::floor((float)k * std::pow(2.0, -1.0 * std::floor((float)(i + 1) / (float)kBlock))));
                    if (n1 < static_cast<int>(hyps.size())) {
                        hyps.resize(n1);
                        nh = n1;
                    }
                }
            } else if ((i + 1) % sub_block == 0) {
                std::sort(hyps.begin(), hyps.end(), cmp);
                int j = nh - 1;
                for (; j > 0; j--)
                    if (hyps[j - 1].score - hyps[j].score > kill_thresh) break;
                nh = nh - j;
                hyps.resize(nh);
            }
            if (nh == 1) break;
            for (size_t j = 0; j < hyps.size(); j++)
                if (bit(hyps[j].id, i)) hyps[j].score += 1.0;
        }
        stats[6] = i, stats[7] = nh;
        return hyps[0].id;
    }
};

}  // namespace

// The estimators of one ARRSAC sample, for tests: all models the device forms for it (before the validity filter) with their flags.
int arrsac_sample_models(mlpl_ctx *ctx, const double *d_p1, const double *d_p2, int n, const int32_t *idx, int m, int kind, double thresh,
                         double *E_out, int32_t *n_models, uint8_t *valid, hipStream_t s) {
    ArrsacRun R;
    R.ctx = ctx, R.s = s, R.L.s = s, R.d_p1 = d_p1, R.d_p2 = d_p2, R.n = n;
    R.flag_points = std::min(n, ctx->opt_arrsac_flag_points ? ctx->opt_arrsac_flag_points : kArrFlagPoints);
    R.thresh2 = thresh * thresh;
    int rc;
    if ((rc = R.alloc())) return rc;
    ArrKey key;
    key.reset(kind);
    for (int i = 0; i < m; ++i) key.push(idx[i]);
    if ((rc = R.run_batch(std::vector<ArrKey>(1, key)))) return rc;
    const int32_t *h_nm = (const int32_t *)R.h_out, *h_valid = (const int32_t *)(R.h_out + 4);  // B = 1: out_nm | out_valid
    const int nm = std::min(h_nm[0], 10);
    *n_models = nm;
    for (int i = 0; i < 10; ++i) valid[i] = i < nm && h_valid[i] ? 1 : 0;
    MLPL_HIP_TRY(hipMemcpy(E_out, R.d_Epool, 720, hipMemcpyDeviceToHost));
    return MLPL_OK;
}

// One ARRSAC problem on a prepared run (Launcher, buffers, trace): estimate, findInliers with the winner, the reference's plausibility test,
// the optional refinement.  Returns 0 / MLPL_E_FAILED / an error; rng_state is advanced only when the estimator itself ran through.
static int arrsac_run_problem(ArrsacRun &R, double thresh, int refine, uint64_t *rng_state, double *E, uint8_t *d_mask, int *n_inliers, long long *stats12) {
    const int n = R.n;
    R.flag_points = std::min(n, R.ctx->opt_arrsac_flag_points ? R.ctx->opt_arrsac_flag_points : kArrFlagPoints);
    R.thresh2 = thresh * thresh;
    R.prosac_rng.state = rng_state[0], R.random_rng.state = rng_state[1];
    int rc;
    if ((rc = R.alloc())) return rc;
    rc = MLPL_OK;
    const int best = R.estimate(&rc);
    if (stats12) std::memcpy(stats12, R.stats, sizeof(R.stats));
    if (rc) return rc;  // an internal limit or a HIP error is not an estimator outcome: the caller's stream states stay where they were
    rng_state[0] = R.prosac_rng.state, rng_state[1] = R.random_rng.state;
    if (n_inliers) *n_inliers = 0;
    if (best < 0) {
        set_error("mlpl_arrsac_essential: no hypothesis passed the sequential test");
        return MLPL_E_FAILED;
    }
    // findInliers with the best model (modelest.cpp:274), then the reference's plausibility test (:275-278).  The winner's matrix is in
    // the device pool already: the count kernel reads it there; the refinement writes next to the counters; one small block comes back
    // through the pinned (mapped) memory.
    double *d_E = R.d_final;
    int32_t *d_info = (int32_t *)(d_E + 18);  // [0] inlier count, [1..2] refinement info
    std::memset(R.h_final, 0, 256);
    hub_copy_bytes(R.L, R.h_final_dev, d_E, 256);  // zeroes the counters (stream-ordered in front of the count kernel)
    MaskCountArgs ma{{(n + 255) / 256, 1}, R.pts, n, (const double *)(R.d_Epool + (size_t)R.pool[best].row * 9), R.thresh2, d_mask, d_info, d_E};
    R.L.launch(HK_ARR_MASK_COUNT, ma);
    if (refine) {  // the kernel itself returns the initial matrix below 50 inliers, as the reference does (modelest.cpp:304-309)
        ArrRefineArgs ra{{1, 1}, R.pts, (const uint8_t *)d_mask, n, (const double *)(R.d_Epool + (size_t)R.pool[best].row * 9), thresh / 50.0, d_E + 9, d_info + 1,
                         (R.ctx->opt_arrsac_refine_warm_start ? 1 : 0) | (R.ctx->opt_eig_inverse_iteration ? 2 : 0)};
        R.L.launch(HK_ARR_REFINE, ra);
    }
    // best model + refined model + counters back: a copy kernel device -> mapped host
    CopyBytesArgs cb{{1, 1}, (const uint4 *)d_E, (uint8_t *)R.h_final_dev, 256};
    R.L.launch(HK_COPY_BYTES, cb);
    if ((rc = R.L.sync())) return rc;
    const double *hE = R.h_final;
    const int32_t *hinfo = (const int32_t *)(hE + 18);
    const int good = hinfo[0];
    if (n_inliers) *n_inliers = good;
    if (stats12) stats12[11] = refine ? hinfo[2] : -1;
    if ((good < 50 && n > 200) || good < 15) {
        set_error("mlpl_arrsac_essential: the best hypothesis has too few inliers (%d)", good);
        return MLPL_E_FAILED;
    }
    // with `refine` the reference always takes the refined matrix (its acceptance test compares the unrefined model's inlier count
    // with itself, modelest.cpp:312-318) and keeps the unrefined model's mask
    std::memcpy(E, (refine && good >= 50) ? hE + 9 : hE, 72);
    return MLPL_OK;
}

int arrsac_essential_dev(mlpl_ctx *ctx, const double *d_p1, const double *d_p2, int n, double thresh, int refine, uint64_t *rng_state,
                         double *E, uint8_t *d_mask, int *n_inliers, hipStream_t s) {
    ArrsacRun R;
    R.ctx = ctx, R.s = s, R.L.s = s, R.d_p1 = d_p1, R.d_p2 = d_p2, R.n = n;
    R.trace_buf = ctx->arrsac_trace, R.trace_cap = ctx->arrsac_trace_cap, R.trace_len = ctx->arrsac_trace_len;
    const int rc = arrsac_run_problem(R, thresh, refine, rng_state, E, d_mask, n_inliers, ctx->last_arrsac_stats);
    ctx->arrsac_trace_len = R.trace_len;
    return rc;
}

// ---- a batch of ARRSAC problems: every run a fiber on a worker thread, every launch merged over the runs (batch_hub.h) -------------------------------
// Problem b: correspondences d_p1 / d_p2 + b * stride * 2 (counts[b] of them), its own pair of cv::RNG states rng_states[2 b .. 2 b + 1]
// (in / out: the reference's samplers draw from process-wide streams, the C ABI hands every problem its own).  Outputs per problem: status
// (0, MLPL_E_FAILED, < 0 errors), E, n_inliers, its mask at d_masks + b * stride.  Every problem's outputs are those of
// mlpl_arrsac_essential_dev on it alone.
constexpr int kArrBatchRuns = 128;

int arrsac_essential_batch_dev(mlpl_ctx *ctx, int B, const double *d_p1, const double *d_p2, int stride, const int32_t *counts, double thresh,
                               int refine, uint64_t *rng_states, double *E, uint8_t *d_masks, int32_t *n_inliers, int32_t *status, hipStream_t s,
                               const CohortFeed *feed = nullptr) {
    if (B <= 0) return MLPL_OK;
    int rc;
    size_t max_dev = 0, max_pin = 0;
    for (int b = 0; b < (feed ? 1 : B); ++b) {  // (a feed delivers the counts later: blocks for `stride` correspondences)
        const ArrsacRun::Layout Y = ArrsacRun::layout(std::max(feed ? stride : counts[b], 6));
        max_dev = std::max(max_dev, Y.dev_total), max_pin = std::max(max_pin, Y.pin_total);
    }
    // cohorts of <= kArrBatchRuns runs, two in flight (batch_hub.h kHubLanes): one cohort's host turns run beside the other's launches
    int n_cohorts = 0, lanes = 0;
    const int cohort = hub_cohort_size(ctx, B, kArrBatchRuns, &n_cohorts, &lanes);
    if (feed && (feed->cohort != cohort || feed->n_cohorts != n_cohorts)) {
        set_error("mlpl_arrsac_essential_batch_dev: the feed's cohorts are not the estimator's");
        return MLPL_E_INTERNAL;
    }
    void *pblk, *dblk;
    if ((rc = pinned_batch_get(ctx, (size_t)lanes * cohort * max_pin, &pblk))) return rc;
    if ((rc = ws_get(ctx, WS_BATCH_RUNS, (size_t)lanes * cohort * max_dev, &dblk))) return rc;
    char *pin = (char *)pblk, *pin_dev = nullptr;
    {
        void *alias = nullptr;
        MLPL_HIP_TRY(hipHostGetDevicePointer(&alias, pin, 0));
        pin_dev = (char *)alias;
    }
    hipStream_t lane_stream[kHubLanes];
    for (int l = 0; l < lanes; ++l)
        if ((rc = hub_lane_stream(ctx, l, s, &lane_stream[l]))) return rc;
    if (lanes > 1 && !feed) MLPL_HIP_TRY(hipStreamSynchronize(s));  // the correspondences were produced on the caller's stream; a lane's own stream has no other ordering
    struct LaneOut {
        long long rounds = 0, merged = 0;
        int first_err = 0;
        std::string first_msg;
    };
    LaneOut lane_out[kHubLanes];
    auto serve_lane = [&](int l) {
        LaneOut &LO = lane_out[l];
        const hipStream_t ls = lane_stream[l];
        if (l > 0 && hipSetDevice(ctx->device) != hipSuccess) {  // a lane's own thread starts on device 0
            LO.first_err = MLPL_E_INTERNAL, LO.first_msg = "batched estimator: hipSetDevice failed on a lane thread";
            return;
        }
        for (int c = l; c < n_cohorts; c += lanes) {
            const int b0 = c * cohort, nb = std::min(cohort, B - b0);
            if (feed) {  // the producer's event (everything this cohort reads on the device is behind it), then its host-side hand-over
                int frc = hipEventSynchronize(feed->ready[c]) == hipSuccess ? MLPL_OK : MLPL_E_HIP;
                if (!frc) frc = feed->on_ready(c, hub_resources(ctx)->lane[l].threads);
                if (frc) {
                    LO.first_err = frc, LO.first_msg = "mlpl_arrsac_essential_batch_dev: the hand-over of a cohort failed";
                    break;
                }
            }
            BatchHub hub(ctx, ls, nb, l);
            std::vector<ArrBufs> bufs((size_t)nb);
            std::vector<std::string> msgs((size_t)nb);
            for (int k = 0; k < nb; ++k) {
                const size_t slot = (size_t)l * cohort + k;
                bufs[k].dev = (char *)dblk + slot * max_dev;
                bufs[k].pin = pin + slot * max_pin, bufs[k].pin_dev = pin_dev + slot * max_pin;
            }
            HubThreads &pool = hub_resources(ctx)->lane[l].threads;
            pool.start(nb, [&](int k) {
                const int b = b0 + k;
                HubRun &hr = hub.run(k);
                int r = MLPL_OK;
                if (counts[b] < 6) {  // (a pair without enough matches in a batch of image pairs: nothing to estimate)
                    status[b] = MLPL_E_FAILED;
                    if (n_inliers) n_inliers[b] = 0;
                    hub.finish(hr);
                    return;
                }
                try {
                    ArrsacRun R;
                    R.ctx = ctx, R.s = ls, R.L.s = ls, R.L.hub = &hub, R.L.run = &hr, R.bufs = &bufs[k];
                    R.d_p1 = d_p1 + (size_t)b * stride * 2, R.d_p2 = d_p2 + (size_t)b * stride * 2, R.n = counts[b];
                    int ninl = 0;
                    r = arrsac_run_problem(R, thresh, refine, rng_states + 2 * (size_t)b, E + (size_t)b * 9, d_masks + (size_t)b * stride, &ninl, nullptr);
                    if (n_inliers) n_inliers[b] = ninl;
                } catch (const std::bad_alloc &) {
                    r = MLPL_E_NOMEM;
                    set_error("mlpl_arrsac_essential_batch_dev: out of host memory");
                } catch (...) {  // (anything else -- a length_error of a vector, say -- must not unwind off the fiber's makecontext frame)
                    r = MLPL_E_INTERNAL;
                    set_error("mlpl_arrsac_essential_batch_dev: a run ended with an unexpected C++ exception");
                }
                if (r && r != MLPL_E_FAILED) msgs[k] = mlpl_last_error();
                status[b] = r;
                hub.finish(hr);
            }, ctx->opt_hub_workers);
            const int hrc = hub.serve();
            pool.wait();
            LO.rounds += hub.rounds(), LO.merged += hub.merged_launches();
            for (int k = 0; k < nb && !LO.first_err; ++k)
                if (status[b0 + k] && status[b0 + k] != MLPL_E_FAILED) LO.first_err = status[b0 + k], LO.first_msg = msgs[k];
            if (hrc && !LO.first_err) LO.first_err = hrc;
            if (LO.first_err) break;
        }
    };
    {
        std::vector<std::thread> others;
        for (int l = 1; l < lanes; ++l) others.emplace_back(serve_lane, l);
        serve_lane(0);
        for (auto &t : others) t.join();
    }
    int first_err = 0;
    std::string first_msg;
    long long rounds = 0, merged = 0;
    for (int l = 0; l < lanes; ++l) {
        rounds += lane_out[l].rounds, merged += lane_out[l].merged;
        if (lane_out[l].first_err && !first_err) first_err = lane_out[l].first_err, first_msg = lane_out[l].first_msg;
    }
    ctx->last_arrsac_stats[8] = rounds, ctx->last_arrsac_stats[9] = merged;
    if (first_err) {
        if (!first_msg.empty()) set_error("%s", first_msg.c_str());
        return first_err;
    }
    return MLPL_OK;
}
