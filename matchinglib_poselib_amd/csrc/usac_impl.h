// usac_impl.h -- USAC (PROSAC + SPRT + LO-RANSAC) essential-matrix estimation with the Nister minimal solver on the MI355X.
// Included by ransac_5pt.hip inside namespace mlpl, after arrsac_impl.h: it reuses solve5pt_kernel / roots_kernel_t, the glibc stream
// and the 9 x 9 Jacobi eigen-solver of that file.
//
// Replaces, under reference poselib/ :
//   source/usac/usac_estimations.cpp:283-470   estimateEssentialMatUsac (POSE_NISTER; configuration: 0.99, 50000 hypotheses, SPRT, LO 5 x 14)
//   include/usac/estimators/USAC.h             solve :335-620 and everything it calls (samplers, SPRT design / history / stopping, LO)
//   include/usac/estimators/EssentialMatEstimator.h   generateMinimalSampleModels :384-520, validateSample :1043, validateModel :1085,
//                                              evaluateModel :1110-1178, generateRefinedModel REFINE_WEIGHTS :540-599, findWeights :2366-2390
//
// USAC is a sequential program: every model is verified by Wald's sequential test, whose start position in a shuffled evaluation order,
// likelihood ratios and decision threshold depend on all earlier verifications.  Its DATA dependence is thin, though: the sample sequence
// is a function of the rand() stream until a model becomes the new best (an "event": local optimisation consumes the stream, PROSAC's
// stopping length changes).  So the host runs the reference's control flow literally and, when it meets a sample it has not seen, plays
// the sampler forward under "no event" and sends the next <= 128 samples to the device as ONE batch:
//   solve5pt_kernel -> roots_kernel_t -> usac_check_kernel (one wave per model: the order key, the oriented-constraint test of
//   validateModel, and the model's inlier BIT for every correspondence in evaluation-pool order, written straight into pinned host
//   memory).  The sequential tests then walk those bit rows on the host (evaluateModel's loop without its arithmetic: the error of a
//   correspondence does not depend on when it is asked for).  An event discards what was speculated, never changes a result.
// Local optimisation (5 inner repetitions x [14-point fit, evaluation, refit on the 2 x threshold inliers, 4 re-weighted refits with a
// shrinking threshold, final evaluation]) runs as ONE launch of usac_lo_kernel, a workgroup per repetition, under the assumption that
// no sequential test inside it rejects (the refined models of a good hypothesis pass); the host verifies that on the bit rows and, where
// a test does reject, resumes that repetition from the reference's state (usac_lo_kernel with a start model).
// Arithmetic that decides inlier bits is the reference's, operation for operation (this file is compiled without FMA contraction).
// Two stated deviations, both where the reference's own numerical routines are inaccurate (tests/test_oracle_usac.py):
// the 5-point models are the exact solutions (OpenGV's Sturm bracketing returns unconverged roots on a share of samples) in the order
// convention of oracle/ref_drivers/usac_ref.cpp, and the 9 x 9 / 3 x 3 decompositions of the refits are Jacobi iterations (ccmath's
// svdu1v / svduv stop early on ~0.2 % of inputs).

namespace {

constexpr int kUsacBatch = 128;
constexpr size_t kUsacCacheBytes = (size_t)256 << 20;  // bound of the host-side sample cache of a run
constexpr int kUsacLoReps = 5, kUsacLoSample = 14, kUsacLoSteps = 4, kUsacLoEvals = 2 + kUsacLoSteps;
constexpr int kUsacLoThreads = 512;
constexpr int kUsacLoMaxRows = 512;  // rows of kUsacLoThreads correspondences a local-optimisation workgroup walks: n <= 262144

struct UsacGeom {
    double T1[9], T2[9], T2t[9], T1i[9], T2ti[9];
};

__device__ __forceinline__ void usac_mul3(double *c, const double *a, const double *b) {  // MathTools::mmul: k ascending from 0.
#pragma unroll
    for (int i = 0; i < 3; ++i)
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            double s = 0.;
#pragma unroll
            for (int k = 0; k < 3; ++k) s += a[3 * i + k] * b[3 * k + j];
            c[3 * i + j] = s;
        }
}
// FTools::normalizePoints' MathTools::vmul on a homogeneous point (x, y, 1)
__device__ __forceinline__ void usac_normalise(const double *T, double x, double y, double *o) {
#pragma unroll
    for (int k = 0; k < 3; ++k) {
        double s = 0.;
        s += T[3 * k] * x;
        s += T[3 * k + 1] * y;
        s += T[3 * k + 2] * 1.0;
        o[k] = s;
    }
}
// evaluateModel's Sampson error (EssentialMatEstimator.h:1133-1139), the reference's operation order
__device__ __forceinline__ double usac_sampson(const double *m, double x1, double y1, double x2, double y2) {
    const double rxc = m[0] * x2 + m[3] * y2 + m[6];
    const double ryc = m[1] * x2 + m[4] * y2 + m[7];
    const double rwc = m[2] * x2 + m[5] * y2 + m[8];
    const double r = (x1 * rxc + y1 * ryc + rwc);
    const double rx = m[0] * x1 + m[1] * y1 + m[2];
    const double ry = m[3] * x1 + m[4] * y1 + m[5];
    return r * r / (rxc * rxc + ryc * ryc + rx * rx + ry * ry);
}

// usac_sampson(...) < thr without the division wherever the answer does not depend on it.  With q = r^2 and D the denominator as the
// reference computes them, the reference's verdict is RN(q / D) < thr.  RN is monotonic and within 2^-53 of q / D, and t = RN(thr D) is
// within 2^-53 of thr D, so q < t (1 - 2^-50) implies RN(q / D) < thr and q > t (1 + 2^-50) implies RN(q / D) > thr; in between (one
// evaluation in ~10^15), and whenever a comparison fails for a NaN or an infinity, the division itself decides.  The IEEE division is a
// quarter of the instructions of an evaluation.
__device__ __forceinline__ bool usac_sampson_less(const double *m, double x1, double y1, double x2, double y2, double thr) {
    const double rxc = m[0] * x2 + m[3] * y2 + m[6];
    const double ryc = m[1] * x2 + m[4] * y2 + m[7];
    const double rwc = m[2] * x2 + m[5] * y2 + m[8];
    const double r = (x1 * rxc + y1 * ryc + rwc);
    const double rx = m[0] * x1 + m[1] * y1 + m[2];
    const double ry = m[3] * x1 + m[4] * y1 + m[5];
    const double q = r * r, D = rxc * rxc + ryc * ryc + rx * rx + ry * ry, t = thr * D;
    if (q < t * (1.0 - 0x1p-50)) return true;
    if (q > t * (1.0 + 0x1p-50)) return false;
    return q / D < thr;
}

struct UsacPoolPackArgs {
    KHdr hdr;
    const double * p1;
    const double * p2;
    const int32_t * pool;
    int n;
    double4 * pts_pool;
};
__device__ __forceinline__ void usac_pool_pack_body(const UsacPoolPackArgs &A, const int vbx, const int vby) {
    const double *__restrict__ p1 = A.p1;
    const double *__restrict__ p2 = A.p2;
    const int32_t *__restrict__ pool = A.pool;
    const int n = A.n;
    double4 *__restrict__ pts_pool = A.pts_pool;

    const int j = vbx * blockDim.x + threadIdx.x;
    if (j >= n) return;
    const int i = pool[j];
    pts_pool[j] = make_double4(p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]);
}
MLPL_HUB_KERNEL(HK_USAC_POOL_PACK, UsacPoolPackArgs, usac_pool_pack_body, 256);

// One wave per (sample, solution slot).  out_* live in pinned, device-mapped host memory.
struct UsacCheckArgs {
    KHdr hdr;
    const double4 * pts_pool;
    int n;
    int words;
    const double * p1;
    const double * p2;
    const int32_t * samples;
    int B;
    const double * E_tab;
    const int32_t * n_models;
    UsacGeom g;
    double thr;
    int32_t * out_nm;
    double * out_key;
    int32_t * out_valid;
    unsigned long long * out_rows;
    double * out_E;
};
__device__ __forceinline__ void usac_check_body(const UsacCheckArgs &A, const int vbx, const int vby) {
    const double4 *__restrict__ pts_pool = A.pts_pool;
    const int n = A.n;
    const int words = A.words;
    const double *__restrict__ p1 = A.p1;
    const double *__restrict__ p2 = A.p2;
    const int32_t *__restrict__ samples = A.samples;
    const int B = A.B;
    const double *__restrict__ E_tab = A.E_tab;
    const int32_t *__restrict__ n_models = A.n_models;
    const UsacGeom g = A.g;
    const double thr = A.thr;
    int32_t *__restrict__ out_nm = A.out_nm;
    double *__restrict__ out_key = A.out_key;
    int32_t *__restrict__ out_valid = A.out_valid;
    unsigned long long *__restrict__ out_rows = A.out_rows;
    double *__restrict__ out_E = A.out_E;

    const int b = vbx / 10, slot = vbx - b * 10;
    const int lane = threadIdx.x;
    if (b >= B) return;
    const int nm = min(n_models[b], 10);
    if (slot == 0 && lane == 0) out_nm[b] = nm;
    if (slot >= nm) return;
    double E[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) E[k] = E_tab[((size_t)b * 10 + slot) * 9 + k];
    if (lane < 9) out_E[((size_t)b * 10 + slot) * 9 + lane] = E_tab[((size_t)b * 10 + slot) * 9 + lane];  // the host's copy of the model
    int valid_model = 0;
    if (lane == 0) {
        // order convention: ascending E(0,0) of the unit-Frobenius matrix whose largest-magnitude element is positive
        double big = 0, n2 = 0;
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            if (fabs(E[k]) > fabs(big)) big = E[k];
            n2 += E[k] * E[k];
        }
        out_key[b * 10 + slot] = (big < 0 ? -E[0] : E[0]) / sqrt(n2);
        // validateModel: oriented epipolar constraint on the five sample points, in normalised coordinates
        double t[9], F[9], e[3];
        usac_mul3(t, g.T2ti, E);
        usac_mul3(F, t, g.T1i);
        e[0] = F[1] * F[8] - F[2] * F[7], e[1] = F[2] * F[6] - F[0] * F[8], e[2] = F[0] * F[7] - F[1] * F[6];  // row 0 x row 2
        bool any = false;
#pragma unroll
        for (int i = 0; i < 3; ++i)
            if ((e[i] > 1.9984e-15) || (e[i] < -1.9984e-15)) any = true;
        if (!any) e[0] = F[4] * F[8] - F[5] * F[7], e[1] = F[5] * F[6] - F[3] * F[8], e[2] = F[3] * F[7] - F[4] * F[6];  // row 1 x row 2
        double sig1 = 0;
        int ok = 1;
        for (int i = 0; i < 5; ++i) {
            const int idx = samples[b * 5 + i];
            double a[3], c[3];
            usac_normalise(g.T1, p1[2 * idx], p1[2 * idx + 1], a);
            usac_normalise(g.T2, p2[2 * idx], p2[2 * idx + 1], c);
            const double s = (F[0] * c[0] + F[3] * c[1] + F[6] * c[2]) * (e[1] * a[2] - e[2] * a[1]);
            if (i == 0)
                sig1 = s;
            else if (sig1 * s < 0) {
                ok = 0;
                break;
            }
        }
        out_valid[b * 10 + slot] = ok;
        valid_model = ok;
    }
    // the control flow never evaluates a model the oriented constraint rejects (validateModel comes first, USAC.h:463-475): no row for it.
    // The rows cross PCIe (pinned host memory): half the models of a batch are rejected here.
    if (!__shfl(valid_model, 0)) return;
    unsigned long long *row = out_rows + ((size_t)b * 10 + slot) * words;
    unsigned long long mine = 0;  // lane l collects word w0 + l: one coalesced 512-byte store per 64 words instead of 64 single stores
    // (Round 5 probe: this loop run TWICE -- the arithmetic of every row doubled, its stores unchanged -- leaves the merged launch at 336-346 us
    // against 303-340: the kernel is bound by its rows crossing PCIe, ~0.58 MB per run and ~295 MB per 512 image pairs, not by the predicate.
    // A packed-fp32 pre-filter as in the RANSAC counting kernel would therefore buy nothing here.  Nor is it the 110 000 workgroups of a cohort's
    // merged launch, most of which find no model in their slot: one four-wave workgroup per SAMPLE, a wave per slot in turns, gives 312 us.
    // Per run 194 samples are solved and 149 consumed with uniform sampling, 186 and 83 with PROSAC: the rows of the rest are speculation.)
    for (int w = 0; w < words; ++w) {
        const int j = w * 64 + lane;
        bool in = false;
        if (j < n) {
            const double4 p = pts_pool[j];
            in = usac_sampson_less(E, p.x, p.y, p.z, p.w, thr);
        }
        const unsigned long long bal = __ballot(in);
        if (lane == (w & 63)) mine = bal;
        if ((w & 63) == 63 || w == words - 1) {
            const int w0 = w & ~63;
            if (w0 + lane <= w) row[w0 + lane] = mine;
        }
    }
}
MLPL_HUB_KERNEL(HK_USAC_CHECK, UsacCheckArgs, usac_check_body, 64);


// ---- degeneracy tests: per-correspondence errors of a rotation / of "no motion" / of a translation / of an upgrade candidate ---------
// EssentialMatEstimator.h: PoseTools::getRotError / getNoMotError (usac/utils/PoseFunctions.cpp:43-141), evaluateModelTrans :1264-1327
// (opengv::triangulation::triangulate2 under (I, t)), evaluateModel :1110-1178.  The adapter's view "1" is the SECOND image
// (CentralRelativeAdapter(bearings2, bearings1), EssentialMatEstimator.h:289-291).
struct UsacDgModel {
    int32_t kind;  // 0 rotation (m = R, row-major), 1 no motion, 2 translation (m[0..2] = t), 3 essential matrix
    int32_t pad;
    double m[9];
};
__device__ __forceinline__ void dg_bearing(double x, double y, double *f) {
    const double nrm = sqrt(x * x + (y * y + 1.0));
    f[0] = x / nrm, f[1] = y / nrm, f[2] = 1.0 / nrm;
}
__device__ __forceinline__ double dg_dot(const double *a, const double *b) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); }
// One workgroup per model: bit rows in evaluation-pool order, rows[(model * 2 + 0) * words ..] = error < thr_pose (kinds 0..2) or
// Sampson error < thr_inl (kind 3); rows[(model * 2 + 1) * words ..] = error < thr_inl (kind 2 only: what storeSolution reads).
struct UsacDgRowsArgs {
    KHdr hdr;
    const double4 * pts_pool;
    int n;
    int words;
    const UsacDgModel * models;
    double thr_pose;
    double thr_inl;
    unsigned long long * rows;
};
__device__ __forceinline__ void usac_degen_rows_body(const UsacDgRowsArgs &A, const int vbx, const int vby) {
    const double4 *__restrict__ pts_pool = A.pts_pool;
    const int n = A.n;
    const int words = A.words;
    const UsacDgModel *__restrict__ models = A.models;
    const double thr_pose = A.thr_pose;
    const double thr_inl = A.thr_inl;
    unsigned long long *__restrict__ rows = A.rows;

    const UsacDgModel M = models[vbx];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    unsigned long long *row_a = rows + ((size_t)vbx * 2) * words, *row_b = row_a + words;
    for (int w = wave; w < words; w += 4) {
        const int j = w * 64 + lane;
        bool in_a = false, in_b = false;
        if (j < n) {
            const double4 p = pts_pool[j];
            if (M.kind == 3) {
                in_a = usac_sampson_less(M.m, p.x, p.y, p.z, p.w, thr_inl);
            } else {
                double f1[3], f2[3], err;
                dg_bearing(p.z, p.w, f1);
                dg_bearing(p.x, p.y, f2);
                if (M.kind == 0) {
                    double u[3];
#pragma unroll
                    for (int r = 0; r < 3; ++r) u[r] = (M.m[3 * r] * f2[0] + M.m[3 * r + 1] * f2[1]) + M.m[3 * r + 2] * f2[2];
                    err = 1.0 - dg_dot(f1, u);
                } else if (M.kind == 1) {
                    err = 1.0 - dg_dot(f1, f2);
                } else {
                    const double *t = M.m;
                    const double b0 = dg_dot(t, f1), b1 = dg_dot(t, f2);
                    const double a00 = dg_dot(f1, f1), a10 = dg_dot(f1, f2), a01 = -a10, a11 = -dg_dot(f2, f2);
                    const double invdet = 1.0 / (a00 * a11 - a10 * a01);
                    const double i00 = a11 * invdet, i10 = -a10 * invdet, i01 = -a01 * invdet, i11 = a00 * invdet;
                    const double l0 = i00 * b0 + i01 * b1, l1 = i10 * b0 + i11 * b1;
                    double pt[3], q[3];
#pragma unroll
                    for (int k = 0; k < 3; ++k) pt[k] = ((l0 * f1[k]) + (t[k] + l1 * f2[k])) / 2;
#pragma unroll
                    for (int k = 0; k < 3; ++k) q[k] = pt[k] + (-t[k]);
                    const double n1 = sqrt(pt[0] * pt[0] + (pt[1] * pt[1] + pt[2] * pt[2])), n2 = sqrt(q[0] * q[0] + (q[1] * q[1] + q[2] * q[2]));
#pragma unroll
                    for (int k = 0; k < 3; ++k) pt[k] = pt[k] / n1, q[k] = q[k] / n2;
                    err = (1.0 - dg_dot(f1, pt)) + (1.0 - dg_dot(f2, q));
                    in_b = err < thr_inl;
                }
                in_a = err < thr_pose;
            }
        }
        const unsigned long long ba = __ballot(in_a), bb = __ballot(in_b);
        if (lane == 0) row_a[w] = ba, row_b[w] = bb;
    }
}
MLPL_HUB_KERNEL(HK_USAC_DG_ROWS, UsacDgRowsArgs, usac_degen_rows_body, 256);

// ---- local optimisation ---------------------------------------------------------------------------------------------------------------
struct UsacLoOut {          // per repetition, pinned host memory; followed by kUsacLoEvals bit rows of `words` words
    int32_t evals;          // evaluations performed (the chain stops when a 2 x threshold inlier set has < 5 members)
    int32_t cnt2;           // members of the 2 x threshold set after the first evaluation (the first refit's point count)
    int32_t sweeps, fits;   // diagnostics: Jacobi sweeps and fits of the chain
    int32_t count[kUsacLoEvals];      // inliers at the threshold per evaluation
    int32_t fit_pts[kUsacLoEvals];    // points the model of evaluation e was fitted to (0 = the start model of a resumed chain)
    double F[kUsacLoEvals][9];        // the evaluated models, normalised coordinates (models_[0])
    double E[kUsacLoEvals][9];        // ... and denormalised (models_denorm_[0])
};
struct UsacLoIn {           // per repetition
    int32_t start_step;     // -1 = from the 14-point sample; j >= 0 = resume at re-weighting step j with the model below
    int32_t sample[kUsacLoSample];
    double F[9], E[9];
};

struct UsacLoLds {
    Jacobi9Lds J;
    double red[8][45];
    double stage[15][kUsacLoThreads];  // usac_reduce45
    double F[9], E[9];
    int scan[kUsacLoThreads / 64 + 1];
    int total, K, sweeps;
    double xv[9], xprev[9], xlambda;  // smallest_eigvec9_wave: this fit's vector, the previous fit's (the next start)
    int x_have, jv_have;              // xprev holds a vector / J.Vv holds the eigenvectors of an earlier Jacobi run of this chain
    int rw[kUsacLoMaxRows * (kUsacLoThreads / 64)];  // members of the fit set per (row, wave), then their exclusive prefix in index order
};

// Fit of REFINE_WEIGHTS from the 45 accumulated products in L.red[0] (wave 0): smallest eigenvector of the covariance matrix, rank-2
// projection (FTools::singulF), denormalisation.
// `warm`: L.J.Vv holds the eigenvectors of the previous fit of this chain.  The covariance matrices of consecutive fits are close (same
// scene, a slightly different inlier set / weights), so the iteration starts from V_prev^T G V_prev, which is nearly diagonal, and
// accumulates its rotations onto V_prev: the same eigen-decomposition to the same tolerance in 2-3 sweeps instead of 8 (a column of V may
// come out with the other sign, which no error or weight sees).
// `inv_iter`: first try smallest_eigvec9_wave (ransac_5pt.hip) from the previous fit's vector -- 3-6 steps of ~0.4 us where the Jacobi
// iteration takes 25-40 us; systems it does not settle on (second-smallest eigenvalue close to the smallest) take the Jacobi path as before.
__device__ __forceinline__ void usac_fit_from_cov(UsacLoLds &L, const UsacGeom &g, int tid, bool warm, bool inv_iter) {
    if (tid < 64) {
        int steps = 0;
        if (inv_iter) steps = smallest_eigvec9_wave(L.red[0], 1.0, (warm && L.x_have) ? L.xprev : nullptr, L.xv, &L.xlambda, tid);  // wave-uniform result
        if (steps > 0) {
            wave_sync();
            if (tid == 0) {
                double F[9], v[3], Fv[3];
                for (int k = 0; k < 9; ++k) F[k] = L.xv[k], L.xprev[k] = L.xv[k];
                L.x_have = 1;
                null_vector_3x3(F, v);
                for (int r = 0; r < 3; ++r) Fv[r] = F[3 * r] * v[0] + F[3 * r + 1] * v[1] + F[3 * r + 2] * v[2];
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) F[3 * r + c] -= Fv[r] * v[c];
                double t[9], E[9];
                usac_mul3(t, g.T2t, F);
                usac_mul3(E, t, g.T1);
                for (int k = 0; k < 9; ++k) L.F[k] = F[k], L.E[k] = E[k];
            }
            return;
        }
        warm = warm && L.jv_have;
        if (!warm) {
            if (tid == 0) {
                int t = 0;
                for (int a = 0; a < 9; ++a)
                    for (int b = a; b < 9; ++b) {
                        L.J.G[a][b] = L.red[0][t];
                        L.J.G[b][a] = L.red[0][t];
                        ++t;
                    }
            }
            for (int e = tid; e < 81; e += 64) L.J.Vv[e / 9][e % 9] = (e / 9 == e % 9) ? 1.0 : 0.0;
        } else {
            // Gn = G V_prev (G read from the packed upper triangle), then G = V_prev^T Gn, upper triangle mirrored
            for (int e = tid; e < 81; e += 64) {
                const int a = e / 9, b = e - a * 9;
                double acc = 0;
                for (int k = 0; k < 9; ++k) {
                    const int lo = a < k ? a : k, hi = a < k ? k : a;
                    acc += L.red[0][lo * 9 - lo * (lo - 1) / 2 + (hi - lo)] * L.J.Vv[k][b];
                }
                L.J.Gn[a][b] = acc;
            }
            wave_sync();
            for (int e = tid; e < 81; e += 64) {
                const int a = e / 9, b = e - a * 9;
                if (a > b) continue;
                double acc = 0;
                for (int k = 0; k < 9; ++k) acc += L.J.Vv[k][a] * L.J.Gn[k][b];
                L.J.G[a][b] = acc;
                L.J.G[b][a] = acc;
            }
        }
        wave_sync();
        const int sweeps = jacobi9_wave(L.J, tid);
        if (tid == 0) {
            L.sweeps += sweeps;
            int m = 0;
            for (int a = 1; a < 9; ++a)
                if (L.J.G[a][a] < L.J.G[m][m]) m = a;
            double F[9], v[3], Fv[3];
            for (int k = 0; k < 9; ++k) F[k] = L.J.Vv[k][m], L.xprev[k] = F[k];
            L.x_have = 1, L.jv_have = 1;
            null_vector_3x3(F, v);
            for (int r = 0; r < 3; ++r) Fv[r] = F[3 * r] * v[0] + F[3 * r + 1] * v[1] + F[3 * r + 2] * v[2];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) F[3 * r + c] -= Fv[r] * v[c];
            double t[9], E[9];
            usac_mul3(t, g.T2t, F);
            usac_mul3(E, t, g.T1);
            for (int k = 0; k < 9; ++k) L.F[k] = F[k], L.E[k] = E[k];
        }
    }
}

// block-wide sum of 45 per-thread accumulators into L.red[0]: in three slices of fifteen, a slice staged value-major in LDS (512
// consecutive doubles per value), thirty-two threads per value add sixteen entries each and finish with five shuffle steps inside their
// half wave.  (Six butterfly steps per value -- 540 ds_bpermute per wave and fit -- cost the kernel ~15 us per fit.)
__device__ __forceinline__ void usac_reduce45(UsacLoLds &L, const double *acc, int tid) {
    const int v = tid >> 5, part = tid & 31;
#pragma unroll
    for (int c = 0; c < 3; ++c) {
#pragma unroll
        for (int k = 0; k < 15; ++k) L.stage[k][tid] = acc[15 * c + k];
        __syncthreads();
        double sacc = 0;
        if (v < 15) {
#pragma unroll
            for (int e = 0; e < kUsacLoThreads / 32; ++e) sacc += L.stage[v][part + 32 * e];
        }
#pragma unroll
        for (int off = 16; off > 0; off >>= 1) sacc += __shfl_xor(sacc, off);
        if (v < 15 && part == 0) L.red[0][15 * c + v] = sacc;
        __syncthreads();
    }
}

__device__ __forceinline__ void usac_cov_add(double *acc, const double *a, const double *c, double w) {
    // row of the data matrix (FTools::computeDataMatrix): entry 3 j + k = x2n[j] * x1n[k]; weighted element by element as the reference
    double q[9];
#pragma unroll
    for (int j = 0; j < 3; ++j)
#pragma unroll
        for (int k = 0; k < 3; ++k) q[3 * j + k] = (c[j] * a[k]) * w;
    int t = 0;
#pragma unroll
    for (int x = 0; x < 9; ++x)
#pragma unroll
        for (int y = x; y < 9; ++y) acc[t++] += q[x] * q[y];
}

struct UsacLoArgs {
    KHdr hdr;
    const double * p1;
    const double * p2;
    const double4 * pts_pool;
    int n;
    int words;
    UsacGeom g;
    double thr;
    double lo_mult;
    const UsacLoIn * in;
    char * out_base;
    size_t out_stride;
    double * err_scratch;
    int warm_start;
};
// -DMLPL_USAC_LO_STAMPS (a probe build, tools/README.md): thread 0 of every workgroup adds the shader-clock ticks it spent in each section of
// the chain to a global table (sections: 0 sample fit, 1 evaluation, 2 bit row, 3 membership + prefix, 4 covariance sums, 5 reduction, 6 fit, 7 workgroups)
#ifdef MLPL_USAC_LO_STAMPS
__device__ unsigned long long g_usac_lo_stamps[8];
#define LO_STAMP(sec)                                                                                  \
    do {                                                                                               \
        if (tid == 0) {                                                                                \
            const unsigned long long now_ = __builtin_amdgcn_s_memtime();                              \
            atomicAdd(&g_usac_lo_stamps[sec], now_ - lo_t_);                                           \
            lo_t_ = now_;                                                                              \
        }                                                                                              \
    } while (0)
#else
#define LO_STAMP(sec) do { } while (0)
#endif
__device__ __forceinline__ void usac_lo_body(const UsacLoArgs &A, const int vbx, const int vby) {
    const double *__restrict__ p1 = A.p1;
    const double *__restrict__ p2 = A.p2;
    const double4 *__restrict__ pts_pool = A.pts_pool;
    const int n = A.n;
    const int words = A.words;
    const UsacGeom g = A.g;
    const double thr = A.thr;
    const double lo_mult = A.lo_mult;
    const UsacLoIn *__restrict__ in = A.in;
    char *__restrict__ out_base = A.out_base;
    const size_t out_stride = A.out_stride;
    double *__restrict__ err_scratch = A.err_scratch;
    const int warm_start = A.warm_start;

    __shared__ UsacLoLds L;
    const int tid = threadIdx.x, rep = vbx;
    const UsacLoIn &I = in[rep];
    UsacLoOut *O = reinterpret_cast<UsacLoOut *>(out_base + (size_t)rep * out_stride);
    unsigned long long *rows = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(O) + sizeof(UsacLoOut));
    double *err = err_scratch + (size_t)rep * n;
    // correspondence i = row * 512 + thread: every pass over the correspondences is coalesced (a contiguous chunk per thread made each
    // wave-wide load touch 64 cache lines, ~10 us per pass on the one CU a repetition runs on)
    const int nrows = (n + kUsacLoThreads - 1) / kUsacLoThreads;
    const int lane = tid & 63, wave = tid >> 6;
    constexpr int kWaves = kUsacLoThreads / 64;
    const double step = (lo_mult * thr - thr) / kUsacLoSteps;
    int eval = 0;
    int fit_pts = 0;
    bool have_fit = false;  // L.J.Vv holds the eigenvectors of an earlier fit of this chain
    if (tid == 0) L.sweeps = 0, L.x_have = 0, L.jv_have = 0;
#ifdef MLPL_USAC_LO_STAMPS
    unsigned long long lo_t_ = __builtin_amdgcn_s_memtime();
    if (tid == 0) atomicAdd(&g_usac_lo_stamps[7], 1ull);
#endif

    if (I.start_step < 0) {  // model of the 14-point sample, unit weights
        double acc[45];
#pragma unroll
        for (int k = 0; k < 45; ++k) acc[k] = 0;
        if (tid < kUsacLoSample) {
            const int i = I.sample[tid];
            double a[3], c[3];
            usac_normalise(g.T1, p1[2 * i], p1[2 * i + 1], a);
            usac_normalise(g.T2, p2[2 * i], p2[2 * i + 1], c);
            usac_cov_add(acc, a, c, 1.0);
        }
        usac_reduce45(L, acc, tid);
        usac_fit_from_cov(L, g, tid, false, (warm_start & 2) != 0);
        have_fit = true;
        fit_pts = kUsacLoSample;
    } else {
        if (tid < 9) L.F[tid] = I.F[tid], L.E[tid] = I.E[tid];
    }
    __syncthreads();
    LO_STAMP(0);

    // phase -1: the evaluation after the sample model + refit on the 2 x threshold set; phases 0..3: re-weighted refits; then the last evaluation
    for (int phase = (I.start_step < 0 ? -1 : I.start_step); phase <= kUsacLoSteps; ++phase) {
        double E[9], F[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) E[k] = L.E[k], F[k] = L.F[k];
        // ---- evaluation: errors in point order, inlier count, bit row in pool order ----
        int cnt = 0;
        for (int k = 0; k < nrows; ++k) {
            const int i = k * kUsacLoThreads + tid;
            if (i < n) {
                const double e = usac_sampson(E, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]);
                err[i] = e;
                cnt += e < thr ? 1 : 0;
            }
        }
        {
            int v = cnt;
#pragma unroll
            for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off);
            if ((tid & 63) == 0) L.scan[tid >> 6] = v;
        }
        __syncthreads();
        LO_STAMP(1);
        if (tid == 0) {
            int s = 0;
            for (int w = 0; w < kUsacLoThreads / 64; ++w) s += L.scan[w];
            L.K = s;
            O->count[eval] = s;
            O->fit_pts[eval] = fit_pts;
            for (int k = 0; k < 9; ++k) O->F[eval][k] = F[k], O->E[eval][k] = E[k];
        }
        {   // the bit row in pool order: the same error from the points gathered in that order (coalesced; err[pool[j]] would be a
            // dependent gather per word)
            unsigned long long *row = rows + (size_t)eval * words;
            for (int w = wave; w < words; w += kWaves) {
                const int j = w * 64 + lane;
                bool inl = false;
                if (j < n) {
                    const double4 p = pts_pool[j];
                    inl = usac_sampson_less(E, p.x, p.y, p.z, p.w, thr);
                }
                const unsigned long long bal = __ballot(inl);
                if (lane == 0) row[w] = bal;
            }
        }
        __syncthreads();
        LO_STAMP(2);
        ++eval;
        if (phase == kUsacLoSteps) break;
        // ---- the point set of the next fit: the first K members (ascending index) of {err < limit} ----
        const double limit = phase < 0 ? lo_mult * thr : (lo_mult * thr) - (phase + 1) * step;
        const int K = phase < 0 ? n : L.K;  // findInliers' own count for the first refit, evaluateModel's count afterwards
        for (int k = 0; k < nrows; ++k) {  // members of the set per (row, wave)
            const int i = k * kUsacLoThreads + tid;
            const unsigned long long b = __ballot(i < n && err[i] < limit);
            if (lane == 0) L.rw[k * kWaves + wave] = __popcll(b);
        }
        __syncthreads();
        if (tid < 64) {  // exclusive prefix of those counts in index order = (row, wave) order; the total
            const int T = nrows * kWaves, per = (T + 63) / 64, b0 = min(tid * per, T), b1 = min(b0 + per, T);
            int sum = 0;
            for (int t = b0; t < b1; ++t) sum += L.rw[t];
            int incl = sum;
#pragma unroll
            for (int off = 1; off < 64; off <<= 1) {
                const int o = __shfl_up(incl, off);
                if (tid >= off) incl += o;
            }
            int run = incl - sum;
            for (int t = b0; t < b1; ++t) {
                const int v = L.rw[t];
                L.rw[t] = run;
                run += v;
            }
            if (tid == 63) L.total = incl;
        }
        __syncthreads();
        LO_STAMP(3);
        const int used = min(L.total, K);
        if (phase < 0 && tid == 0) O->cnt2 = L.total;
        if (used < 5) {  // generateRefinedModel refuses (or, after the first evaluation, the repetition ends: temp_inliers < min sample)
            if (phase < 0) break;
            __syncthreads();
            continue;  // the model stays
        }
        double acc[45];
#pragma unroll
        for (int k = 0; k < 45; ++k) acc[k] = 0;
        for (int k = 0; k < nrows; ++k) {
            if (L.rw[k * kWaves + wave] >= K) break;  // wave-uniform: everything from here on is beyond the first K members
            const int i = k * kUsacLoThreads + tid;
            const bool member = i < n && err[i] < limit;
            const unsigned long long b = __ballot(member);
            if (!member) continue;
            const int rank = L.rw[k * kWaves + wave] + __popcll(b & ((1ull << lane) - 1ull));
            if (rank >= K) continue;
            double a[3], c[3];
            usac_normalise(g.T1, p1[2 * i], p1[2 * i + 1], a);
            usac_normalise(g.T2, p2[2 * i], p2[2 * i + 1], c);
            double w = 1.0;
            if (phase >= 0) {  // findWeights, REFINE_WEIGHTS: Torr's weight from the normalised model
                const double rxc = F[0] * c[0] + F[3] * c[1] + F[6];
                const double ryc = F[1] * c[0] + F[4] * c[1] + F[7];
                const double rx = F[0] * a[0] + F[1] * a[1] + F[2];
                const double ry = F[3] * a[0] + F[4] * a[1] + F[5];
                w = 1 / sqrt(rxc * rxc + ryc * ryc + rx * rx + ry * ry);
            }
            usac_cov_add(acc, a, c, w);
        }
        LO_STAMP(4);
        usac_reduce45(L, acc, tid);
        LO_STAMP(5);
        usac_fit_from_cov(L, g, tid, have_fit && (warm_start & 1) != 0, (warm_start & 2) != 0);
        have_fit = true;
        fit_pts = used;
        __syncthreads();
        LO_STAMP(6);
    }
    if (tid == 0) O->evals = eval, O->sweeps = L.sweeps;
}
// Measured and not kept (round 4): the merged launch compiled for four waves per SIMD (128 VGPRs, staging cut to 48 KB so that two
// workgroups share a compute unit -- at 256 VGPRs a workgroup owns its CU and the 640 workgroups of a cohort run in 2.5 rounds while nothing
// else runs beside them): 295 spilled registers, 1.1 -> 3.2 ms per merged launch, 512 problems 13.6 -> 17.8 ms.  The 45 running sums per thread
// are what the registers hold.
// Round 5, the same question without the spills: the covariance of a fit accumulated in TWO passes over the fit set (24 sums, then 21 -- the
// products and each sum's order of additions unchanged, so the fits came out bit-identical: the records of 512 pairs had the same hash), the
// staging cut to 32 KB, the model in scalar registers; compiled for 128 registers (still 111 spilled dwords, most of them in the one-wave
// eigenvector code) and for 256.  Merged launch of a cohort of 86 runs (430 workgroups), same box, alternating builds: this kernel 747-749 us,
// two passes at 256 registers 806-866 us, at 128 registers (two workgroups per compute unit) 807-940 us -- a second resident workgroup does not
// raise a compute unit's throughput here, the repeated weights and products cost 10-15 %, and the C5 call took 15.0-15.6 ms with all three
// (it is not bound by this kernel's throughput: tools/c5_opt_ab.py under rocprofv3, gpurun_out/r5/usac_lo_kernel_ab3.log).  Not kept.
MLPL_HUB_KERNEL(HK_USAC_LO, UsacLoArgs, usac_lo_body, kUsacLoThreads);

// ---- local optimisation with the refinements of the 5-point family (poselib::RefineAlg REF_STEWENIUS(_WEIGHTS), REF_NISTER(_WEIGHTS)) ------
// EssentialMatEstimator.h generateRefinedModel :640-850, findWeights :2404-2428; P/source/usac/utils/weightingEssential.cpp:56-206
// (fivept_*_weight, computePseudoHuberWeight), P/source/BA_driver.cpp:2639-2648 (costPseudoHuber).  ConfigUSAC's own default is
// REF_STEWENIUS_WEIGHTS (pose_estim.h:99-100).  A fit is OpenGV's five-point solver on ALL points of the set: unit bearing vectors, row
// i = f2_i (x) f1_i, for a re-weighted step of the _WEIGHTS forms scaled by the pseudo-Huber weight of the current model -- the four
// right singular vectors of the smallest singular values span the solver's input, i.e. the four smallest eigenvectors of the 9 x 9 Gram
// matrix -- then, of its real solutions, the one with the smallest Sampson-error sum over the inliers of the best model so far (with the
// reference's early exit).  One exact solver serves Nister and Stewenius as it does for the minimal sample (solve_from_basis +
// roots_kernel_t).  A repetition is a CHAIN of launches with (block of correspondences, chain) as the grid, so that the per-correspondence
// passes use the chip and several chains -- the five repetitions of a local optimisation, later the chains of many problems -- share
// every launch:
//   usac5_begin_kernel                      state; the 14-point sample's Gram matrix
//   [refit_solve_kernel, roots_kernel_t, usac5_choose_kernel]            first model
//   per phase (-1: refit on the 2 x threshold set, 0..3: re-weighted refits): usac5_eval_kernel (errors, inlier bits in pool order,
//   per-block set counts), usac5_gram_kernel (the set's weighted Gram matrix in per-block parts), refit_solve_kernel (parts summed in
//   block order -> Jacobi -> basis -> elimination), roots_kernel_t, usac5_choose_kernel; then the last usac5_eval_kernel.
// As with REF_WEIGHTS the chain assumes that no sequential test inside it rejects; the host replays the reference's logic on the bit
// rows and resumes a chain from the reference's state where one does.  The solution choice reads the inlier flags of the best model,
// which a repetition that stores a new best changes: the host then re-runs the repetitions behind it with the new flags.
constexpr int kLo5Threads = 256, kLo5MaxBlocks = 64;  // rows of 256 correspondences per block: n <= 64 * 16 * 256 (= the REF_WEIGHTS kernel's limit)

struct UsacLo5State {  // device, one per chain
    double E[9];       // current model (denormalised)
    int32_t alive, step_fit, fit_pts, pad;
    int32_t cnt_inl[kLo5MaxBlocks], cnt_mem[kLo5MaxBlocks];
    double warm[82];   // [0] != 0: [1..81] hold the eigenvectors of the chain's previous fit (refit_solve_body starts its iteration there)
};
struct UsacLo5Out {    // pinned host memory, one per chain, followed by kUsacLoEvals bit rows of `words` words
    int32_t evals, cnt2, first_fit, pad;  // first_fit: 1 = the sample's fit gave a model, 2 = it gave none (fresh chains)
    int32_t fit_pts[kUsacLoEvals];        // points the model of evaluation e was fitted to
    int32_t fit_state[kUsacLoEvals];      // the fit that follows evaluation e: 0 = none (fewer than 5 points), 1 = model, 2 = no solution
    double E[kUsacLoEvals][9];            // the evaluated models (denormalised)
    // per fit of the chain (0 = the sample's): solutions found (-1: no fit), the one kept, the solutions -- the host re-takes the choices
    // when the flags of the best model change (UsacRun::lo5_pick_host)
    int32_t hist_nm[kUsacLoEvals], hist_take[kUsacLoEvals];
    double hist_E[kUsacLoEvals][10][9];
};

__device__ __forceinline__ void usac5_row(double x1, double y1, double x2, double y2, double *q) {
    double f1[3], f2[3];
    dg_bearing(x1, y1, f1);
    dg_bearing(x2, y2, f2);
#pragma unroll
    for (int a = 0; a < 3; ++a)
#pragma unroll
        for (int b = 0; b < 3; ++b) q[3 * a + b] = f1[b] * f2[a];
}
// computePseudoHuberWeight(f, fprime, E, th): SampsonL1_Eigen on unit bearing vectors, costPseudoHuber of the distance, times 1 / denominator
__device__ __forceinline__ double usac5_weight(const double *E, double x1, double y1, double x2, double y2, double th) {
    double f[3], fp[3];
    dg_bearing(x1, y1, f);
    dg_bearing(x2, y2, fp);
    double xpE[3];
#pragma unroll
    for (int c = 0; c < 3; ++c) xpE[c] = fp[0] * E[c] + fp[1] * E[3 + c] + fp[2] * E[6 + c];
    const double num = xpE[0] * f[0] + xpE[1] * f[1] + xpE[2] * f[2];
    const double e0 = E[0] * f[0] + E[1] * f[1] + E[2] * f[2], e1 = E[3] * f[0] + E[4] * f[1] + E[5] * f[2];
    const double denom1 = 1 / (sqrt(e0 * e0 + e1 * e1 + xpE[0] * xpE[0] + xpE[1] * xpE[1]) + 1e-8);
    const double d_abs = fabs(num * denom1) + 1e-12, q = d_abs / th;
    return denom1 * (sqrt(2 * (th * th) * (sqrt(1 + q * q) - 1)) / d_abs);
}

// grid = chains, 64 threads.  in[c].start_step < 0: a fresh chain -- Gram matrix of its 14-point sample into part 0; else the chain
// resumes with in[c].E as its model.
struct Usac5BeginArgs {
    KHdr hdr;
    const double * p1;
    const double * p2;
    const UsacLoIn * in;
    UsacLo5State * st;
    double * gram_part;
    size_t part_stride;
    char * out_base;
    size_t out_stride;
};
__device__ __forceinline__ void usac5_begin_body(const Usac5BeginArgs &A, const int vbx, const int vby) {
    const double *__restrict__ p1 = A.p1;
    const double *__restrict__ p2 = A.p2;
    const UsacLoIn *__restrict__ in = A.in;
    UsacLo5State *__restrict__ st = A.st;
    double *__restrict__ gram_part = A.gram_part;
    const size_t part_stride = A.part_stride;
    char *__restrict__ out_base = A.out_base;
    const size_t out_stride = A.out_stride;

    __shared__ double q[kUsacLoSample][9];
    const int c = vbx, lane = threadIdx.x;
    const UsacLoIn &I = in[c];
    UsacLo5State &S = st[c];
    UsacLo5Out *O = reinterpret_cast<UsacLo5Out *>(out_base + (size_t)c * out_stride);
    if (lane == 0) {
        S.alive = 1, S.step_fit = I.start_step < 0 ? 1 : 0, S.fit_pts = I.start_step < 0 ? kUsacLoSample : 0;
        S.warm[0] = 0.0;  // a chain's first fit after (re)start is a cold one
        O->evals = 0, O->cnt2 = 0, O->first_fit = 0;
    }
    if (lane < kUsacLoEvals) O->fit_pts[lane] = 0, O->fit_state[lane] = 0, O->hist_nm[lane] = -1;
    if (I.start_step >= 0) {
        if (lane < 9) S.E[lane] = I.E[lane];
        return;
    }
    if (lane < kUsacLoSample) {
        const int i = I.sample[lane];
        usac5_row(p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1], q[lane]);
    }
    wave_sync();
    if (lane < 45) {
        int a = 0, rem = lane;
        while (rem >= 9 - a) rem -= 9 - a, ++a;
        const int b = a + rem;
        double sacc = 0;
        for (int k = 0; k < kUsacLoSample; ++k) sacc += q[k][a] * q[k][b];
        gram_part[(size_t)c * part_stride + lane] = sacc;
    }
}
MLPL_HUB_KERNEL(HK_USAC5_BEGIN, Usac5BeginArgs, usac5_begin_body, 64);

__device__ __forceinline__ double usac5_wave_min(double v) {
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v = fmin(v, __shfl_xor(v, off));
    return v;
}

// The solution generateRefinedModel keeps (:697-752 / :793-848): lanes j < nm hold solution j (E, and its key in the order convention);
// Sampson-error sums over the inliers of the best model so far in index order, every fourth index a check whether the smallest sum is
// below 0.66 of the second smallest; then std::min_element over the sums in the convention's order.  One wave (the block); returns the lane.
// The sums are one dependent chain per solution and the exit test couples the solutions, but neither needs the ERRORS in sequence: per
// chunk of 64 indices the flagged ones are compacted, lane = correspondence evaluates its error under every solution (LDS), lane =
// solution turns its row into running sums by adding in index order (the reference's additions, one after the other), and lane =
// correspondence again evaluates the exit test on the running sums of its index; the first index whose test holds ends the loop with
// the sums as they stood there.  Same sums, same exit, same choice as the one-index-at-a-time loop (lo5_pick_host keeps that form) at
// one global-load latency per 64 indices instead of one per inlier: 81 -> ~5 us per choice in a merged launch, whose duration is its
// slowest chain's.
struct Usac5PickLds {
    double err[10][64];  // [solution][compacted member]: errors, then running sums
    double E[10][9];
    int idx[64];
};
__device__ __forceinline__ int usac5_pick(const double *E, double key, int nm, int lane, const double *__restrict__ p1,
                                          const double *__restrict__ p2, int n, const uint8_t *__restrict__ flags) {
    if (nm <= 1) return 0;
    __shared__ Usac5PickLds L;
    int pos = 0;  // position in the convention (ascending key, stable): it only breaks ties of the error sums
    for (int k = 0; k < nm; ++k) {
        const double kk = __shfl(key, k);
        pos += (kk < key || (kk == key && k < lane)) ? 1 : 0;
    }
    if (lane < nm) {
#pragma unroll
        for (int k = 0; k < 9; ++k) L.E[lane][k] = E[k];
    }
    wave_sync();
    double sum = 0;
    for (int i0 = 0; i0 < n; i0 += 64) {
        const int ii = i0 + lane;
        const bool in_range = ii < n;
        const uint8_t f = in_range ? flags[ii] : (uint8_t)0;
        double x1 = 0, y1 = 0, x2 = 0, y2 = 0;
        if (in_range) x1 = p1[2 * ii], y1 = p1[2 * ii + 1], x2 = p2[2 * ii], y2 = p2[2 * ii + 1];
        const unsigned long long mask = __ballot(f != 0);
        if (!mask) continue;
        const int cnt = __popcll(mask);
        if (f != 0) {
            const int r = __popcll(mask & ((1ull << lane) - 1ull));
            L.idx[r] = ii;
            for (int sidx = 0; sidx < nm; ++sidx) L.err[sidx][r] = usac_sampson(L.E[sidx], x1, y1, x2, y2);
        }
        wave_sync();
        if (lane < nm) {
            double run = sum;
            for (int r = 0; r < cnt; ++r) {
                run += L.err[lane][r];
                L.err[lane][r] = run;
            }
        }
        wave_sync();
        bool hit = false;
        if (lane < cnt) {
            const int i = L.idx[lane];
            if ((i > 3) && (i % 4 == 0)) {
                // smallest sum, the first solution holding it, smallest of the others (a lane without a solution counts as +inf, as in
                // the wave-wide form; fmin ignores a NaN sum as it did there)
                double m1 = INFINITY;
                for (int sidx = 0; sidx < nm; ++sidx) m1 = fmin(m1, L.err[sidx][lane]);
                int first = nm;
                for (int sidx = nm - 1; sidx >= 0; --sidx)
                    if (L.err[sidx][lane] == m1) first = sidx;
                double m2 = INFINITY;
                for (int sidx = 0; sidx < nm; ++sidx)
                    if (sidx != first) m2 = fmin(m2, L.err[sidx][lane]);
                hit = m1 < 0.66 * m2;
            }
        }
        const unsigned long long hits = __ballot(hit);
        const int last = hits ? __ffsll((long long)hits) - 1 : cnt - 1;
        if (lane < nm) sum = L.err[lane][last];
        wave_sync();
        if (hits) break;
    }
    const double v = lane < nm ? sum : INFINITY;
    const double m1 = usac5_wave_min(v);
    int cand = (lane < nm && v == m1) ? pos : 64;
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) cand = min(cand, __shfl_xor(cand, off));
    const unsigned long long who = __ballot(lane < nm && pos == cand);
    return who ? __ffsll((long long)who) - 1 : 0;  // (all sums NaN: solution 0, which no test can tell from any other)
}
__device__ __forceinline__ double usac5_key(const double *E) {  // ascending E(0,0) of the unit-Frobenius matrix whose largest-magnitude element is positive
    double big = 0, n2 = 0;
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        if (fabs(E[k]) > fabs(big)) big = E[k];
        n2 += E[k] * E[k];
    }
    return (big < 0 ? -E[0] : E[0]) / sqrt(n2);
}

// grid = chains, 64 threads: of the solutions roots_kernel_t left for the chain's system, the one generateRefinedModel keeps.
// `fit_eval` = index of the evaluation this fit follows, -1 for the sample's fit; `ends_chain`: a fit without a solution ends the
// repetition (the sample's fit and the refit on the 2 x threshold set; in a re-weighted step the model stays).  The solutions and the
// choice are kept per (chain, fit) in the output block: the host re-takes the choices when the flags change (UsacRun::lo5_pick_host).
struct Usac5ChooseArgs {
    KHdr hdr;
    const double * p1;
    const double * p2;
    int n;
    const uint8_t * flags;
    const double * E_tab;
    const int32_t * n_models;
    UsacLo5State * st;
    char * out_base;
    size_t out_stride;
    int fit_eval;
    int ends_chain;
};
__device__ __forceinline__ void usac5_choose_body(const Usac5ChooseArgs &A, const int vbx, const int vby) {
    const double *__restrict__ p1 = A.p1;
    const double *__restrict__ p2 = A.p2;
    const int n = A.n;
    const uint8_t *__restrict__ flags = A.flags;
    const double *__restrict__ E_tab = A.E_tab;
    const int32_t *__restrict__ n_models = A.n_models;
    UsacLo5State *__restrict__ st = A.st;
    char *__restrict__ out_base = A.out_base;
    const size_t out_stride = A.out_stride;
    const int fit_eval = A.fit_eval;
    const int ends_chain = A.ends_chain;

    const int c = vbx, lane = threadIdx.x;
    UsacLo5State &S = st[c];
    UsacLo5Out *O = reinterpret_cast<UsacLo5Out *>(out_base + (size_t)c * out_stride);
    const int fit = fit_eval + 1;
    if (!S.alive || !S.step_fit) {
        if (lane == 0) O->hist_nm[fit] = -1;
        return;
    }
    const int nm = min(n_models[c], 10);
    if (lane == 0) O->hist_nm[fit] = nm;
    if (nm <= 0) {
        if (lane == 0) {
            if (fit_eval < 0) O->first_fit = 2;
            else O->fit_state[fit_eval] = 2;
            if (ends_chain) S.alive = 0;
        }
        return;
    }
    double E[9] = {0, 0, 0, 0, 0, 0, 0, 0, 0};
    double key = INFINITY;
    if (lane < nm) {
#pragma unroll
        for (int k = 0; k < 9; ++k) {
            E[k] = E_tab[((size_t)c * 10 + lane) * 9 + k];
            O->hist_E[fit][lane][k] = E[k];
        }
        key = usac5_key(E);
    }
    const int take = usac5_pick(E, key, nm, lane, p1, p2, n, flags);
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const double v = __shfl(E[k], take);
        if (lane == 0) S.E[k] = v;
    }
    if (lane == 0) {
        O->hist_take[fit] = take;
        if (fit_eval < 0) O->first_fit = 1;
        else O->fit_state[fit_eval] = 1;
    }
}
MLPL_HUB_KERNEL(HK_USAC5_CHOOSE, Usac5ChooseArgs, usac5_choose_body, 64);

// A chain's whole fit in ONE launch (round 5): refit_solve_body -> roots_body -> usac5_choose_body of chain vbx, one after the
// other in the chain's wave.  The three were launches of their own with one wave per chain (roots: six chains per wave), each waiting for
// the one before; the data they hand over (the chain's PolyRec, its <= 10 solutions) is written and read back by the same wave.  The
// arithmetic is that of the separate kernels, body for body (a hypothesis's lanes in roots_body never look at another hypothesis), so
// the models, choices and traces are the same bit for bit (option usac_lo5_fused_fit = 0: the three launches, for A/B and tests).
struct Usac5FitArgs {
    KHdr hdr;
    const double * gram_part;
    int nparts;
    PolyRec * recs;
    size_t part_stride;
    int warm_on;
    int polish;
    double * E_tab;
    int32_t * n_models;
    Usac5ChooseArgs ch;
};
__device__ __forceinline__ void usac5_fit_body(const Usac5FitArgs &A, const int vbx, const int vby) {
    const int c = vbx;
    char *stb = reinterpret_cast<char *>(A.ch.st);
    refit_solve_body(A.gram_part, A.nparts, A.recs, A.part_stride, stb + offsetof(UsacLo5State, step_fit), sizeof(UsacLo5State), c,
                     A.warm_on ? stb + offsetof(UsacLo5State, warm) : nullptr, sizeof(UsacLo5State));
    // (the record and the solutions go through memory as between the launches: the wave's own stores, ordered before its loads)
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
    wave_sync();
    if (A.polish) roots_body<true>(A.recs + c, 0, 1, A.E_tab + (size_t)c * 90, A.n_models + c, nullptr, nullptr, nullptr, nullptr, 0, 0);
    else roots_body<false>(A.recs + c, 0, 1, A.E_tab + (size_t)c * 90, A.n_models + c, nullptr, nullptr, nullptr, nullptr, 0, 0);
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "agent");
    wave_sync();
    usac5_choose_body(A.ch, c, vby);
}
MLPL_HUB_KERNEL(HK_USAC5_FIT, Usac5FitArgs, usac5_fit_body, 64);


// grid = (blocks, chains), 256 threads.  Evaluation `e` of the chain's model: errors in point order, per-block counts of the inliers and of
// the members of {err < limit}, the inlier bit row in pool order.
struct Usac5EvalArgs {
    KHdr hdr;
    const double * p1;
    const double * p2;
    const double4 * pts_pool;
    int n;
    int words;
    int rows_per_block;
    double thr;
    double limit;
    UsacLo5State * st;
    double * err_all;
    char * out_base;
    size_t out_stride;
    int e;
};
__device__ __forceinline__ void usac5_eval_body(const Usac5EvalArgs &A, const int vbx, const int vby) {
    const double *__restrict__ p1 = A.p1;
    const double *__restrict__ p2 = A.p2;
    const double4 *__restrict__ pts_pool = A.pts_pool;
    const int n = A.n;
    const int words = A.words;
    const int rows_per_block = A.rows_per_block;
    const double thr = A.thr;
    const double limit = A.limit;
    UsacLo5State *__restrict__ st = A.st;
    double *__restrict__ err_all = A.err_all;
    char *__restrict__ out_base = A.out_base;
    const size_t out_stride = A.out_stride;
    const int e = A.e;

    __shared__ int red[2][kLo5Threads / 64];
    const int c = vby, b = vbx, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    UsacLo5State &S = st[c];
    if (!S.alive) return;
    UsacLo5Out *O = reinterpret_cast<UsacLo5Out *>(out_base + (size_t)c * out_stride);
    unsigned long long *rows = reinterpret_cast<unsigned long long *>(reinterpret_cast<char *>(O) + sizeof(UsacLo5Out));
    double *err = err_all + (size_t)c * n;
    double E[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) E[k] = S.E[k];
    int inl = 0, mem = 0;
    for (int r = 0; r < rows_per_block; ++r) {
        const int i = (b * rows_per_block + r) * kLo5Threads + tid;
        if (i < n) {
            const double ev = usac_sampson(E, p1[2 * i], p1[2 * i + 1], p2[2 * i], p2[2 * i + 1]);
            err[i] = ev;
            inl += ev < thr ? 1 : 0;
            mem += ev < limit ? 1 : 0;
        }
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) inl += __shfl_xor(inl, off), mem += __shfl_xor(mem, off);
    if (lane == 0) red[0][wave] = inl, red[1][wave] = mem;
    __syncthreads();
    if (tid == 0) {
        int a = 0, m = 0;
        for (int w = 0; w < kLo5Threads / 64; ++w) a += red[0][w], m += red[1][w];
        S.cnt_inl[b] = a, S.cnt_mem[b] = m;
        if (b == 0) {
            O->evals = e + 1;
            O->fit_pts[e] = S.fit_pts;
            for (int k = 0; k < 9; ++k) O->E[e][k] = E[k];
        }
    }
    unsigned long long *row = rows + (size_t)e * words;
    for (int w = b * (kLo5Threads / 64) + wave; w < words; w += A.hdr.gx * (kLo5Threads / 64)) {
        const int j = w * 64 + lane;
        bool in = false;
        if (j < n) {
            const double4 p = pts_pool[j];
            in = usac_sampson_less(E, p.x, p.y, p.z, p.w, thr);
        }
        const unsigned long long bal = __ballot(in);
        if (lane == 0) row[w] = bal;
    }
}
MLPL_HUB_KERNEL(HK_USAC5_EVAL, Usac5EvalArgs, usac5_eval_body, kLo5Threads);

// grid = (blocks, chains), 256 threads.  The fit set of this step: the first K members (ascending index) of {err < limit}, K = n for the
// refit after the first evaluation (findInliers' own count), the inlier count of the evaluation afterwards (USAC.h:1027-1040).  Each
// block adds the (weighted) rows of its members into its own 45-value part; a block without members writes zeros.
struct Usac5GramArgs {
    KHdr hdr;
    const double * p1;
    const double * p2;
    int n;
    int rows_per_block;
    double limit;
    int first_refit;
    int weighted;
    double th_ph;
    UsacLo5State * st;
    const double * err_all;
    double * gram_part;
    size_t part_stride;
    char * out_base;
    size_t out_stride;
    int e;
};
__device__ __forceinline__ void usac5_gram_body(const Usac5GramArgs &A, const int vbx, const int vby) {
    const double *__restrict__ p1 = A.p1;
    const double *__restrict__ p2 = A.p2;
    const int n = A.n;
    const int rows_per_block = A.rows_per_block;
    const double limit = A.limit;
    const int first_refit = A.first_refit;
    const int weighted = A.weighted;
    const double th_ph = A.th_ph;
    UsacLo5State *__restrict__ st = A.st;
    const double *__restrict__ err_all = A.err_all;
    double *__restrict__ gram_part = A.gram_part;
    const size_t part_stride = A.part_stride;
    char *__restrict__ out_base = A.out_base;
    const size_t out_stride = A.out_stride;
    const int e = A.e;

    __shared__ int cnt[16 * (kLo5Threads / 64) + 1];
    __shared__ double red[kLo5Threads / 64][45];
    const int c = vby, b = vbx, tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    constexpr int kWaves = kLo5Threads / 64;
    UsacLo5State &S = st[c];
    if (!S.alive) {
        if (b == 0 && tid == 0) S.step_fit = 0;
        return;
    }
    UsacLo5Out *O = reinterpret_cast<UsacLo5Out *>(out_base + (size_t)c * out_stride);
    const double *err = err_all + (size_t)c * n;
    int total = 0, K = 0, before = 0;
    for (int k = 0; k < A.hdr.gx; ++k) {
        const int m = S.cnt_mem[k];
        total += m, K += S.cnt_inl[k];
        before += k < b ? m : 0;
    }
    if (first_refit) K = n;
    const int used = min(total, K);
    double *part = gram_part + (size_t)c * part_stride + (size_t)b * 45;
    if (used < 5) {  // generateRefinedModel refuses; after the first evaluation the repetition ends
        if (b == 0 && tid == 0) {
            if (first_refit) O->cnt2 = total, S.alive = 0;
            S.step_fit = 0;
            O->fit_state[e] = 0;
        }
        return;
    }
    if (b == 0 && tid == 0) {
        if (first_refit) O->cnt2 = total;
        S.step_fit = 1, S.fit_pts = used;
    }
    double E[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) E[k] = S.E[k];
    // members of this block per (row, wave) in index order, then their exclusive prefix
    for (int r = 0; r < rows_per_block; ++r) {
        const int i = (b * rows_per_block + r) * kLo5Threads + tid;
        const unsigned long long bal = __ballot(i < n && err[i] < limit);
        if (lane == 0) cnt[r * kWaves + wave] = __popcll(bal);
    }
    __syncthreads();
    if (tid == 0) {
        int run = before;
        for (int t = 0; t < rows_per_block * kWaves; ++t) {
            const int v = cnt[t];
            cnt[t] = run;
            run += v;
        }
    }
    __syncthreads();
    double acc[45];
#pragma unroll
    for (int k = 0; k < 45; ++k) acc[k] = 0;
    for (int r = 0; r < rows_per_block; ++r) {
        if (cnt[r * kWaves + wave] >= K) break;  // wave-uniform
        const int i = (b * rows_per_block + r) * kLo5Threads + tid;
        const bool member = i < n && err[i] < limit;
        const unsigned long long bal = __ballot(member);
        if (!member) continue;
        const int rank = cnt[r * kWaves + wave] + __popcll(bal & ((1ull << lane) - 1ull));
        if (rank >= K) continue;
        const double x1 = p1[2 * i], y1 = p1[2 * i + 1], x2 = p2[2 * i], y2 = p2[2 * i + 1];
        double q[9];
        usac5_row(x1, y1, x2, y2, q);
        if (weighted) {
            const double w = usac5_weight(E, x1, y1, x2, y2, th_ph);
#pragma unroll
            for (int k = 0; k < 9; ++k) q[k] *= w;
        }
        int t = 0;
#pragma unroll
        for (int x = 0; x < 9; ++x)
#pragma unroll
            for (int y = x; y < 9; ++y) acc[t++] += q[x] * q[y];
    }
    // wave-wide sums of the 45 accumulators by a transposing butterfly: at distance `off` a lane keeps one half of the values it still
    // holds and hands the other half to its partner (23 + 12 + 6 + 3 + 2 + 1 = 47 exchanges instead of 45 x 6); afterwards lane l holds
    // the total of value usac5_red_slot(l) -- a fixed order of additions, the same in every launch form
    {
        double buf[48];
#pragma unroll
        for (int k = 0; k < 48; ++k) buf[k] = k < 45 ? acc[k] : 0.0;
        int len = 48;
#pragma unroll
        for (int off = 32; off > 0; off >>= 1) {
            const int half = len >> 1;  // 24, 12, 6, 3 -> then 3 is odd: handled below
            if (len & 1) break;
            const bool upper = (lane & off) != 0;
#pragma unroll
            for (int k = 0; k < 24; ++k) {
                if (k >= half) break;
                const double send = upper ? buf[k] : buf[k + half];
                const double keep = upper ? buf[k + half] : buf[k];
                buf[k] = keep + __shfl_xor(send, off);
            }
            len = half;
        }
        // len == 3 after the distances 32, 16, 8, 4: lanes with equal (lane >> 2) hold partial sums of the same three values
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            buf[k] += __shfl_xor(buf[k], 2);
            buf[k] += __shfl_xor(buf[k], 1);
        }
        // value index held by this lane group: bit 5 of the lane selects the upper 24 of 48, bit 4 the upper 12 of those, ...
        if ((lane & 3) == 0) {
            const int base = ((lane >> 5) & 1) * 24 + ((lane >> 4) & 1) * 12 + ((lane >> 3) & 1) * 6 + ((lane >> 2) & 1) * 3;
#pragma unroll
            for (int k = 0; k < 3; ++k)
                if (base + k < 45) red[wave][base + k] = buf[k];
        }
    }
    __syncthreads();
    if (tid < 45) part[tid] = (red[0][tid] + red[1][tid]) + (red[2][tid] + red[3][tid]);
}
MLPL_HUB_KERNEL(HK_USAC5_GRAM, Usac5GramArgs, usac5_gram_body, kLo5Threads);

struct UsacWald {
    double epsilon, delta, A;
    unsigned k;
};
struct UsacKey {
    uint32_t v[5];
    bool operator==(const UsacKey &o) const { return std::memcmp(v, o.v, sizeof(v)) == 0; }
};
struct UsacKeyHash {
    size_t operator()(const UsacKey &k) const {
        uint64_t h = 1469598103934665603ull;
        for (int i = 0; i < 5; ++i) h = (h ^ k.v[i]) * 1099511628211ull;
        return (size_t)h;
    }
};
struct UsacModel {  // one minimal model as the host holds it
    double E[9];
    int valid;
    const uint64_t *bits;  // pool order; in the run's row arena (UsacRun::row_alloc), null for a rejected model
};
// Bit rows of the cached models: chunks that are kept and refilled when the cache is dropped (a heap allocation per model -- 250 per
// batch of samples -- was a tenth of a run's host time).
struct UsacRowArena {
    static constexpr size_t kChunkWords = 1u << 16;
    std::vector<std::unique_ptr<uint64_t[]>> chunks;
    size_t chunk = 0, used = 0;
    uint64_t *alloc(size_t words) {
        if (words > kChunkWords) return nullptr;
        if (chunks.empty() || used + words > kChunkWords) {
            if (!chunks.empty()) ++chunk;
            if (chunk >= chunks.size()) chunks.emplace_back(new uint64_t[kChunkWords]);
            used = 0;
        }
        uint64_t *p = chunks[chunk].get() + used;
        used += words;
        return p;
    }
    void reset() { chunk = 0, used = 0; }
};
struct UsacSampleModels {
    int n = 0;
    UsacModel m[10];
};

unsigned usac_to_uint(double v) {  // (unsigned int) of a double as x86-64 compilers convert: cvttsd2si to 64 bits, low half
    const long long w = (v > -9.2233720368547758e18 && v < 9.2233720368547758e18) ? (long long)v : (long long)0x8000000000000000ull;
    return (unsigned)(unsigned long long)w;
}

inline std::mutex &usac_prosac_tab_mutex() {
    static std::mutex m;
    return m;
}

struct UsacBufs {  // a run's device / pinned buffers when the caller provides them (batched runs: one slice per run, usac_batch.h)
    char *dev = nullptr;      // device block of dev_bytes(n, refine)
    char *pin = nullptr;      // pinned, device-mapped block of pin_bytes(...)
    char *pin_dev = nullptr;  // its device alias
};

struct UsacRun {
    mlpl_ctx *ctx;
    hipStream_t s;
    Launcher L;                  // launches now (a run alone) or records for the hub (a run of a batch)
    const UsacBufs *bufs = nullptr;
    double *trace_buf = nullptr;  // decision trace of this run (batched runs: one buffer per run; else the context's)
    int trace_cap = 0, trace_len = 0;
    const double *d_p1, *d_p2;
    std::vector<double> hp1, hp2;  // host copies (sample validation, normalisation)
    unsigned n = 0, max_hyp = 50000;
    double conf = 0.99, thr = 0;
    bool prosac = false;
    bool sprt_fast = true;  // option usac_sprt_fast (tests run both)
    unsigned prosac_max_samples = 1000, prosac_min_stop = 20;
    double prosac_beta = 0.09, prosac_non_rand_conf = 0.99;
    std::vector<unsigned> sorted_idx;
    double sprt_tM = 2314.0, sprt_mS = 8.5, sprt_delta = 0.05, sprt_epsilon = 0.15, sprt_A = 0;
    double lo_mult = 2.0;
    int lo_stepwise = 0;
    UsacGeom g;
    GlibcRand rng;
    int words = 0;
    // device / pinned buffers
    double4 *d_pts_pool = nullptr;
    int32_t *h_smp = nullptr, *d_smp = nullptr;  // pinned + its device alias
    char *h_out = nullptr, *h_out_dev = nullptr;
    PolyRec *d_recs = nullptr;
    double *d_Etab = nullptr;
    int32_t *d_nm = nullptr;
    double *d_err = nullptr;
    UsacLoIn *h_lo_in = nullptr, *d_lo_in = nullptr;
    int refine = 0;  // poselib::RefineAlg: 0 = REF_WEIGHTS (usac_lo_kernel), 4..7 = the 5-point family (usac5_* chains)
    UsacLo5State *d_lo5_state = nullptr;
    double *d_lo5_gram = nullptr;
    uint8_t *d_lo5_flags = nullptr, *h_lo5_flags = nullptr;
    const uint8_t *d_lo5_flags_src = nullptr;  // device alias of h_lo5_flags
    int32_t *h_lo5_diff = nullptr, *d_lo5_diff = nullptr;
    int lo5_blocks = 1, lo5_rows = 1;
    int batch_cap = kUsacBatch;
    // state
    std::vector<unsigned> min_sample, pool;
    unsigned pool_index = 0;
    std::vector<UsacWald> history;
    unsigned last_wald_update = 0;
    unsigned subset_size = 5, largest_size = 5, stop_len = 0;
    std::vector<unsigned> growth, non_random, maximality;
    std::unordered_map<UsacKey, UsacSampleModels, UsacKeyHash> cache;
    double cache_thr = 0;
    UsacRowArena rows_arena;
    size_t cache_bytes = 0;  // bit rows held by the cache; bounded by kUsacCacheBytes (a run that never stops early would keep 50000 samples' rows)
    // results
    unsigned hyp_count = 0, model_count = 0, rejected_samples = 0, rejected_models = 0, best = 0, points_verified = 0, num_lo = 0;
    std::vector<uint8_t> flags;
    std::vector<uint64_t> best_bits;  // pool order: the errors of the best model as far as anything reads them
    double final_model[9] = {0};
    long long stats[8] = {0};  // [0] batches, [1] samples solved, [2] samples consumed, [3] LO launches, [4] LO resumes

    void emit(double type, const double *v, int nv) {
        if (!trace_buf) return;
        if (trace_len < trace_cap) {
            double *r = trace_buf + (size_t)trace_len * 16;
            std::memset(r, 0, 128);
            r[0] = type;
            for (int i = 0; i < nv && i < 15; ++i) r[1 + i] = v[i];
        }
        trace_len++;
    }

    static void mul3(double *c, const double *a, const double *b) {
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double sacc = 0.;
                for (int k = 0; k < 3; ++k) sacc += a[3 * i + k] * b[3 * k + j];
                c[3 * i + j] = sacc;
            }
    }
    static void inv_similarity(const double *T, double *Ti) {
        std::memset(Ti, 0, 72);
        const double sc = T[0];
        Ti[0] = 1.0 / sc, Ti[4] = 1.0 / sc, Ti[8] = 1.0;
        if (T[2] != 0 || T[5] != 0) Ti[2] = -T[2] / sc, Ti[5] = -T[5] / sc;
        if (T[6] != 0 || T[7] != 0) Ti[6] = -T[6] / sc, Ti[7] = -T[7] / sc;
    }

    // FTools::normalizePoints (FundmatrixFunctions.cpp:7-62): means and mean distances in the reference's (sequential) summation order
    void normalisation() {
        double *T1 = g.T1, *T2 = g.T2;
        std::memset(T1, 0, 72), std::memset(T2, 0, 72);
        double m1[2] = {0, 0}, m2[2] = {0, 0};
        for (unsigned i = 0; i < n; ++i) m1[0] += hp1[2 * i], m1[1] += hp1[2 * i + 1], m2[0] += hp2[2 * i], m2[1] += hp2[2 * i + 1];
        m1[0] /= (double)n, m2[0] /= (double)n, m1[1] /= (double)n, m2[1] /= (double)n;
        double d1 = 0, d2 = 0;
        for (unsigned i = 0; i < n; ++i) {
            d1 += sqrt((hp1[2 * i] - m1[0]) * (hp1[2 * i] - m1[0]) + (hp1[2 * i + 1] - m1[1]) * (hp1[2 * i + 1] - m1[1]));
            d2 += sqrt((hp2[2 * i] - m2[0]) * (hp2[2 * i] - m2[0]) + (hp2[2 * i + 1] - m2[1]) * (hp2[2 * i + 1] - m2[1]));
        }
        d1 /= (double)n, d2 /= (double)n;
        const double s1 = sqrt(2.0) / d1, s2 = sqrt(2.0) / d2;
        T1[0] = s1, T1[2] = -s1 * m1[0], T1[4] = s1, T1[5] = -s1 * m1[1], T1[8] = 1.0;
        T2[0] = s2, T2[2] = -s2 * m2[0], T2[4] = s2, T2[5] = -s2 * m2[1], T2[8] = 1.0;
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) g.T2t[3 * i + j] = T2[3 * j + i];
        inv_similarity(T1, g.T1i), inv_similarity(g.T2t, g.T2ti);
    }
    void norm1(unsigned idx, double *o) const {  // normalised first-image point (vmul's order)
        for (int k = 0; k < 3; ++k) {
            double sacc = 0.;
            sacc += g.T1[3 * k] * hp1[2 * idx];
            sacc += g.T1[3 * k + 1] * hp1[2 * idx + 1];
            sacc += g.T1[3 * k + 2] * 1.0;
            o[k] = sacc;
        }
    }

    // host-time account of a run (debug: MLPL_USAC_PROF=1 prints the sums of a batch): TSC ticks per section; a run's waits for the device
    // (during which its worker thread serves other runs) are taken out through sync_timed()
    enum { PF_SETUP = 0, PF_BUILD, PF_BATCH_HOST, PF_WAIT, PF_LO, PF_SOLVE, PF_NUM };
    uint64_t prof[PF_NUM] = {0, 0, 0, 0, 0, 0};
    static uint64_t tsc() { return __builtin_ia32_rdtsc(); }
    int sync_timed() {
        const uint64_t t0 = tsc();
        const int rc = L.sync();
        prof[PF_WAIT] += tsc() - t0;
        return rc;
    }

    int setup() {
        const uint64_t pf_t0 = tsc(), pf_w0 = prof[PF_WAIT];
        struct PfSetup {
            UsacRun &r;
            uint64_t t0, w0;
            ~PfSetup() { r.prof[PF_SETUP] += (tsc() - t0) - (r.prof[PF_WAIT] - w0); }
        } pf_setup{*this, pf_t0, pf_w0};
        normalisation();
        min_sample.assign(5, 0);
        if (prosac) init_prosac();
        last_wald_update = 0, history.clear();
        design_sprt();
        pool_index = 0;
        pool.resize(n);
        for (unsigned i = 0; i < n; ++i) pool[i] = i;
        for (unsigned i = 1; i < n; ++i) {  // std::random_shuffle (libstdc++): swap(i, rand() % (i + 1))
            const unsigned j = (unsigned)rng.next() % (i + 1);
            if (i != j) std::swap(pool[i], pool[j]);
        }
        if (n > (unsigned)kUsacLoMaxRows * kUsacLoThreads) {
            set_error("mlpl_usac_essential: more than %d correspondences", kUsacLoMaxRows * kUsacLoThreads);
            return MLPL_E_UNSUPPORTED;
        }
        flags.assign(n, 0);
        words = (int)((n + 63) / 64);
        best_bits.assign(words, 0);
        // buffers: one device block and one pinned, device-mapped block, laid out by usac_layout (a run alone takes them from the
        // context, a run of a batch gets its slices from the batch driver)
        const UsacLayout Y = usac_layout(n, refine, dg_on);
        batch_cap = Y.batch_cap, dg_cap = Y.dg_cap, lo5_blocks = Y.lo5_blocks, lo5_rows = Y.lo5_rows;
        char *dev, *pin, *pin_dev;
        if (bufs) {
            dev = bufs->dev, pin = bufs->pin, pin_dev = bufs->pin_dev;
        } else {
            void *p;
            int rc;
            if ((rc = ws_get(ctx, WS_AUX3, Y.dev_total, &p))) return rc;
            dev = (char *)p;
            if ((rc = pinned_get(ctx, Y.pin_total, &p))) return rc;
            pin = (char *)p;
            void *alias = nullptr;
            MLPL_HIP_TRY(hipHostGetDevicePointer(&alias, pin, 0));
            pin_dev = (char *)alias;
        }
        d_pts_pool = (double4 *)(dev + Y.d_pts);
        d_recs = (PolyRec *)(dev + Y.d_recs);
        d_Etab = (double *)(dev + Y.d_Etab);
        d_nm = (int32_t *)(dev + Y.d_nm);
        d_err = (double *)(dev + Y.d_err);
        d_lo5_state = (UsacLo5State *)(dev + Y.d_lo5_state);
        d_lo5_gram = (double *)(dev + Y.d_lo5_gram);
        d_lo5_flags = (uint8_t *)(dev + Y.d_lo5_flags);
        h_out = pin + Y.p_out, h_out_dev = pin_dev + Y.p_out;
        h_smp = (int32_t *)(pin + Y.p_smp), d_smp = (int32_t *)(pin_dev + Y.p_smp);
        h_lo_in = (UsacLoIn *)(pin + Y.p_lo_in), d_lo_in = (UsacLoIn *)(pin_dev + Y.p_lo_in);
        int32_t *h_pool = (int32_t *)(pin + Y.p_pool);
        h_dg = (UsacDgModel *)(pin + Y.p_dg), d_dg = (UsacDgModel *)(pin_dev + Y.p_dg);
        h_lo5_flags = (uint8_t *)(pin + Y.p_flags), d_lo5_flags_src = (const uint8_t *)(pin_dev + Y.p_flags);
        h_lo5_diff = (int32_t *)(pin + Y.p_diff), d_lo5_diff = (int32_t *)(pin_dev + Y.p_diff);
        if (dg_on) {
            pool_pos.resize(n);
            for (unsigned j = 0; j < n; ++j) pool_pos[pool[j]] = j;
            dg_in_rot.assign(n, 0), dg_out_rot.assign(n, 0), dg_in_nomot.assign(n, 0), dg_out_nomot.assign(n, 0);
        }
        // the evaluation pool: the points gathered in its order (the kernel reads the permutation from the mapped block)
        for (unsigned i = 0; i < n; ++i) h_pool[i] = (int32_t)pool[i];
        UsacPoolPackArgs pa{{(int)((n + 255) / 256), 1}, d_p1, d_p2, (const int32_t *)(pin_dev + Y.p_pool), (int)n, d_pts_pool};
        L.launch(HK_USAC_POOL_PACK, pa);
        return MLPL_OK;
    }
    struct UsacLayout {
        size_t d_pts, d_recs, d_Etab, d_nm, d_err, d_lo5_state, d_lo5_gram, d_lo5_flags, dev_total;
        size_t p_out, p_smp, p_lo_in, p_pool, p_dg, p_flags, p_diff, pin_total;
        int batch_cap, dg_cap, words, lo5_blocks, lo5_rows;
    };
    static UsacLayout usac_layout(unsigned n, int refine, bool dg) {
        UsacLayout Y;
        auto up = [](size_t v) { return (v + 255) & ~(size_t)255; };
        const int words = (int)((n + 63) / 64);
        Y.words = words;
        Y.batch_cap = std::max(8, std::min(kUsacBatch, (int)((size_t)(6u << 20) / ((size_t)10 * words * 8))));
        const int rows = (int)((n + kLo5Threads - 1) / kLo5Threads);
        Y.lo5_blocks = std::max(1, std::min(kLo5MaxBlocks, rows));
        Y.lo5_rows = std::max(1, (rows + Y.lo5_blocks - 1) / Y.lo5_blocks);
        Y.lo5_blocks = std::max(1, (rows + Y.lo5_rows - 1) / Y.lo5_rows);
        size_t o = 0;
        Y.d_pts = o, o = up(o + (size_t)n * sizeof(double4));
        Y.d_recs = o, o = up(o + (size_t)Y.batch_cap * sizeof(PolyRec));
        Y.d_Etab = o, o = up(o + (size_t)Y.batch_cap * 90 * 8);
        Y.d_nm = o, o = up(o + (size_t)Y.batch_cap * 4);
        Y.d_err = o, o = up(o + (size_t)kUsacLoReps * n * 8);
        Y.d_lo5_state = o, o = up(o + (refine ? (size_t)kUsacLoReps * sizeof(UsacLo5State) : 0));
        Y.d_lo5_gram = o, o = up(o + (refine ? (size_t)kUsacLoReps * kLo5MaxBlocks * 45 * 8 : 0));
        Y.d_lo5_flags = o, o = up(o + (refine ? (size_t)n + 16 : 0));
        Y.dev_total = o + 256;
        const size_t batch_out = (size_t)Y.batch_cap * 4 + 64 + (size_t)Y.batch_cap * 10 * (8 + 4) + 64 + (size_t)Y.batch_cap * 10 * words * 8 + 64 + (size_t)Y.batch_cap * 720;
        const size_t lo_stride = (sizeof(UsacLoOut) + (size_t)kUsacLoEvals * words * 8 + 63) & ~(size_t)63;
        const size_t lo5_stride = (sizeof(UsacLo5Out) + (size_t)kUsacLoEvals * words * 8 + 63) & ~(size_t)63;
        const size_t out_bytes = std::max(batch_out, (size_t)kUsacLoReps * std::max(lo_stride, lo5_stride));
        Y.dg_cap = dg ? (int)std::max<size_t>(10, std::min<size_t>(128, out_bytes / ((size_t)2 * words * 8))) : 0;
        o = 0;
        Y.p_out = o, o = up(o + out_bytes);
        Y.p_smp = o, o = up(o + (size_t)Y.batch_cap * 5 * 4);
        Y.p_lo_in = o, o = up(o + (size_t)kUsacLoReps * sizeof(UsacLoIn));
        Y.p_pool = o, o = up(o + (size_t)n * 4);
        Y.p_dg = o, o = up(o + (size_t)Y.dg_cap * sizeof(UsacDgModel));
        Y.p_flags = o, o = up(o + (size_t)n + 16);   // the inlier flags of the best model (5-point refinements) / the final mask on its way to the device
        Y.p_diff = o, o = up(o + 64);
        Y.pin_total = o + 256;
        return Y;
    }
    size_t batch_out_bytes(int B) const { return (size_t)B * 4 + 64 + (size_t)B * 10 * (8 + 4) + 64 + (size_t)B * 10 * words * 8 + 64 + (size_t)B * 720; }
    size_t lo_out_stride() const { return (sizeof(UsacLoOut) + (size_t)kUsacLoEvals * words * 8 + 63) & ~(size_t)63; }
    size_t lo5_out_stride() const { return (sizeof(UsacLo5Out) + (size_t)kUsacLoEvals * words * 8 + 63) & ~(size_t)63; }

    void uniform_sample(GlibcRand &r, unsigned data_size, unsigned sample_size, std::vector<unsigned> &sample) const {
        unsigned count = 0;
        do {
            const unsigned index = (unsigned)r.next() % data_size;
            if (std::find(sample.begin(), sample.begin() + count, index) == sample.begin() + count) sample[count++] = index;
        } while (count < sample_size);
    }

    void init_prosac() {
        growth.assign(n, 0);
        double T_n = prosac_max_samples;
        unsigned T_n_p = 1;
        for (unsigned i = 0; i < 5; ++i) T_n *= (double)(5 - i) / (n - i);
        for (unsigned i = 0; i < n; ++i) {
            if (i + 1 <= 5) {
                growth[i] = T_n_p;
                continue;
            }
            const double temp = (double)(i + 1) * T_n / (i + 1 - 5);
            growth[i] = T_n_p + (unsigned)ceil(temp - T_n);
            T_n = temp;
            T_n_p = growth[i];
        }
        non_random.assign(n, 0);
        // Non-randomness table (USAC.h initProsac): for every subset size nn <= 1001 the binomial terms v[5 .. nn-1] by a forward
        // recurrence, then their tail sum from the top.  Entry nn depends on (nn, beta) only, and each recurrence is one serial chain of
        // two multiplications per step -- ~500 000 dependent steps at n >= 1001, a millisecond on the host, as long as the rest of a PROSAC
        // call.  Four sizes are therefore advanced side by side (four independent chains in flight); every chain performs the reference's
        // operations in the reference's order (element-wise vector arithmetic is IEEE arithmetic per element), so the table is the same
        // bit for bit.  (The reference also allocates and zeroes n doubles per nn; the buffer here is reused: every entry is written
        // before the tail sum reads it.)
        // Entry nn of the table depends on (nn, beta, confidence) only: the context keeps the last table and a call with the same two
        // parameters (ConfigUSAC::noAutomaticProsacParamters: beta stays 0.09) copies it -- the same numbers, half a millisecond less.
        const unsigned top = std::min(n, 1001u);
        std::lock_guard<std::mutex> tab_lock(usac_prosac_tab_mutex());  // the runs of a batch share the context's table
        if (ctx->usac_prosac_tab && ctx->usac_prosac_tab_top >= top && ctx->usac_prosac_tab_beta == prosac_beta &&
            ctx->usac_prosac_tab_conf == prosac_non_rand_conf) {
            for (unsigned nn = 6; nn <= top; ++nn) non_random[nn - 1] = ctx->usac_prosac_tab[nn - 1];
            for (unsigned nn = top + 1; nn <= n; ++nn) non_random[nn - 1] = non_random[nn - 2];
        } else {
            constexpr int W = 4;
            typedef double vecW __attribute__((vector_size(W * sizeof(double))));  // element-wise IEEE operations: same bits as scalar code
            std::vector<vecW> vbuf((size_t)top + 1);                                // vbuf[i - 1][k]: term i of size nn0 + k
            const double ratio_c = prosac_beta / (1 - prosac_beta);
            for (unsigned nn0 = 6; nn0 <= top; nn0 += W) {
                vecW pn, nnv;
                unsigned nnk[W];
                for (int k = 0; k < W; ++k) {  // sizes beyond `top` are computed along and dropped
                    const unsigned nn = nn0 + k;
                    nnk[k] = nn;
                    nnv[k] = (double)nn;
                    pn[k] = prosac_beta * std::pow((double)1 - prosac_beta, (double)nn - 5 - 1) * (nn - 5);
                }
                vbuf[5] = pn;
                // steps every size takes: i < nn0 (the integers nn - i and i - 4 are exact in double, so the quotient is the reference's)
                for (unsigned i = 7; i < nn0; ++i) {
                    const double di = (double)i, den = (double)(i - 5 + 1);
                    pn = pn * ratio_c * ((nnv - di) / den);
                    vbuf[i - 1] = pn;
                }
                // the last steps, size by size: term nn is beta^(nn - 5)
                for (int k = 0; k < W; ++k) {
                    const unsigned nn = nnk[k];
                    if (nn > top) continue;
                    double p = pn[k];
                    for (unsigned i = std::max(7u, nn0); i <= nn; ++i) {
                        if (i == nn) {
                            vbuf[nn - 1][k] = std::pow((double)prosac_beta, (double)nn - 5);
                            break;
                        }
                        vbuf[i - 1][k] = p * (prosac_beta / (1 - prosac_beta)) * ((double)(nn - i) / (i - 5 + 1));
                        p = vbuf[i - 1][k];
                    }
                    double accp = 0.0;
                    unsigned i_min = 0;
                    for (unsigned i = nn; i >= 6; --i) {
                        accp += vbuf[i - 1][k];
                        if (accp < 1 - prosac_non_rand_conf)
                            i_min = i;
                        else
                            break;
                    }
                    non_random[nn - 1] = i_min;
                }
            }
            for (unsigned nn = top + 1; nn <= n; ++nn) non_random[nn - 1] = non_random[nn - 2];
            if (!ctx->usac_prosac_tab) ctx->usac_prosac_tab = (unsigned *)std::calloc(1001, sizeof(unsigned));
            if (ctx->usac_prosac_tab) {
                for (unsigned nn = 6; nn <= top; ++nn) ctx->usac_prosac_tab[nn - 1] = non_random[nn - 1];
                ctx->usac_prosac_tab_top = top, ctx->usac_prosac_tab_beta = prosac_beta, ctx->usac_prosac_tab_conf = prosac_non_rand_conf;
            }
        }
        maximality.assign(n, max_hyp);
        largest_size = 5, subset_size = 5, stop_len = n;
    }
    // generatePROSACMinSample on explicit state, so that the speculation can run it on copies
    void prosac_sample(GlibcRand &r, unsigned &subset, unsigned &largest, unsigned stop, unsigned hyp, std::vector<unsigned> &sample) const {
        if (hyp > prosac_max_samples) {
            uniform_sample(r, n, 5, sample);
            return;
        }
        if (subset > stop) uniform_sample(r, stop, 5, sample);  // no return in the reference: overwritten below, the stream is consumed
        if (hyp > growth[subset - 1]) {
            ++subset;
            if (subset > n) subset = n;
            if (largest < subset) largest = subset;
        }
        uniform_sample(r, subset - 1, 4, sample);
        sample[4] = subset - 1;
        for (auto &i : sample) i = sorted_idx[i];
    }

    unsigned standard_stopping(unsigned num_inliers, unsigned tot, unsigned sample_size) const {
        double n_inl = 1.0, n_pts = 1.0;
        for (unsigned i = 0; i < sample_size; ++i) {
            n_inl *= num_inliers - i;  // unsigned arithmetic, as the reference
            n_pts *= tot - i;
        }
        const double p = n_inl / n_pts;
        if (p < std::numeric_limits<double>::epsilon()) return max_hyp;
        if (1 - p < std::numeric_limits<double>::epsilon()) return 1;
        return usac_to_uint(ceil(log(1 - conf) / log(1 - p)));
    }
    unsigned prosac_stopping(unsigned hyp) {
        unsigned max_samples = maximality[stop_len - 1];
        unsigned inl = 0;
        for (unsigned i = 0; i < prosac_min_stop; ++i) inl += flags[sorted_idx[i]];
        for (unsigned i = prosac_min_stop; i < n; ++i) {
            inl += flags[sorted_idx[i]];
            if (non_random[i] < inl) {
                non_random[i] = inl;
                if ((i == n - 1) || (flags[sorted_idx[i]] && !flags[sorted_idx[i + 1]])) {
                    unsigned ns = standard_stopping(inl, i + 1, 5);
                    if (i + 1 < largest_size) ns += hyp - growth[i];
                    if (ns < maximality[i]) {
                        maximality[i] = ns;
                        if ((ns < max_samples) || ((ns == max_samples) && (i + 1 >= stop_len))) {
                            stop_len = i + 1;
                            max_samples = ns;
                        }
                    }
                }
            }
        }
        return max_samples;
    }
    void design_sprt() {
        const double C = (1 - sprt_delta) * log((1 - sprt_delta) / (1 - sprt_epsilon)) + sprt_delta * (log(sprt_delta / sprt_epsilon));
        const double K = (sprt_tM * C) / sprt_mS + 1;
        double An_1 = K, An = 0;
        for (unsigned i = 0; i < 10; ++i) {
            An = K + log(An_1);
            if (An - An_1 < 1.5e-8) break;
            An_1 = An;
        }
        sprt_A = An;
    }
    void add_history(unsigned num_hyp) {
        history.push_back(UsacWald{sprt_epsilon, sprt_delta, sprt_A, num_hyp - last_wald_update});
        last_wald_update = num_hyp;
    }
    static double exp_sprt(double new_eps, double epsilon, double delta) {
        const double al = log(delta / epsilon), be = log((1.0 - delta) / (1.0 - epsilon));
        const double x0 = log(1.0 / (1.0 - new_eps)) / be;
        const double v0 = new_eps * exp(x0 * al);
        const double x1 = log((1.0 - 2.0 * v0) / (1.0 - new_eps)) / be;
        const double v1 = new_eps * exp(x1 * al) + (1.0 - new_eps) * exp(x1 * be);
        return x0 - (x0 - x1) / (1.0 + v0 - v1) * v0;
    }
    unsigned sprt_stopping(unsigned num_inliers, unsigned tot) const {
        double n_inl = 1.0, n_pts = 1.0, k = 0.0, log_eta = 0.0;
        const double new_eps = (double)num_inliers / tot;
        for (unsigned i = 0; i < 5; ++i) {
            n_inl *= (double)(num_inliers - i);
            n_pts *= (double)(tot - i);
        }
        const double p = n_inl / n_pts;
        if (p < std::numeric_limits<double>::epsilon()) return max_hyp;
        if (1.0 - p < std::numeric_limits<double>::epsilon()) return 1;
        for (size_t t = history.size(); t-- > 0;) {
            const UsacWald &w = history[t];
            k += w.k;
            const double h = exp_sprt(new_eps, w.epsilon, w.delta);
            const double reject = 1.0 / (exp(h * log(w.A)));
            log_eta += (double)w.k * log(1.0 - p * (1.0 - reject));
        }
        const double ns = k + (log(1.0 - conf) - log_eta) / log(1.0 - p * (1.0 - (1.0 / sprt_A)));
        return usac_to_uint(ceil(ns));
    }

    bool validate_sample(const std::vector<unsigned> &smp) const {
        int i, j, k;
        for (i = 0; i < 5; i++) {
            for (j = 0; j < i; j++) {
                double a[3], b[3];
                norm1(smp[i], a), norm1(smp[j], b);
                const double pix = a[0] / a[2], piy = a[1] / a[2], pjx = b[0] / b[2], pjy = b[1] / b[2];
                const double dx1 = pjx - pix, dy1 = pjy - piy;
                for (k = 0; k < j; k++) {
                    double c[3];
                    norm1(smp[k], c);
                    const double dx2 = c[0] / c[2] - pix, dy2 = c[1] / c[2] - piy;
                    if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) break;
                }
                if (k < j) break;
            }
            if (j < i) break;
        }
        return i >= 4;
    }

    // ---- device batches ----
    int run_batch(const std::vector<UsacKey> &keys) {
        const int B = (int)keys.size();
        if (B == 0) return MLPL_OK;
        struct PfBatch {
            UsacRun &r;
            uint64_t t0, w0;
            ~PfBatch() { r.prof[PF_BATCH_HOST] += (tsc() - t0) - (r.prof[PF_WAIT] - w0); }
        } pf_batch{*this, tsc(), prof[PF_WAIT]};
        for (int b = 0; b < B; ++b)
            for (int i = 0; i < 5; ++i) h_smp[b * 5 + i] = (int32_t)keys[b].v[i];
        const size_t off_key = ((size_t)B * 4 + 63) & ~(size_t)63, off_valid = off_key + (size_t)B * 80;
        const size_t off_rows = (off_valid + (size_t)B * 40 + 63) & ~(size_t)63;
        int32_t *o_nm = (int32_t *)h_out_dev, *o_valid = (int32_t *)(h_out_dev + off_valid);
        double *o_key = (double *)(h_out_dev + off_key);
        unsigned long long *o_rows = (unsigned long long *)(h_out_dev + off_rows);
        const size_t off_E = (off_rows + (size_t)B * 10 * words * 8 + 63) & ~(size_t)63;
        hub_launch_solver(L, ctx, B, d_p1, d_p2, (const int32_t *)d_smp, d_recs, d_Etab, d_nm);
        UsacCheckArgs ca{{B * 10, 1}, (const double4 *)d_pts_pool, (int)n, words, d_p1, d_p2, (const int32_t *)d_smp, B, (const double *)d_Etab,
                         (const int32_t *)d_nm, g, thr, o_nm, o_key, o_valid, o_rows, (double *)(h_out_dev + off_E)};
        L.launch(HK_USAC_CHECK, ca);
        int rcw;
        if ((rcw = sync_timed())) return rcw;
        const int32_t *h_nm = (const int32_t *)h_out, *h_valid = (const int32_t *)(h_out + off_valid);
        const double *h_key = (const double *)(h_out + off_key);
        const uint64_t *h_rows = (const uint64_t *)(h_out + off_rows);
        const double *hE = (const double *)(h_out + off_E);
        for (int b = 0; b < B; ++b) {
            UsacSampleModels &sm = cache[keys[b]];  // (filled in place: the entry is a kilobyte)
            const int nm = std::min(h_nm[b], 10);
            int order[10];
            for (int i = 0; i < nm; ++i) {  // ascending key, equal keys in slot order (a stable insertion sort: no temporary buffer)
                int j = i;
                while (j > 0 && h_key[b * 10 + i] < h_key[b * 10 + order[j - 1]]) order[j] = order[j - 1], --j;
                order[j] = i;
            }
            sm.n = nm;
            for (int oi = 0; oi < nm; ++oi) {
                const int slot = order[oi];
                UsacModel &m = sm.m[oi];
                std::memcpy(m.E, hE + ((size_t)b * 10 + slot) * 9, 72);
                m.valid = h_valid[b * 10 + slot];
                m.bits = nullptr;
                if (m.valid) {  // (no row for a rejected model)
                    uint64_t *row = rows_arena.alloc((size_t)words);
                    if (!row) return MLPL_E_INTERNAL;
                    std::memcpy(row, h_rows + ((size_t)b * 10 + slot) * words, (size_t)words * 8);
                    m.bits = row;
                }
            }
            cache_bytes += sizeof(UsacSampleModels) + (size_t)nm * words * 8;
        }
        stats[0]++, stats[1] += B;
        return MLPL_OK;
    }

    // evaluateModel's sequential test on a bit row (pool order): the likelihood ratio is multiplied point by point (inlier: delta / epsilon,
    // outlier: (1 - delta) / (1 - epsilon)), clamped from below at 10 DBL_EPSILON, and the model is rejected the moment it exceeds A.
    // The steps are one dependent chain (multiply, clamp, compare: ~2.5 ns each on the host), and a model that is NOT rejected walks all
    // n correspondences -- 35 such walks per local optimisation, a third of a USAC call.  A walk therefore takes its first kSprtExact steps
    // as the reference does (a bad model is rejected there) and then tries to PROVE that no later step can reject: per bit-row word, with
    // ones / zeros counted by popcount, every intermediate ratio is at most max(U, floor) max(up, 1)^ones max(down, 1)^zeros for an upper
    // bound U of the ratio at the word's start, and U advances to max(U up^ones down^zeros, floor max(up, 1)^ones max(down, 1)^zeros)
    // (the clamp only ever raises the ratio to the floor; powers rounded up).  If every word stays below A the verdict is "accepted" with all
    // n correspondences tested and the row's inlier count -- exactly what the step-by-step walk returns; if any word cannot be cleared
    // the walk resumes step by step from the saved exact state.  Same verdicts, counts and pool positions by construction; the decision
    // traces of tests/test_gpu_usac*.py compare all three with the reference-built runs.
    static constexpr unsigned kSprtExact = 128;
    struct SprtBounds {
        double up = -1, down = -1;
        double Pu[65], Pz[65], Gu[65], Gz[65];
    } sb;
    void sprt_bounds(double up, double down) {
        if (sb.up == up && sb.down == down) return;
        sb.up = up, sb.down = down;
        const double gu = up > 1 ? up : 1.0, gz = down > 1 ? down : 1.0, slack = 1.0 + 0x1p-40;
        double pu = 1, pz = 1, qu = 1, qz = 1;
        for (int k = 0; k <= 64; ++k) {
            sb.Pu[k] = pu * slack, sb.Pz[k] = pz * slack, sb.Gu[k] = qu * slack, sb.Gz[k] = qz * slack;
            pu *= up, pz *= down, qu *= gu, qz *= gz;
        }
    }
    bool sprt_walk(const uint64_t *bits, unsigned *num_inl, unsigned *tested) {
        bool good = true;
        double lj, lj1 = 1.0;
        *num_inl = 0, *tested = 0;
        const double up = sprt_delta / sprt_epsilon, down = (1 - sprt_delta) / (1 - sprt_epsilon);
        unsigned i = 0;
        auto steps = [&](unsigned end) {  // the reference's loop over [i, end)
            for (; i < end; ++i) {
                if (pool_index > n - 1) pool_index = 0;
                const unsigned j = pool_index;
                ++pool_index;
                const bool in = (bits[j >> 6] >> (j & 63)) & 1;
                if (in) {
                    ++(*num_inl);
                    lj = lj1 * up;
                } else
                    lj = lj1 * down;
                if (lj <= DBL_EPSILON) lj = DBL_EPSILON * 10;
                if (lj > sprt_A) {
                    good = false;
                    *tested = i + 1;
                    return;
                }
                lj1 = lj;
            }
        };
        steps(std::min(n, kSprtExact));
        if (!good) return false;
        if (i < n && sprt_fast) {
            sprt_bounds(up, down);
            const double floor_v = DBL_EPSILON * 10, slack = 1.0 + 0x1p-40, A_safe = sprt_A * (1.0 - 0x1p-30);
            double U = lj1;
            unsigned pi = pool_index, at = i, ones_rest = 0;
            bool cleared = true;
            while (at < n) {
                if (pi > n - 1) pi = 0;
                const unsigned off = pi & 63, len = std::min(std::min(64u - off, n - pi), n - at);
                uint64_t w = bits[pi >> 6] >> off;
                if (len < 64) w &= (1ull << len) - 1ull;
                const unsigned ones = (unsigned)__builtin_popcountll(w), zeros = len - ones;
                const double grow = sb.Gu[ones] * sb.Gz[zeros] * slack;
                const double base = U > floor_v ? U : floor_v;
                if (!(base * grow * slack < A_safe)) {  // also taken for NaN / inf parameters
                    cleared = false;
                    break;
                }
                const double through = U * sb.Pu[ones] * sb.Pz[zeros] * slack, from_floor = floor_v * grow;
                U = (through > from_floor ? through : from_floor) * slack;
                ones_rest += ones, pi += len, at += len;
            }
            if (cleared) {
                *num_inl += ones_rest, *tested = n, pool_index = pi;
                return true;
            }
        }
        steps(n);
        if (good) *tested = n;
        return good;
    }
    void emit_eval(unsigned mi, unsigned start, unsigned inl, unsigned tested, bool good) {
        double v[11] = {(double)hyp_count, (double)mi, (double)start, (double)inl, (double)tested, good ? 1.0 : 0.0,
                        sprt_delta, sprt_epsilon, sprt_A, thr, (double)num_lo};
        emit(2, v, 11);
    }

    void store_solution(unsigned mi, unsigned num_inl, const uint64_t *bits, const double *E) {
        best = num_inl;
        std::memset(flags.data(), 0, n);  // the set bits only: half the scattered stores of a loop over all positions
        for (int w = 0; w < words; ++w) {
            uint64_t m = bits[w];
            if (w == words - 1 && (n & 63)) m &= (1ull << (n & 63)) - 1ull;
            while (m) {
                const unsigned j = (unsigned)w * 64u + (unsigned)__builtin_ctzll(m);
                m &= m - 1;
                flags[pool[j]] = 1;
            }
        }
        best_bits.assign(bits, bits + words);
        std::memcpy(final_model, E, 72);
        double v[3] = {(double)hyp_count, (double)mi, (double)num_inl};
        emit(4, v, 3);
    }

    // ---- local optimisation ----
    int launch_lo(int reps_from, int reps_to) {  // h_lo_in[reps_from .. reps_to) are filled
        UsacLoArgs la{{reps_to - reps_from, 1}, d_p1, d_p2, (const double4 *)d_pts_pool, (int)n, words, g, thr, lo_mult, (const UsacLoIn *)(d_lo_in + reps_from),
                      h_out_dev + (size_t)reps_from * lo_out_stride(), lo_out_stride(), d_err + (size_t)reps_from * n,
                      (ctx->opt_usac_lo_warm_start ? 1 : 0) | (ctx->opt_eig_inverse_iteration ? 2 : 0)};
        L.launch(HK_USAC_LO, la);
        int rcw;
        if ((rcw = sync_timed())) return rcw;
        stats[3]++;
        return MLPL_OK;
    }
    void emit_refined(unsigned pts, bool weighted, const double *E, bool ok = true) {
        double v[13] = {0};
        v[0] = hyp_count, v[1] = pts, v[2] = weighted ? 1 : 0, v[3] = ok ? 1 : 0;
        if (ok) std::memcpy(v + 4, E, 72);
        emit(3, v, 13);
    }

    // locallyOptimizeSolution (USAC.h:947-1073) over the device chains
    int local_optimization(unsigned best_inliers, unsigned *out) {
        struct PfLo {
            UsacRun &r;
            uint64_t t0, w0;
            ~PfLo() { r.prof[PF_LO] += (tsc() - t0) - (r.prof[PF_WAIT] - w0); }
        } pf_lo{*this, tsc(), prof[PF_WAIT]};
        *out = 0;
        if (best_inliers < 2 * kUsacLoSample) return MLPL_OK;
        std::vector<unsigned> orig(n), sample(kUsacLoSample);
        unsigned c = 0;
        for (unsigned i = 0; i < n; ++i)
            if (flags[i]) orig[c++] = i;  // findInliers(err_ptr_[1], threshold): the flags are that predicate
        unsigned lo_inliers = best_inliers;
        ++num_lo;
        // all samples are known up front: nothing inside a repetition consumes the stream
        for (int r = 0; r < kUsacLoReps; ++r) {
            uniform_sample(rng, best_inliers, kUsacLoSample, sample);
            h_lo_in[r].start_step = -1;
            for (int j = 0; j < kUsacLoSample; ++j) h_lo_in[r].sample[j] = (int32_t)orig[sample[j]];
        }
        int rc;
        if ((rc = launch_lo(0, kUsacLoReps))) return rc;
        const size_t stride = lo_out_stride();
        for (int r = 0; r < kUsacLoReps; ++r) {
            // the chain of this repetition: evaluation record e of O holds the model evaluated there, its inlier count and its bit row
            const UsacLoOut *O = (const UsacLoOut *)(h_out + (size_t)r * stride);
            stats[7] += O->sweeps;
            const uint64_t *rows = (const uint64_t *)((const char *)O + sizeof(UsacLoOut));
            int e = 0;
            unsigned tmp = 0, tested;
            auto evaluate = [&]() {
                const unsigned start = pool_index;
                const bool good = sprt_walk(rows + (size_t)e * words, &tmp, &tested);
                emit_eval(0, start, tmp, tested, good);
                return good;
            };
            // restart the chain at re-weighting step `step` (kUsacLoSteps = only the last evaluation) from the model of record e
            auto resume = [&](int step) {
                UsacLoIn &I = h_lo_in[r];
                I.start_step = step;
                std::memcpy(I.F, O->F[e], 72), std::memcpy(I.E, O->E[e], 72);
                e = 0;
                stats[4]++;
                return launch_lo(r, r + 1);
            };
            emit_refined(kUsacLoSample, false, O->E[0]);
            if (!evaluate()) continue;
            if ((unsigned)O->cnt2 < 5) continue;  // findInliers at 2 x threshold < minimal sample
            e = 1;
            emit_refined((unsigned)O->cnt2, false, O->E[1]);
            for (int j = 0; j < kUsacLoSteps; ++j) {
                if (lo_stepwise && (rc = resume(j))) return rc;  // tests: every step through the resume path (same model, same results)
                if (!evaluate()) {  // rejected: the model stays and is evaluated again by the next step
                    if ((rc = resume(j + 1))) return rc;
                    continue;
                }
                ++e;  // the refitted model (or the same one again when fewer than 5 points were left)
                if (tmp >= 5) emit_refined((unsigned)O->fit_pts[e], true, O->E[e]);
            }
            if (!evaluate()) continue;
            if (tmp > lo_inliers) {
                lo_inliers = tmp;
                store_solution(0, lo_inliers, rows + (size_t)e * words, O->E[e]);
            }
        }
        *out = lo_inliers;
        return MLPL_OK;
    }


    // ---- local optimisation, refinements of the 5-point family (the usac5_* chains above) ----
    // chains [c0, c1): their h_lo_in are filled, all start at `start_phase` (-1 = from the 14-point sample)
    int lo5_run(int c0, int c1, int start_phase) {
        const int C = c1 - c0;
        const size_t stride = lo5_out_stride(), part_stride = (size_t)kLo5MaxBlocks * 45;
        UsacLo5State *st = d_lo5_state + c0;
        double *gram = d_lo5_gram + (size_t)c0 * part_stride, *err = d_err + (size_t)c0 * n, *Etab = d_Etab + (size_t)c0 * 90;
        char *outp = h_out_dev + (size_t)c0 * stride;
        PolyRec *recs = d_recs + c0;
        int32_t *nm = d_nm + c0;
        const char *gate = reinterpret_cast<const char *>(st) + offsetof(UsacLo5State, step_fit);
        Usac5BeginArgs ba{{C, 1}, d_p1, d_p2, (const UsacLoIn *)(d_lo_in + c0), st, gram, part_stride, outp, stride};
        L.launch(HK_USAC5_BEGIN, ba);
        auto fit = [&](int nparts, int fit_eval, int ends) {
            if (ctx->opt_usac_lo5_fused_fit) {
                Usac5FitArgs fa{{C, 1}, (const double *)gram, nparts, recs, part_stride, ctx->opt_usac_lo_warm_start ? 1 : 0, ctx->opt_solver_polish ? 1 : 0, Etab, nm,
                                Usac5ChooseArgs{{C, 1}, d_p1, d_p2, (int)n, (const uint8_t *)d_lo5_flags, (const double *)Etab, (const int32_t *)nm, st, outp, stride, fit_eval, ends}};
                L.launch(HK_USAC5_FIT, fa);
                return;
            }
            RefitSolveArgs ra{{C, 1}, (const double *)gram, nparts, recs, part_stride, gate, sizeof(UsacLo5State),
                              ctx->opt_usac_lo_warm_start ? reinterpret_cast<char *>(st) + offsetof(UsacLo5State, warm) : nullptr, sizeof(UsacLo5State)};
            L.launch(HK_REFIT_SOLVE, ra);
            RootsArgs ro{{(C + kHypPerWave - 1) / kHypPerWave, 1}, (const PolyRec *)recs, C, Etab, nm};
            L.launch(ctx->opt_solver_polish ? HK_ROOTS_POLISH : HK_ROOTS_PLAIN, ro);
            Usac5ChooseArgs ch{{C, 1}, d_p1, d_p2, (int)n, (const uint8_t *)d_lo5_flags, (const double *)Etab, (const int32_t *)nm, st, outp, stride, fit_eval, ends};
            L.launch(HK_USAC5_CHOOSE, ch);
        };
        if (start_phase < 0) fit(1, -1, 1);
        const double step = (lo_mult * thr - thr) / kUsacLoSteps, th_ph = sqrt(thr) / 50.0;
        const bool weights = refine == 5 || refine == 7;
        for (int phase = start_phase; phase <= kUsacLoSteps; ++phase) {
            const int e = phase - start_phase;
            const double limit = phase < 0 ? lo_mult * thr : (phase < kUsacLoSteps ? (lo_mult * thr) - (phase + 1) * step : 0.0);
            Usac5EvalArgs ea{{lo5_blocks, C}, d_p1, d_p2, (const double4 *)d_pts_pool, (int)n, words, lo5_rows, thr, limit, st, err, outp, stride, e};
            L.launch(HK_USAC5_EVAL, ea);
            if (phase == kUsacLoSteps) break;
            Usac5GramArgs ga{{lo5_blocks, C}, d_p1, d_p2, (int)n, lo5_rows, limit, phase < 0 ? 1 : 0, (phase >= 0 && weights) ? 1 : 0, th_ph, st,
                             (const double *)err, gram, part_stride, outp, stride, e};
            L.launch(HK_USAC5_GRAM, ga);
            fit(lo5_blocks, e, phase < 0 ? 1 : 0);
        }
        int rcw;
        if ((rcw = sync_timed())) return rcw;
        stats[3]++;
        return MLPL_OK;
    }
    // usac5_pick on the host: the same sums in the same order, the same early exit, the same tie rule
    static double sampson_host(const double *m, double x1, double y1, double x2, double y2) {
        const double rxc = m[0] * x2 + m[3] * y2 + m[6];
        const double ryc = m[1] * x2 + m[4] * y2 + m[7];
        const double rwc = m[2] * x2 + m[5] * y2 + m[8];
        const double r = (x1 * rxc + y1 * ryc + rwc);
        const double rx = m[0] * x1 + m[1] * y1 + m[2];
        const double ry = m[3] * x1 + m[4] * y1 + m[5];
        return r * r / (rxc * rxc + ryc * ryc + rx * rx + ry * ry);
    }
    int lo5_pick_host(const double (*Es)[9], int nm) const {
        if (nm <= 1) return 0;
        double key[10], sum[10];
        int pos[10];
        for (int j = 0; j < nm; ++j) {
            double big = 0, n2 = 0;
            for (int k = 0; k < 9; ++k) {
                if (fabs(Es[j][k]) > fabs(big)) big = Es[j][k];
                n2 += Es[j][k] * Es[j][k];
            }
            key[j] = (big < 0 ? -Es[j][0] : Es[j][0]) / sqrt(n2);
            sum[j] = 0;
        }
        for (int j = 0; j < nm; ++j) {
            pos[j] = 0;
            for (int k = 0; k < nm; ++k) pos[j] += (key[k] < key[j] || (key[k] == key[j] && k < j)) ? 1 : 0;
        }
        auto two_smallest = [&](double &m1, double &m2) {
            m1 = INFINITY;
            int first = -1;
            for (int j = 0; j < nm; ++j) m1 = fmin(m1, sum[j]);
            for (int j = 0; j < nm && first < 0; ++j)
                if (sum[j] == m1) first = j;
            m2 = INFINITY;
            for (int j = 0; j < nm; ++j)
                if (j != first) m2 = fmin(m2, sum[j]);
        };
        for (unsigned i = 0; i < n; ++i) {
            if (!flags[i]) continue;
            for (int j = 0; j < nm; ++j) sum[j] += sampson_host(Es[j], hp1[2 * i], hp1[2 * i + 1], hp2[2 * i], hp2[2 * i + 1]);
            if ((i > 3) && (i % 4 == 0)) {
                double m1, m2;
                two_smallest(m1, m2);
                if (m1 < 0.66 * m2) break;
            }
        }
        double m1, m2;
        two_smallest(m1, m2);
        int take = 0, best_pos = 64;
        for (int j = 0; j < nm; ++j)
            if (sum[j] == m1 && pos[j] < best_pos) best_pos = pos[j], take = j;
        return take;
    }
    int lo5_upload_flags() {  // the inlier flags of the best model, point order (pinned staging: the copy is stream-ordered)
        std::memcpy(h_lo5_flags, flags.data(), n);
        hub_copy_bytes(L, d_lo5_flags_src, d_lo5_flags, n);
        return MLPL_OK;
    }

    // locallyOptimizeSolution (USAC.h:947-1073) over the usac5 chains
    int local_optimization5(unsigned best_inliers, unsigned *out) {
        struct PfLo {
            UsacRun &r;
            uint64_t t0, w0;
            ~PfLo() { r.prof[PF_LO] += (tsc() - t0) - (r.prof[PF_WAIT] - w0); }
        } pf_lo{*this, tsc(), prof[PF_WAIT]};
        *out = 0;
        if (best_inliers < 2 * kUsacLoSample) return MLPL_OK;
        std::vector<unsigned> orig(n), sample(kUsacLoSample);
        unsigned c = 0;
        for (unsigned i = 0; i < n; ++i)
            if (flags[i]) orig[c++] = i;
        unsigned lo_inliers = best_inliers;
        ++num_lo;
        for (int r = 0; r < kUsacLoReps; ++r) {
            uniform_sample(rng, best_inliers, kUsacLoSample, sample);
            h_lo_in[r].start_step = -1;
            for (int j = 0; j < kUsacLoSample; ++j) h_lo_in[r].sample[j] = (int32_t)orig[sample[j]];
        }
        int rc;
        if ((rc = lo5_upload_flags()) || (rc = lo5_run(0, kUsacLoReps, -1))) return rc;
        const size_t stride = lo5_out_stride();
        for (int r = 0; r < kUsacLoReps; ++r) {
            const UsacLo5Out *O = (const UsacLo5Out *)(h_out + (size_t)r * stride);
            const uint64_t *rows = (const uint64_t *)((const char *)O + sizeof(UsacLo5Out));
            int e = 0;
            unsigned tmp = 0, tested;
            auto evaluate = [&]() {
                const unsigned start = pool_index;
                const bool good = sprt_walk(rows + (size_t)e * words, &tmp, &tested);
                emit_eval(0, start, tmp, tested, good);
                return good;
            };
            auto resume = [&](int step) {  // restart the chain at re-weighting step `step` with the model of record e
                UsacLoIn &I = h_lo_in[r];
                I.start_step = step;
                std::memcpy(I.E, O->E[e], 72);
                e = 0;
                stats[4]++;
                return lo5_run(r, r + 1, step);
            };
            if (O->first_fit != 1) {
                emit_refined(kUsacLoSample, false, nullptr, false);
                continue;
            }
            emit_refined(kUsacLoSample, false, O->E[0]);
            if (!evaluate()) continue;
            if ((unsigned)O->cnt2 < 5) continue;
            if (O->fit_state[0] != 1) {
                emit_refined((unsigned)O->cnt2, false, nullptr, false);
                continue;
            }
            e = 1;
            emit_refined((unsigned)O->cnt2, false, O->E[1]);
            for (int j = 0; j < kUsacLoSteps; ++j) {
                if (lo_stepwise && (rc = resume(j))) return rc;
                if (!evaluate()) {
                    if ((rc = resume(j + 1))) return rc;
                    continue;
                }
                const int fs = O->fit_state[e];
                ++e;
                if (fs == 1) emit_refined((unsigned)O->fit_pts[e], true, O->E[e]);
                else if (fs == 2) emit_refined(tmp, true, nullptr, false);
            }
            if (!evaluate()) continue;
            if (tmp > lo_inliers) {
                lo_inliers = tmp;
                store_solution(0, lo_inliers, rows + (size_t)e * words, O->E[e]);
                if (r + 1 < kUsacLoReps) {
                    // the repetitions behind this one chose their solutions under the old flags: the host re-takes every choice of
                    // theirs on the solutions the chain recorded (usually a handful of error evaluations: the early exit), and a chain
                    // whose choice would now differ is run again.  The device's flags follow (later resumes and re-runs read them).
                    if ((rc = lo5_upload_flags())) return rc;
                    stats[5]++;
                    for (int k = r + 1; k < kUsacLoReps; ++k) {
                        const UsacLo5Out *Ok = (const UsacLo5Out *)(h_out + (size_t)k * stride);
                        bool differs = false;
                        for (int f = 0; f < kUsacLoEvals && !differs; ++f)
                            if (Ok->hist_nm[f] > 1) differs = lo5_pick_host(Ok->hist_E[f], Ok->hist_nm[f]) != Ok->hist_take[f];
                        if (differs) {
                            stats[6]++;
                            h_lo_in[k].start_step = -1;
                            if ((rc = lo5_run(k, k + 1, -1))) return rc;
                        }
                    }
                }
            }
        }
        *out = lo_inliers;
        return MLPL_OK;
    }

    // ---- degeneracy tests and model upgrade (ConfigUSAC::degeneracyCheck = DEGEN_USAC_INTERNAL) ---------------------------------------
    // EssentialMatEstimator.h: testSolutionDegeneracy :1334-1362, testSolutionDegeneracyRot :1511-1663, testSolutionDegeneracyNoMot
    // :1838-1911, upgradeDegenerateModel :1917-2365 (its two pose branches), as estimateEssentialMatUsac configures them
    // (usac_estimations.cpp:443-456: 8000 upgrade samples, enableUpgradeDegenPose).  The homography test of the 8-point refinements is
    // not part of it.  Errors of all correspondences come from usac_degen_rows_kernel as bit rows in pool order; what the
    // reference reads from its two error arrays afterwards (errors below a threshold, entries never written = DBL_MAX) is kept per
    // array as three bits per correspondence.
    enum { DG_NOT_FOUND = 0x1, DG_H = 0x2, DG_ROT_TRANS = 0x4, DG_NO_MOT = 0x8, DG_UPGRADE = 0x10 };
    bool dg_on = false, dg_losac = false;
    double dg_thr = 0;  // poseDegenTheshold
    unsigned dg_type = DG_NOT_FOUND, dg_cnt_rot = 0, dg_cnt_nomot = 0, dg_cnt_trans = 0;
    unsigned dg_max_rot = 8000, dg_max_nomot = 8000;
    std::vector<uint8_t> dg_in_rot, dg_out_rot, dg_in_nomot, dg_out_nomot;
    double dg_R[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::vector<int> dg_sample_rot, dg_sample_nomot;
    std::vector<unsigned> pool_pos;
    UsacDgModel *h_dg = nullptr, *d_dg = nullptr;
    int dg_cap = 0;
    struct DgErrs {  // one of the reference's two error arrays, as far as anything reads it
        std::vector<uint8_t> touched, below_count, below_inl;
    };

    static bool near_zero(double d) { return (d < 1e-3) && (d > -1e-3); }  // poselib::nearZero
    void view1(unsigned i, double *f) const { dgm::bearing(hp2[2 * i], hp2[2 * i + 1], f); }  // adapter view 1 = second image
    void view2(unsigned i, double *f) const { dgm::bearing(hp1[2 * i], hp1[2 * i + 1], f); }
    static bool row_bit(const uint64_t *row, unsigned j) { return (row[j >> 6] >> (j & 63)) & 1; }
    unsigned row_count(const uint64_t *row) const {
        unsigned c = 0;
        for (int w = 0; w < words; ++w) c += (unsigned)__builtin_popcountll(row[w]);
        return c;
    }
    // rows of h_dg[0 .. B): returns the pinned block, 2 rows of `words` words per model
    int dg_rows(int B, const uint64_t **rows) {
        UsacDgRowsArgs da{{B, 1}, (const double4 *)d_pts_pool, (int)n, words, (const UsacDgModel *)d_dg, dg_thr, thr, (unsigned long long *)h_out_dev};
        L.launch(HK_USAC_DG_ROWS, da);
        int rcw;
        if ((rcw = sync_timed())) return rcw;
        *rows = (const uint64_t *)h_out;
        stats[5]++;
        return MLPL_OK;
    }
    void rotation_only(const std::vector<int> &idx, double *R) const {  // opengv rotationOnly: Arun on the centred bearing vectors
        double c1[3] = {0, 0, 0}, c2[3] = {0, 0, 0}, H[9] = {0};
        for (int i : idx) {
            double a[3], b[3];
            view1((unsigned)i, a), view2((unsigned)i, b);
            for (int k = 0; k < 3; ++k) c1[k] += a[k], c2[k] += b[k];
        }
        const double m = (double)idx.size();
        for (int k = 0; k < 3; ++k) c1[k] = c1[k] / m, c2[k] = c2[k] / m;
        for (int i : idx) {
            double a[3], b[3];
            view1((unsigned)i, a), view2((unsigned)i, b);
            dgm::cross_cov_add(H, a, b, c1, c2);
        }
        dgm::arun(H, R);
    }

    int test_rotation(bool *degenerate) {
        static const unsigned pair_of[20] = {0, 1, 0, 2, 0, 3, 0, 4, 1, 2, 1, 3, 1, 4, 2, 3, 2, 4, 3, 4};
        static const unsigned rest_of[30] = {2, 3, 4, 1, 3, 4, 1, 2, 4, 1, 2, 3, 0, 3, 4, 0, 2, 4, 0, 2, 3, 0, 1, 4, 0, 1, 3, 0, 1, 2};
        int rc;
        for (unsigned i = 0; i < 10; ++i) {  // the ten two-point rotations of the sample, evaluated on all correspondences at once
            double a0[3], b0[3], a1[3], b1[3];
            const unsigned s0 = min_sample[pair_of[2 * i]], s1 = min_sample[pair_of[2 * i + 1]];
            view1(s0, a0), view2(s0, b0), view1(s1, a1), view2(s1, b1);
            h_dg[i].kind = 0;
            dgm::twopt_rotation(a0, b0, a1, b1, h_dg[i].m);
        }
        const uint64_t *rows;
        if ((rc = dg_rows(10, &rows))) return rc;
        std::vector<uint64_t> first((size_t)10 * words);
        for (unsigned i = 0; i < 10; ++i) std::memcpy(&first[(size_t)i * words], rows + (size_t)2 * i * words, (size_t)words * 8);
        std::vector<int> sample(5, 0), inl;
        for (unsigned i = 0; i < 10; ++i) {
            const uint64_t *row = &first[(size_t)i * words];
            for (unsigned j = 0; j < 2; ++j) sample[j] = (int)min_sample[pair_of[2 * i + j]];
            unsigned count1 = 2, num = 0;
            for (unsigned j = 0; j < 3; ++j) {
                const unsigned idx = min_sample[rest_of[3 * i + j]];
                if (row_bit(row, pool_pos[idx])) sample[count1++] = (int)idx, ++num;
            }
            if (num == 0) continue;
            num = row_count(row);
            const unsigned first_count = num;
            if (num < 2) continue;
            inl.clear();
            for (unsigned j = 0; j < n; ++j)
                if (row_bit(row, j)) {
                    inl.push_back((int)pool[j]);
                    if (count1 < 5) sample[count1++] = (int)pool[j];
                }
            h_dg[0].kind = 0;
            rotation_only(inl, h_dg[0].m);
            if ((rc = dg_rows(1, &rows))) return rc;
            num = row_count(rows);
            double v[5] = {(double)hyp_count, (double)i, (double)first_count, (double)num, 0};
            if (num < best / 5) {
                emit(8, v, 5);
                continue;
            }
            *degenerate = true;
            if (dg_type != (dg_type & (DG_UPGRADE | DG_ROT_TRANS))) dg_type = DG_ROT_TRANS;
            if (num > dg_cnt_rot) {
                dg_type |= DG_UPGRADE;
                inl.clear();
                for (unsigned j = 0; j < n; ++j)
                    if (row_bit(rows, j)) inl.push_back((int)pool[j]);
                rotation_only(inl, dg_R);
                dg_cnt_rot = num;
                for (unsigned j = 0; j < n; ++j) {
                    const bool in = row_bit(rows, j);
                    dg_in_rot[pool[j]] = in ? 1 : 0, dg_out_rot[pool[j]] = in ? 0 : 1;
                }
                dg_sample_rot = sample;
                v[4] = 1;
            }
            emit(8, v, 5);
        }
        return MLPL_OK;
    }

    int test_no_motion(bool *degenerate) {
        if (dg_cnt_nomot > 0) return MLPL_OK;
        int rc;
        const uint64_t *rows;
        h_dg[0].kind = 1;
        if ((rc = dg_rows(1, &rows))) return rc;
        dg_sample_nomot.clear();
        unsigned num = 0;
        for (unsigned j = 0; j < 5; ++j)
            if (row_bit(rows, pool_pos[min_sample[j]])) dg_sample_nomot.push_back((int)min_sample[j]), ++num;
        if (num == 0) return MLPL_OK;
        num = row_count(rows);
        if (num < best / 5) return MLPL_OK;
        *degenerate = true;
        const bool dominant = (double)num > 0.7 * (double)dg_cnt_rot;
        if (dominant)
            if (dg_type != (dg_type & (DG_UPGRADE | DG_NO_MOT))) dg_type = DG_NO_MOT;
        if (num > dg_cnt_nomot) {
            if (dominant) dg_type |= DG_UPGRADE;
            dg_cnt_nomot = num;
            for (unsigned j = 0; j < n; ++j) {
                const bool in = row_bit(rows, j);
                dg_in_nomot[pool[j]] = in ? 1 : 0, dg_out_nomot[pool[j]] = in ? 0 : 1;
                if (in && dg_sample_nomot.size() < 5) dg_sample_nomot.push_back((int)pool[j]);
            }
        }
        return MLPL_OK;
    }

    int test_degeneracy(bool *degenerate, bool *upgrade) {
        *degenerate = false, *upgrade = false;
        dg_type = DG_H;  // :1346, taken whenever the homography test is off
        int rc;
        if ((rc = test_rotation(degenerate))) return rc;
        if (dg_type & DG_UPGRADE) *upgrade = true;
        if (dg_type == (unsigned)(DG_ROT_TRANS | DG_UPGRADE))
            if ((rc = test_no_motion(degenerate))) return rc;
        double v[7] = {(double)hyp_count, *degenerate ? 1.0 : 0.0, *upgrade ? 1.0 : 0.0, (double)dg_type, (double)dg_cnt_rot, (double)dg_cnt_nomot, (double)best};
        emit(7, v, 7);
        return MLPL_OK;
    }

    // storeSolution inside the upgrade loops: the inlier flags are whatever the current error array holds below the inlier threshold
    void store_from_errs(const DgErrs &A, unsigned num_inl, const double *E) {
        best = num_inl;
        for (unsigned i = 0; i < n; ++i) flags[i] = A.touched[i] ? A.below_inl[i] : 0;
        for (int w = 0; w < words; ++w) best_bits[w] = 0;
        for (unsigned j = 0; j < n; ++j)
            if (flags[pool[j]]) best_bits[j >> 6] |= 1ull << (j & 63);
        std::memcpy(final_model, E, 72);
        double v[3] = {(double)hyp_count, 0.0, (double)num_inl};
        emit(4, v, 3);
    }
    unsigned standard_stopping_on(const DgErrs &A, const std::vector<unsigned> &idx, unsigned num_outliers) const {
        unsigned c = 0, untouched = 0;
        for (unsigned j : idx) {
            if (A.touched[j] && A.below_count[j])
                ++c;
            else if (!A.touched[j])
                ++untouched;
        }
        return standard_stopping(c, num_outliers - untouched, 1);
    }

    int upgrade_model(unsigned *result) {
        unsigned best_up = best, best_up_rot = dg_cnt_rot, best_up_trans = dg_cnt_trans;
        *result = 0;
        if (n < 2) return MLPL_OK;
        if (!(dg_type & DG_UPGRADE)) {
            double v[4] = {(double)hyp_count, 0.0, 0.0, (double)best_up};
            emit(9, v, 4);
            *result = best_up;
            return MLPL_OK;
        }
        const bool nomot = (dg_type & DG_NO_MOT) != 0;
        const unsigned num_outliers = n - (nomot ? dg_cnt_nomot : dg_cnt_rot);
        if (num_outliers < (nomot ? 1u : 3u)) return MLPL_OK;
        std::vector<unsigned> outlier_indices(num_outliers, 0);
        {
            unsigned c = 0;
            const std::vector<uint8_t> &of = nomot ? dg_out_nomot : dg_out_rot;
            for (unsigned i = 0; i < n; ++i)
                if (of[i]) outlier_indices[c++] = i;
        }
        // the two error arrays: errs[cur] plays err_ptr_[0] (filled with DBL_MAX), errs[1 - cur] holds the errors of the best model
        DgErrs errs[2];
        for (auto &e : errs) e.touched.assign(n, 0), e.below_count.assign(n, 0), e.below_inl.assign(n, 0);
        int cur = 0;
        const int counted = 0;  // current_err_array: the array that is err_ptr_[0] on entry, whatever the swaps do afterwards
        for (unsigned i = 0; i < n; ++i) errs[1].touched[i] = 1, errs[1].below_inl[i] = flags[i];
        unsigned &limit = nomot ? dg_max_nomot : dg_max_rot;
        const unsigned size_nomot = (unsigned)dg_sample_nomot.size();
        const unsigned k_draw = nomot ? 1u : 3u;
        std::vector<unsigned> smp(k_draw);
        struct Cand {
            GlibcRand after;  // the stream after this candidate's draws
            int index[5];
            bool skip;        // translation too short: the loop continues without an evaluation
            double E[9], t[3];
            int model;        // its row block in the launch, -1 = none
        };
        std::vector<Cand> cands;
        unsigned tried = 0;
        int rc;
        for (unsigned i = 0; i < limit;) {
            // candidates do not depend on what is accepted: draw and solve a batch ahead, evaluate it in one launch
            const unsigned want = std::min<unsigned>((unsigned)dg_cap, limit - i);
            cands.clear();
            GlibcRand r2 = rng;
            int B = 0;
            for (unsigned c = 0; c < want; ++c) {
                Cand cd;
                uniform_sample(r2, num_outliers, k_draw, smp);
                cd.skip = false, cd.model = -1;
                if (nomot) {
                    cd.index[0] = (int)outlier_indices[smp[0]];
                    cd.index[1] = dg_sample_nomot[(unsigned)r2.next() % size_nomot];
                    double a0[3], b0[3], a1[3], b1[3];
                    view1((unsigned)cd.index[0], a0), view2((unsigned)cd.index[0], b0), view1((unsigned)cd.index[1], a1), view2((unsigned)cd.index[1], b1);
                    dgm::twopt_translation(a0, b0, a1, b1, cd.t);
                    const double len = std::sqrt(cd.t[0] * cd.t[0] + (cd.t[1] * cd.t[1] + cd.t[2] * cd.t[2]));
                    cd.skip = near_zero(len * 100);
                    if (!cd.skip) {
                        const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
                        dgm::e_from_rt(I3, cd.t, cd.E);
                        h_dg[B].kind = 2;
                        std::memset(h_dg[B].m, 0, 72);
                        std::memcpy(h_dg[B].m, cd.t, 24);
                        cd.model = B++;
                    }
                } else {
                    for (int j = 0; j < 3; ++j) cd.index[j] = (int)outlier_indices[smp[j]];
                    cd.index[3] = dg_sample_rot[0], cd.index[4] = dg_sample_rot[1];
                    std::memset(cd.E, 0, 72);  // solved below, several candidates at a time
                }
                cd.after = r2;
                cands.push_back(cd);
            }
            if (!nomot) {
                // The eigensolver of a candidate (Levenberg-Marquardt on five correspondences, ~6 us) is independent of every other
                // candidate: the batch is solved by a few host threads, each on its own slice.
                auto solve_slice = [&](size_t c0, size_t c1) {
                    for (size_t c = c0; c < c1; ++c) {
                        Cand &cd = cands[c];
                        double f1[5][3], f2[5][3], R[9], t[3];
                        for (int j = 0; j < 5; ++j) view1((unsigned)cd.index[j], f1[j]), view2((unsigned)cd.index[j], f2[j]);
                        dgm::eigensolver(f1, f2, 5, dg_R, R, t);
                        const double len = std::sqrt(t[0] * t[0] + (t[1] * t[1] + t[2] * t[2]));
                        cd.skip = near_zero(len * 100);
                        if (!cd.skip) {
                            for (int k = 0; k < 3; ++k) cd.t[k] = t[k] / len;
                            dgm::e_from_rt(R, cd.t, cd.E);
                        }
                    }
                };
                const size_t nc = cands.size();
                const unsigned hw = std::thread::hardware_concurrency();
                const size_t workers = std::min<size_t>(std::min<size_t>(hw ? hw : 1, 8), nc / 16);
                if (workers >= 2) {
                    std::vector<std::thread> pool;
                    for (size_t w = 1; w < workers; ++w) pool.emplace_back(solve_slice, nc * w / workers, nc * (w + 1) / workers);
                    solve_slice(0, nc / workers);
                    for (auto &th : pool) th.join();
                } else
                    solve_slice(0, nc);
                for (Cand &cd : cands)
                    if (!cd.skip) {
                        h_dg[B].kind = 3;
                        std::memcpy(h_dg[B].m, cd.E, 72);
                        cd.model = B++;
                    }
            }
            const uint64_t *rows = nullptr;
            if (B > 0 && (rc = dg_rows(B, &rows))) return rc;
            unsigned used = 0;
            for (; used < cands.size() && i < limit; ++used, ++i) {
                const Cand &cd = cands[used];
                ++tried;
                {
                    double v[12] = {(double)hyp_count, nomot ? 1.0 : 2.0, (double)i};
                    if (nomot)
                        v[3] = cd.t[0], v[4] = cd.t[1], v[5] = cd.t[2];
                    else
                        std::memcpy(v + 3, cd.E, 72);
                    emit(10, v, 12);
                }
                if (cd.skip) continue;
                const uint64_t *row_a = rows + (size_t)2 * cd.model * words, *row_b = nomot ? row_a + words : row_a;
                unsigned num = 0, tested = 0;
                const unsigned start = pool_index;
                const bool good = sprt_walk(row_a, &num, &tested);
                {
                    double v[11] = {(double)hyp_count, nomot ? -1.0 : 0.0, (double)start, (double)num, (double)tested, good ? 1.0 : 0.0,
                                    sprt_delta, sprt_epsilon, sprt_A, nomot ? dg_thr : thr, (double)num_lo};
                    emit(2, v, 11);
                }
                DgErrs &W = errs[cur];  // evaluateModel / evaluateModelTrans write err_ptr_[0] for the correspondences they reach
                for (unsigned q = 0, j = start; q < tested; ++q, ++j) {
                    if (j > n - 1) j = 0;
                    const unsigned pt = pool[j];
                    W.touched[pt] = 1, W.below_count[pt] = row_bit(row_a, j), W.below_inl[pt] = row_bit(row_b, j);
                }
                if (num > (nomot ? best_up_trans : best_up_rot)) {
                    if (!nomot)
                        for (int j = 2, c = 0; j < 5; ++j) dg_sample_rot[j] = cd.index[c++];
                    if (num > best_up || (near_zero(final_model[0] * 100) && near_zero(final_model[4] * 100) && near_zero(final_model[8] * 100))) {
                        store_from_errs(errs[cur], num, cd.E);
                        cur = 1 - cur;
                        best_up = num;
                        if (nomot) {
                            if (size_nomot > 3) {
                                unsigned k = 0;
                                for (size_t j = 0; j < 3; j++) {
                                    if (dg_sample_nomot[k] == cd.index[1]) {
                                        j--;
                                        k++;
                                        continue;
                                    }
                                    min_sample[j] = (unsigned)dg_sample_nomot[k];
                                    k++;
                                }
                                min_sample[3] = (unsigned)cd.index[1], min_sample[4] = (unsigned)cd.index[0];
                            }
                            dg_cnt_trans = num;
                        } else {
                            for (size_t j = 0; j < 5; j++) min_sample[j] = (unsigned)dg_sample_rot[j];
                        }
                    }
                    (nomot ? best_up_trans : best_up_rot) = num;
                    const unsigned ns = standard_stopping_on(errs[counted], outlier_indices, num_outliers);
                    if (ns < limit) limit = ns;
                }
            }
            // the stream stands where the reference's loop left it
            if (used > 0) rng = cands[used - 1].after;
            if (used < cands.size()) break;
        }
        double v[4] = {(double)hyp_count, nomot ? 1.0 : 2.0, (double)tried, (double)best_up};
        emit(9, v, 4);
        *result = best_up;
        return MLPL_OK;
    }

    int solve(bool *ok) {
        struct PfSolve {
            UsacRun &r;
            uint64_t t0, w0;
            ~PfSolve() { r.prof[PF_SOLVE] += (tsc() - t0) - (r.prof[PF_WAIT] - w0); }
        } pf_solve{*this, tsc(), prof[PF_WAIT]};
        unsigned adaptive = max_hyp;
        bool update_sprt_stopping = true;
        *ok = false;
        if (n < 5 || (prosac && n < prosac_min_stop)) return MLPL_OK;
        const unsigned max2 = max_hyp / 2, max3 = 2 * max_hyp / 3;
        int rc;
        while (hyp_count < adaptive && hyp_count < max_hyp) {
            ++hyp_count;
            if ((hyp_count == max2) && (best == 0))
                thr *= 1.33;
            else if ((hyp_count == max3) && (best == 0))
                thr *= 1.13;
            if (thr != cache_thr) cache.clear(), rows_arena.reset(), cache_bytes = 0, cache_thr = thr;  // bit rows are per threshold
            if (prosac)
                prosac_sample(rng, subset_size, largest_size, stop_len, hyp_count, min_sample);
            else
                uniform_sample(rng, n, 5, min_sample);
            if (!validate_sample(min_sample)) {
                double v[7] = {(double)hyp_count, (double)min_sample[0], (double)min_sample[1], (double)min_sample[2], (double)min_sample[3],
                               (double)min_sample[4], -1.0};
                emit(1, v, 7);
                ++rejected_samples;
                continue;
            }
            UsacKey key;
            for (int i = 0; i < 5; ++i) key.v[i] = min_sample[i];
            auto it = cache.find(key);
            if (it == cache.end()) {
                // nothing refers to a cache entry here: a cache that has outgrown its bound is dropped whole (its entries are speculation and
                // samples already consumed; a sample that recurs is solved again, with the same result)
                if (cache_bytes > kUsacCacheBytes) cache.clear(), rows_arena.reset(), cache_bytes = 0;
                // play the sampler forward under "no event" and solve what is coming in one batch
                const uint64_t pf_b0 = tsc();
                std::vector<UsacKey> batch(1, key);
                std::unordered_set<UsacKey, UsacKeyHash> in_batch;
                in_batch.insert(key);
                GlibcRand r2 = rng;
                unsigned sub = subset_size, lar = largest_size;
                std::vector<unsigned> smp(5);
                unsigned hyp = hyp_count;
                const unsigned horizon = std::min(std::min(adaptive, max_hyp), best == 0 ? (hyp_count < max2 ? max2 - 1 : (hyp_count < max3 ? max3 - 1 : max_hyp)) : max_hyp);
                // Option usac_first_batch (default 0 = as every batch, up to 128): a run's FIRST speculative batch may be smaller -- the best
                // model changes most often at the start, every event throws the rest of a batch away, and the rows of a batch cross PCIe (with
                // 128, PROSAC consumes 83 of the 186 samples it solves per run; with 32 it solves 126).  Speculation only: results cannot depend
                // on it.  Measured on two boxes, 512 image pairs, alternating (tools/c5_opt_ab.py usac_first_batch ...): uniform 14.8-15.3 ->
                // 14.5-15.3 ms at 64, PROSAC 16.0-16.6 -> 15.4-16.6 ms at 32, ConfigUSAC's default refinement 45.4-46.7 -> 44.6-47.6 ms at 32:
                // inside the spread between runs.  Not enabled.
                const int first_cap = ctx->opt_usac_first_batch > 0 ? ctx->opt_usac_first_batch : batch_cap;
                const int cap_now = stats[0] == 0 ? std::min(batch_cap, first_cap) : batch_cap;
                while (hyp < horizon && (int)batch.size() < cap_now) {
                    ++hyp;
                    if (prosac)
                        prosac_sample(r2, sub, lar, stop_len, hyp, smp);
                    else
                        uniform_sample(r2, n, 5, smp);
                    if (!validate_sample(smp)) continue;
                    UsacKey k2;
                    for (int i = 0; i < 5; ++i) k2.v[i] = smp[i];
                    if (cache.find(k2) == cache.end() && in_batch.insert(k2).second) batch.push_back(k2);
                }
                prof[PF_BUILD] += tsc() - pf_b0;
                if ((rc = run_batch(batch))) return rc;
                it = cache.find(key);
            }
            stats[2]++;
            const UsacSampleModels &sm = it->second;
            const unsigned ns = (unsigned)sm.n;
            {
                double v[8] = {(double)hyp_count, (double)min_sample[0], (double)min_sample[1], (double)min_sample[2], (double)min_sample[3],
                               (double)min_sample[4], (double)ns, 0};
                emit(1, v, 7);
                for (unsigned i = 0; i < ns; ++i) {
                    double w[11];
                    w[0] = hyp_count, w[1] = i;
                    std::memcpy(w + 2, sm.m[i].E, 72);
                    emit(5, w, 11);
                }
            }
            model_count += ns;
            bool update_best = false;
            for (unsigned i = 0; i < ns; ++i) {
                const UsacModel &m = sm.m[i];
                if (!m.valid) {
                    double v[2] = {(double)hyp_count, (double)i};
                    emit(6, v, 2);
                    ++rejected_models;
                    continue;
                }
                unsigned inl, tested;
                const unsigned start = pool_index;
                const bool good = sprt_walk(m.bits, &inl, &tested);
                emit_eval(i, start, inl, tested, good);
                if (!good) {
                    points_verified += tested;
                    const double delta_new = (double)inl / tested;
                    if (delta_new > 0 && fabs(sprt_delta - delta_new) / sprt_delta > 0.1) {
                        add_history(hyp_count);
                        sprt_delta = delta_new;
                        design_sprt();
                    }
                } else {
                    points_verified += n;
                    if (inl > best) {
                        update_best = true;
                        best = inl;
                        add_history(hyp_count);
                        sprt_epsilon = (double)best / n;
                        design_sprt();
                        update_sprt_stopping = true;
                        store_solution(i, best, m.bits, m.E);
                    }
                }
            }
            if (update_best && dg_on) {  // USAC.h:509-528
                bool degenerate = false, upgrade = false;
                if ((rc = test_degeneracy(&degenerate, &upgrade))) return rc;
                if (degenerate && upgrade) {
                    unsigned up = 0;
                    if ((rc = upgrade_model(&up))) return rc;
                    if (up > best) best = up;
                }
            }
            if (update_best) {
                unsigned lo = 0;
                // the cache entry `sm` may be invalidated by nothing below (no insertion during LO)
                if ((rc = refine ? local_optimization5(best, &lo) : local_optimization(best, &lo))) return rc;
                if (dg_losac) {  // USAC.h:540-556 (testDegeneracyLOSAC: the 8-point refinements)
                    bool degenerate = false, upgrade = false;
                    if ((rc = test_degeneracy(&degenerate, &upgrade))) return rc;
                    if (degenerate && upgrade) {
                        unsigned up = 0;
                        if ((rc = upgrade_model(&up))) return rc;
                        if (up > lo) lo = up;
                    }
                }
                if (lo > best) best = lo;
                if (prosac && hyp_count <= prosac_max_samples)
                    adaptive = prosac_stopping(hyp_count);
                else
                    adaptive = standard_stopping(best, n, 5);
            }
            if (!prosac) {
                if (hyp_count >= adaptive && update_sprt_stopping) {
                    adaptive = sprt_stopping(best, n);
                    update_sprt_stopping = false;
                }
            }
        }
        *ok = true;
        return MLPL_OK;
    }
};

}  // namespace

// A run's configuration from the caller's parameters (estimateEssentialMatUsac, usac_estimations.cpp:283-470)
static void usac_configure(UsacRun &R, mlpl_ctx *ctx, const double *d_p1, const double *d_p2, int n, const mlpl_usac_params *P, hipStream_t s) {
    R.ctx = ctx, R.s = s, R.L.s = s, R.d_p1 = d_p1, R.d_p2 = d_p2, R.n = (unsigned)n;
    R.max_hyp = (unsigned)P->max_hyp, R.conf = P->conf, R.thr = P->th * P->th;
    R.prosac = P->sorted_idx != nullptr;
    if (R.prosac) R.sorted_idx.assign(P->sorted_idx, P->sorted_idx + n);
    R.prosac_beta = P->prosac_beta, R.sprt_delta = P->sprt_delta, R.sprt_epsilon = P->sprt_epsilon;
    R.sprt_mS = P->sprt_mS, R.sprt_tM = P->sprt_tM;
    R.lo_stepwise = ctx->opt_usac_lo_stepwise;
    R.refine = P->refine;
    R.sprt_fast = ctx->opt_usac_sprt_fast != 0;
    R.dg_on = P->check_degeneracy != 0;
    R.dg_losac = R.dg_on && (P->check_degeneracy & 2) != 0;
    if (R.dg_on) R.dg_thr = 1.0 - std::cos(std::atan(P->th_pixels / P->focal_length));  // EssentialMatEstimator.h:349
    R.rng.seed(P->seed);
}
static void usac_results(const UsacRun &R, bool ok, double *results) {
    const double fin[12] = {ok ? 1.0 : 0.0,
                            (double)R.hyp_count,
                            (double)R.model_count,
                            (double)R.rejected_samples,
                            (double)R.rejected_models,
                            (double)R.best,
                            (double)R.points_verified,
                            (double)R.num_lo,
                            R.history.empty() ? 0.0 : R.history.back().delta,
                            R.history.empty() ? 0.0 : R.history.back().epsilon,
                            R.sprt_delta,
                            R.sprt_epsilon};
    std::memcpy(results, fin, sizeof(fin));
}
// the inlier mask of the run's model to device memory (through the run's pinned block; the run waits for the copy)
static int usac_mask_to_device(UsacRun &R, uint8_t *d_mask) {
    std::memcpy(R.h_lo5_flags, R.flags.data(), (size_t)R.n);
    hub_copy_bytes(R.L, R.d_lo5_flags_src, d_mask, (size_t)R.n);
    return R.L.sync();
}

// h_p1 / h_p2: the caller's host copies of the correspondences when it has them (else they are fetched); h_mask: host destination of the
// inlier mask (d_mask is then not written)
int usac_essential_dev(mlpl_ctx *ctx, const double *d_p1, const double *d_p2, int n, const mlpl_usac_params *P, double *E, uint8_t *d_mask,
                       double *results, hipStream_t s, const double *h_p1 = nullptr, const double *h_p2 = nullptr, uint8_t *h_mask = nullptr) {
    UsacRun R;
    usac_configure(R, ctx, d_p1, d_p2, n, P, s);
    R.trace_buf = ctx->usac_trace, R.trace_cap = ctx->usac_trace_cap, R.trace_len = ctx->usac_trace_len;
    if (h_p1 && h_p2) {
        R.hp1.assign(h_p1, h_p1 + (size_t)2 * n), R.hp2.assign(h_p2, h_p2 + (size_t)2 * n);
    } else {
        R.hp1.resize((size_t)2 * n), R.hp2.resize((size_t)2 * n);
        MLPL_HIP_TRY(hipMemcpyAsync(R.hp1.data(), d_p1, (size_t)n * 16, hipMemcpyDeviceToHost, s));
        MLPL_HIP_TRY(hipMemcpyAsync(R.hp2.data(), d_p2, (size_t)n * 16, hipMemcpyDeviceToHost, s));
        MLPL_HIP_TRY(hipStreamSynchronize(s));
    }
    int rc;
    bool ok = false;
    if (n >= 5) {
        if ((rc = R.setup())) return rc;
        if ((rc = R.solve(&ok))) return rc;
    }
    ctx->usac_trace_len = R.trace_len;
    std::memcpy(ctx->last_usac_stats, R.stats, sizeof(R.stats));
    {  // what estimateEssentialMatUsac reports of the degeneracy tests (usac_estimations.cpp:564-636, 689-726)
        double *d = ctx->last_usac_degen;
        std::memset(d, 0, sizeof(ctx->last_usac_degen));
        d[0] = R.dg_on ? 1.0 : 0.0, d[1] = R.dg_cnt_rot, d[2] = R.dg_cnt_nomot, d[3] = R.dg_type;
        std::memcpy(d + 4, R.dg_R, 72);
        std::free(ctx->last_usac_flags);
        ctx->last_usac_flags = nullptr, ctx->last_usac_flags_n = 0;
        if (R.dg_on && n > 0 && R.dg_in_rot.size() == (size_t)n) {
            ctx->last_usac_flags = (uint8_t *)std::malloc((size_t)2 * n);
            if (!ctx->last_usac_flags) return MLPL_E_NOMEM;
            std::memcpy(ctx->last_usac_flags, R.dg_in_rot.data(), (size_t)n);
            std::memcpy(ctx->last_usac_flags + n, R.dg_in_nomot.data(), (size_t)n);
            ctx->last_usac_flags_n = n;
        }
    }
    if (results) usac_results(R, ok, results);
    if (!ok) {
        set_error("mlpl_usac_essential: too few correspondences (%d)", n);
        return MLPL_E_FAILED;
    }
    std::memcpy(E, R.final_model, 72);
    if (h_mask) {
        std::memcpy(h_mask, R.flags.data(), (size_t)n);
    } else if (d_mask) {
        if ((rc = usac_mask_to_device(R, d_mask))) return rc;
    }
    return MLPL_OK;
}

// ---- a batch of USAC problems: every run a fiber on a worker thread, every launch merged over the runs (batch_hub.h) -----------------------------------
// Problem b: correspondences d_p1 / d_p2 + b * stride * 2 (counts[b] of them), parameters params[b] (its seed, its PROSAC order, ...).
// Outputs per problem: status (0, MLPL_E_FAILED = solve() refused, other < 0 = error), E, results[12], its inlier mask in d_masks + b *
// stride (optional), degen (optional, 16 doubles as mlpl_usac_last_degeneracy), its decision trace (optional).  Every problem's outputs
// are those of mlpl_usac_essential_dev on it alone: a run sees exactly its own launches' results, whatever ran beside them.
constexpr int kUsacBatchRuns = 128;
  // runs advancing together (threads of an internal batch)

struct UsacBatchTrace {
    double *buf;       // [B][cap][16]
    int cap;
    int32_t *lens;     // [B]
};

int usac_essential_batch_dev(mlpl_ctx *ctx, int B, const double *d_p1, const double *d_p2, int stride, const int32_t *counts,
                             const mlpl_usac_params *params, double *E, uint8_t *d_masks, double *results, int32_t *status, double *degen,
                             const UsacBatchTrace *trace, hipStream_t s, const CohortFeed *feed = nullptr) {
    if (B <= 0) return MLPL_OK;
    int rc;
    // the sequential parts read the correspondences on the host: one copy of the whole block through the batch's pinned memory
    size_t max_dev = 0, max_pin = 0;
    for (int b = 0; b < B; ++b) {  // (a feed delivers the counts later: blocks for `stride` correspondences)
        const UsacRun::UsacLayout Y = UsacRun::usac_layout((unsigned)std::max(feed ? stride : counts[b], 1), params[b].refine, params[b].check_degeneracy != 0);
        max_dev = std::max(max_dev, Y.dev_total), max_pin = std::max(max_pin, Y.pin_total);
    }
    // Cohorts of <= kUsacBatchRuns runs, two of them in flight (batch_hub.h kHubLanes): while one cohort's merged launches execute, the
    // other cohort's runs walk their bit rows on the host.  A batch that fits one cohort is split in two halves for the same reason.
    int n_cohorts = 0, lanes = 0;
    const int cohort = hub_cohort_size(ctx, B, kUsacBatchRuns, &n_cohorts, &lanes, hub_usac_lanes_default(params[0].refine));
    if (feed && (feed->cohort != cohort || feed->n_cohorts != n_cohorts)) {
        set_error("mlpl_usac_essential_batch_dev: the feed's cohorts are not the estimator's");
        return MLPL_E_INTERNAL;
    }
    const size_t pts_bytes = ((size_t)B * stride * 16 + 255) & ~(size_t)255;
    void *pblk, *dblk;
    if ((rc = pinned_batch_get(ctx, 2 * pts_bytes + (size_t)lanes * cohort * max_pin, &pblk))) return rc;
    if ((rc = ws_get(ctx, WS_BATCH_RUNS, (size_t)lanes * cohort * max_dev, &dblk))) return rc;
    char *pin = (char *)pblk, *pin_dev = nullptr;
    {
        void *alias = nullptr;
        MLPL_HIP_TRY(hipHostGetDevicePointer(&alias, pin, 0));
        pin_dev = (char *)alias;
    }
    double *h_p1 = (double *)pin, *h_p2 = (double *)(pin + pts_bytes);
    // The sequential parts read the correspondences on the host (normalisation, sample validation, the choices of the 5-point refinements,
    // the degeneracy tests): they cross PCIe once, cohort by cohort on a stream of their own, so that the first cohort starts after its
    // own slice has arrived (512 problems of 5000: 3.4 ms for all, 0.85 ms for the first quarter) and the rest travels beside its work.
    if (!feed) MLPL_HIP_TRY(hipStreamSynchronize(s));  // everything the caller queued before is done: the lanes' own streams need no other ordering
    HubStreams *hres = hub_resources(ctx);
    if (!hres->copy) MLPL_HIP_TRY(hipStreamCreateWithFlags(&hres->copy, hipStreamNonBlocking));
    std::vector<hipEvent_t> arrived((size_t)n_cohorts, nullptr);
    struct EventsGuard {
        std::vector<hipEvent_t> &ev;
        ~EventsGuard() {
            for (hipEvent_t e : ev)
                if (e) (void)hipEventDestroy(e);
        }
    } events_guard{arrived};
    for (int c = 0; c < n_cohorts && !feed; ++c) {
        const int b0 = c * cohort, nb = std::min(cohort, B - b0);
        int max_n = 1;  // only the rows in use: counts[b] <= max_n of the stride rows of a problem
        for (int b = b0; b < b0 + nb; ++b) max_n = std::max(max_n, counts[b]);
        const size_t at = (size_t)b0 * stride * 2;
        MLPL_HIP_TRY(hipMemcpy2DAsync(h_p1 + at, (size_t)stride * 16, d_p1 + at, (size_t)stride * 16, (size_t)max_n * 16, (size_t)nb, hipMemcpyDeviceToHost, hres->copy));
        MLPL_HIP_TRY(hipMemcpy2DAsync(h_p2 + at, (size_t)stride * 16, d_p2 + at, (size_t)stride * 16, (size_t)max_n * 16, (size_t)nb, hipMemcpyDeviceToHost, hres->copy));
        MLPL_HIP_TRY(hipEventCreateWithFlags(&arrived[(size_t)c], hipEventDisableTiming));
        MLPL_HIP_TRY(hipEventRecord(arrived[(size_t)c], hres->copy));
    }
    char *run_pin = pin + 2 * pts_bytes, *run_pin_dev = pin_dev + 2 * pts_bytes;
    const auto t_all = std::chrono::steady_clock::now();
    const uint64_t tsc_all0 = UsacRun::tsc();
    struct LaneOut {
        long long rounds = 0, merged = 0, host_us = 0, device_us = 0, spawn_us = 0;
        int first_err = 0;
        std::string first_msg;
    };
    LaneOut lane_out[kHubLanes];
    hipStream_t lane_stream[kHubLanes];
    for (int l = 0; l < lanes; ++l)
        if ((rc = hub_lane_stream(ctx, l, s, &lane_stream[l]))) return rc;
    std::atomic<uint64_t> prof_sum[UsacRun::PF_NUM], run_stats[5];
    for (auto &v : prof_sum) v.store(0);
    for (auto &v : run_stats) v.store(0);
    std::vector<double> cohort_ms((size_t)n_cohorts * 3, 0.0);  // debug (MLPL_USAC_PROF): when a cohort was handed over, when its runs started, when they had all finished
    auto since = [&]() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_all).count(); };
    auto serve_lane = [&](int l) {
        LaneOut &LO = lane_out[l];
        const hipStream_t ls = lane_stream[l];
        if (l > 0 && hipSetDevice(ctx->device) != hipSuccess) {  // a lane's own thread starts on device 0
            LO.first_err = MLPL_E_INTERNAL, LO.first_msg = "batched estimator: hipSetDevice failed on a lane thread";
            return;
        }
        for (int c = l; c < n_cohorts; c += lanes) {
            const int b0 = c * cohort, nb = std::min(cohort, B - b0);
            if (feed) {  // the producer's event, its host-side hand-over (counts, orders), then this cohort's correspondences to the host
                int frc = hipEventSynchronize(feed->ready[c]) == hipSuccess ? MLPL_OK : MLPL_E_HIP;
                if (!frc) frc = feed->on_ready(c, hub_resources(ctx)->lane[l].threads);
                if (!frc) {
                    int max_n = 1;
                    for (int b = b0; b < b0 + nb; ++b) max_n = std::max(max_n, counts[b]);
                    const size_t at = (size_t)b0 * stride * 2;
                    if (hipMemcpy2DAsync(h_p1 + at, (size_t)stride * 16, d_p1 + at, (size_t)stride * 16, (size_t)max_n * 16, (size_t)nb, hipMemcpyDeviceToHost, ls) != hipSuccess ||
                        hipMemcpy2DAsync(h_p2 + at, (size_t)stride * 16, d_p2 + at, (size_t)stride * 16, (size_t)max_n * 16, (size_t)nb, hipMemcpyDeviceToHost, ls) != hipSuccess ||
                        hipStreamSynchronize(ls) != hipSuccess)
                        frc = MLPL_E_HIP;
                }
                if (frc) {
                    LO.first_err = frc, LO.first_msg = "mlpl_usac_essential_batch_dev: the hand-over of a cohort failed";
                    break;
                }
            } else if (hipEventSynchronize(arrived[(size_t)c]) != hipSuccess) {
                LO.first_err = MLPL_E_INTERNAL, LO.first_msg = "mlpl_usac_essential_batch_dev: the copy of the correspondences failed";
                break;
            }
            cohort_ms[(size_t)c * 3] = since();
            BatchHub hub(ctx, ls, nb, l);
            const auto t_spawn = std::chrono::steady_clock::now();
            std::vector<UsacBufs> bufs((size_t)nb);
            std::vector<std::string> msgs((size_t)nb);
            for (int k = 0; k < nb; ++k) {
                const size_t slot = (size_t)l * cohort + k;
                bufs[k].dev = (char *)dblk + slot * max_dev;
                bufs[k].pin = run_pin + slot * max_pin, bufs[k].pin_dev = run_pin_dev + slot * max_pin;
            }
            HubThreads &pool = hub_resources(ctx)->lane[l].threads;
            pool.start(nb, [&](int k) {
                const int b = b0 + k;
                const int n = counts[b];
                HubRun &hr = hub.run(k);
                int r = MLPL_OK;
                bool ok = false;
                try {
                    UsacRun R;
                    usac_configure(R, ctx, d_p1 + (size_t)b * stride * 2, d_p2 + (size_t)b * stride * 2, n, &params[b], ls);
                    R.L.hub = &hub, R.L.run = &hr, R.bufs = &bufs[k];
                    if (trace && trace->buf) R.trace_buf = trace->buf + (size_t)b * trace->cap * 16, R.trace_cap = trace->cap;
                    R.hp1.assign(h_p1 + (size_t)b * stride * 2, h_p1 + (size_t)b * stride * 2 + (size_t)2 * std::max(n, 0));
                    R.hp2.assign(h_p2 + (size_t)b * stride * 2, h_p2 + (size_t)b * stride * 2 + (size_t)2 * std::max(n, 0));
                    if (n >= 5) {
                        r = R.setup();
                        if (!r) r = R.solve(&ok);
                    }
                    for (int q = 0; q < UsacRun::PF_NUM; ++q) prof_sum[q].fetch_add(R.prof[q], std::memory_order_relaxed);
                    for (int q = 0; q < 5; ++q) run_stats[q].fetch_add((uint64_t)R.stats[q], std::memory_order_relaxed);
                    if (!r) {
                        usac_results(R, ok, results + (size_t)b * 12);
                        if (trace && trace->lens) trace->lens[b] = R.trace_len;
                        if (degen) {
                            double *d = degen + (size_t)b * 16;
                            std::memset(d, 0, 128);
                            d[0] = R.dg_on ? 1.0 : 0.0, d[1] = R.dg_cnt_rot, d[2] = R.dg_cnt_nomot, d[3] = R.dg_type;
                            std::memcpy(d + 4, R.dg_R, 72);
                        }
                        if (ok) {
                            std::memcpy(E + (size_t)b * 9, R.final_model, 72);
                            if (d_masks) r = usac_mask_to_device(R, d_masks + (size_t)b * stride);
                        } else
                            r = MLPL_E_FAILED;
                    }
                } catch (const std::bad_alloc &) {
                    r = MLPL_E_NOMEM;
                    set_error("mlpl_usac_essential_batch_dev: out of host memory");
                } catch (...) {  // (anything else -- a length_error of a vector, say -- must not unwind off the fiber's makecontext frame)
                    r = MLPL_E_INTERNAL;
                    set_error("mlpl_usac_essential_batch_dev: a run ended with an unexpected C++ exception");
                }
                if (r && r != MLPL_E_FAILED) msgs[k] = mlpl_last_error();
                status[b] = r;
                hub.finish(hr);
            }, ctx->opt_hub_workers);
            LO.spawn_us += std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_spawn).count();
            cohort_ms[(size_t)c * 3 + 1] = since();
            const int hrc = hub.serve();
            pool.wait();
            cohort_ms[(size_t)c * 3 + 2] = since();
            LO.rounds += hub.rounds(), LO.merged += hub.merged_launches(), LO.host_us += hub.host_us(), LO.device_us += hub.device_us();
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
    long long rounds = 0, merged = 0, host_us = 0, device_us = 0, spawn_us = 0;
    int first_err = 0;
    std::string first_msg;
    for (int l = 0; l < lanes; ++l) {
        const LaneOut &LO = lane_out[l];
        rounds += LO.rounds, merged += LO.merged, host_us += LO.host_us, device_us += LO.device_us, spawn_us += LO.spawn_us;
        if (LO.first_err && !first_err) first_err = LO.first_err, first_msg = LO.first_msg;
    }
    ctx->last_usac_stats[0] = rounds, ctx->last_usac_stats[1] = merged, ctx->last_usac_stats[2] = host_us, ctx->last_usac_stats[3] = device_us;
    ctx->last_usac_stats[4] = spawn_us;
    ctx->last_usac_stats[6] = lanes, ctx->last_usac_stats[7] = n_cohorts;   // the configuration the call actually ran with (bench.py records it)
    ctx->last_usac_stats[5] = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_all).count();
    if (const char *pe = getenv("MLPL_USAC_PROF"); pe && *pe == '1') {  // debug: where the runs' host time goes (TSC ticks summed over the runs; waits excluded)
        static const char *names[UsacRun::PF_NUM] = {"setup", "sampler play-forward", "batch hand-over (host)", "waiting for the device", "local optimisation (host)", "solve() total"};
        const double ticks_per_us = (double)(UsacRun::tsc() - tsc_all0) / std::max<long long>(1, ctx->last_usac_stats[5]);
        std::fprintf(stderr, "[mlpl usac prof] %d runs, %d lanes, call %.2f ms:", B, lanes, ctx->last_usac_stats[5] / 1e3);
        for (int q = 0; q < UsacRun::PF_NUM; ++q) std::fprintf(stderr, " %s %.1f us per run;", names[q], (double)prof_sum[q].load() / ticks_per_us / B);
#ifdef MLPL_USAC_LO_STAMPS
        {
            unsigned long long h[8];
            if (hipMemcpyFromSymbol(h, HIP_SYMBOL(g_usac_lo_stamps), sizeof(h)) == hipSuccess && h[7]) {
                static const char *sec[7] = {"sample fit", "evaluation", "bit row", "membership + prefix", "covariance sums", "reduction", "fit"};
                std::fprintf(stderr, " usac_lo, 100 MHz ticks per workgroup (%llu workgroups):", h[7]);
                for (int q = 0; q < 7; ++q) std::fprintf(stderr, " %s %.1f;", sec[q], (double)h[q] / (double)h[7]);
                const unsigned long long z[8] = {0, 0, 0, 0, 0, 0, 0, 0};
                (void)hipMemcpyToSymbol(HIP_SYMBOL(g_usac_lo_stamps), z, sizeof(z));
            }
        }
#endif
        std::fprintf(stderr, " per run: %.2f speculative batches, %.1f samples solved, %.1f consumed, %.2f local optimisations, %.2f resumed chains;",
                     (double)run_stats[0].load() / B, (double)run_stats[1].load() / B, (double)run_stats[2].load() / B, (double)run_stats[3].load() / B,
                     (double)run_stats[4].load() / B);
        std::fprintf(stderr, " cohorts (handed over / runs started / all finished, ms):");
        for (int c = 0; c < n_cohorts; ++c) std::fprintf(stderr, " [%.2f %.2f %.2f]", cohort_ms[(size_t)c * 3], cohort_ms[(size_t)c * 3 + 1], cohort_ms[(size_t)c * 3 + 2]);
        std::fprintf(stderr, "\n");
    }
    if (first_err) {
        if (!first_msg.empty()) set_error("%s", first_msg.c_str());
        return first_err;
    }
    return MLPL_OK;
}
