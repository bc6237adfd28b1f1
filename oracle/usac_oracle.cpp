// usac_oracle.cpp -- CPU restatement of the reference's USAC essential-matrix estimation with the Nister minimal solver
// (second half).  TEST INFRASTRUCTURE (see oracle.h): the checker of the device path; never linked into the product.
//
// Follows, under /root/reference/matchinglib_poselib/source/poselib/ :
//   include/usac/estimators/USAC.h               solve :335-620, generateUniformRandomSample :628-645, initPROSAC :699-808,
//                                                generatePROSACMinSample :815-852, updatePROSACStopping :859-913, designSPRTTest :920-942,
//                                                locallyOptimizeSolution :947-1073, findInliers :1079-1093, updateStandardStopping
//                                                :1099-1126, updateSPRTStopping :1133-1171, computeExpSPRT :1178-1192, storeSolution :1217-1239
//   include/usac/estimators/EssentialMatEstimator.h   initProblem :189-352, generateMinimalSampleModels :384-398 + :505-520,
//                                                generateRefinedModel REFINE_WEIGHTS :540-599 / REFINE_NISTER :757-850, validateSample
//                                                :1043-1078, validateModel :1085-1104, evaluateModel :1110-1178, findWeights :2366-2390
//   source/usac/usac_estimations.cpp:283-470     estimateEssentialMatUsac: the configuration (0.99, 50000 hypotheses, LO 5 x 14, 2.0, 4;
//                                                PROSAC 1000 samples, stop length 20, 0.99) and srand(seed) before the pool shuffle
//   source/usac/utils/FundmatrixFunctions.cpp    normalizePoints :7-62, computeDataMatrix :64-88, formCovMat :312-332, singulF :334-361,
//                                                computeEpipole :363-373, getOriSign :375-381
//   include/usac/estimators/EssentialMatEstimator.h   with check_degeneracy: testSolutionDegeneracy :1334-1362, testSolutionDegeneracyRot
//                                                :1511-1663, testSolutionDegeneracyNoMot :1838-1911, evaluateModelTrans :1264-1327,
//                                                upgradeDegenerateModel :1917-2365 (its two pose branches), as configured by
//                                                usac_estimations.cpp:367-375, 443-456 without the homography test
//   source/usac/utils/PoseFunctions.cpp          getRotError :43-70, getNoMotError :118-141
//   thirdparty/opengv/src/                       relative_pose/methods.cpp twopt :57-86, twopt_rotationOnly :98-122, rotationOnly :128-160,
//                                                eigensolver :496-551; math/arun.cpp :33-56; triangulation/methods.cpp triangulate2 :94-117
// The degeneracy part is pinned by oracle/_ref/usac_ref with check_degeneracy (tests/golden/usac_degen_trace.npz).  Its eigensolver is
// restated as what it computes, not how: a damped Newton iteration on the smallest eigenvalue of M(R) (Jacobi eigenvalues, central
// differences) from the same start rotation.  OpenGV's own Levenberg-Marquardt follows the rounding noise of its forward-difference
// Jacobian (tests/test_usac_degen_math.py), so no restatement can follow it step by step; the translation is the eigenvector OpenGV
// takes, column 0 of Eigen::EigenSolver's unordered decomposition (eigen_order3 below).
//
// Pinned by oracle/_ref/usac_ref: the reference's USAC.h + usac/utils + vendored OpenGV compiled in place, turn by turn
// (tests/golden/usac_trace.npz).  What is restated from published algorithms rather than compiled from the reference: the smallest
// singular vector of the 9 x 9 covariance matrix and the rank-2 projection (reference: ccmath svdu1v / svduv; here one-sided Jacobi --
// the vector is unique up to sign, and a sign does not change a Sampson error), the 5-point solver (reference: OpenGV fivept_nister;
// here oracle_run5point -- the same solution set), std::random_shuffle (libstdc++: for i in 1..n-1 swap(i, rand() % (i+1))).
// ORDER CONVENTION for the solutions of one sample: ascending E(0,0) of the unit-Frobenius matrix whose largest-magnitude element is
// positive -- see oracle/ref_drivers/usac_ref.cpp for why OpenGV's own order cannot be restated.
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <limits>
#include <vector>

#include "oracle.h"

namespace {

struct WaldTest {
    double epsilon, delta, A;
    unsigned k;
};

// (unsigned int) of a double the way x86-64 compilers do it: cvttsd2si to 64 bits, low half
unsigned to_uint(double v) {
    long long w = (v > -9.2233720368547758e18 && v < 9.2233720368547758e18) ? (long long)v : (long long)0x8000000000000000ull;
    return (unsigned)(unsigned long long)w;
}

// The position Eigen::EigenSolver<Matrix3d> gives each eigenvalue (Eigen::RealSchur: Hessenberg form, EISPACK hqr with deflation from the
// bottom; thirdparty/opengv/third_party_notuse/Eigen/src/Eigenvalues/RealSchur.h:245-525, HessenbergDecomposition.h): OpenGV's eigensolver
// takes its translation from position 0 (modules/main.cpp:646-659), which holds the smallest eigenvalue in about a third of the cases only.
// Reflectors in Eigen's formulation and operation order: after convergence, rounding-level entries decide the next one.  Pinned by
// tests/golden/eigen_order3.npz (oracle/_ref/eigen_svd3 eig).
// Householder reflector H = I - tau (1, ess)(1, ess)^T with H v = beta e1, and its application to a block, in the formulation (and
// the operation order) of Eigen's makeHouseholder / applyHouseholderOnTheLeft / OnTheRight: entries at rounding-noise level decide
// what the next reflector does once the iteration has converged, so the arithmetic has to be the same.
struct Refl {
    double ess[2], tau, beta;
    int n;  // length of v
};
inline Refl make_reflector(const double *v, int n) {
    Refl h;
    h.n = n, h.ess[0] = h.ess[1] = 0;
    double tail = 0;
    for (int i = 1; i < n; ++i) tail += v[i] * v[i];
    const double c0 = v[0];
    if (tail == 0.0) {
        h.tau = 0, h.beta = c0;
    } else {
        h.beta = std::sqrt(c0 * c0 + tail);
        if (c0 >= 0) h.beta = -h.beta;
        for (int i = 1; i < n; ++i) h.ess[i - 1] = v[i] / (c0 - h.beta);
        h.tau = (h.beta - c0) / h.beta;
    }
    return h;
}
// rows r0 .. r0 + n - 1, columns c0 .. c1 - 1
inline void apply_left(double T[3][3], const Refl &h, int r0, int c0, int c1) {
    for (int j = c0; j < c1; ++j) {
        double tmp = 0;
        for (int i = 1; i < h.n; ++i) tmp = (i == 1) ? h.ess[0] * T[r0 + 1][j] : tmp + h.ess[i - 1] * T[r0 + i][j];
        tmp += T[r0][j];
        T[r0][j] -= h.tau * tmp;
        for (int i = 1; i < h.n; ++i) T[r0 + i][j] -= (h.tau * h.ess[i - 1]) * tmp;
    }
}
// columns k0 .. k0 + n - 1, rows r0 .. r1 - 1
inline void apply_right(double T[3][3], const Refl &h, int k0, int r0, int r1) {
    for (int i = r0; i < r1; ++i) {
        double tmp = 0;
        for (int j = 1; j < h.n; ++j) tmp = (j == 1) ? T[i][k0 + 1] * h.ess[0] : tmp + T[i][k0 + j] * h.ess[j - 1];
        tmp += T[i][k0];
        T[i][k0] -= h.tau * tmp;
        for (int j = 1; j < h.n; ++j) T[i][k0 + j] -= (h.tau * tmp) * h.ess[j - 1];
    }
}
inline void eigen_order3(const double *M, double *d) {
    double T[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) T[r][c] = M[3 * r + c];
    {  // Hessenberg form (HessenbergDecomposition): the reflector of (T10, T20) on rows / columns 1, 2
        const double v[2] = {T[1][0], T[2][0]};
        const Refl h = make_reflector(v, 2);
        T[1][0] = h.beta;
        apply_left(T, h, 1, 1, 3);
        apply_right(T, h, 1, 0, 3);
        T[2][0] = 0.0;
    }
    int iu = 2, iter = 0, total = 0;
    double exshift = 0;
    double norm = 0;
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < (j + 2 < 3 ? j + 2 : 3); ++i) norm += std::fabs(T[i][j]);
    if (norm != 0)
        while (iu >= 0) {
            int il = iu;
            while (il > 0) {
                double s = std::fabs(T[il - 1][il - 1]) + std::fabs(T[il][il]);
                if (s == 0.0) s = norm;
                if (std::fabs(T[il][il - 1]) < DBL_EPSILON * s) break;
                il--;
            }
            if (il == iu) {
                T[iu][iu] += exshift;
                if (iu > 0) T[iu][iu - 1] = 0.0;
                iu--;
                iter = 0;
            } else if (il == iu - 1) {
                const double a = T[iu - 1][iu - 1], dd = T[iu][iu], b = T[iu - 1][iu], c = T[iu][iu - 1];
                const double p = 0.5 * (a - dd), q = p * p + c * b;
                T[iu][iu] += exshift, T[iu - 1][iu - 1] += exshift;
                if (q >= 0) {  // two real eigenvalues: the rotation puts the eigenvalue of the eigenvector (p +- z, c) first
                    const double z = std::sqrt(std::fabs(q)), mean = 0.5 * (a + dd) + exshift;
                    if (p >= 0)
                        T[iu - 1][iu - 1] = mean + z, T[iu][iu] = mean - z;
                    else
                        T[iu - 1][iu - 1] = mean - z, T[iu][iu] = mean + z;
                    T[iu][iu - 1] = 0.0;
                }
                if (iu > 1) T[iu - 1][iu - 2] = 0.0;
                iu -= 2;
                iter = 0;
            } else {  // iu = 2, il = 0: one Francis step on the whole matrix
                double s0 = T[2][2], s1 = T[1][1], s2 = T[2][1] * T[1][2];
                if (iter == 10) {
                    exshift += s0;
                    for (int i = 0; i <= iu; ++i) T[i][i] -= s0;
                    const double s = std::fabs(T[2][1]) + std::fabs(T[1][0]);
                    s0 = 0.75 * s, s1 = 0.75 * s, s2 = -0.4375 * s * s;
                }
                if (iter == 30) {
                    double s = (s1 - s0) / 2.0;
                    s = s * s + s2;
                    if (s > 0) {
                        s = std::sqrt(s);
                        if (s1 < s0) s = -s;
                        s = s + (s1 - s0) / 2.0;
                        s = s0 - s2 / s;
                        exshift += s;
                        for (int i = 0; i <= iu; ++i) T[i][i] -= s;
                        s0 = s1 = s2 = 0.964;
                    }
                }
                ++iter, ++total;
                if (total > 120) break;
                const double Tmm = T[0][0], r = s0 - Tmm, s = s1 - Tmm;
                const double v[3] = {(r * s - s2) / T[1][0] + T[0][1], T[1][1] - Tmm - r - s, T[2][1]};
                const Refl h1 = make_reflector(v, 3);
                if (h1.beta != 0.0) {
                    apply_left(T, h1, 0, 0, 3);
                    apply_right(T, h1, 0, 0, 3);
                }
                const double v2[2] = {T[1][0], T[2][0]};
                const Refl h2 = make_reflector(v2, 2);
                if (h2.beta != 0.0) {
                    T[1][0] = h2.beta;
                    apply_left(T, h2, 1, 1, 3);
                    apply_right(T, h2, 1, 0, 3);
                }
                T[2][0] = 0.0;
            }
        }
    d[0] = T[0][0], d[1] = T[1][1], d[2] = T[2][2];
}

struct Usac {
    // configuration
    unsigned n = 0, max_hyp = 50000;
    double conf = 0.99, thr = 0;
    int refine = 0;
    bool prosac = false;
    unsigned prosac_max_samples = 1000, prosac_min_stop = 20;
    double prosac_beta = 0.09, prosac_non_rand_conf = 0.99;
    std::vector<unsigned> sorted_idx;
    double sprt_tM = 2314.0, sprt_mS = 8.5, sprt_delta = 0.05, sprt_epsilon = 0.15, sprt_A = 0;
    unsigned lo_sample = 14, lo_reps = 5, lo_steps = 4;
    double lo_mult = 2.0;
    // data
    const double *p1 = nullptr, *p2 = nullptr;
    std::vector<double> pd, pn;  // 6 per point: denormalised / normalised
    double T1[9], T2[9], T2t[9], T1i[9], T2ti[9];
    std::vector<double> data_matrix;
    oracle_glibc_rand rng;
    // state
    std::vector<unsigned> min_sample, pool;
    unsigned pool_index = 0;
    std::vector<double> errs[2];
    int cur = 0;  // errs[cur] = scratch (err_ptr_[0]), errs[1 - cur] = errors of the best model
    double models[10][9], models_denorm[10][9];
    std::vector<WaldTest> history;
    unsigned last_wald_update = 0;
    unsigned subset_size = 5, largest_size = 5, stop_len = 0;
    std::vector<unsigned> growth, non_random, maximality;
    // results
    unsigned hyp_count = 0, model_count = 0, rejected_samples = 0, rejected_models = 0, best = 0, points_verified = 0, num_lo = 0;
    std::vector<unsigned> flags, best_sample;
    double final_model[9];
    unsigned num_prev_best_lo = 0;
    // trace
    double *events = nullptr;
    int event_cap = 0, n_events = 0;

    void emit(double type, const double *v, int nv) {
        if (!events || n_events >= event_cap) {
            n_events++;
            return;
        }
        double *r = events + (size_t)n_events * 16;
        std::memset(r, 0, 128);
        r[0] = type;
        for (int i = 0; i < nv && i < 15; ++i) r[1 + i] = v[i];
        n_events++;
    }

    static void mul3(double *c, const double *a, const double *b) {  // MathTools::mmul: row times column, k ascending
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) {
                double s = 0.;
                for (int k = 0; k < 3; ++k) s += a[3 * i + k] * b[3 * k + j];
                c[3 * i + j] = s;
            }
    }
    static void inv_similarity(const double *T, double *Ti) {  // inverse of [s 0 a; 0 s b; 0 0 1] or of its transpose
        std::memset(Ti, 0, 72);
        const double s = T[0];
        Ti[0] = 1.0 / s, Ti[4] = 1.0 / s, Ti[8] = 1.0;
        if (T[2] != 0 || T[5] != 0) Ti[2] = -T[2] / s, Ti[5] = -T[5] / s;
        if (T[6] != 0 || T[7] != 0) Ti[6] = -T[6] / s, Ti[7] = -T[7] / s;
    }

    void init() {
        // estimateEssentialMatUsac: homogeneous 6-vectors; FTools::normalizePoints
        pd.resize((size_t)6 * n), pn.resize((size_t)6 * n);
        for (unsigned i = 0; i < n; ++i) {
            pd[6 * i] = p1[2 * i], pd[6 * i + 1] = p1[2 * i + 1], pd[6 * i + 2] = 1.0;
            pd[6 * i + 3] = p2[2 * i], pd[6 * i + 4] = p2[2 * i + 1], pd[6 * i + 5] = 1.0;
        }
        std::memset(T1, 0, 72), std::memset(T2, 0, 72);
        double m1[2] = {0, 0}, m2[2] = {0, 0};
        for (unsigned i = 0; i < n; ++i) m1[0] += pd[6 * i], m1[1] += pd[6 * i + 1], m2[0] += pd[6 * i + 3], m2[1] += pd[6 * i + 4];
        m1[0] /= (double)n, m2[0] /= (double)n, m1[1] /= (double)n, m2[1] /= (double)n;
        double d1 = 0, d2 = 0;
        for (unsigned i = 0; i < n; ++i) {
            d1 += sqrt((pd[6 * i] - m1[0]) * (pd[6 * i] - m1[0]) + (pd[6 * i + 1] - m1[1]) * (pd[6 * i + 1] - m1[1]));
            d2 += sqrt((pd[6 * i + 3] - m2[0]) * (pd[6 * i + 3] - m2[0]) + (pd[6 * i + 4] - m2[1]) * (pd[6 * i + 4] - m2[1]));
        }
        d1 /= (double)n, d2 /= (double)n;
        const double s1 = sqrt(2.0) / d1, s2 = sqrt(2.0) / d2;
        T1[0] = s1, T1[2] = -s1 * m1[0], T1[4] = s1, T1[5] = -s1 * m1[1], T1[8] = 1.0;
        T2[0] = s2, T2[2] = -s2 * m2[0], T2[4] = s2, T2[5] = -s2 * m2[1], T2[8] = 1.0;
        for (unsigned i = 0; i < n; ++i)
            for (int h = 0; h < 2; ++h) {
                const double *T = h ? T2 : T1, *v = &pd[6 * i + 3 * h];
                for (int k = 0; k < 3; ++k) {
                    double s = 0.;
                    for (int c = 0; c < 3; ++c) s += T[3 * k + c] * v[c];
                    pn[6 * i + 3 * h + k] = s;
                }
            }
        for (int i = 0; i < 3; ++i)
            for (int j = 0; j < 3; ++j) T2t[3 * i + j] = T2[3 * j + i];
        inv_similarity(T1, T1i), inv_similarity(T2t, T2ti);
        data_matrix.resize((size_t)9 * n);
        for (unsigned i = 0; i < n; ++i)
            for (int j = 0; j < 3; ++j)
                for (int k = 0; k < 3; ++k) data_matrix[(size_t)(3 * j + k) * n + i] = pn[6 * i + j + 3] * pn[6 * i + k];
        // initDataUSAC
        min_sample.assign(5, 0);
        if (prosac) init_prosac();
        last_wald_update = 0, history.clear();
        design_sprt();
        num_prev_best_lo = 0;
        errs[0].assign(n, 0.0), errs[1].assign(n, 0.0), cur = 0;
        pool_index = 0;
        pool.resize(n);
        for (unsigned i = 0; i < n; ++i) pool[i] = i;
        for (unsigned i = 1; i < n; ++i) {  // std::random_shuffle (libstdc++)
            const unsigned j = (unsigned)oracle_rand(&rng) % (i + 1);
            if (i != j) std::swap(pool[i], pool[j]);
        }
        flags.assign(n, 0), best_sample.assign(5, 0);
        std::memset(final_model, 0, sizeof(final_model));
    }

    void uniform_sample(unsigned data_size, unsigned sample_size, std::vector<unsigned> &sample) {
        unsigned count = 0;
        do {
            const unsigned index = (unsigned)oracle_rand(&rng) % data_size;
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
        double pn_i = 1.0;
        for (unsigned nn = 6; nn <= n; ++nn) {
            if (nn - 1 > 1000) {
                non_random[nn - 1] = non_random[nn - 2];
                continue;
            }
            std::vector<double> v(n, 0);
            v[5] = prosac_beta * std::pow((double)1 - prosac_beta, (double)nn - 5 - 1) * (nn - 5);
            pn_i = v[5];
            for (unsigned i = 7; i <= nn; ++i) {
                if (i == nn) {
                    v[nn - 1] = std::pow((double)prosac_beta, (double)nn - 5);
                    break;
                }
                v[i - 1] = pn_i * (prosac_beta / (1 - prosac_beta)) * ((double)(nn - i) / (i - 5 + 1));
                pn_i = v[i - 1];
            }
            double acc = 0.0;
            unsigned i_min = 0;
            for (unsigned i = nn; i >= 6; --i) {
                acc += v[i - 1];
                if (acc < 1 - prosac_non_rand_conf)
                    i_min = i;
                else
                    break;
            }
            non_random[nn - 1] = i_min;
        }
        maximality.assign(n, max_hyp);
        largest_size = 5, subset_size = 5, stop_len = n;
    }

    void prosac_sample(unsigned hyp, std::vector<unsigned> &sample) {
        if (hyp > prosac_max_samples) {
            uniform_sample(n, 5, sample);
            return;
        }
        if (subset_size > stop_len) uniform_sample(stop_len, 5, sample);  // no return in the reference: overwritten below
        if (hyp > growth[subset_size - 1]) {
            ++subset_size;
            if (subset_size > n) subset_size = n;
            if (largest_size < subset_size) largest_size = subset_size;
        }
        uniform_sample(subset_size - 1, 4, sample);
        sample[4] = subset_size - 1;
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
        return to_uint(ceil(log(1 - conf) / log(1 - p)));
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
        history.push_back(WaldTest{sprt_epsilon, sprt_delta, sprt_A, num_hyp - last_wald_update});
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
        for (size_t t = history.size(); t-- > 0;) {  // newest first
            const WaldTest &w = history[t];
            k += w.k;
            const double h = exp_sprt(new_eps, w.epsilon, w.delta);
            const double reject = 1.0 / (exp(h * log(w.A)));
            log_eta += (double)w.k * log(1.0 - p * (1.0 - reject));
        }
        const double ns = k + (log(1.0 - conf) - log_eta) / log(1.0 - p * (1.0 - (1.0 / sprt_A)));
        return to_uint(ceil(ns));
    }

    // ---- the problem class -------------------------------------------------------------------------------------------------------
    bool validate_sample() const {
        int i, j, k;
        for (i = 0; i < 5; i++) {
            for (j = 0; j < i; j++) {
                const double *a = &pn[min_sample[i] * 6], *b = &pn[min_sample[j] * 6];
                const double pix = a[0] / a[2], piy = a[1] / a[2], pjx = b[0] / b[2], pjy = b[1] / b[2];
                const double dx1 = pjx - pix, dy1 = pjy - piy;
                for (k = 0; k < j; k++) {
                    const double *c = &pn[min_sample[k] * 6];
                    const double dx2 = c[0] / c[2] - pix, dy2 = c[1] / c[2] - piy;
                    if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) break;
                }
                if (k < j) break;
            }
            if (j < i) break;
        }
        return i >= 4;
    }

    static double order_key(const double *E) {
        double big = 0, n2 = 0;
        for (int k = 0; k < 9; ++k) {
            if (std::fabs(E[k]) > std::fabs(big)) big = E[k];
            n2 += E[k] * E[k];
        }
        return (big < 0 ? -E[0] : E[0]) / sqrt(n2);
    }

    void set_model(int slot, const double *E) {  // models_denorm = E, models = T2^-T E T1^-1
        double t[9];
        std::memcpy(models_denorm[slot], E, 72);
        mul3(t, T2ti, E);
        mul3(models[slot], t, T1i);
    }

    unsigned minimal_models() {
        double q1[10], q2[10], Es[90];
        for (int i = 0; i < 5; ++i) {
            q1[2 * i] = p1[2 * min_sample[i]], q1[2 * i + 1] = p1[2 * min_sample[i] + 1];
            q2[2 * i] = p2[2 * min_sample[i]], q2[2 * i + 1] = p2[2 * min_sample[i] + 1];
        }
        const int ns = oracle_run5point(q1, q2, 5, Es);
        double v[8] = {(double)hyp_count, (double)min_sample[0], (double)min_sample[1], (double)min_sample[2], (double)min_sample[3],
                       (double)min_sample[4], (double)ns, 0};
        emit(1, v, 7);
        if (ns > 10) return 0;
        int order[10];
        double key[10];
        for (int i = 0; i < ns; ++i) order[i] = i, key[i] = order_key(Es + 9 * i);
        std::stable_sort(order, order + ns, [&](int a, int b) { return key[a] < key[b]; });
        for (int i = 0; i < ns; ++i) {
            set_model(i, Es + 9 * order[i]);
            double w[11];
            w[0] = hyp_count, w[1] = i;
            std::memcpy(w + 2, models_denorm[i], 72);
            emit(5, w, 11);
        }
        return (unsigned)ns;
    }

    bool validate_model(unsigned mi) {
        const double *F = models[mi];
        double e[3];
        auto cross = [](double *o, const double *a, const double *b) {
            o[0] = a[1] * b[2] - a[2] * b[1], o[1] = a[2] * b[0] - a[0] * b[2], o[2] = a[0] * b[1] - a[1] * b[0];
        };
        cross(e, F, F + 6);
        bool any = false;
        for (int i = 0; i < 3; ++i)
            if ((e[i] > 1.9984e-15) || (e[i] < -1.9984e-15)) any = true;
        if (!any) cross(e, F + 3, F + 6);
        auto ori = [&](const double *pt) { return (F[0] * pt[3] + F[3] * pt[4] + F[6] * pt[5]) * (e[1] * pt[2] - e[2] * pt[1]); };
        const double sig1 = ori(&pn[6 * min_sample[0]]);
        for (unsigned i = 1; i < 5; ++i)
            if (sig1 * ori(&pn[6 * min_sample[i]]) < 0) {
                double v[2] = {(double)hyp_count, (double)mi};
                emit(6, v, 2);
                return false;
            }
        return true;
    }

    static double sampson(const double *m, const double *pt) {  // PoseTools::getSampsonError / evaluateModel :1133-1139
        const double rxc = m[0] * pt[3] + m[3] * pt[4] + m[6];
        const double ryc = m[1] * pt[3] + m[4] * pt[4] + m[7];
        const double rwc = m[2] * pt[3] + m[5] * pt[4] + m[8];
        const double r = (pt[0] * rxc + pt[1] * ryc + rwc);
        const double rx = m[0] * pt[0] + m[1] * pt[1] + m[2];
        const double ry = m[3] * pt[0] + m[4] * pt[1] + m[5];
        return r * r / (rxc * rxc + ryc * ryc + rx * rx + ry * ry);
    }

    bool evaluate(unsigned mi, unsigned *num_inl, unsigned *tested) {
        const double *model = models_denorm[mi];
        double *err = errs[cur].data();
        bool good = true;
        double lj, lj1 = 1.0;
        *num_inl = 0, *tested = 0;
        const unsigned start = pool_index;
        for (unsigned i = 0; i < n; ++i) {
            if (pool_index > n - 1) pool_index = 0;
            const unsigned pt = pool[pool_index];
            ++pool_index;
            const double e = sampson(model, &pd[6 * pt]);
            err[pt] = e;
            if (e < thr) ++(*num_inl);
            if (e < thr)
                lj = lj1 * (sprt_delta / sprt_epsilon);
            else
                lj = lj1 * ((1 - sprt_delta) / (1 - sprt_epsilon));
            if (lj <= DBL_EPSILON) lj = DBL_EPSILON * 10;
            if (lj > sprt_A) {
                good = false;
                *tested = i + 1;
                break;
            }
            lj1 = lj;
        }
        if (good) *tested = n;
        double v[11] = {(double)hyp_count, (double)mi, (double)start, (double)*num_inl, (double)*tested, good ? 1.0 : 0.0,
                        sprt_delta, sprt_epsilon, sprt_A, thr, (double)num_lo};
        emit(2, v, 11);
        return good;
    }

    unsigned find_inliers(const std::vector<double> &err, double threshold, std::vector<unsigned> &out) const {
        unsigned c = 0;
        for (unsigned i = 0; i < n; ++i)
            if (err[i] < threshold) out[c++] = i;
        return c;
    }

    void store_solution(unsigned mi, unsigned num_inl) {
        best = num_inl;
        const std::vector<double> &err = errs[cur];
        for (unsigned i = 0; i < n; ++i) flags[i] = err[i] < thr ? 1 : 0;
        best_sample = min_sample;
        cur = 1 - cur;
        std::memcpy(final_model, models_denorm[mi], 72);
        double v[3] = {(double)hyp_count, (double)mi, (double)num_inl};
        emit(4, v, 3);
    }

    bool refined_model(const std::vector<unsigned> &sample, unsigned m, bool weighted, const double *weights) {
        if (m < 5) return false;
        bool ok = true;
        if (refine == 0) {  // REFINE_WEIGHTS: smallest singular vector of the (weighted) 9 x 9 covariance, rank-2 projection
            std::vector<double> A((size_t)m * 9);
            for (unsigned i = 0; i < m; ++i)
                for (unsigned j = 0; j < 9; ++j) {
                    const double s = data_matrix[(size_t)j * n + sample[i]];
                    A[(size_t)i * 9 + j] = weighted ? s * weights[i] : s;
                }
            double Cv[81];
            for (unsigned i = 0; i < 9; ++i)
                for (unsigned j = 0; j <= i; ++j) {
                    double val = 0;
                    for (unsigned k = 0; k < m; ++k) val += A[(size_t)k * 9 + i] * A[(size_t)k * 9 + j];
                    Cv[9 * i + j] = val, Cv[i + 9 * j] = val;
                }
            double w[9], V[81];
            oracle_jacobi_svd(Cv, 9, 9, w, V);
            double F[9];
            for (int i = 0; i < 9; ++i) F[i] = V[9 * i + 8];
            double w3[3], V3[9];
            oracle_jacobi_svd(F, 3, 3, w3, V3);
            double Fv[3];  // F v_min
            for (int r = 0; r < 3; ++r) Fv[r] = F[3 * r] * V3[2] + F[3 * r + 1] * V3[5] + F[3 * r + 2] * V3[8];
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) F[3 * r + c] -= Fv[r] * V3[3 * c + 2];
            std::memcpy(models[0], F, 72);
            double t[9];
            mul3(t, T2t, F);
            mul3(models_denorm[0], t, T1);
        } else {
            // REFINE_NISTER / _STEWENIUS (6, 4) and their _WEIGHTS forms (7, 5), EssentialMatEstimator.h:640-850: OpenGV's five-point
            // solver on ALL sample points -- unit bearing vectors of the denormalised points, row i = f2_i (x) f1_i
            // (methods.cpp:183-268) -- and, for a re-weighted step of the _WEIGHTS forms, row i scaled by w_i / |w|
            // (fivept_nister_weight / fivept_stewenius_weight, weightingEssential.cpp:62-148); then the solution with the smallest
            // Sampson-error sum over the inliers of the best model so far, with the reference's early exit.  One exact solver serves
            // Nister and Stewenius as it does for the minimal sample.
            const bool use_w = weighted && (refine == 5 || refine == 7);
            std::vector<double> rows((size_t)9 * m);
            double wnorm = 0;
            if (use_w) {
                for (unsigned i = 0; i < m; ++i) wnorm += std::pow(weights[i], 2);
                wnorm = std::sqrt(wnorm);
            }
            for (unsigned i = 0; i < m; ++i) {
                double f1[3], f2[3];
                bearing(&pd[6 * sample[i]], f1), bearing(&pd[6 * sample[i] + 3], f2);
                double *r = &rows[(size_t)9 * i];
                for (int a = 0; a < 3; ++a)
                    for (int b = 0; b < 3; ++b) r[3 * a + b] = f1[b] * f2[a];
                if (use_w) {
                    const double sc = weights[i] / wnorm;
                    for (int k = 0; k < 9; ++k) r[k] *= sc;
                }
            }
            double Es[90];
            const int ns = oracle_run5point_rows(rows.data(), (int)m, Es);
            int take = 0;
            if (ns > 1) {
                int order[10];
                double key[10];
                for (int i = 0; i < ns; ++i) order[i] = i, key[i] = order_key(Es + 9 * i);
                std::stable_sort(order, order + ns, [&](int a, int b) { return key[a] < key[b]; });
                std::vector<double> sums(ns, 0.0);
                for (unsigned i = 0; i < n; ++i) {
                    if (!flags[i]) continue;
                    for (int j = 0; j < ns; ++j) sums[j] += sampson(Es + 9 * order[j], &pd[6 * i]);
                    if ((i > 3) && (i % 4 == 0)) {
                        std::vector<double> t = sums;
                        std::partial_sort(t.begin(), t.begin() + 2, t.end());
                        if (t[0] < 0.66 * t[1]) break;
                    }
                }
                take = order[std::min_element(sums.begin(), sums.end()) - sums.begin()];
            } else if (ns != 1)
                ok = false;
            if (ok) set_model(0, Es + 9 * take);
        }
        double v[13];
        v[0] = hyp_count, v[1] = m, v[2] = weighted ? 1 : 0, v[3] = ok ? 1 : 0;
        for (int k = 0; k < 9; ++k) v[4 + k] = ok ? models_denorm[0][k] : 0.0;
        emit(3, v, 13);
        return ok;
    }

    static void bearing(const double *pt, double *f) {  // (x, y, 1) / |.| (EssentialMatEstimator.h:273-279)
        const double nrm = sqrt(pt[0] * pt[0] + (pt[1] * pt[1] + pt[2] * pt[2]));
        f[0] = pt[0] / nrm, f[1] = pt[1] / nrm, f[2] = pt[2] / nrm;
    }

    void find_weights(const std::vector<unsigned> &inl, unsigned cnt, double *weights) const {
        if (refine == 5 || refine == 7) {
            // findWeights, REFINE_STEWENIUS_WEIGHTS / REFINE_NISTER_WEIGHTS (:2404-2428): computePseudoHuberWeight
            // (weightingEssential.cpp:190-206) of the denormalised model on unit bearing vectors, threshold sqrt(thr) / 50;
            // costPseudoHuber is P/source/BA_driver.cpp:2639-2648
            const double *E = models_denorm[0];
            const double th = std::sqrt(thr) / 50.0, b_sq = th * th;
            for (unsigned i = 0; i < cnt; ++i) {
                double f[3], fp[3];
                bearing(&pd[6 * inl[i]], f), bearing(&pd[6 * inl[i] + 3], fp);
                double xpE[3], Ex1[2];
                for (int c = 0; c < 3; ++c) xpE[c] = fp[0] * E[c] + fp[1] * E[3 + c] + fp[2] * E[6 + c];
                const double num = xpE[0] * f[0] + xpE[1] * f[1] + xpE[2] * f[2];
                for (int r = 0; r < 2; ++r) Ex1[r] = E[3 * r] * f[0] + E[3 * r + 1] * f[1] + E[3 * r + 2] * f[2];
                const double denom1 = 1 / (std::sqrt(Ex1[0] * Ex1[0] + Ex1[1] * Ex1[1] + xpE[0] * xpE[0] + xpE[1] * xpE[1]) + 1e-8);
                const double d_abs = std::abs(num * denom1) + 1e-12;
                const double q = d_abs / th;
                weights[i] = denom1 * (std::sqrt(2 * b_sq * (std::sqrt(1 + q * q) - 1)) / d_abs);
            }
            return;
        }
        if (refine != 0) return;
        const double *m = models[0];
        for (unsigned i = 0; i < cnt; ++i) {
            const double *pt = &pn[6 * inl[i]];
            const double rxc = m[0] * pt[3] + m[3] * pt[4] + m[6], ryc = m[1] * pt[3] + m[4] * pt[4] + m[7];
            const double rx = m[0] * pt[0] + m[1] * pt[1] + m[2], ry = m[3] * pt[0] + m[4] * pt[1] + m[5];
            weights[i] = 1 / sqrt(rxc * rxc + ryc * ryc + rx * rx + ry * ry);
        }
    }

    unsigned local_optimization(unsigned best_inliers) {
        if (best_inliers < 2 * lo_sample) return 0;
        const unsigned ss = std::min(lo_sample, best_inliers / 2);
        std::vector<unsigned> sample(ss), orig(n), iter(n);
        unsigned lo_inliers = best_inliers, tmp = 0, tested;
        find_inliers(errs[1 - cur], thr, orig);
        ++num_lo;
        std::vector<double> weights(n);
        const double step = (lo_mult * thr - thr) / lo_steps;
        for (unsigned i = 0; i < lo_reps; ++i) {
            uniform_sample(best_inliers, ss, sample);
            for (unsigned j = 0; j < ss; ++j) sample[j] = orig[sample[j]];
            if (!refined_model(sample, ss, false, nullptr)) continue;
            if (!evaluate(0, &tmp, &tested)) continue;
            tmp = find_inliers(errs[cur], lo_mult * thr, iter);
            if (tmp < 5) continue;
            if (!refined_model(iter, tmp, false, nullptr)) continue;
            for (unsigned j = 0; j < lo_steps; ++j) {
                if (!evaluate(0, &tmp, &tested)) continue;
                find_inliers(errs[cur], (lo_mult * thr) - (j + 1) * step, iter);
                find_weights(iter, tmp, weights.data());
                if (!refined_model(iter, tmp, true, weights.data())) continue;
            }
            if (!evaluate(0, &tmp, &tested)) continue;
            if (tmp > lo_inliers) {
                lo_inliers = tmp;
                store_solution(0, lo_inliers);
            }
        }
        return lo_inliers;
    }


    // ---- degeneracy tests and model upgrade --------------------------------------------------------------------------------------
    enum { DG_NOT_FOUND = 0x1, DG_H = 0x2, DG_ROT_TRANS = 0x4, DG_NO_MOT = 0x8, DG_UPGRADE = 0x10 };
    int check_degeneracy = 0;  // 0 off, 1 after every new best model, 3 also after every local optimisation
    double dg_thr = 0;         // poseDegenTheshold = 1 - cos atan(th_pixels / focal length)
    unsigned dg_type = DG_NOT_FOUND, cnt_rot = 0, cnt_nomot = 0, cnt_trans = 0, max_up_rot = 8000, max_up_nomot = 8000;
    std::vector<unsigned> in_rot, out_rot, in_nomot, out_nomot;
    std::vector<int> sample_rot, sample_nomot;
    double R_degen[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    std::vector<double> v1, v2;  // bearing vectors: adapter view 1 = second image, view 2 = first image (:289-291)

    static bool near_zero(double d) { return (d < 1e-3) && (d > -1e-3); }
    static double dot3(const double *a, const double *b) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); }
    static void cross3(const double *a, const double *b, double *o) {
        o[0] = a[1] * b[2] - a[2] * b[1], o[1] = a[2] * b[0] - a[0] * b[2], o[2] = a[0] * b[1] - a[1] * b[0];
    }
    void init_degeneracy() {
        v1.resize((size_t)3 * n), v2.resize((size_t)3 * n);
        for (unsigned i = 0; i < n; ++i) {
            const double a[3] = {p2[2 * i], p2[2 * i + 1], 1.0}, b[3] = {p1[2 * i], p1[2 * i + 1], 1.0};
            const double na = sqrt(a[0] * a[0] + (a[1] * a[1] + a[2] * a[2])), nb = sqrt(b[0] * b[0] + (b[1] * b[1] + b[2] * b[2]));
            for (int k = 0; k < 3; ++k) v1[3 * i + k] = a[k] / na, v2[3 * i + k] = b[k] / nb;
        }
        in_rot.assign(n, 0), out_rot.assign(n, 0), in_nomot.assign(n, 0), out_nomot.assign(n, 0);
    }
    double rot_error(const double *R, unsigned i) const {
        const double *f2 = &v2[3 * i];
        double u[3];
        for (int r = 0; r < 3; ++r) u[r] = (R[3 * r] * f2[0] + R[3 * r + 1] * f2[1]) + R[3 * r + 2] * f2[2];
        return 1.0 - dot3(&v1[3 * i], u);
    }
    double nomot_error(unsigned i) const { return 1.0 - dot3(&v1[3 * i], &v2[3 * i]); }
    double trans_error(const double *t, unsigned i) const {  // evaluateModelTrans: midpoint triangulation under (I, t), both reprojections
        const double *f1 = &v1[3 * i], *f2 = &v2[3 * i];
        const double b0 = dot3(t, f1), b1 = dot3(t, f2);
        const double a00 = dot3(f1, f1), a10 = dot3(f1, f2), a01 = -a10, a11 = -dot3(f2, f2);
        const double invdet = 1.0 / (a00 * a11 - a10 * a01);
        const double l0 = (a11 * invdet) * b0 + (-a01 * invdet) * b1, l1 = (-a10 * invdet) * b0 + (a00 * invdet) * b1;
        double pt[3], q[3];
        for (int k = 0; k < 3; ++k) pt[k] = ((l0 * f1[k]) + (t[k] + l1 * f2[k])) / 2, q[k] = pt[k] + (-t[k]);
        const double n1 = sqrt(pt[0] * pt[0] + (pt[1] * pt[1] + pt[2] * pt[2])), n2 = sqrt(q[0] * q[0] + (q[1] * q[1] + q[2] * q[2]));
        for (int k = 0; k < 3; ++k) pt[k] = pt[k] / n1, q[k] = q[k] / n2;
        return (1.0 - dot3(f1, pt)) + (1.0 - dot3(f2, q));
    }
    static void arun(const double *H, double *R) {  // math/arun.cpp: V U^T, third column of V negated when the determinant is -1
        double sv[3], U[9], V[9];
        oracle_eigen_svd3(H, sv, U, V);
        for (int pass = 0; pass < 2; ++pass) {
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) R[3 * r + c] = V[3 * r] * U[3 * c] + V[3 * r + 1] * U[3 * c + 1] + V[3 * r + 2] * U[3 * c + 2];
            const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
            if (det >= 0) break;
            for (int r = 0; r < 3; ++r) V[3 * r + 2] = -V[3 * r + 2];
        }
    }
    void cross_cov(const std::vector<int> &idx, const double *c1, const double *c2, double *H) const {
        std::memset(H, 0, 72);
        for (int i : idx)
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) H[3 * r + c] += (v2[3 * i + r] - c2[r]) * (v1[3 * i + c] - c1[c]);
    }
    void twopt_rotation(int i0, int i1, double *R) const {
        double c1[3], c2[3], H[9];
        for (int k = 0; k < 3; ++k) c1[k] = (v1[3 * i0 + k] + v1[3 * i1 + k]) / 3.0, c2[k] = (v2[3 * i0 + k] + v2[3 * i1 + k]) / 3.0;
        cross_cov({i0, i1}, c1, c2, H);
        arun(H, R);
    }
    void rotation_only(const std::vector<int> &idx, double *R) const {
        double c1[3] = {0, 0, 0}, c2[3] = {0, 0, 0}, H[9];
        for (int i : idx)
            for (int k = 0; k < 3; ++k) c1[k] += v1[3 * i + k], c2[k] += v2[3 * i + k];
        for (int k = 0; k < 3; ++k) c1[k] = c1[k] / (double)idx.size(), c2[k] = c2[k] / (double)idx.size();
        cross_cov(idx, c1, c2, H);
        arun(H, R);
    }
    void twopt_translation(int i0, int i1, double *t) const {
        double n1[3], n2[3];
        cross3(&v1[3 * i0], &v2[3 * i0], n1), cross3(&v1[3 * i1], &v2[3 * i1], n2);
        cross3(n1, n2, t);
        const double nrm = sqrt(t[0] * t[0] + (t[1] * t[1] + t[2] * t[2]));
        double flow[3];
        for (int k = 0; k < 3; ++k) t[k] = t[k] / nrm, flow[k] = v1[3 * i0 + k] - v2[3 * i0 + k];
        if (dot3(flow, t) < 0)
            for (int k = 0; k < 3; ++k) t[k] = -t[k];
    }
    // the eigensolver's objective from its definition: M(c) = sum (f1 x R f2)(f1 x R f2)^T, R = (1 + |c|^2) x the Cayley rotation
    static void cayley_unscaled(const double *c, double *R) {
        R[0] = 1 + c[0] * c[0] - c[1] * c[1] - c[2] * c[2], R[1] = 2 * (c[0] * c[1] - c[2]), R[2] = 2 * (c[0] * c[2] + c[1]);
        R[3] = 2 * (c[0] * c[1] + c[2]), R[4] = 1 - c[0] * c[0] + c[1] * c[1] - c[2] * c[2], R[5] = 2 * (c[1] * c[2] - c[0]);
        R[6] = 2 * (c[0] * c[2] - c[1]), R[7] = 2 * (c[1] * c[2] + c[0]), R[8] = 1 - c[0] * c[0] - c[1] * c[1] + c[2] * c[2];
    }
    void compose_M(const int *idx, const double *c, double *M) const {
        double R[9];
        cayley_unscaled(c, R);
        std::memset(M, 0, 72);
        for (int k = 0; k < 5; ++k) {
            const double *f1 = &v1[3 * idx[k]], *f2 = &v2[3 * idx[k]];
            double u[3], nv[3];
            for (int r = 0; r < 3; ++r) u[r] = R[3 * r] * f2[0] + R[3 * r + 1] * f2[1] + R[3 * r + 2] * f2[2];
            cross3(f1, u, nv);
            for (int r = 0; r < 3; ++r)
                for (int q = 0; q < 3; ++q) M[3 * r + q] += nv[r] * nv[q];
        }
    }
    static void sym_eig3(const double *M, double *w, double *V) {  // ascending eigenvalues, V columns (via the singular pairs of a PSD matrix)
        double s[3], Vv[9];
        oracle_jacobi_svd(M, 3, 3, s, Vv);
        for (int k = 0; k < 3; ++k) {
            w[k] = s[2 - k];
            for (int r = 0; r < 3; ++r) V[3 * r + k] = Vv[3 * r + (2 - k)];
        }
    }
    double lambda_min(const int *idx, const double *c) const {
        double M[9], w[3], V[9];
        compose_M(idx, c, M);
        sym_eig3(M, w, V);
        return w[0];
    }
    void eigensolver(const int *idx, const double *R0, double *R, double *t) const {
        // start: Cayley parameters of R0, C = (R0 - I)(R0 + I)^-1
        double A[9], B[9], Bi[9], c[3];
        for (int k = 0; k < 9; ++k) A[k] = R0[k] - (k % 4 == 0), B[k] = R0[k] + (k % 4 == 0);
        const double det = B[0] * (B[4] * B[8] - B[5] * B[7]) - B[1] * (B[3] * B[8] - B[5] * B[6]) + B[2] * (B[3] * B[7] - B[4] * B[6]);
        Bi[0] = (B[4] * B[8] - B[5] * B[7]) / det, Bi[1] = (B[2] * B[7] - B[1] * B[8]) / det, Bi[2] = (B[1] * B[5] - B[2] * B[4]) / det;
        Bi[3] = (B[5] * B[6] - B[3] * B[8]) / det, Bi[4] = (B[0] * B[8] - B[2] * B[6]) / det, Bi[5] = (B[2] * B[3] - B[0] * B[5]) / det;
        Bi[6] = (B[3] * B[7] - B[4] * B[6]) / det, Bi[7] = (B[1] * B[6] - B[0] * B[7]) / det, Bi[8] = (B[0] * B[4] - B[1] * B[3]) / det;
        auto Cm = [&](int r, int k) { return A[3 * r] * Bi[k] + A[3 * r + 1] * Bi[3 + k] + A[3 * r + 2] * Bi[6 + k]; };
        c[0] = -Cm(1, 2), c[1] = Cm(0, 2), c[2] = -Cm(0, 1);
        // damped Newton on lambda_min with central differences (gradient h = 1e-6, Hessian from gradient differences)
        auto grad = [&](const double *x, double *g) {
            for (int k = 0; k < 3; ++k) {
                double a[3] = {x[0], x[1], x[2]}, b[3] = {x[0], x[1], x[2]};
                a[k] += 1e-6, b[k] -= 1e-6;
                g[k] = (lambda_min(idx, a) - lambda_min(idx, b)) / 2e-6;
            }
        };
        double f = lambda_min(idx, c), mu = 1e-9;
        for (int it = 0; it < 40; ++it) {
            double g[3], Hs[9];
            grad(c, g);
            for (int k = 0; k < 3; ++k) {
                double a[3] = {c[0], c[1], c[2]}, b[3] = {c[0], c[1], c[2]}, ga[3], gb[3];
                a[k] += 1e-4, b[k] -= 1e-4;
                grad(a, ga), grad(b, gb);
                for (int r = 0; r < 3; ++r) Hs[3 * r + k] = (ga[r] - gb[r]) / 2e-4;
            }
            bool moved = false;
            for (int tries = 0; tries < 12 && !moved; ++tries) {
                double Hd[9], step[3];
                for (int k = 0; k < 9; ++k) Hd[k] = 0.5 * (Hs[k] + Hs[3 * (k % 3) + k / 3]) + (k % 4 == 0 ? mu : 0.0);
                const double d = Hd[0] * (Hd[4] * Hd[8] - Hd[5] * Hd[7]) - Hd[1] * (Hd[3] * Hd[8] - Hd[5] * Hd[6]) + Hd[2] * (Hd[3] * Hd[7] - Hd[4] * Hd[6]);
                if (d != 0 && std::isfinite(d)) {
                    step[0] = ((Hd[4] * Hd[8] - Hd[5] * Hd[7]) * g[0] + (Hd[2] * Hd[7] - Hd[1] * Hd[8]) * g[1] + (Hd[1] * Hd[5] - Hd[2] * Hd[4]) * g[2]) / d;
                    step[1] = ((Hd[5] * Hd[6] - Hd[3] * Hd[8]) * g[0] + (Hd[0] * Hd[8] - Hd[2] * Hd[6]) * g[1] + (Hd[2] * Hd[3] - Hd[0] * Hd[5]) * g[2]) / d;
                    step[2] = ((Hd[3] * Hd[7] - Hd[4] * Hd[6]) * g[0] + (Hd[1] * Hd[6] - Hd[0] * Hd[7]) * g[1] + (Hd[0] * Hd[4] - Hd[1] * Hd[3]) * g[2]) / d;
                    const double x[3] = {c[0] - step[0], c[1] - step[1], c[2] - step[2]};
                    const double fx = lambda_min(idx, x);
                    if (fx <= f && std::isfinite(fx)) {
                        moved = fabs(step[0]) + fabs(step[1]) + fabs(step[2]) > 1e-13;
                        f = fx, c[0] = x[0], c[1] = x[1], c[2] = x[2];
                        mu = std::max(mu / 10, 1e-12);
                        if (!moved) break;
                        continue;
                    }
                }
                mu *= 10;
            }
            if (!moved) break;
        }
        double Rr[9], M[9], w[3], V[9];
        cayley_unscaled(c, Rr);
        const double scale = 1 + c[0] * c[0] + c[1] * c[1] + c[2] * c[2];
        for (int k = 0; k < 9; ++k) R[k] = Rr[k] / scale;
        compose_M(idx, c, M);
        sym_eig3(M, w, V);
        double d[3];
        eigen_order3(M, d);  // OpenGV: column 0 of Eigen::EigenSolver, length from positions 1 and 2
        int k0 = 0;
        for (int k = 1; k < 3; ++k)
            if (fabs(w[k] - d[0]) < fabs(w[k0] - d[0])) k0 = k;
        const double mag = sqrt(d[1] * d[1] + d[2] * d[2]);
        for (int k = 0; k < 3; ++k) t[k] = mag * V[3 * k + k0];
        const double *f1 = &v1[3 * idx[0]], *f2 = &v2[3 * idx[0]];
        double flow[3];
        for (int r = 0; r < 3; ++r) flow[r] = f1[r] - (R[3 * r] * f2[0] + R[3 * r + 1] * f2[1] + R[3 * r + 2] * f2[2]);
        if (flow[0] * t[0] + flow[1] * t[1] + flow[2] * t[2] < 0)
            for (int k = 0; k < 3; ++k) t[k] = -t[k];
    }
    static void e_from_rt(const double *R, const double *t, double *E) {  // poselib::getEfromRT
        const double s = 1.0 / sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
        const double a = t[0] * s, b = t[1] * s, c = t[2] * s;
        const double Sk[9] = {0, -c, b, c, 0, -a, -b, a, 0};
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 3; ++k) {
                double acc = 0;
                for (int m = 0; m < 3; ++m) acc += Sk[3 * r + m] * R[3 * m + k];
                E[3 * r + k] = acc;
            }
    }

    void test_rotation(bool *degenerate) {
        static const unsigned pair_of[20] = {0, 1, 0, 2, 0, 3, 0, 4, 1, 2, 1, 3, 1, 4, 2, 3, 2, 4, 3, 4};
        static const unsigned rest_of[30] = {2, 3, 4, 1, 3, 4, 1, 2, 4, 1, 2, 3, 0, 3, 4, 0, 2, 4, 0, 2, 3, 0, 1, 4, 0, 1, 3, 0, 1, 2};
        std::vector<int> sample(5, 0), inl;
        std::vector<double> e(n);
        double R[9];
        for (unsigned i = 0; i < 10; ++i) {
            for (unsigned j = 0; j < 2; ++j) sample[j] = (int)min_sample[pair_of[2 * i + j]];
            twopt_rotation(sample[0], sample[1], R);
            unsigned count1 = 2, num = 0;
            for (unsigned j = 0; j < 3; ++j) {
                const unsigned idx = min_sample[rest_of[3 * i + j]];
                if (rot_error(R, idx) < dg_thr) sample[count1++] = (int)idx, ++num;
            }
            if (num == 0) continue;
            num = 0;
            for (unsigned j = 0; j < n; ++j) {
                e[j] = rot_error(R, pool[j]);
                if (e[j] < dg_thr) ++num;
            }
            const unsigned first_count = num;
            if (num < 2) continue;
            inl.clear();
            for (unsigned j = 0; j < n; ++j)
                if (e[j] < dg_thr) {
                    inl.push_back((int)pool[j]);
                    if (count1 < 5) sample[count1++] = (int)pool[j];
                }
            rotation_only(inl, R);
            num = 0;
            for (unsigned j = 0; j < n; ++j) {
                e[j] = rot_error(R, pool[j]);
                if (e[j] < dg_thr) ++num;
            }
            double v[5] = {(double)hyp_count, (double)i, (double)first_count, (double)num, 0};
            if (num < best / 5) {
                emit(8, v, 5);
                continue;
            }
            *degenerate = true;
            if (dg_type != (dg_type & (DG_UPGRADE | DG_ROT_TRANS))) dg_type = DG_ROT_TRANS;
            if (num > cnt_rot) {
                dg_type |= DG_UPGRADE;
                inl.clear();
                for (unsigned j = 0; j < n; ++j)
                    if (e[j] < dg_thr) inl.push_back((int)pool[j]);
                rotation_only(inl, R_degen);
                cnt_rot = num;
                for (unsigned j = 0; j < n; ++j) in_rot[pool[j]] = e[j] < dg_thr ? 1 : 0, out_rot[pool[j]] = e[j] < dg_thr ? 0 : 1;
                sample_rot = sample;
                v[4] = 1;
            }
            emit(8, v, 5);
        }
    }
    void test_no_motion(bool *degenerate) {
        if (cnt_nomot > 0) return;
        sample_nomot.clear();
        unsigned num = 0;
        for (unsigned j = 0; j < 5; ++j)
            if (nomot_error(min_sample[j]) < dg_thr) sample_nomot.push_back((int)min_sample[j]), ++num;
        if (num == 0) return;
        std::vector<double> e(n);
        num = 0;
        for (unsigned j = 0; j < n; ++j) {
            e[j] = nomot_error(pool[j]);
            if (e[j] < dg_thr) ++num;
        }
        if (num < best / 5) return;
        *degenerate = true;
        const bool dominant = (double)num > 0.7 * (double)cnt_rot;
        if (dominant)
            if (dg_type != (dg_type & (DG_UPGRADE | DG_NO_MOT))) dg_type = DG_NO_MOT;
        if (num > cnt_nomot) {
            if (dominant) dg_type |= DG_UPGRADE;
            cnt_nomot = num;
            for (unsigned j = 0; j < n; ++j) {
                const bool in = e[j] < dg_thr;
                in_nomot[pool[j]] = in ? 1 : 0, out_nomot[pool[j]] = in ? 0 : 1;
                if (in && sample_nomot.size() < 5) sample_nomot.push_back((int)pool[j]);
            }
        }
    }
    void test_degeneracy(bool *degenerate, bool *upgrade) {
        *degenerate = false, *upgrade = false;
        dg_type = DG_H;
        test_rotation(degenerate);
        if (dg_type & DG_UPGRADE) *upgrade = true;
        if (dg_type == (unsigned)(DG_ROT_TRANS | DG_UPGRADE)) test_no_motion(degenerate);
        double v[7] = {(double)hyp_count, *degenerate ? 1.0 : 0.0, *upgrade ? 1.0 : 0.0, (double)dg_type, (double)cnt_rot, (double)cnt_nomot, (double)best};
        emit(7, v, 7);
    }
    bool evaluate_translation(const double *t, unsigned *num_inl, unsigned *tested) {
        double *err = errs[cur].data();
        bool good = true;
        double lj, lj1 = 1.0;
        *num_inl = 0, *tested = 0;
        const unsigned start = pool_index;
        for (unsigned i = 0; i < n; ++i) {
            if (pool_index > n - 1) pool_index = 0;
            const unsigned pt = pool[pool_index];
            ++pool_index;
            const double e = trans_error(t, pt);
            err[pt] = e;
            if (e < dg_thr) ++(*num_inl);
            lj = e < dg_thr ? lj1 * (sprt_delta / sprt_epsilon) : lj1 * ((1 - sprt_delta) / (1 - sprt_epsilon));
            if (lj <= DBL_EPSILON) lj = DBL_EPSILON * 10;
            if (lj > sprt_A) {
                good = false;
                *tested = i + 1;
                break;
            }
            lj1 = lj;
        }
        if (good) *tested = n;
        double v[11] = {(double)hyp_count, -1.0, (double)start, (double)*num_inl, (double)*tested, good ? 1.0 : 0.0,
                        sprt_delta, sprt_epsilon, sprt_A, dg_thr, (double)num_lo};
        emit(2, v, 11);
        return good;
    }
    unsigned stopping_on(const std::vector<double> &err, const std::vector<unsigned> &idx, double limit, unsigned num_outliers) const {
        unsigned c = 0, untouched = 0;
        for (unsigned j : idx) {
            if (err[j] < limit)
                ++c;
            else if (std::round(err[j] - DBL_MAX) == 0)
                ++untouched;
        }
        return standard_stopping(c, num_outliers - untouched, 1);
    }
    unsigned upgrade_model() {
        unsigned best_up = best, best_up_rot = cnt_rot, best_up_trans = cnt_trans;
        if (n < 2) return 0;
        unsigned tried = 0, branch = 0;
        if (dg_type & DG_UPGRADE) {
            const bool nomot = (dg_type & DG_NO_MOT) != 0;
            branch = nomot ? 1 : 2;
            const unsigned num_outliers = n - (nomot ? cnt_nomot : cnt_rot);
            if (num_outliers < (nomot ? 1u : 3u)) return 0;
            std::vector<unsigned> outlier_indices(num_outliers, 0), smp(nomot ? 1 : 3);
            unsigned c = 0;
            for (unsigned i = 0; i < n; ++i)
                if ((nomot ? out_nomot : out_rot)[i]) outlier_indices[c++] = i;
            std::fill(errs[cur].begin(), errs[cur].end(), DBL_MAX);
            const int counted = cur;  // current_err_array: a pointer, it does not follow the swaps of storeSolution
            unsigned &limit = nomot ? max_up_nomot : max_up_rot;
            const unsigned size_nomot = (unsigned)sample_nomot.size();
            for (unsigned i = 0; i < limit; ++i) {
                ++tried;
                uniform_sample(num_outliers, nomot ? 1 : 3, smp);
                int index[5];
                double E[9], t[3], R[9];
                unsigned num = 0, tested = 0;
                if (nomot) {
                    index[0] = (int)outlier_indices[smp[0]];
                    index[1] = sample_nomot[(unsigned)oracle_rand(&rng) % size_nomot];
                    twopt_translation(index[0], index[1], t);
                    double v[12] = {(double)hyp_count, 1.0, (double)i, t[0], t[1], t[2]};
                    emit(10, v, 12);
                    if (near_zero(sqrt(t[0] * t[0] + (t[1] * t[1] + t[2] * t[2])) * 100)) continue;
                    evaluate_translation(t, &num, &tested);
                } else {
                    for (int j = 0; j < 3; ++j) index[j] = (int)outlier_indices[smp[j]];
                    index[3] = sample_rot[0], index[4] = sample_rot[1];
                    eigensolver(index, R_degen, R, t);
                    const double len = sqrt(t[0] * t[0] + (t[1] * t[1] + t[2] * t[2]));
                    double v[12] = {(double)hyp_count, 2.0, (double)i};
                    if (near_zero(len * 100)) {
                        emit(10, v, 12);
                        continue;
                    }
                    for (int k = 0; k < 3; ++k) t[k] = t[k] / len;
                    e_from_rt(R, t, E);
                    std::memcpy(v + 3, E, 72);
                    emit(10, v, 12);
                    set_model(0, E);
                    evaluate(0, &num, &tested);
                }
                if (num > (nomot ? best_up_trans : best_up_rot)) {
                    if (!nomot)
                        for (int j = 2, q = 0; j < 5; ++j) sample_rot[j] = index[q++];
                    if (num > best_up || (near_zero(final_model[0] * 100) && near_zero(final_model[4] * 100) && near_zero(final_model[8] * 100))) {
                        if (nomot) {
                            const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
                            e_from_rt(I3, t, E);
                            set_model(0, E);
                        }
                        store_solution(0, num);
                        best_up = num;
                        if (nomot) {
                            if (size_nomot > 3) {
                                unsigned k = 0;
                                for (size_t j = 0; j < 3; j++) {
                                    if (sample_nomot[k] == index[1]) {
                                        j--;
                                        k++;
                                        continue;
                                    }
                                    min_sample[j] = (unsigned)sample_nomot[k];
                                    k++;
                                }
                                min_sample[3] = (unsigned)index[1], min_sample[4] = (unsigned)index[0];
                            }
                            cnt_trans = num;
                        } else {
                            for (size_t j = 0; j < 5; j++) min_sample[j] = (unsigned)sample_rot[j];
                        }
                    }
                    (nomot ? best_up_trans : best_up_rot) = num;
                    const unsigned ns = stopping_on(errs[counted], outlier_indices, nomot ? dg_thr : thr, num_outliers);
                    if (ns < limit) limit = ns;
                }
            }
        }
        double v[4] = {(double)hyp_count, (double)branch, (double)tried, (double)best_up};
        emit(9, v, 4);
        return best_up;
    }

    bool solve() {
        unsigned adaptive = max_hyp;
        bool update_sprt_stopping = true;
        if (n < 5 || (prosac && n < prosac_min_stop)) return false;
        const unsigned max2 = max_hyp / 2, max3 = 2 * max_hyp / 3;
        while (hyp_count < adaptive && hyp_count < max_hyp) {
            ++hyp_count;
            if ((hyp_count == max2) && (best == 0))
                thr *= 1.33;
            else if ((hyp_count == max3) && (best == 0))
                thr *= 1.13;
            if (prosac)
                prosac_sample(hyp_count, min_sample);
            else
                uniform_sample(n, 5, min_sample);
            if (!validate_sample()) {
                double v[7] = {(double)hyp_count, (double)min_sample[0], (double)min_sample[1], (double)min_sample[2], (double)min_sample[3],
                               (double)min_sample[4], -1.0};
                emit(1, v, 7);
                ++rejected_samples;
                continue;
            }
            const unsigned ns = minimal_models();
            model_count += ns;
            bool update_best = false;
            for (unsigned i = 0; i < ns; ++i) {
                if (!validate_model(i)) {
                    ++rejected_models;
                    continue;
                }
                unsigned inl, tested;
                const bool good = evaluate(i, &inl, &tested);
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
                        store_solution(i, best);
                    }
                }
            }
            if (update_best && check_degeneracy) {  // USAC.h:509-528
                bool degenerate = false, upgrade = false;
                test_degeneracy(&degenerate, &upgrade);
                if (degenerate && upgrade) {
                    const unsigned up = upgrade_model();
                    if (up > best) best = up;
                }
            }
            if (update_best) {
                unsigned lo = local_optimization(best);
                if (check_degeneracy & 2) {  // USAC.h:540-556
                    bool degenerate = false, upgrade = false;
                    test_degeneracy(&degenerate, &upgrade);
                    if (degenerate && upgrade) {
                        const unsigned up = upgrade_model();
                        if (up > lo) lo = up;
                    }
                }
                if (lo > best) best = lo;
                if (num_prev_best_lo < best) num_prev_best_lo = best;
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
        return true;
    }
};

}  // namespace

extern "C" int oracle_usac_essential(const double *p1, const double *p2, int n, double th, unsigned seed, int refine,
                                     const uint32_t *sorted_idx, int max_hyp, double conf, double prosac_beta, double sprt_delta,
                                     double sprt_epsilon, double sprt_mS, double sprt_tM, double *E, uint8_t *inlier_flags, double *results,
                                     double *events, int event_cap, int *n_events) {
    Usac u;
    u.n = (unsigned)n, u.max_hyp = (unsigned)max_hyp, u.conf = conf, u.thr = th * th, u.refine = refine;
    u.p1 = p1, u.p2 = p2;
    u.prosac = sorted_idx != nullptr;
    if (u.prosac) u.sorted_idx.assign(sorted_idx, sorted_idx + n);
    u.prosac_beta = prosac_beta, u.sprt_delta = sprt_delta, u.sprt_epsilon = sprt_epsilon, u.sprt_mS = sprt_mS, u.sprt_tM = sprt_tM;
    u.events = events, u.event_cap = event_cap;
    oracle_srand(&u.rng, seed);
    u.init();
    const bool ok = u.solve();
    if (n_events) *n_events = u.n_events;
    if (results) {
        const double fin[12] = {ok ? 1.0 : 0.0,
                                (double)u.hyp_count,
                                (double)u.model_count,
                                (double)u.rejected_samples,
                                (double)u.rejected_models,
                                (double)u.best,
                                (double)u.points_verified,
                                (double)u.num_lo,
                                u.history.empty() ? 0.0 : u.history.back().delta,
                                u.history.empty() ? 0.0 : u.history.back().epsilon,
                                u.sprt_delta,
                                u.sprt_epsilon};
        std::memcpy(results, fin, sizeof(fin));
    }
    if (E) std::memcpy(E, u.final_model, 72);
    if (inlier_flags)
        for (int i = 0; i < n; ++i) inlier_flags[i] = (uint8_t)u.flags[i];
    return ok ? 1 : 0;
}

// With the degeneracy handling of DEGEN_USAC_INTERNAL: check_degeneracy 1 (after every new best model) or 3 (also after every local
// optimisation); degen[16] = {1, inliers of the rotation, of "no motion", type, R_degenerate[9]}; flags_rot / flags_nomot: n bytes each.
extern "C" int oracle_usac_essential_degen(const double *p1, const double *p2, int n, double th, unsigned seed, int refine,
                                           const uint32_t *sorted_idx, int max_hyp, double conf, double prosac_beta, double sprt_delta,
                                           double sprt_epsilon, double sprt_mS, double sprt_tM, int check_degeneracy, double th_pixels,
                                           double focal_length, double *E, uint8_t *inlier_flags, double *results, double *events,
                                           int event_cap, int *n_events, double *degen, uint8_t *flags_rot, uint8_t *flags_nomot) {
    Usac u;
    u.n = (unsigned)n, u.max_hyp = (unsigned)max_hyp, u.conf = conf, u.thr = th * th, u.refine = refine;
    u.p1 = p1, u.p2 = p2;
    u.prosac = sorted_idx != nullptr;
    if (u.prosac) u.sorted_idx.assign(sorted_idx, sorted_idx + n);
    u.prosac_beta = prosac_beta, u.sprt_delta = sprt_delta, u.sprt_epsilon = sprt_epsilon, u.sprt_mS = sprt_mS, u.sprt_tM = sprt_tM;
    u.events = events, u.event_cap = event_cap;
    u.check_degeneracy = check_degeneracy;
    u.dg_thr = 1.0 - cos(atan(th_pixels / focal_length));
    oracle_srand(&u.rng, seed);
    u.init();
    if (check_degeneracy) u.init_degeneracy();
    const bool ok = u.solve();
    if (n_events) *n_events = u.n_events;
    if (results) {
        const double fin[12] = {ok ? 1.0 : 0.0, (double)u.hyp_count, (double)u.model_count, (double)u.rejected_samples, (double)u.rejected_models,
                                (double)u.best, (double)u.points_verified, (double)u.num_lo, u.history.empty() ? 0.0 : u.history.back().delta,
                                u.history.empty() ? 0.0 : u.history.back().epsilon, u.sprt_delta, u.sprt_epsilon};
        std::memcpy(results, fin, sizeof(fin));
    }
    if (E) std::memcpy(E, u.final_model, 72);
    if (inlier_flags)
        for (int i = 0; i < n; ++i) inlier_flags[i] = (uint8_t)u.flags[i];
    if (degen) {
        std::memset(degen, 0, 128);
        degen[0] = check_degeneracy ? 1.0 : 0.0, degen[1] = u.cnt_rot, degen[2] = u.cnt_nomot, degen[3] = u.dg_type;
        std::memcpy(degen + 4, u.R_degen, 72);
    }
    if (check_degeneracy && ok)
        for (int i = 0; i < n; ++i) {
            if (flags_rot) flags_rot[i] = (uint8_t)u.in_rot[i];
            if (flags_nomot) flags_nomot[i] = (uint8_t)u.in_nomot[i];
        }
    return ok ? 1 : 0;
}

// Eigen::EigenSolver<Matrix3d>'s eigenvalue order on a 3 x 3 matrix (row-major), for tests (tests/golden/eigen_order3.npz)
extern "C" void oracle_eigen_order3(const double *M, double *d) { eigen_order3(M, d); }
