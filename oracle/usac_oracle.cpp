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
        } else {  // REFINE_NISTER: the solver on all sample points, the solution with the smallest error sum over the current inliers
            std::vector<double> q1((size_t)2 * m), q2((size_t)2 * m);
            for (unsigned i = 0; i < m; ++i) {
                q1[2 * i] = p1[2 * sample[i]], q1[2 * i + 1] = p1[2 * sample[i] + 1];
                q2[2 * i] = p2[2 * sample[i]], q2[2 * i + 1] = p2[2 * sample[i] + 1];
            }
            double Es[90];
            const int ns = oracle_run5point(q1.data(), q2.data(), (int)m, Es);
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

    void find_weights(const std::vector<unsigned> &inl, unsigned cnt, double *weights) const {
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
            if (update_best) {
                const unsigned lo = local_optimization(best);
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
