/*
 * arrsac_oracle.cpp -- CPU restatement of the reference's ARRSAC path of estimateEssentialMat.
 *
 * THIS IS TEST INFRASTRUCTURE, NOT PRODUCT CODE (see oracle.h).
 *
 * A literal, sequential restatement (one hypothesis at a time, as the reference runs it) of
 *   P/source/five-point-nister/modelest.cpp:111-341      EssentialMatEstimatorTheia + CvModelEstimator3::runARRSAC
 *   P/include/arrsac/arrsac.h:236-547                     theia::Arrsac::GenerateInitialHypothesisSet / Estimate
 *   P/include/arrsac/prosac_sampler.h:85-157              ProsacSampler::Sample
 *   P/include/arrsac/random_sampler.h:57-78               RandomSampler::Sample
 *   P/include/arrsac/sequential_probability_ratio.h:88-126, P/source/arrsac/sequential_probability_ratio.cc:38-62
 *   P/source/five-point-nister/five-point.cpp:534-601     CvEMEstimator::ValidModel
 *   P/source/pose_estim.cpp:337-792                       robustEssentialRefine (model 0, no normalisation, no mask)
 *   P/source/pose_helper.cpp:115-182                      SampsonL1, getClosestE;  P/source/BA_driver.cpp:2639-2648 costPseudoHuber
 *
 * Third-party arithmetic that is not under /root/reference and is restated from its published algorithm ("parity unpinned"
 * at that boundary, as for cv::SVD in oracle.h):
 *   cv::RNG (OpenCV 4.2.0 core: multiply-with-carry, state*4164903690 + carry; uniform(a,b) = a + next() % (b-a))
 *   cv::findFundamentalMat(FM_8POINT) (OpenCV 4.2.0 calib3d fundam.cpp run8Point, float32 input points)
 *   Eigen::JacobiSVD<Matrix3d> (two-sided Jacobi; the SIGN of V's columns matters to ValidModel): restated from Eigen's
 *     algorithm and pinned against the Eigen 3.2.0 the reference vendors (oracle/_ref/eigen_svd3, tests marker `ref`); the
 *     reference builds against Eigen 3.3.7, whose 2x2 kernel differs in the LEFT rotation only, V is the same.
 *   Eigen::EigenSolver<9x9> on a symmetric matrix (robustEssentialRefine): replaced by the symmetric Jacobi SVD; the reference
 *     rejects a refinement when an eigenvalue below DBL_EPSILON sits at a position < 8 of EigenSolver's unspecified order; here
 *     that is "the second smallest eigenvalue is below DBL_EPSILON".
 *   the sign of the 5-point solutions (an artefact of cv::SVD's null-space basis, see canonical_sign below)
 *   std::sort: the C++ library's own (the preemption step sorts tied integer scores with it, so the permutation is the
 *     library's; the reference built on this machine gets the same one).
 * The float -> int conversions of the hypothesis-count formulas overflow in the reference (undefined behaviour; x86 cvttsd2si
 * yields INT_MIN): restated explicitly as INT_MIN.
 */
#include <algorithm>
#include <cfloat>
#include <climits>
#include <cmath>
#include <cstdint>
#include <cstring>
#include <vector>

#include "oracle.h"

namespace {

// ---- cv::RNG ----------------------------------------------------------------------------------------------------
struct CvRng {
    uint64_t state;
    unsigned next() {
        state = (uint64_t)(unsigned)state * 4164903690U + (unsigned)(state >> 32);
        return (unsigned)state;
    }
    int uniform(int a, int b) { return a == b ? a : (int)(next() % (unsigned)(b - a) + a); }
};

int to_int_x86(double v) {  // static_cast<int>(double) as cvttsd2si does it
    if (!(v > -2147483649.0 && v < 2147483648.0)) return INT_MIN;
    return (int)v;
}

// ---- Eigen::JacobiSVD<Matrix3d>: two-sided Jacobi, U flips make the values positive, then a descending sort ----------
struct Rot {
    double c, s;
};
// apply_rotation_in_the_plane(x, y, j): x' = c x + s y, y' = -s x + c y
void rot_rows(double *M, int p, int q, Rot j) {
    for (int i = 0; i < 3; ++i) {
        const double x = M[p * 3 + i], y = M[q * 3 + i];
        M[p * 3 + i] = j.c * x + j.s * y;
        M[q * 3 + i] = -j.s * x + j.c * y;
    }
}
void rot_cols(double *M, int p, int q, Rot j) {  // applyOnTheRight(p, q, j) = rotation of the columns with j.transpose()
    const Rot t = {j.c, -j.s};
    for (int i = 0; i < 3; ++i) {
        const double x = M[i * 3 + p], y = M[i * 3 + q];
        M[i * 3 + p] = t.c * x + t.s * y;
        M[i * 3 + q] = -t.s * x + t.c * y;
    }
}
Rot make_jacobi(double x, double y, double z) {
    const double deno = 2.0 * std::fabs(y);
    if (deno < DBL_MIN) return {1.0, 0.0};
    const double tau = (x - z) / deno;
    const double w = std::sqrt(tau * tau + 1.0);
    const double t = tau > 0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
    const double sign_t = t > 0 ? 1.0 : -1.0;
    const double n = 1.0 / std::sqrt(t * t + 1.0);
    return {n, -sign_t * (y / std::fabs(y)) * std::fabs(t) * n};
}
void eigen_svd3(const double *Min, double *sv, double *U, double *V) {
    double W[9];
    double scale = 0;
    for (int i = 0; i < 9; ++i) scale = std::max(scale, std::fabs(Min[i]));
    if (scale == 0) scale = 1;
    for (int i = 0; i < 9; ++i) {
        W[i] = Min[i] / scale;
        U[i] = V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    }
    const double precision = 2.0 * DBL_EPSILON;
    double max_diag = std::max(std::fabs(W[0]), std::max(std::fabs(W[4]), std::fabs(W[8])));
    bool finished = false;
    for (int guard = 0; !finished && guard < 1000; ++guard) {
        finished = true;
        for (int p = 1; p < 3; ++p)
            for (int q = 0; q < p; ++q) {
                const double threshold = std::max(DBL_MIN, precision * max_diag);
                if (std::fabs(W[p * 3 + q]) > threshold || std::fabs(W[q * 3 + p]) > threshold) {
                    finished = false;
                    // real_2x2_jacobi_svd on the block (p,p) (p,q) / (q,p) (q,q)
                    double m00 = W[p * 3 + p], m01 = W[p * 3 + q], m10 = W[q * 3 + p], m11 = W[q * 3 + q];
                    Rot rot1;
                    const double t = m00 + m11, d = m10 - m01;
                    if (std::fabs(d) < DBL_MIN) {
                        rot1 = {1.0, 0.0};
                    } else {
                        const double u = t / d, tmp = std::sqrt(1.0 + u * u);
                        rot1 = {u / tmp, 1.0 / tmp};
                    }
                    {  // m.applyOnTheLeft(0, 1, rot1)
                        const double a0 = rot1.c * m00 + rot1.s * m10, a1 = rot1.c * m01 + rot1.s * m11;
                        const double b0 = -rot1.s * m00 + rot1.c * m10, b1 = -rot1.s * m01 + rot1.c * m11;
                        m00 = a0, m01 = a1, m10 = b0, m11 = b1;
                    }
                    const Rot jr = make_jacobi(m00, m01, m11);
                    const Rot jrt = {jr.c, -jr.s};
                    const Rot jl = {rot1.c * jrt.c - rot1.s * jrt.s, rot1.c * jrt.s + rot1.s * jrt.c};  // rot1 * j_right^T
                    rot_rows(W, p, q, jl);
                    rot_cols(U, p, q, Rot{jl.c, -jl.s});
                    rot_cols(W, p, q, jr);
                    rot_cols(V, p, q, jr);
                    max_diag = std::max(max_diag, std::max(std::fabs(W[p * 3 + p]), std::fabs(W[q * 3 + q])));
                }
            }
    }
    for (int i = 0; i < 3; ++i) {
        const double a = W[i * 3 + i];
        sv[i] = std::fabs(a);
        if (a < 0)
            for (int r = 0; r < 3; ++r) U[r * 3 + i] = -U[r * 3 + i];
    }
    for (int i = 0; i < 3; ++i) sv[i] *= scale;
    for (int i = 0; i < 3; ++i) {  // descending sort by swapping with the largest of the tail
        int pos = i;
        for (int k = i + 1; k < 3; ++k)
            if (sv[k] > sv[pos]) pos = k;
        if (sv[pos] == 0) break;
        if (pos != i) {
            std::swap(sv[i], sv[pos]);
            for (int r = 0; r < 3; ++r) {
                std::swap(U[r * 3 + i], U[r * 3 + pos]);
                std::swap(V[r * 3 + i], V[r * 3 + pos]);
            }
        }
    }
}

bool is_zero(double d) { return d < 1e-3 && d > -1e-3; }  // five-point.hpp:76-81, pose_helper.h:82-87

// The SIGN of a 5-point solution.  The reference's E = x E0 + y E1 + z E2 + E3 inherits its sign from E3, the last right singular
// vector cv::SVD returns for the 5 x 9 system -- one of four vectors spanning a null space, whose individual directions are decided by
// rounding noise inside OpenCV's Jacobi iteration (rotations between columns that have already converged to zero).  ValidModel is
// not sign-symmetric when a sample violates the oriented epipolar constraint in both orientations (it tries the given sign first), so the
// reference's verdict on such a model is an artefact no restatement can reproduce.  This oracle and the device code fix the sign
// instead: the element of largest magnitude is positive.  The ORDER of the up to ten solutions of a sample (cv::solvePoly's root order
// for the polynomial in that same noise-dependent basis) decides which of them triggers the inner RANSAC first; it is fixed to
// ascending E(0,0) after the sign convention.
void canonical_sign(double *E) {
    int at = 0;
    for (int k = 1; k < 9; ++k)
        if (std::fabs(E[k]) > std::fabs(E[at])) at = k;
    if (E[at] < 0)
        for (int k = 0; k < 9; ++k) E[k] = -E[k];
}

// CvEMEstimator::ValidModel (five-point.cpp:534-601); pts = m x 2 sample coordinates
bool valid_model(const double *q1, const double *q2, int m, const double *Ein) {
    double E[9];
    std::memcpy(E, Ein, sizeof(E));
    bool emult = false;
    for (;;) {
        double Et[9], sv[3], U[9], V[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) Et[r * 3 + c] = E[c * 3 + r];
        eigen_svd3(Et, sv, U, V);
        if (sv[0] / sv[1] > 1.2) return false;
        if (!is_zero(0.01 * sv[2] / sv[1])) return false;
        const double e2[3] = {V[2], V[5], V[8]};
        int fail = 0;
        bool again = false;
        for (int i = 0; i < m && !again; ++i) {
            const double x1[3] = {q1[2 * i], q1[2 * i + 1], 1.0}, x2[3] = {q2[2 * i], q2[2 * i + 1], 1.0};
            const double l1[3] = {e2[1] * x2[2] - e2[2] * x2[1], e2[2] * x2[0] - e2[0] * x2[2], e2[0] * x2[1] - e2[1] * x2[0]};
            double l2[3];
            for (int r = 0; r < 3; ++r) l2[r] = E[r * 3] * x1[0] + E[r * 3 + 1] * x1[1] + E[r * 3 + 2] * x1[2];
            for (int j = 0; j < 3; ++j) {
                if (is_zero(0.1 * l1[j]) || is_zero(0.1 * l2[j])) continue;
                if (l1[j] * l2[j] < 0) {
                    if (!emult) {
                        emult = true;
                        for (int k = 0; k < 9; ++k) E[k] = E[k] * -1.0;
                        again = true;
                        break;
                    }
                    fail++;
                    break;
                }
            }
        }
        if (again) continue;
        return !((float)fail / (float)m >= 0.4);
    }
}

// cv::findFundamentalMat(m1, m2, FM_8POINT) for 8 <= m points given as doubles (converted to float32 first, as OpenCV does)
bool cv_fm_8point(const double *q1, const double *q2, int m, double *F) {
    std::vector<float> a(2 * m), b(2 * m);
    for (int i = 0; i < 2 * m; ++i) a[i] = (float)q1[i], b[i] = (float)q2[i];
    double m1cx = 0, m1cy = 0, m2cx = 0, m2cy = 0, scale1 = 0, scale2 = 0;
    for (int i = 0; i < m; ++i) m1cx += a[2 * i], m1cy += a[2 * i + 1], m2cx += b[2 * i], m2cy += b[2 * i + 1];
    const double t = 1. / m;
    m1cx *= t, m1cy *= t, m2cx *= t, m2cy *= t;
    for (int i = 0; i < m; ++i) {
        const double dx1 = a[2 * i] - m1cx, dy1 = a[2 * i + 1] - m1cy, dx2 = b[2 * i] - m2cx, dy2 = b[2 * i + 1] - m2cy;
        scale1 += std::sqrt(dx1 * dx1 + dy1 * dy1);
        scale2 += std::sqrt(dx2 * dx2 + dy2 * dy2);
    }
    scale1 *= t, scale2 *= t;
    if (scale1 < FLT_EPSILON || scale2 < FLT_EPSILON) return false;
    scale1 = std::sqrt(2.) / scale1, scale2 = std::sqrt(2.) / scale2;
    double A[81] = {0};
    for (int i = 0; i < m; ++i) {
        const double x1 = (a[2 * i] - m1cx) * scale1, y1 = (a[2 * i + 1] - m1cy) * scale1;
        const double x2 = (b[2 * i] - m2cx) * scale2, y2 = (b[2 * i + 1] - m2cy) * scale2;
        const double r[9] = {x2 * x1, x2 * y1, x2, y2 * x1, y2 * y1, y2, x1, y1, 1};
        for (int j = 0; j < 9; ++j)
            for (int k = 0; k < 9; ++k) A[j * 9 + k] += r[j] * r[k];
    }
    double W[9], V[81];
    oracle_jacobi_svd(A, 9, 9, W, V);  // symmetric positive semi-definite: singular pairs = eigen pairs, descending
    if (std::fabs(W[7]) < DBL_EPSILON) return false;
    double F0[9];
    for (int k = 0; k < 9; ++k) F0[k] = V[k * 9 + 8];
    // rank 2: U diag(w0, w1, 0) V^T = F0 - (F0 v2) v2^T with v2 the right singular vector of the smallest singular value
    double w3[3], V3[9];
    oracle_jacobi_svd(F0, 3, 3, w3, V3);
    const double v2[3] = {V3[2], V3[5], V3[8]};
    double Fv[3];
    for (int r = 0; r < 3; ++r) Fv[r] = F0[r * 3] * v2[0] + F0[r * 3 + 1] * v2[1] + F0[r * 3 + 2] * v2[2];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) F0[r * 3 + c] -= Fv[r] * v2[c];
    // F = T2^T F0 T1
    const double T1[9] = {scale1, 0, -scale1 * m1cx, 0, scale1, -scale1 * m1cy, 0, 0, 1};
    const double T2[9] = {scale2, 0, -scale2 * m2cx, 0, scale2, -scale2 * m2cy, 0, 0, 1};
    double tmp[9];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += T2[k * 3 + r] * F0[k * 3 + c];
            tmp[r * 3 + c] = s;
        }
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            double s = 0;
            for (int k = 0; k < 3; ++k) s += tmp[r * 3 + k] * T1[k * 3 + c];
            F[r * 3 + c] = s;
        }
    if (std::fabs(F[8]) > FLT_EPSILON) {
        const double inv = 1. / F[8];
        for (int k = 0; k < 9; ++k) F[k] *= inv;
    }
    return true;
}

// CalculateSPRTDecisionThreshold (sequential_probability_ratio.cc:38-62)
double sprt_threshold(double sigma, double epsilon, double time_ratio, int num_models_verified) {
    const double c = (1.0 - sigma) * std::log((1.0 - sigma) / (1.0 - epsilon)) + sigma * std::log(sigma / epsilon);
    const double a_0 = time_ratio * c / static_cast<double>(num_models_verified) + 1.0;
    double th = a_0;
    for (int i = 0; i < 1000; i++) {
        const double nt = a_0 + std::log(th);
        const double step = std::fabs(nt - th);
        th = nt;
        if (step < 1e-4) break;
    }
    return th;
}

struct Model {
    double E[9];
};
struct Scored {
    Model m;
    double score;
};
bool cmp_scored(Scored i, Scored j) { return i.score > j.score; }

int32_t *g_trace = nullptr;  // optional turn-by-turn record of the first stage (tests): 20 ints per turn
int g_trace_cap = 0, g_trace_len = 0;
void trace_turn(int k, int inner, const std::vector<int> &subset, int nvalid, const int *res) {
    if (!g_trace || g_trace_len + 20 > g_trace_cap) return;
    int32_t *r = g_trace + g_trace_len;
    r[0] = k, r[1] = inner, r[2] = (int)subset.size();
    for (int i = 0; i < 5; ++i) r[3 + i] = subset[i];
    r[8] = nvalid;
    for (int i = 0; i < 10; ++i) r[9 + i] = i < nvalid ? res[i] : -1;
    r[19] = (subset.size() > 5 ? subset[5] : 0) + 100 * (subset.size() > 6 ? subset[6] : 0);  // indices of the first stage are < 100
    g_trace_len += 20;
}

struct Arrsac {
    const double *p1, *p2;
    int n;
    double error_thresh;
    CvRng prosac_rng, random_rng;
    // Arrsac(5, thr^2, 500, 100, 14, 8) (modelest.cpp:270)
    const int min_sample_size = 5, max_candidate_hyps = 500, block_size = 100, nonmin_sample_size = 14, nonmin_sample_size_min = 8;
    double sigma = 0.05, epsilon = 0.1;
    const double inlier_confidence = 0.95, time_ratio = 250.0;
    int num_models_verified_accum = 0, num_rejected = 0;
    double rejected_accum = 0.0;
    const int max_inner_ransac_its = 20;
    int64_t stats[8] = {0};

    double error(int i, const Model &m) const {
        float e;
        oracle_sampson_err(p1 + 2 * i, p2 + 2 * i, 1, m.E, &e);
        return (double)e;
    }
    void gather(const std::vector<int> &s, std::vector<double> &a, std::vector<double> &b) const {
        a.resize(2 * s.size()), b.resize(2 * s.size());
        for (size_t i = 0; i < s.size(); ++i) {
            a[2 * i] = p1[2 * s[i]], a[2 * i + 1] = p1[2 * s[i] + 1];
            b[2 * i] = p2[2 * s[i]], b[2 * i + 1] = p2[2 * s[i] + 1];
        }
    }
    // EssentialMatEstimatorTheia::EstimateModel (modelest.cpp:111-148)
    bool estimate_model(const std::vector<int> &s, std::vector<Model> &out) const {
        std::vector<double> a, b;
        gather(s, a, b);
        double Es[90];
        const int nm = oracle_run5point(a.data(), b.data(), (int)s.size(), Es);
        if (nm <= 0) return false;
        int order[10];
        for (int i = 0; i < nm; ++i) {
            canonical_sign(Es + 9 * i);
            order[i] = i;
        }
        // ... and the ORDER of a sample's solutions (the reference: cv::solvePoly's root order for ITS null-space basis): ascending E(0,0)
        std::stable_sort(order, order + nm, [&](int x, int y) { return Es[9 * x] < Es[9 * y]; });
        for (int oi = 0; oi < nm; ++oi) {
            const int i = order[oi];
            if (!valid_model(a.data(), b.data(), (int)s.size(), Es + 9 * i)) continue;
            Model m;
            std::memcpy(m.E, Es + 9 * i, sizeof(m.E));
            out.push_back(m);
        }
        return !out.empty();
    }
    // EstimateModelNonminimal (modelest.cpp:151-178)
    bool estimate_nonminimal(const std::vector<int> &s, std::vector<Model> &out) const {
        std::vector<double> a, b;
        gather(s, a, b);
        Model m;
        if (!cv_fm_8point(a.data(), b.data(), (int)s.size(), m.E)) return false;
        if (!valid_model(a.data(), b.data(), (int)s.size(), m.E)) return false;
        out.push_back(m);
        return true;
    }
    // SequentialProbabilityRatioTest over the first `count` correspondences
    bool sprt(int count, const Model &h, double decision_threshold, int *num_tested, double *ratio, std::vector<char> &inl,
              int *num_inl) const {
        *num_inl = 0;
        double lr = 1.0;
        inl.assign(count, 0);
        for (int i = 0; i < count; ++i) {
            if (error(i, h) < error_thresh) {
                lr *= sigma / epsilon;
                *num_inl += 1;
                inl[i] = 1;
            } else {
                lr *= (1.0 - sigma) / (1.0 - epsilon);
            }
            if (lr > decision_threshold) {
                *ratio = static_cast<double>(*num_inl) / static_cast<double>(i + 1);
                *num_tested = i + 1;
                return false;
            }
        }
        *ratio = static_cast<double>(*num_inl) / static_cast<double>(count);
        *num_tested = count;
        return true;
    }
    // ProsacSampler::Sample for the k-th sample over the first `count` correspondences
    void prosac_sample(int count, int k, std::vector<int> &subset) {
        double t_n = 200000;
        int nn = min_sample_size;
        for (int i = 0; i < min_sample_size; i++) t_n *= static_cast<double>(nn - i) / (double)((size_t)count - i);
        double t_n_prime = 1.0;
        for (int t = 1; t <= k; t++) {
            if (t > t_n_prime && nn < count) {
                const double t_n_plus1 = (t_n * ((double)nn + 1.0)) / ((double)nn + 1.0 - (double)min_sample_size);
                t_n_prime += std::ceil(t_n_plus1 - t_n);
                t_n = t_n_plus1;
                nn++;
            }
        }
        subset.clear();
        std::vector<int> used;
        if (t_n_prime < k) {
            for (int i = 0; i < min_sample_size; i++) {
                int r;
                while (std::find(used.begin(), used.end(), (r = prosac_rng.uniform(0, nn))) != used.end()) {
                }
                used.push_back(r);
                subset.push_back(r);
            }
        } else {
            for (int i = 0; i < min_sample_size - 1; i++) {
                int r;
                while (std::find(used.begin(), used.end(), (r = prosac_rng.uniform(0, nn - 1))) != used.end()) {
                }
                used.push_back(r);
                subset.push_back(r);
            }
            subset.push_back(nn - 1);
        }
    }
    // RandomSampler::Sample: `size` distinct positions of `universe`
    void random_sample(const std::vector<int> &universe, int size, std::vector<int> &subset) {
        subset.assign(size, 0);
        std::vector<int> used;
        for (int i = 0; i < size; i++) {
            int r;
            while (std::find(used.begin(), used.end(), (r = random_rng.uniform(0, (int)universe.size()))) != used.end()) {
            }
            used.push_back(r);
            subset[i] = universe[r];
        }
    }
    int hyps_needed(double eps, double power) const {  // ceil(log(1 - conf) / log(1 - eps^power)), capped at M
        const int v = to_int_x86(std::ceil(std::log(1.0 - inlier_confidence) / std::log(1.0 - std::pow(eps, power))));
        return std::min(max_candidate_hyps, v);
    }

    // Arrsac::GenerateInitialHypothesisSet (arrsac.h:236-372) on the first `count` correspondences
    int initial_set(int count, std::vector<Scored> &accepted) {
        int k = 1, k2 = 0, m_prime = max_candidate_hyps, inner_its = 0, max_num_inliers = 0;
        bool inner = false;
        int random_size = nonmin_sample_size;
        std::vector<int> data;
        while (k <= m_prime) {
            std::vector<Model> hyps;
            std::vector<int> subset;
            const bool turn_inner = inner;
            if (!inner) {
                prosac_sample(count, k, subset);
                stats[2]++;
                if (!estimate_model(subset, hyps)) {
                    trace_turn(k, 0, subset, 0, nullptr);
                    k2++, k++;
                    continue;
                }
            } else {
                random_sample(data, random_size, subset);
                stats[3]++;
                bool valid;
                if (random_size == min_sample_size || random_size < nonmin_sample_size_min)
                    valid = estimate_model(subset, hyps);
                else
                    valid = estimate_nonminimal(subset, hyps);
                inner_its++;
                if (inner_its == max_inner_ransac_its) inner_its = 0, inner = false;
                if (!valid) {
                    trace_turn(k, 1, subset, 0, nullptr);
                    k2++, k++;
                    continue;
                }
            }
            num_models_verified_accum += (int)hyps.size();
            const double dt = sprt_threshold(sigma, epsilon, time_ratio, num_models_verified_accum / (k - k2));
            int res[10];
            for (size_t j = 0; j < hyps.size(); j++) {
                int tested, num_inl;
                double ratio;
                std::vector<char> inl;
                const bool ok = sprt(count, hyps[j], dt, &tested, &ratio, inl, &num_inl);
                if (j < 10) res[j] = (ok ? 1000 : 0) + num_inl;
                if (!ok) {
                    rejected_accum += ratio;
                    num_rejected++;
                    const double st = rejected_accum / static_cast<double>(num_rejected);
                    if (st > 0) sigma = st;
                } else if (num_inl > max_num_inliers) {
                    max_num_inliers = num_inl;
                    accepted.push_back(Scored{hyps[j], (double)num_inl});
                    if (num_inl > min_sample_size) {
                        inner = true;
                        inner_its = 0;
                        random_size = std::max(std::min(nonmin_sample_size, (int)std::floor((float)max_num_inliers / 2.0f)), min_sample_size);
                        data.clear();
                        for (int i = 0; i < count; i++)
                            if (inl[i]) data.push_back(i);
                        epsilon = ratio == 1.0 ? 0.9999 : ratio;
                        m_prime = hyps_needed(epsilon, (double)min_sample_size);
                        stats[4]++;
                    }
                } else {
                    accepted.push_back(Scored{hyps[j], (double)num_inl});
                }
            }
            trace_turn(k, turn_inner ? 1 : 0, subset, (int)std::min<size_t>(hyps.size(), 10), res);
            k++;
        }
        if (accepted.empty()) return 0;
        return k - k2 - 1;
    }

    // Arrsac::Estimate (arrsac.h:375-547)
    bool estimate(Model *best) {
        const int sub_block = (int)std::floor((float)block_size / 5.0);
        const int kill_thresh = (int)std::floor(7.0 * (float)sub_block / 12.0);
        std::vector<Scored> hyps;
        int k = initial_set(std::min(n, block_size), hyps);
        stats[0] = k, stats[1] = (int64_t)hyps.size();
        if (k == 0) return false;
        if (n <= block_size) {
            double hi = 0.0;
            int at = 0;
            for (size_t i = 0; i < hyps.size(); i++)
                if (hyps[i].score > hi) hi = hyps[i].score, at = (int)i;
            *best = hyps[at].m;
            return true;
        }
        std::vector<int> all(n);
        for (int i = 0; i < n; ++i) all[i] = i;
        int nh = (int)hyps.size();
        int i = block_size;
        for (; i < n; i++) {
            if ((i + 1) % block_size == 0) {
                std::sort(hyps.begin(), hyps.end(), cmp_scored);
                double max_inliers = hyps[0].score;
                epsilon = max_inliers / static_cast<double>(i + 1);
                if (epsilon == 1.0) epsilon = 0.9999;
                int temp_max = hyps_needed(epsilon, (double)(i + 1));
                if (temp_max > k) {
                    int k2 = 0;
                    // `temp_max - k` is re-evaluated every turn (k grows with each generated hypothesis)
                    for (int j = 0; j < (int64_t)temp_max - k; j++) {
                        std::vector<int> subset;
                        random_sample(all, min_sample_size, subset);
                        stats[5]++;
                        std::vector<Model> est;
                        if (!estimate_model(subset, est)) {
                            k2++;
                            continue;
                        }
                        num_models_verified_accum += (int)est.size();
                        const double dt = sprt_threshold(sigma, epsilon, time_ratio, num_models_verified_accum / (k + j + 1 - k2));
                        for (size_t m = 0; m < est.size(); m++) {
                            int tested, num_inl;
                            double ratio;
                            std::vector<char> inl;
                            const bool ok = sprt(i + 1, est[m], dt, &tested, &ratio, inl, &num_inl);
                            if (!ok) {
                                rejected_accum += ratio;
                                num_rejected++;
                                const double st = rejected_accum / static_cast<double>(num_rejected);
                                if (st > 0) sigma = st;
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
                    const int n1 = std::max(1, (int)std::floor((float)k * std::pow(2.0, -1.0 * std::floor((float)(i + 1) / (float)block_size))));
                    if (n1 < static_cast<int>(hyps.size())) {
                        hyps.resize(n1);
                        nh = n1;
                    }
                }
            } else if ((i + 1) % sub_block == 0) {
                std::sort(hyps.begin(), hyps.end(), cmp_scored);
                int j = nh - 1;
                for (; j > 0; j--)
                    if (hyps[j - 1].score - hyps[j].score > kill_thresh) break;
                nh = nh - j;
                hyps.resize(nh);
            }
            if (nh == 1) break;
            for (size_t j = 0; j < hyps.size(); j++)
                if (error(i, hyps[j].m) < error_thresh) hyps[j].score += 1.0;
        }
        stats[6] = i, stats[7] = nh;
        *best = hyps[0].m;
        return true;
    }
};

// pose_helper.cpp:115-143
void sampson_l1(const double *x1, const double *x2, const double *E, double &denom1, double &num) {
    const double X1[3] = {x1[0], x1[1], 1.0}, X2[3] = {x2[0], x2[1], 1.0};
    double xpE[3], Ex1[3];
    for (int c = 0; c < 3; ++c) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += X2[k] * E[k * 3 + c];
        xpE[c] = s;
    }
    num = xpE[0] * X1[0] + xpE[1] * X1[1] + xpE[2] * X1[2];
    for (int r = 0; r < 3; ++r) {
        double s = 0;
        for (int k = 0; k < 3; ++k) s += E[r * 3 + k] * X1[k];
        Ex1[r] = s;
    }
    const double a = Ex1[0] * Ex1[0], b = Ex1[1] * Ex1[1], c = xpE[0] * xpE[0], d = xpE[1] * xpE[1];
    denom1 = 1 / (std::sqrt(a + b + c + d) + 1e-8);
}
double cost_pseudo_huber(double d, double thresh) {
    const double b_sq = thresh * thresh;
    const double d_abs = std::fabs(d) + 1e-12;
    const double q = d_abs / thresh;
    return std::sqrt(2 * b_sq * (std::sqrt(1 + q * q) - 1)) / d_abs;
}

}  // namespace

extern "C" {

void oracle_cv_rng_stream(uint64_t *state, int count, uint32_t *out) {
    CvRng r{*state};
    for (int i = 0; i < count; ++i) out[i] = r.next();
    *state = r.state;
}

void oracle_eigen_svd3(const double *M, double *sv, double *U, double *V) { eigen_svd3(M, sv, U, V); }

int oracle_valid_model(const double *q1, const double *q2, int m, const double *E) { return valid_model(q1, q2, m, E) ? 1 : 0; }

int oracle_cv_fm_8point(const double *q1, const double *q2, int m, double *F) { return cv_fm_8point(q1, q2, m, F) ? 1 : 0; }

double oracle_sprt_threshold(double sigma, double epsilon, double time_ratio, int num_models_verified) {
    return sprt_threshold(sigma, epsilon, time_ratio, num_models_verified);
}

/* robustEssentialRefine(points1, points2, E_init, E_refined, th, 0, true, ...) (pose_estim.cpp:337-792) for model 0 without
 * normalisation and without a mask: iteratively re-weighted (pseudo-Huber on the Sampson distance) linear fit, the closest essential
 * matrix after every step.  Returns the number of iterations executed; err[2] = {first, last sum of squared errors}. */
int oracle_robust_essential_refine(const double *p1, const double *p2, int n, const double *E_init, double th, double *E_refined,
                                   double *err2) {
    std::memcpy(E_refined, E_init, 72);
    if (err2) err2[0] = err2[1] = 999.0;
    if (n < 50) return 0;
    const double min_diff = th / 10, min_err = th * th / 100 * n;
    double F3[9], err = -9999.0, err_old = 1e12;
    std::memcpy(F3, E_init, 72);
    std::vector<double> w(n), d1(n), A1((size_t)n * 9);
    int j = 0;
    for (; j < 50; j++) {
        for (int i = 0; i < n; i++) {
            double denom1, num;
            sampson_l1(p1 + 2 * i, p2 + 2 * i, F3, denom1, num);
            w[i] = cost_pseudo_huber(num * denom1, th);
            d1[i] = denom1;
        }
        double wn = 0;
        for (int i = 0; i < n; i++) wn += std::pow(w[i] * d1[i], 2);
        wn = std::sqrt(wn);
        for (int i = 0; i < n; i++) {
            const double x0 = p1[2 * i], y0 = p1[2 * i + 1], x1 = p2[2 * i], y1 = p2[2 * i + 1];
            const double r[9] = {x1 * x0, x1 * y0, x1, y1 * x0, y1 * y0, y1, x0, y0, 1};
            const double f = d1[i] * w[i] / wn;
            for (int k = 0; k < 9; ++k) A1[(size_t)i * 9 + k] = r[k] * f;
        }
        double A2[81] = {0};
        for (int i = 0; i < n; i++)
            for (int a = 0; a < 9; ++a)
                for (int b = 0; b < 9; ++b) A2[a * 9 + b] += A1[(size_t)i * 9 + a] * A1[(size_t)i * 9 + b];
        double W[9], V[81];
        oracle_jacobi_svd(A2, 9, 9, W, V);
        if (std::fabs(W[7]) < DBL_EPSILON) {
            std::memcpy(E_refined, E_init, 72);
            return -1;
        }
        double last[9], F2[9];
        for (int k = 0; k < 9; ++k) last[k] = V[k * 9 + 8];
        std::memcpy(F2, last, 72);
        {  // getClosestE (pose_helper.cpp:152-177)
            double sv[3], U[9], Vv[9];
            eigen_svd3(F2, sv, U, Vv);
            if (!is_zero(sv[2])) break;
            if (sv[0] / sv[1] > 1.5 || sv[0] / sv[1] < 0.66) break;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) F2[r * 3 + c] = U[r * 3] * sv[0] * Vv[c * 3] + U[r * 3 + 1] * sv[1] * Vv[c * 3 + 1];
            if (!(sv[1] > 0)) break;  // validateEssential without the full check: the kernel of E^T must be one-dimensional
        }
        std::memcpy(F3, F2, 72);
        err = 0;
        for (int i = 0; i < n; i++) {
            double s = 0;
            for (int k = 0; k < 9; ++k) s += A1[(size_t)i * 9 + k] * last[k];
            err += s * s;
        }
        const double diff = std::fabs(err_old - err);
        if (j > 1 && (diff < min_diff || err < min_err)) break;
        err_old = err;
        if (!j && err2) err2[0] = err;
    }
    if (err2 && j) err2[1] = err;
    std::memcpy(E_refined, F3, 72);
    return j;
}

/* CvModelEstimator3::runARRSAC as findEssentialMat drives it (five-point.cpp:120-124, modelest.cpp:197-341).
 * rng_state[2]: cv::RNG state of ProsacSampler::Sample's and of RandomSampler::Sample's function-local static (in/out; both are
 * 0xffffffff in a fresh process).  refine = `lesqu` with robustEssentialRefine as the callback.  stats (may be NULL) receives
 * {k of the initial set, hypotheses after it, PROSAC samples, inner-RANSAC samples, inner-RANSAC restarts, samples generated in the
 * preemptive phase, correspondence index at the end of it, hypotheses left}.  Returns 1 on success. */
int oracle_arrsac_essential(const double *p1, const double *p2, int n, double thresh, int refine, uint64_t *rng_state, double *E,
                            uint8_t *mask, int *n_inliers, int64_t *stats) {
    if (n < 5) return 0;
    std::vector<float> err(n);
    double sum;
    if (n == 5) {  // modelest.cpp:219-261
        double Es[90];
        const int nm = oracle_run5point(p1, p2, 5, Es);
        if (nm <= 0) return 0;
        std::vector<uint8_t> tm(n);
        int max_good = 0;
        bool result = false;
        double errmin = DBL_MAX;
        for (int i = 0; i < nm; ++i) {
            const int good = oracle_find_inliers(p1, p2, n, Es + 9 * i, thresh, err.data(), tm.data(), &sum);
            if (good > std::max(max_good, 4)) {
                std::memcpy(mask, tm.data(), n), std::memcpy(E, Es + 9 * i, 72);
                max_good = good, errmin = sum, result = true;
            } else if (good == std::max(max_good, 5) && errmin < DBL_MAX && errmin > sum) {
                std::memcpy(mask, tm.data(), n), std::memcpy(E, Es + 9 * i, 72);
                errmin = sum;
            }
        }
        if (n_inliers) *n_inliers = max_good;
        return result ? 1 : 0;
    }
    Arrsac a;
    a.p1 = p1, a.p2 = p2, a.n = n, a.error_thresh = thresh * thresh;
    a.prosac_rng.state = rng_state[0], a.random_rng.state = rng_state[1];
    Model best;
    const bool ok = a.estimate(&best);
    rng_state[0] = a.prosac_rng.state, rng_state[1] = a.random_rng.state;
    if (stats) std::memcpy(stats, a.stats, sizeof(a.stats));
    if (!ok) return 0;
    int good = oracle_find_inliers(p1, p2, n, best.E, thresh, err.data(), mask, &sum);
    if (n_inliers) *n_inliers = good;
    if ((good < 50 && n > 200) || good < 15) return 0;
    std::memcpy(E, best.E, 72);
    if (refine && good >= 50) {
        std::vector<double> a1((size_t)good * 2), a2((size_t)good * 2);
        int j = 0;
        for (int i = 0; i < n; ++i)
            if (mask[i]) {
                a1[2 * j] = p1[2 * i], a1[2 * j + 1] = p1[2 * i + 1], a2[2 * j] = p2[2 * i], a2[2 * j + 1] = p2[2 * i + 1];
                j++;
            }
        // the reference's acceptance test compares the inlier count of the UNREFINED model with itself (ratio 1 > 0.66): the refined
        // matrix is always taken, the mask stays the unrefined model's (modelest.cpp:312-318)
        oracle_robust_essential_refine(a1.data(), a2.data(), good, best.E, thresh / 50.0, E, nullptr);
    }
    return 1;
}

/* Test hook: record the turns of the first stage of the following oracle_arrsac_essential calls into buf (20 ints per turn:
 * k, inner, sample size, first five indices, valid models, per model 1000 * accepted + inliers seen, sixth + 100 * seventh index);
 * returns the ints written so far. */
int oracle_arrsac_trace(int32_t *buf, int cap) {
    const int len = g_trace_len;
    g_trace = buf, g_trace_cap = cap, g_trace_len = 0;
    return len;
}

/* std::sort with the reference's comparator (arrsac.h:82-84): perm[i] = original position of the i-th element afterwards. */
void oracle_std_sort_desc(const double *score, int n, int32_t *perm) {
    struct It {
        int id;
        double score;
    };
    std::vector<It> v(n);
    for (int i = 0; i < n; ++i) v[i] = {i, score[i]};
    std::sort(v.begin(), v.end(), [](It a, It b) { return a.score > b.score; });
    for (int i = 0; i < n; ++i) perm[i] = v[i].id;
}

}  // extern "C"
