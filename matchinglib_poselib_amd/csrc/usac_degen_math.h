// usac_degen_math.h -- the fixed-size numerics behind USAC's degeneracy tests and model upgrade (host side; included by usac_impl.h).
//
// Everything here works on two, a handful or five bearing vectors and 3 x 3 matrices: the per-correspondence work of the same tests
// (the angular error of every correspondence under a rotation / a translation / no motion, the Sampson error of upgrade candidates)
// runs on the device (usac_degen_rows_kernel in usac_impl.h).
//
// Restates, from their published formulations, what the reference takes from its vendored OpenGV (thirdparty/opengv/src/):
//   relative_pose/methods.cpp:57-86    twopt                 translation from two correspondences of known rotation
//   relative_pose/methods.cpp:98-122   twopt_rotationOnly    Arun's rotation from two correspondences ("centred" on a third of their sum)
//   relative_pose/methods.cpp:128-160  rotationOnly          Arun's rotation from n correspondences
//   math/arun.cpp:33-56                arun                  R = V U^T of the cross-covariance (Eigen::JacobiSVD), determinant forced to +1
//   relative_pose/methods.cpp:496-551 + modules/main.cpp:619-666 + modules/eigensolver/modules.cpp   eigensolver (Kneip & Lynen): rotation
//                                      minimising the smallest eigenvalue of M(R) = sum (f1 x R f2)(f1 x R f2)^T, found by
//                                      Levenberg-Marquardt on the gradient with a forward-difference Jacobian (Eigen's port of MINPACK
//                                      lmdif: ftol 5e-5, xtol 10 eps, at most 100 evaluations), translation = eigenvector
//   math/cayley.cpp                    the Cayley parameters
// The translation is the eigenvector OpenGV takes: column 0 of Eigen::EigenSolver's unordered decomposition (eigen_diag_order3 below).
#pragma once
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstring>

namespace dgm {

inline void bearing(double x, double y, double *f) {  // (x, y, 1) / |.| as Eigen evaluates it: x^2 + (y^2 + 1)
    const double nrm = std::sqrt(x * x + (y * y + 1.0));
    f[0] = x / nrm, f[1] = y / nrm, f[2] = 1.0 / nrm;
}
inline void cross(const double *a, const double *b, double *o) {
    o[0] = a[1] * b[2] - a[2] * b[1], o[1] = a[2] * b[0] - a[0] * b[2], o[2] = a[0] * b[1] - a[1] * b[0];
}
inline double dot(const double *a, const double *b) { return a[0] * b[0] + (a[1] * b[1] + a[2] * b[2]); }
inline void matvec(const double *R, const double *v, double *o) {
    for (int r = 0; r < 3; ++r) o[r] = (R[3 * r] * v[0] + R[3 * r + 1] * v[1]) + R[3 * r + 2] * v[2];
}

// Eigen::JacobiSVD of a 3 x 3 matrix (row-major), as OpenGV's arun calls it (ComputeFullU | ComputeFullV): two-sided Jacobi on the pairs
// (1,0), (2,0), (2,1) -- a rotation that symmetrises the 2 x 2 block, then the symmetric Jacobi rotation --, the signs of the singular
// values moved into U, descending order by column swaps (Eigen/src/SVD/JacobiSVD.h; the same restatement as the device's svd3_eigen in
// arrsac_impl.h and the oracle's, which tests/golden/eigen_svd3.npz pins against the Eigen the reference vendors).  It matters which
// decomposition this is: for a cross-covariance of rank one or two (two correspondences) the rotation V U^T depends on the basis the
// algorithm leaves in the null space.
inline void svd3_pair(double *W, double *U, double *V, int p, int q, double &max_diag, bool &any) {
    const double threshold = std::fmax(DBL_MIN, 2.0 * DBL_EPSILON * max_diag);
    if (!(std::fabs(W[p * 3 + q]) > threshold || std::fabs(W[q * 3 + p]) > threshold)) return;
    any = true;
    double m00 = W[p * 3 + p], m01 = W[p * 3 + q], m10 = W[q * 3 + p], m11 = W[q * 3 + q];
    double r1c = 1.0, r1s = 0.0;
    const double t = m00 + m11, d = m10 - m01;
    if (!(std::fabs(d) < DBL_MIN)) {
        const double u = t / d, tmp = std::sqrt(1.0 + u * u);
        r1s = 1.0 / tmp, r1c = u / tmp;
    }
    {
        const double a0 = r1c * m00 + r1s * m10, a1 = r1c * m01 + r1s * m11, b1 = -r1s * m01 + r1c * m11;
        m00 = a0, m01 = a1, m11 = b1;
    }
    double jrc = 1.0, jrs = 0.0;
    const double deno = 2.0 * std::fabs(m01);
    if (!(deno < DBL_MIN)) {
        const double tau = (m00 - m11) / deno, w = std::sqrt(tau * tau + 1.0);
        const double tt = tau > 0 ? 1.0 / (tau + w) : 1.0 / (tau - w);
        const double sign_t = tt > 0 ? 1.0 : -1.0, nn = 1.0 / std::sqrt(tt * tt + 1.0);
        jrc = nn, jrs = -sign_t * (m01 / std::fabs(m01)) * std::fabs(tt) * nn;
    }
    const double jlc = r1c * jrc + r1s * jrs, jls = -r1c * jrs + r1s * jrc;  // rot1 * jr^T
    for (int i = 0; i < 3; ++i) {  // rows p, q of W turn with jl
        const double x = W[p * 3 + i], y = W[q * 3 + i];
        W[p * 3 + i] = jlc * x + jls * y, W[q * 3 + i] = -jls * x + jlc * y;
    }
    for (int i = 0; i < 3; ++i) {  // columns of U with jl^T, columns of W and V with jr
        double x = U[i * 3 + p], y = U[i * 3 + q];
        U[i * 3 + p] = jlc * x + jls * y, U[i * 3 + q] = -jls * x + jlc * y;
        x = W[i * 3 + p], y = W[i * 3 + q];
        W[i * 3 + p] = jrc * x - jrs * y, W[i * 3 + q] = jrs * x + jrc * y;
        x = V[i * 3 + p], y = V[i * 3 + q];
        V[i * 3 + p] = jrc * x - jrs * y, V[i * 3 + q] = jrs * x + jrc * y;
    }
    max_diag = std::fmax(max_diag, std::fmax(std::fabs(W[p * 3 + p]), std::fabs(W[q * 3 + q])));
}
inline void svd3(const double *A, double *U, double *S, double *V) {
    double W[9], scale = 0;
    for (int i = 0; i < 9; ++i) scale = std::fmax(scale, std::fabs(A[i]));
    if (scale == 0) scale = 1;
    for (int i = 0; i < 9; ++i) W[i] = A[i] / scale, U[i] = V[i] = (i % 4 == 0) ? 1.0 : 0.0;
    double max_diag = std::fmax(std::fabs(W[0]), std::fmax(std::fabs(W[4]), std::fabs(W[8])));
    for (int guard = 0; guard < 100; ++guard) {
        bool any = false;
        svd3_pair(W, U, V, 1, 0, max_diag, any);
        svd3_pair(W, U, V, 2, 0, max_diag, any);
        svd3_pair(W, U, V, 2, 1, max_diag, any);
        if (!any) break;
    }
    for (int i = 0; i < 3; ++i) {
        const double a = W[i * 4];
        S[i] = std::fabs(a) * scale;
        if (a < 0)
            for (int r = 0; r < 3; ++r) U[r * 3 + i] = -U[r * 3 + i];
    }
    auto swap_cols = [&](int a, int b) {
        double tmp = S[a];
        S[a] = S[b], S[b] = tmp;
        for (int r = 0; r < 3; ++r) {
            tmp = U[r * 3 + a], U[r * 3 + a] = U[r * 3 + b], U[r * 3 + b] = tmp;
            tmp = V[r * 3 + a], V[r * 3 + a] = V[r * 3 + b], V[r * 3 + b] = tmp;
        }
    };
    // descending order, first maximum wins (Eigen: swap with the largest of the tail, stop at a zero)
    if (S[1] > S[0] && S[1] >= S[2]) swap_cols(0, 1);
    else if (S[2] > S[0] && S[2] > S[1]) swap_cols(0, 2);
    if (S[0] != 0 && S[2] > S[1]) swap_cols(1, 2);
}

// arun (math/arun.cpp:33-56): R = V U^T; when its determinant is negative, the third column of V is negated
inline void arun(const double *H, double *R) {
    double U[9], S[3], V[9];
    svd3(H, U, S, V);
    for (int pass = 0; pass < 2; ++pass) {
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) R[3 * r + c] = V[3 * r] * U[3 * c] + V[3 * r + 1] * U[3 * c + 1] + V[3 * r + 2] * U[3 * c + 2];
        const double det = R[0] * (R[4] * R[8] - R[5] * R[7]) - R[1] * (R[3] * R[8] - R[5] * R[6]) + R[2] * (R[3] * R[7] - R[4] * R[6]);
        if (!(det < 0)) break;
        for (int r = 0; r < 3; ++r) V[3 * r + 2] = -V[3 * r + 2];
    }
}

// Cross-covariance sum (f' - c')(f - c)^T over correspondences given as (f = vector of view "1" of the adapter, f' = of view "2")
inline void cross_cov_add(double *H, const double *f, const double *fp, const double *c, const double *cp) {
    double a[3], b[3];
    for (int k = 0; k < 3; ++k) a[k] = f[k] - c[k], b[k] = fp[k] - cp[k];
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) H[3 * r + k] += b[r] * a[k];
}
inline void twopt_rotation(const double *f_a, const double *fp_a, const double *f_b, const double *fp_b, double *R) {
    double c[3], cp[3], H[9] = {0};
    for (int k = 0; k < 3; ++k) c[k] = (f_a[k] + f_b[k]) / 3.0, cp[k] = (fp_a[k] + fp_b[k]) / 3.0;  // OpenGV divides by three here
    cross_cov_add(H, f_a, fp_a, c, cp);
    cross_cov_add(H, f_b, fp_b, c, cp);
    arun(H, R);
}
inline void twopt_translation(const double *f1, const double *f1p, const double *f2, const double *f2p, double *t) {
    double n1[3], n2[3], flow[3];
    cross(f1, f1p, n1), cross(f2, f2p, n2);
    cross(n1, n2, t);
    const double nrm = std::sqrt(t[0] * t[0] + (t[1] * t[1] + t[2] * t[2]));
    for (int k = 0; k < 3; ++k) t[k] = t[k] / nrm, flow[k] = f1[k] - f1p[k];
    if (dot(flow, t) < 0)
        for (int k = 0; k < 3; ++k) t[k] = -t[k];
}

// E = [t / |t|]_x R as poselib::getEfromRT forms it (pose_helper.cpp:785-805: the vector times the reciprocal of its norm)
inline void e_from_rt(const double *R, const double *t, double *E) {
    const double s = 1.0 / std::sqrt(t[0] * t[0] + t[1] * t[1] + t[2] * t[2]);
    const double a = t[0] * s, b = t[1] * s, c = t[2] * s;
    const double Sk[9] = {0, -c, b, c, 0, -a, -b, a, 0};
    for (int r = 0; r < 3; ++r)
        for (int k = 0; k < 3; ++k) {
            double acc = 0;
            for (int m = 0; m < 3; ++m) acc += Sk[3 * r + m] * R[3 * m + k];
            E[3 * r + k] = acc;
        }
}

// ---- the eigensolver --------------------------------------------------------------------------------------------------------------
struct EigSums {
    double G[3][3][9];  // G[b][e] = sum f1_b f1_e (f2 f2^T): xxF = G[0][0], xyF = G[0][1], ...
};
inline void eig_sums(const double (*f1)[3], const double (*f2)[3], int n, EigSums &Sx) {
    std::memset(&Sx, 0, sizeof(Sx));
    for (int i = 0; i < n; ++i) {
        double F[9];
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) F[3 * r + c] = f2[i][r] * f2[i][c];
        for (int b = 0; b < 3; ++b)
            for (int e = 0; e < 3; ++e) {
                const double w = f1[i][b] * f1[i][e];
                for (int k = 0; k < 9; ++k) Sx.G[b][e][k] += w * F[k];
            }
    }
}
inline void cayley_reduced(const double *c, double *R) {  // (1 + |c|^2) times the rotation
    R[0] = 1 + c[0] * c[0] - c[1] * c[1] - c[2] * c[2], R[1] = 2 * (c[0] * c[1] - c[2]), R[2] = 2 * (c[0] * c[2] + c[1]);
    R[3] = 2 * (c[0] * c[1] + c[2]), R[4] = 1 - c[0] * c[0] + c[1] * c[1] - c[2] * c[2], R[5] = 2 * (c[1] * c[2] - c[0]);
    R[6] = 2 * (c[0] * c[2] - c[1]), R[7] = 2 * (c[1] * c[2] + c[0]), R[8] = 1 - c[0] * c[0] - c[1] * c[1] + c[2] * c[2];
}
inline void cayley_reduced_jac(const double *c, int k, double *J) {  // d cayley_reduced / d c_k
    if (k == 0) {
        const double v[9] = {2 * c[0], 2 * c[1], 2 * c[2], 2 * c[1], -2 * c[0], -2, 2 * c[2], 2, -2 * c[0]};
        std::memcpy(J, v, 72);
    } else if (k == 1) {
        const double v[9] = {-2 * c[1], 2 * c[0], 2, 2 * c[0], 2 * c[1], 2 * c[2], -2, 2 * c[2], -2 * c[1]};
        std::memcpy(J, v, 72);
    } else {
        const double v[9] = {-2 * c[2], -2, 2 * c[0], 2, -2 * c[2], 2 * c[1], 2 * c[0], 2 * c[1], 2 * c[2]};
        std::memcpy(J, v, 72);
    }
}
inline double bilinear(const double *a, const double *G, const double *b) {  // a^T G b, rows a, b of a 3 x 3 matrix
    double s = 0;
    for (int r = 0; r < 3; ++r) s += a[r] * (G[3 * r] * b[0] + G[3 * r + 1] * b[1] + G[3 * r + 2] * b[2]);
    return s;
}
// M(R)_ad = sum eps_abc eps_def  r_c^T G_be r_f  (r_c = row c of R); with dR != nullptr the derivative along dR instead
inline void compose_M(const EigSums &Sx, const double *R, const double *dR, double *M) {
    static const int perm[3][2][2] = {{{1, 2}, {2, 1}}, {{2, 0}, {0, 2}}, {{0, 1}, {1, 0}}};  // (b, c) with eps_abc = +1, -1
    for (int a = 0; a < 3; ++a)
        for (int d = a; d < 3; ++d) {
            double s = 0;
            for (int x = 0; x < 2; ++x)
                for (int y = 0; y < 2; ++y) {
                    const int b = perm[a][x][0], c = perm[a][x][1], e = perm[d][y][0], f = perm[d][y][1];
                    const double sign = (x == y) ? 1.0 : -1.0;
                    if (!dR)
                        s += sign * bilinear(R + 3 * c, Sx.G[b][e], R + 3 * f);
                    else
                        s += sign * (bilinear(dR + 3 * c, Sx.G[b][e], R + 3 * f) + bilinear(R + 3 * c, Sx.G[b][e], dR + 3 * f));
                }
            M[3 * a + d] = s, M[3 * d + a] = s;
        }
}
// Gradient of the smallest eigenvalue of M(cayley) w.r.t. the Cayley parameters: the closed form of the smallest root of the
// characteristic polynomial (b, c, d; trigonometric solution) differentiated by the chain rule, as OpenGV does it -- including its
// behaviour where two eigenvalues meet (acos of an argument at the edge of [-1, 1]).
inline void smallest_ev_gradient(const EigSums &Sx, const double *cay, double *grad) {
    double R[9], M[9], dM[3][9];
    cayley_reduced(cay, R);
    compose_M(Sx, R, nullptr, M);
    for (int k = 0; k < 3; ++k) {
        double dR[9];
        cayley_reduced_jac(cay, k, dR);
        compose_M(Sx, R, dR, dM[k]);
    }
    const double m00 = M[0], m01 = M[1], m02 = M[2], m11 = M[4], m12 = M[5], m22 = M[8];
    const double b = -m00 - m11 - m22;
    const double c = -m02 * m02 - m12 * m12 - m01 * m01 + m00 * m11 + m00 * m22 + m11 * m22;
    const double d = m11 * m02 * m02 + m00 * m12 * m12 + m22 * m01 * m01 - m00 * m11 * m22 - 2 * m01 * m12 * m02;
    const double s = 2 * b * b * b - 9 * b * c + 27 * d;
    const double q = b * b - 3 * c;
    const double t = 4 * q * q * q;
    const double alpha = std::acos(s / std::sqrt(t));
    const double beta = alpha / 3, y = std::cos(beta);
    const double r = 0.5 * std::sqrt(t), w = std::pow(r, 1.0 / 3.0);
    for (int k = 0; k < 3; ++k) {
        const double *J = dM[k];
        const double j00 = J[0], j01 = J[1], j02 = J[2], j11 = J[4], j12 = J[5], j22 = J[8];
        const double bj = -j00 - j11 - j22;
        const double cj = -2.0 * m02 * j02 - 2.0 * m12 * j12 - 2.0 * m01 * j01 + j00 * m11 + m00 * j11 + j00 * m22 + m00 * j22 + j11 * m22 + m11 * j22;
        const double dj = j11 * m02 * m02 + m11 * 2 * m02 * j02 + j00 * m12 * m12 + m00 * 2.0 * m12 * j12 + j22 * m01 * m01 + m22 * 2.0 * m01 * j01 -
                          j00 * m11 * m22 - m00 * j11 * m22 - m00 * m11 * j22 - 2.0 * (j01 * m12 * m02 + m01 * j12 * m02 + m01 * m12 * j02);
        const double sj = 2.0 * 3.0 * b * b * bj - 9.0 * bj * c - 9.0 * b * cj + 27.0 * dj;
        const double tj = 4.0 * 3.0 * q * q * (2.0 * b * bj - 3.0 * cj);
        const double alpha_j = -1.0 / std::sqrt(1.0 - (s * s / t)) * (sj * std::sqrt(t) - s * 0.5 * std::pow(t, -0.5) * tj) / t;
        const double beta_j = alpha_j / 3.0;
        const double yj = -std::sin(beta) * beta_j;
        const double rj = 0.25 * std::pow(t, -0.5) * tj;
        const double wj = (1.0 / 3.0) * std::pow(r, -2.0 / 3.0) * rj;
        const double kj = wj * y + w * yj;
        grad[k] = (-bj - 2.0 * kj) / 3.0;
    }
}

inline double norm3(const double *v) {
    const double m = std::max(std::fabs(v[0]), std::max(std::fabs(v[1]), std::fabs(v[2])));
    if (!(m > 0) || !std::isfinite(m)) return std::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const double a = v[0] / m, b = v[1] / m, c = v[2] / m;
    return m * std::sqrt(a * a + b * b + c * c);
}

// MINPACK's qrsolv for n = 3: given the upper triangle of r (column-pivoted QR of the Jacobian), the permutation, a diagonal d and
// Q^T b, solves the least-squares system [J; D] x = [b; 0]; sdiag receives the diagonal of the triangular factor S, the strict lower
// triangle of s its off-diagonal part (transposed).
inline void lm_qrsolv(double s[3][3], const int *ipvt, const double *diag, const double *qtb, double *x, double *sdiag) {
    const int n = 3;
    double wa[3], keep[3];
    for (int j = 0; j < n; ++j) {
        for (int i = j; i < n; ++i) s[i][j] = s[j][i];
        keep[j] = s[j][j];
        wa[j] = qtb[j];
    }
    for (int j = 0; j < n; ++j) {
        const int l = ipvt[j];
        if (diag[l] != 0.0) {
            for (int k = j; k < n; ++k) sdiag[k] = 0.0;
            sdiag[j] = diag[l];
            double qtbpj = 0.0;
            for (int k = j; k < n; ++k) {
                if (sdiag[k] == 0.0) continue;
                double co, si;
                if (std::fabs(s[k][k]) < std::fabs(sdiag[k])) {
                    const double cot = s[k][k] / sdiag[k];
                    si = 0.5 / std::sqrt(0.25 + 0.25 * (cot * cot));
                    co = si * cot;
                } else {
                    const double tn = sdiag[k] / s[k][k];
                    co = 0.5 / std::sqrt(0.25 + 0.25 * (tn * tn));
                    si = co * tn;
                }
                s[k][k] = co * s[k][k] + si * sdiag[k];
                const double temp = co * wa[k] + si * qtbpj;
                qtbpj = -si * wa[k] + co * qtbpj;
                wa[k] = temp;
                for (int i = k + 1; i < n; ++i) {
                    const double t2 = co * s[i][k] + si * sdiag[i];
                    sdiag[i] = -si * s[i][k] + co * sdiag[i];
                    s[i][k] = t2;
                }
            }
        }
        sdiag[j] = s[j][j];
        s[j][j] = keep[j];
    }
    int nsing = n;
    for (int j = 0; j < n; ++j) {
        if (sdiag[j] == 0.0 && nsing == n) nsing = j;
        if (nsing < n) wa[j] = 0.0;
    }
    for (int k = nsing - 1; k >= 0; --k) {
        double sum = 0.0;
        for (int i = k + 1; i < nsing; ++i) sum += s[i][k] * wa[i];
        wa[k] = (wa[k] - sum) / sdiag[k];
    }
    for (int j = 0; j < n; ++j) x[ipvt[j]] = wa[j];
}

// MINPACK's lmpar for n = 3 (the Levenberg-Marquardt parameter for the step bound delta)
inline void lm_par(const double r[3][3], int rank, const int *ipvt, const double *diag, const double *qtb, double delta, double &par, double *x) {
    const int n = 3;
    const double dwarf = DBL_MIN;
    double wa1[3], wa2[3], sdiag[3];
    for (int j = 0; j < n; ++j) wa1[j] = j < rank ? qtb[j] : 0.0;
    for (int k = rank - 1; k >= 0; --k) {  // Gauss-Newton direction: back substitution on the leading rank x rank triangle
        double sum = 0;
        for (int i = k + 1; i < rank; ++i) sum += r[k][i] * wa1[i];
        wa1[k] = (wa1[k] - sum) / r[k][k];
    }
    for (int j = 0; j < n; ++j) x[ipvt[j]] = wa1[j];
    int iter = 0;
    for (int j = 0; j < n; ++j) wa2[j] = diag[j] * x[j];
    double dxnorm = norm3(wa2);
    double fp = dxnorm - delta;
    if (fp <= 0.1 * delta) {
        par = 0;
        return;
    }
    double parl = 0.0;
    if (rank == n) {
        for (int j = 0; j < n; ++j) {
            const int l = ipvt[j];
            wa1[j] = diag[l] * (wa2[l] / dxnorm);
        }
        for (int j = 0; j < n; ++j) {  // solve R^T z = wa1
            double sum = 0;
            for (int i = 0; i < j; ++i) sum += r[i][j] * wa1[i];
            wa1[j] = (wa1[j] - sum) / r[j][j];
        }
        const double temp = norm3(wa1);
        parl = fp / delta / temp / temp;
    }
    for (int j = 0; j < n; ++j) {
        double sum = 0;
        for (int i = 0; i <= j; ++i) sum += r[i][j] * qtb[i];
        wa1[j] = sum / diag[ipvt[j]];
    }
    const double gnorm = norm3(wa1);
    double paru = gnorm / delta;
    if (paru == 0.0) paru = dwarf / std::min(delta, 0.1);
    par = std::max(par, parl);
    par = std::min(par, paru);
    if (par == 0.0) par = gnorm / dxnorm;
    double s[3][3];
    for (;;) {
        ++iter;
        if (par == 0.0) par = std::max(dwarf, 0.001 * paru);
        const double sq = std::sqrt(par);
        for (int j = 0; j < n; ++j) wa1[j] = sq * diag[j];
        std::memcpy(s, r, sizeof(s));
        lm_qrsolv(s, ipvt, wa1, qtb, x, sdiag);
        for (int j = 0; j < n; ++j) wa2[j] = diag[j] * x[j];
        dxnorm = norm3(wa2);
        double temp = fp;
        fp = dxnorm - delta;
        if (std::fabs(fp) <= 0.1 * delta || (parl == 0.0 && fp <= temp && temp < 0.0) || iter == 10) break;
        for (int j = 0; j < n; ++j) {
            const int l = ipvt[j];
            wa1[j] = diag[l] * (wa2[l] / dxnorm);
        }
        for (int j = 0; j < n; ++j) {
            wa1[j] /= sdiag[j];
            temp = wa1[j];
            for (int i = j + 1; i < n; ++i) wa1[i] -= s[i][j] * temp;
        }
        temp = norm3(wa1);
        const double parc = fp / delta / temp / temp;
        if (fp > 0.0) parl = std::max(parl, par);
        if (fp < 0.0) paru = std::min(paru, par);
        par = std::max(parl, par + parc);
    }
    if (iter == 0) par = 0.0;
}

// Householder QR with column pivoting of a 3 x 3 matrix (a[row][col]); on return a holds R in its upper triangle, qtb = Q^T b.
inline void lm_qr(double a[3][3], int *ipvt, double *colnorm, double *qtb, int *rank) {
    const int n = 3;
    double rem[3];
    for (int j = 0; j < n; ++j) {
        const double col[3] = {a[0][j], a[1][j], a[2][j]};
        colnorm[j] = norm3(col);
        rem[j] = colnorm[j];
        ipvt[j] = j;
    }
    double maxpivot = 0;
    for (int k = 0; k < n; ++k) {
        int big = k;
        for (int j = k + 1; j < n; ++j)
            if (rem[j] > rem[big]) big = j;
        if (big != k) {
            for (int i = 0; i < n; ++i) std::swap(a[i][k], a[i][big]);
            std::swap(rem[k], rem[big]);
            std::swap(ipvt[k], ipvt[big]);
        }
        double nrm = 0;
        for (int i = k; i < n; ++i) nrm += a[i][k] * a[i][k];
        nrm = std::sqrt(nrm);
        if (nrm == 0.0) continue;
        const double alpha = a[k][k] >= 0 ? -nrm : nrm;  // R_kk
        double v[3] = {0, 0, 0};
        for (int i = k; i < n; ++i) v[i] = a[i][k];
        v[k] -= alpha;
        double vv = 0;
        for (int i = k; i < n; ++i) vv += v[i] * v[i];
        if (vv > 0) {
            for (int j = k; j < n; ++j) {
                double dt = 0;
                for (int i = k; i < n; ++i) dt += v[i] * a[i][j];
                const double f = 2.0 * dt / vv;
                for (int i = k; i < n; ++i) a[i][j] -= f * v[i];
            }
            double dt = 0;
            for (int i = k; i < n; ++i) dt += v[i] * qtb[i];
            const double f = 2.0 * dt / vv;
            for (int i = k; i < n; ++i) qtb[i] -= f * v[i];
        }
        a[k][k] = alpha;
        for (int i = k + 1; i < n; ++i) a[i][k] = 0.0;
        maxpivot = std::max(maxpivot, std::fabs(alpha));
        for (int j = k + 1; j < n; ++j) {  // remaining column norms
            double rr = 0;
            for (int i = k + 1; i < n; ++i) rr += a[i][j] * a[i][j];
            rem[j] = std::sqrt(rr);
        }
    }
    int rk = 0;
    for (int k = 0; k < n; ++k)
        if (std::fabs(a[k][k]) > maxpivot * (DBL_EPSILON * 3)) ++rk;  // ColPivHouseholderQR::rank() with its default threshold
    *rank = rk;
}

// Levenberg-Marquardt on g(x) = grad lambda_min(M(x)), three unknowns, forward-difference Jacobian (Eigen's LevenbergMarquardt over
// NumericalDiff, i.e. MINPACK lmdif with the Jacobian's own evaluation of g(x) counted: ftol 5e-5, xtol 10 eps, gtol 0, factor 100,
// at most 100 evaluations)
inline void lm_minimise_gradient(const EigSums &Sx, double *x) {
    const int n = 3;
    const double ftol = 0.00005, xtol = 10.0 * DBL_EPSILON, gtol = 0.0, factor = 100.0;
    const int maxfev = 100;
    double fvec[3], diag[3] = {0, 0, 0}, qtf[3], wa1[3], wa2[3], wa3[3], wa4[3];
    double fjac[3][3];
    int ipvt[3];
    smallest_ev_gradient(Sx, x, fvec);
    int nfev = 1, iter = 1;
    double fnorm = norm3(fvec), par = 0.0, delta = 0.0, xnorm = 0.0;
#ifdef DGM_TRACE
    std::printf("ME  init x %.15g %.15g %.15g fnorm %.15g\n", x[0], x[1], x[2], fnorm);
#endif
    for (;;) {
        {  // forward differences (the evaluation at x itself is repeated and counted, as NumericalDiff does)
            const double eps = std::sqrt(DBL_EPSILON);
            double v1[3], v2[3], xx[3] = {x[0], x[1], x[2]};
            smallest_ev_gradient(Sx, xx, v1);
            ++nfev;
            for (int j = 0; j < n; ++j) {
                double h = eps * std::fabs(xx[j]);
                if (h == 0.0) h = eps;
                xx[j] += h;
                smallest_ev_gradient(Sx, xx, v2);
                ++nfev;
                xx[j] = x[j];
                for (int i = 0; i < n; ++i) fjac[i][j] = (v2[i] - v1[i]) / h;
            }
        }
        int rank;
        for (int i = 0; i < n; ++i) wa4[i] = fvec[i];
        lm_qr(fjac, ipvt, wa2, wa4, &rank);
        if (iter == 1) {
            for (int j = 0; j < n; ++j) diag[j] = (wa2[j] == 0.0) ? 1.0 : wa2[j];
            for (int j = 0; j < n; ++j) wa3[j] = diag[j] * x[j];
            xnorm = norm3(wa3);
            delta = factor * xnorm;
            if (delta == 0.0) delta = factor;
        }
        for (int i = 0; i < n; ++i) qtf[i] = wa4[i];
        double gnorm = 0.0;
        if (fnorm != 0.0)
            for (int j = 0; j < n; ++j)
                if (wa2[ipvt[j]] != 0.0) {
                    double sum = 0;
                    for (int i = 0; i <= j; ++i) sum += fjac[i][j] * (qtf[i] / fnorm);
                    gnorm = std::max(gnorm, std::fabs(sum / wa2[ipvt[j]]));
                }
        if (gnorm <= gtol) return;
        for (int j = 0; j < n; ++j) diag[j] = std::max(diag[j], wa2[j]);
        double ratio = 0.0;
        do {
            lm_par(fjac, rank, ipvt, diag, qtf, delta, par, wa1);
            for (int j = 0; j < n; ++j) {
                wa1[j] = -wa1[j];
                wa2[j] = x[j] + wa1[j];
                wa3[j] = diag[j] * wa1[j];
            }
            const double pnorm = norm3(wa3);
            if (iter == 1) delta = std::min(delta, pnorm);
            smallest_ev_gradient(Sx, wa2, wa4);
            ++nfev;
            const double fnorm1 = norm3(wa4);
            double actred = -1.0;
            if (0.1 * fnorm1 < fnorm) actred = 1.0 - (fnorm1 / fnorm) * (fnorm1 / fnorm);
            for (int i = 0; i < n; ++i) {  // R P^T p
                double sum = 0;
                for (int j = i; j < n; ++j) sum += fjac[i][j] * wa1[ipvt[j]];
                wa3[i] = sum;
            }
            const double temp1 = (norm3(wa3) / fnorm) * (norm3(wa3) / fnorm);
            const double temp2 = (std::sqrt(par) * pnorm / fnorm) * (std::sqrt(par) * pnorm / fnorm);
            const double prered = temp1 + temp2 / 0.5;
            const double dirder = -(temp1 + temp2);
            ratio = 0.0;
            if (prered != 0.0) ratio = actred / prered;
            if (ratio <= 0.25) {
                double temp = 0.5;
                if (actred < 0.0) temp = 0.5 * dirder / (dirder + 0.5 * actred);
                if (0.1 * fnorm1 >= fnorm || temp < 0.1) temp = 0.1;
                delta = temp * std::min(delta, pnorm / 0.1);
                par /= temp;
            } else if (!(par != 0.0 && ratio < 0.75)) {
                delta = pnorm / 0.5;
                par = 0.5 * par;
            }
#ifdef DGM_TRACE
            std::printf("ME  inner: iter %d nfev %d ratio %.6g actred %.6g prered %.6g pnorm %.6g delta %.6g par %.6g fnorm1 %.15g cand %.15g %.15g %.15g\n", iter, nfev, ratio, actred, prered, pnorm, delta, par, fnorm1, wa2[0], wa2[1], wa2[2]);
#endif
            if (ratio >= 1e-4) {
                for (int j = 0; j < n; ++j) {
                    x[j] = wa2[j];
                    wa2[j] = diag[j] * x[j];
                    fvec[j] = wa4[j];
                }
                xnorm = norm3(wa2);
                fnorm = fnorm1;
                ++iter;
            }
            if (std::fabs(actred) <= ftol && prered <= ftol && 0.5 * ratio <= 1.0) return;
            if (delta <= xtol * xnorm) return;
            if (nfev >= maxfev) return;
            if (std::fabs(actred) <= DBL_EPSILON && prered <= DBL_EPSILON && 0.5 * ratio <= 1.0) return;
            if (delta <= DBL_EPSILON * xnorm) return;
            if (gnorm <= DBL_EPSILON) return;
        } while (ratio < 1e-4);
    }
}

inline void rot_to_cayley(const double *R, double *c) {  // C = (R - I)(R + I)^-1; (-C12, C02, -C01)
    double A[9], B[9], Bi[9];
    for (int k = 0; k < 9; ++k) A[k] = R[k] - (k % 4 == 0 ? 1.0 : 0.0), B[k] = R[k] + (k % 4 == 0 ? 1.0 : 0.0);
    const double det = B[0] * (B[4] * B[8] - B[5] * B[7]) - B[1] * (B[3] * B[8] - B[5] * B[6]) + B[2] * (B[3] * B[7] - B[4] * B[6]);
    const double id = 1.0 / det;
    Bi[0] = (B[4] * B[8] - B[5] * B[7]) * id, Bi[1] = (B[2] * B[7] - B[1] * B[8]) * id, Bi[2] = (B[1] * B[5] - B[2] * B[4]) * id;
    Bi[3] = (B[5] * B[6] - B[3] * B[8]) * id, Bi[4] = (B[0] * B[8] - B[2] * B[6]) * id, Bi[5] = (B[2] * B[3] - B[0] * B[5]) * id;
    Bi[6] = (B[3] * B[7] - B[4] * B[6]) * id, Bi[7] = (B[1] * B[6] - B[0] * B[7]) * id, Bi[8] = (B[0] * B[4] - B[1] * B[3]) * id;
    auto C = [&](int r, int k) { return A[3 * r] * Bi[k] + A[3 * r + 1] * Bi[3 + k] + A[3 * r + 2] * Bi[6 + k]; };
    c[0] = -C(1, 2), c[1] = C(0, 2), c[2] = -C(0, 1);
}

// Eigen-decomposition of a symmetric 3 x 3 matrix by Jacobi rotations: ev ascending, vec[k] the eigenvector of ev[k]
inline void sym_eig3(const double *Min, double *ev, double (*vec)[3]) {
    double A[3][3], V[3][3] = {{1, 0, 0}, {0, 1, 0}, {0, 0, 1}};
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) A[r][c] = Min[3 * r + c];
    for (int sweep = 0; sweep < 60; ++sweep) {
        const double off = std::fabs(A[0][1]) + std::fabs(A[0][2]) + std::fabs(A[1][2]);
        if (off == 0.0) break;
        bool rotated = false;
        for (int p = 0; p < 2; ++p)
            for (int q = p + 1; q < 3; ++q) {
                if (std::fabs(A[p][q]) <= 1e-18 * (std::fabs(A[p][p]) + std::fabs(A[q][q]))) {
                    A[p][q] = A[q][p] = 0.0;
                    continue;
                }
                rotated = true;
                const double theta = (A[q][q] - A[p][p]) / (2.0 * A[p][q]);
                const double t = (theta >= 0 ? 1.0 : -1.0) / (std::fabs(theta) + std::sqrt(theta * theta + 1.0));
                const double c = 1.0 / std::sqrt(t * t + 1.0), s = t * c;
                for (int k = 0; k < 3; ++k) {
                    const double akp = A[k][p], akq = A[k][q];
                    A[k][p] = c * akp - s * akq, A[k][q] = s * akp + c * akq;
                }
                for (int k = 0; k < 3; ++k) {
                    const double apk = A[p][k], aqk = A[q][k];
                    A[p][k] = c * apk - s * aqk, A[q][k] = s * apk + c * aqk;
                }
                for (int k = 0; k < 3; ++k) {
                    const double vkp = V[k][p], vkq = V[k][q];
                    V[k][p] = c * vkp - s * vkq, V[k][q] = s * vkp + c * vkq;
                }
            }
        if (!rotated) break;
    }
    int order[3] = {0, 1, 2};
    std::stable_sort(order, order + 3, [&](int a, int b) { return A[a][a] < A[b][b]; });
    for (int k = 0; k < 3; ++k) {
        ev[k] = A[order[k]][order[k]];
        for (int i = 0; i < 3; ++i) vec[k][i] = V[i][order[k]];
    }
}

// ---- the order of Eigen::EigenSolver's eigenvalues on a 3 x 3 matrix ----------------------------------------------------------------
// OpenGV takes the translation from COLUMN 0 of Eigen::EigenSolver's eigenvectors and its length from eigenvalues 1 and 2
// (modules/main.cpp:646-659).  EigenSolver does not order its eigenvalues: on these symmetric matrices the smallest one -- the
// eigenvector that would be the translation -- sits at position 0 in about a third of the cases.  Which position an eigenvalue gets is
// a property of Eigen::RealSchur's iteration (Householder reduction to Hessenberg form, EISPACK hqr: Francis double-shift QR steps with
// deflation from the bottom, Wilkinson / MATLAB exceptional shifts after 10 / 30 iterations, 2 x 2 blocks with real eigenvalues split by
// a rotation), restated here for three rows.  Once the iteration has converged, entries at rounding-noise level decide what the next
// reflector does, so the reflectors are formed and applied in Eigen's formulation and operation order; on identical input the diagonal
// comes out in Eigen's order (probe against the Eigen the reference vendors: 60 000 of 60 000 matrices, values to 7e-16).
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
inline void eigen_diag_order3(const double *M, double *d) {
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

// opengv::relative_pose::eigensolver on n correspondences (f1 = adapter view 1, f2 = view 2), starting from R_init.  t is not normalised:
// its length is what OpenGV returns (the root sum of squares of the eigenvalues at positions 1 and 2).
inline void eigensolver(const double (*f1)[3], const double (*f2)[3], int n, const double *R_init, double *R, double *t) {
    EigSums Sx;
    eig_sums(f1, f2, n, Sx);
    double x[3];
    rot_to_cayley(R_init, x);
    lm_minimise_gradient(Sx, x);
    double Rr[9], M[9];
    cayley_reduced(x, Rr);
    const double scale = 1 + x[0] * x[0] + x[1] * x[1] + x[2] * x[2];
    for (int k = 0; k < 9; ++k) R[k] = (1 / scale) * Rr[k];
    compose_M(Sx, Rr, nullptr, M);
    double ev[3], vec[3][3], d[3];
    sym_eig3(M, ev, vec);          // accurate eigenpairs ...
    eigen_diag_order3(M, d);       // ... and the position Eigen::EigenSolver gives each eigenvalue
    int k0 = 0;
    for (int k = 1; k < 3; ++k)
        if (std::fabs(ev[k] - d[0]) < std::fabs(ev[k0] - d[0])) k0 = k;
    const double mag = std::sqrt(d[1] * d[1] + d[2] * d[2]);
    for (int k = 0; k < 3; ++k) t[k] = mag * vec[k0][k];
    double f2r[3], flow[3];
    matvec(R, f2[0], f2r);
    for (int k = 0; k < 3; ++k) flow[k] = f1[0][k] - f2r[k];
    if (flow[0] * t[0] + flow[1] * t[1] + flow[2] * t[2] < 0.0)
        for (int k = 0; k < 3; ++k) t[k] = -t[k];
}

}  // namespace dgm
