/*
 * pose_oracle.c -- CPU restatement of the reference's robust essential-matrix path.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).  Plain C, single thread, no dependencies beyond libm.
 *
 * Follows, under /root/reference/matchinglib_poselib/source/poselib/ :
 *   source/five-point-nister/modelest.cpp:52-66    srand in the estimator ctor / setSeed
 *   source/five-point-nister/modelest.cpp:69-83    findInliers
 *   source/five-point-nister/modelest.cpp:86-109   cvRANSACUpdateNumIters1
 *   source/five-point-nister/modelest.cpp:343-474  runRANSAC (incl. the `lesqu` refit :420-464)
 *   source/five-point-nister/modelest.cpp:567-650  getSubset / checkSubset
 *   source/five-point-nister/five-point.cpp:366-471 run5Point
 *   source/five-point-nister/five-point.cpp:476-503 computeReprojError3 (Sampson, fp64 -> float)
 *   source/five-point-nister/five-point.cpp:603-824 getCoeffMat (the 10x20 constraint matrix)
 *   source/five-point-nister/five-point.cpp:150-352 recoverPose / decomposeEssentialMat
 *   source/pose_estim.cpp:857-946                   estimateEssentialMat / getPoseTriangPts wrappers
 *
 * What is restated from a dependency that is NOT vendored in the reference (OpenCV 4.2.0, ci/make_opencv.sh:6) and
 * is therefore "parity unpinned" at that boundary (the reference holds no vectors for it):
 *   cv::SVD::compute  -> one-sided (Hestenes) Jacobi SVD, the algorithm OpenCV's JacobiSVDImpl_ uses.  For the 5x9
 *                        minimal case any orthonormal basis of the 4-dim null space gives the same set of E; the
 *                        reference's own basis depends on OpenCV's pseudo-random completion of V.
 *   cv::Mat::inv()*   -> Gauss elimination with partial pivoting (DECOMP_LU) solving A1 X = A2.
 *   cv::solvePoly     -> Durand-Kerner (Weierstrass) iteration, start values (1+i)^k, <= 1000 sweeps, Gauss-Seidel
 *                        order, stop when the largest correction is exactly 0 (restated from OpenCV's mathfuncs.cpp;
 *                        its special branch for exactly coinciding estimates is NOT restated).
 *   cv::SVD::solveZ   -> right singular vector of the smallest singular value.
 *   cv::triangulatePoints -> per point the 4x4 DLT system  [x*P(2,:) - P(0,:); y*P(2,:) - P(1,:)] (both views),
 *                        solution = right singular vector of the smallest singular value.
 *   cv::sum(err)      -> double accumulation of the float errors.  OpenCV's sum kernel for CV_32F widens to double and keeps
 *                        SIMD-lane accumulators; restated here as the SSE2 form: four interleaved double accumulators
 *                        (element i goes to accumulator i mod 4), combined as (s0 + s2) + (s1 + s3).  The exact lane
 *                        count/tail handling of the reference's OpenCV build is not knowable (it only matters for the
 *                        equal-inlier-count tie-break, modelest.cpp:408-414).
 *
 * The constraint matrix of five-point.cpp:603-824 is 200 machine-generated expressions; it is NOT copied here.
 * It is re-derived from its definition: with E = x*E0 + y*E1 + z*E2 + E3 the ten cubic constraints
 * det(E) = 0 and E E^T E - 1/2 trace(E E^T) E = 0 are expanded by polynomial arithmetic over the 20 monomials,
 * in the reference's column order (after its perm[] step, five-point.cpp:813-823):
 *   x^3 y^3 x^2y xy^2 x^2z x^2 y^2z y^2 xyz xy | xz^2 xz x yz^2 yz y z^3 z^2 z 1
 * Row order and row scaling do not matter because only inv(A[:, :10]) * A[:, 10:] is used (five-point.cpp:394).
 * Likewise c[0..10] (five-point.cpp:416-428) is the determinant of the 3x3 polynomial matrix B(z), computed here by
 * polynomial multiplication rather than from the expanded formula.
 *
 * Reference quirk kept on purpose: checkSubset() returns `i >= i1` with i0 = i1 = count-1 (modelest.cpp:622-649),
 * so it is true for every input -- the collinearity test never rejects a sample.  The rand() stream consumption is
 * therefore: one draw per pick, duplicates redrawn.
 */
#include "oracle.h"

#include <float.h>
#include <math.h>
#include <stdlib.h>
#include <string.h>

/* ------------------------------------------------------------------------------------------------------------
 * glibc srand()/rand(): TYPE_3 additive feedback generator (degree 31, separation 3), restated.
 * ---------------------------------------------------------------------------------------------------------- */
void oracle_srand(oracle_glibc_rand *st, unsigned seed) {
    int32_t *r = st->r;
    if (seed == 0) seed = 1;
    r[0] = (int32_t)seed;
    for (int i = 1; i < 31; ++i) {
        /* r[i] = (16807 * r[i-1]) % 2147483647 without overflow (Schrage) */
        long hi = r[i - 1] / 127773;
        long lo = r[i - 1] % 127773;
        long word = 16807 * lo - 2836 * hi;
        if (word < 0) word += 2147483647;
        r[i] = (int32_t)word;
    }
    st->f = 3; /* front = r[sep], rear = r[0] over the 31-word state */
    st->b = 0;
    for (int i = 0; i < 310; ++i) (void)oracle_rand(st);
}

int oracle_rand(oracle_glibc_rand *st) {
    uint32_t *r = (uint32_t *)st->r;
    r[st->f] += r[st->b];
    const uint32_t result = r[st->f] >> 1;
    if (++st->f >= 31) st->f = 0;
    if (++st->b >= 31) st->b = 0;
    return (int)result;
}

/* ------------------------------------------------------------------------------------------------------------
 * One-sided Jacobi SVD.  A: m x n row-major.  w[n] descending, V: n x n row-major (columns = right singular
 * vectors), optionally AV = A*V (m x n, columns = sigma_j * u_j) when AV != NULL.
 * ---------------------------------------------------------------------------------------------------------- */
#define SVD_MAX_M 16384
static void jacobi_svd_impl(const double *A, int m, int n, double *w, double *V, double *AV_out) {
    double *G = (double *)malloc(sizeof(double) * (size_t)m * n);
    memcpy(G, A, sizeof(double) * (size_t)m * n);
    for (int i = 0; i < n; ++i)
        for (int j = 0; j < n; ++j) V[i * n + j] = (i == j) ? 1.0 : 0.0;
    const double eps = DBL_EPSILON * 2;
    for (int sweep = 0; sweep < 60; ++sweep) {
        int rotated = 0;
        for (int p = 0; p < n - 1; ++p) {
            for (int q = p + 1; q < n; ++q) {
                double alpha = 0, beta = 0, gamma = 0;
                for (int i = 0; i < m; ++i) {
                    const double gp = G[i * n + p], gq = G[i * n + q];
                    alpha += gp * gp;
                    beta += gq * gq;
                    gamma += gp * gq;
                }
                if (fabs(gamma) <= eps * sqrt(alpha * beta) || gamma == 0.0) continue;
                rotated = 1;
                const double zeta = (beta - alpha) / (2.0 * gamma);
                const double t = (zeta >= 0 ? 1.0 : -1.0) / (fabs(zeta) + sqrt(1.0 + zeta * zeta));
                const double c = 1.0 / sqrt(1.0 + t * t), s = c * t;
                for (int i = 0; i < m; ++i) {
                    const double gp = G[i * n + p], gq = G[i * n + q];
                    G[i * n + p] = c * gp - s * gq;
                    G[i * n + q] = s * gp + c * gq;
                }
                for (int i = 0; i < n; ++i) {
                    const double vp = V[i * n + p], vq = V[i * n + q];
                    V[i * n + p] = c * vp - s * vq;
                    V[i * n + q] = s * vp + c * vq;
                }
            }
        }
        if (!rotated) break;
    }
    for (int j = 0; j < n; ++j) {
        double s = 0;
        for (int i = 0; i < m; ++i) s += G[i * n + j] * G[i * n + j];
        w[j] = sqrt(s);
    }
    /* sort descending (selection sort, stable on ties), permuting the columns of V and G */
    for (int a = 0; a < n - 1; ++a) {
        int best = a;
        for (int b = a + 1; b < n; ++b)
            if (w[b] > w[best]) best = b;
        if (best != a) {
            double tw = w[a];
            w[a] = w[best];
            w[best] = tw;
            for (int i = 0; i < n; ++i) {
                double tv = V[i * n + a];
                V[i * n + a] = V[i * n + best];
                V[i * n + best] = tv;
            }
            for (int i = 0; i < m; ++i) {
                double tg = G[i * n + a];
                G[i * n + a] = G[i * n + best];
                G[i * n + best] = tg;
            }
        }
    }
    if (AV_out) memcpy(AV_out, G, sizeof(double) * (size_t)m * n);
    free(G);
}

void oracle_jacobi_svd(const double *A, int m, int n, double *w, double *V) { jacobi_svd_impl(A, m, n, w, V, NULL); }

/* ------------------------------------------------------------------------------------------------------------
 * cv::solvePoly restated (Durand-Kerner).  coeffs ascending.  Returns number of roots written.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    double re, im;
} cplx;
static cplx cmul(cplx a, cplx b) {
    cplx r = {a.re * b.re - a.im * b.im, a.re * b.im + a.im * b.re};
    return r;
}
static cplx cadd(cplx a, cplx b) {
    cplx r = {a.re + b.re, a.im + b.im};
    return r;
}
static cplx csub(cplx a, cplx b) {
    cplx r = {a.re - b.re, a.im - b.im};
    return r;
}
static cplx cdiv(cplx a, cplx b) {
    const double t = 1. / (b.re * b.re + b.im * b.im);
    cplx r = {(a.re * b.re + a.im * b.im) * t, (-a.re * b.im + a.im * b.re) * t};
    return r;
}

int oracle_solve_poly(const double *coeffs_in, int deg, double *roots_out, int max_iters) {
    cplx coeffs[32], roots[32];
    int n = deg;
    if (deg > 30) return -1;
    for (int i = 0; i <= n; ++i) {
        coeffs[i].re = coeffs_in[i];
        coeffs[i].im = 0;
    }
    for (; n > 1; n--)
        if (fabs(coeffs[n].re) + fabs(coeffs[n].im) > DBL_EPSILON) break;
    cplx p = {1, 0}, r = {1, 1};
    for (int i = 0; i < n; ++i) {
        roots[i] = p;
        p = cmul(p, r);
    }
    if (max_iters <= 0) max_iters = 1000;
    for (int iter = 0; iter < max_iters; ++iter) {
        double maxDiff = 0;
        for (int i = 0; i < n; ++i) {
            p = roots[i];
            cplx num = coeffs[n], denom = coeffs[n];
            for (int j = 0; j < n; ++j) {
                num = cadd(cmul(num, p), coeffs[n - j - 1]);
                if (j != i) {
                    const cplx d = csub(p, roots[j]);
                    if (d.re != 0 || d.im != 0) denom = cmul(denom, d);
                }
            }
            num = cdiv(num, denom);
            roots[i] = csub(p, num);
            const double a = sqrt(num.re * num.re + num.im * num.im);
            if (a > maxDiff) maxDiff = a;
        }
        if (maxDiff <= 0) break;
    }
    for (int i = 0; i < n; ++i) {
        if (fabs(roots[i].im) < 1e-100) roots[i].im = 0;
        roots_out[2 * i] = roots[i].re;
        roots_out[2 * i + 1] = roots[i].im;
    }
    /* OpenCV pads the trimmed leading roots by repeating the last one (roots[n+1] = roots[n]) */
    int nn = n;
    for (; nn < deg; ++nn) {
        roots_out[2 * nn] = roots_out[2 * (nn - 1)];
        roots_out[2 * nn + 1] = roots_out[2 * (nn - 1) + 1];
    }
    return deg;
}

/* ------------------------------------------------------------------------------------------------------------
 * Trivariate polynomials of total degree <= 3 in (x, y, z), coefficient c[i][j][k] of x^i y^j z^k.
 * ---------------------------------------------------------------------------------------------------------- */
typedef struct {
    double c[4][4][4];
} poly3;

static void p_zero(poly3 *p) { memset(p, 0, sizeof(*p)); }
static void p_linear(poly3 *p, double cx, double cy, double cz, double c1) {
    p_zero(p);
    p->c[1][0][0] = cx;
    p->c[0][1][0] = cy;
    p->c[0][0][1] = cz;
    p->c[0][0][0] = c1;
}
static void p_addmul(poly3 *acc, const poly3 *a, const poly3 *b, double scale) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j + i < 4; ++j)
            for (int k = 0; k + j + i < 4; ++k) {
                const double av = a->c[i][j][k];
                if (av == 0.0) continue;
                for (int l = 0; l + i < 4; ++l)
                    for (int m = 0; m + j < 4; ++m)
                        for (int n = 0; n + k < 4; ++n) {
                            if (i + j + k + l + m + n > 3) continue;
                            acc->c[i + l][j + m][k + n] += scale * av * b->c[l][m][n];
                        }
            }
}
static void p_axpy(poly3 *acc, const poly3 *a, double scale) {
    for (int i = 0; i < 4; ++i)
        for (int j = 0; j < 4; ++j)
            for (int k = 0; k < 4; ++k) acc->c[i][j][k] += scale * a->c[i][j][k];
}

/* column order of the reference's coefficient matrix (after perm[]) */
static const int kMono[20][3] = {{3, 0, 0}, {0, 3, 0}, {2, 1, 0}, {1, 2, 0}, {2, 0, 1}, {2, 0, 0}, {0, 2, 1},
                                 {0, 2, 0}, {1, 1, 1}, {1, 1, 0}, {1, 0, 2}, {1, 0, 1}, {1, 0, 0}, {0, 1, 2},
                                 {0, 1, 1}, {0, 1, 0}, {0, 0, 3}, {0, 0, 2}, {0, 0, 1}, {0, 0, 0}};

/* EE: 4 basis matrices, EE[b*9 + r*3 + c].  A: 10 x 20 row-major. */
static void coeff_matrix(const double *EE, double *A) {
    poly3 E[3][3];
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) p_linear(&E[r][c], EE[0 * 9 + r * 3 + c], EE[1 * 9 + r * 3 + c], EE[2 * 9 + r * 3 + c],
                                            EE[3 * 9 + r * 3 + c]);
    poly3 rows[10];
    /* row 0: det(E) */
    {
        poly3 m01, m02, m12, det;
        /* 2x2 minors of rows 1,2 */
        p_zero(&m12);
        p_addmul(&m12, &E[1][1], &E[2][2], 1.0);
        p_addmul(&m12, &E[1][2], &E[2][1], -1.0);
        p_zero(&m02);
        p_addmul(&m02, &E[1][0], &E[2][2], 1.0);
        p_addmul(&m02, &E[1][2], &E[2][0], -1.0);
        p_zero(&m01);
        p_addmul(&m01, &E[1][0], &E[2][1], 1.0);
        p_addmul(&m01, &E[1][1], &E[2][0], -1.0);
        p_zero(&det);
        p_addmul(&det, &E[0][0], &m12, 1.0);
        p_addmul(&det, &E[0][1], &m02, -1.0);
        p_addmul(&det, &E[0][2], &m01, 1.0);
        rows[0] = det;
    }
    /* rows 1..9: E E^T E - 1/2 trace(E E^T) E */
    poly3 EEt[3][3], tr;
    p_zero(&tr);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            p_zero(&EEt[r][c]);
            for (int k = 0; k < 3; ++k) p_addmul(&EEt[r][c], &E[r][k], &E[c][k], 1.0);
        }
    for (int r = 0; r < 3; ++r) p_axpy(&tr, &EEt[r][r], 1.0);
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) {
            poly3 acc;
            p_zero(&acc);
            for (int k = 0; k < 3; ++k) p_addmul(&acc, &EEt[r][k], &E[k][c], 1.0);
            p_addmul(&acc, &tr, &E[r][c], -0.5);
            rows[1 + r * 3 + c] = acc;
        }
    for (int r = 0; r < 10; ++r)
        for (int m = 0; m < 20; ++m) A[r * 20 + m] = rows[r].c[kMono[m][0]][kMono[m][1]][kMono[m][2]];
}

/* Solves A1 X = A2 (A = [A1 | A2], 10 x 20) in place by Gauss-Jordan with partial pivoting; on return columns
 * 10..19 hold X = inv(A1) * A2.  Returns 0 if singular. */
static int reduce_10x20(double *A) {
    for (int col = 0; col < 10; ++col) {
        int piv = col;
        for (int r = col + 1; r < 10; ++r)
            if (fabs(A[r * 20 + col]) > fabs(A[piv * 20 + col])) piv = r;
        if (fabs(A[piv * 20 + col]) < DBL_EPSILON * 1e-3) return 0;
        if (piv != col)
            for (int j = 0; j < 20; ++j) {
                double t = A[col * 20 + j];
                A[col * 20 + j] = A[piv * 20 + j];
                A[piv * 20 + j] = t;
            }
        const double inv = 1.0 / A[col * 20 + col];
        for (int j = 0; j < 20; ++j) A[col * 20 + j] *= inv;
        for (int r = 0; r < 10; ++r) {
            if (r == col) continue;
            const double f = A[r * 20 + col];
            if (f == 0.0) continue;
            for (int j = 0; j < 20; ++j) A[r * 20 + j] -= f * A[col * 20 + j];
        }
    }
    return 1;
}

/* polynomial helpers in one variable (ascending coefficients) */
static void poly_mul(const double *a, int da, const double *b, int db, double *out) {
    for (int i = 0; i <= da + db; ++i) out[i] = 0;
    for (int i = 0; i <= da; ++i)
        for (int j = 0; j <= db; ++j) out[i + j] += a[i] * b[j];
}

/* c(z) = det B(z), degree 10, ascending coefficients (five-point.cpp:416-428 write the same determinant out term by term).
 * b: 3 x 13 row-major.  Exposed so that tests can pin it against the reference's own expressions (oracle/_ref/libfivept_ref.so). */
void oracle_detpoly(const double *b, double *c) {
    double P[3][3][5];
    int deg[3] = {3, 3, 4};
    for (int i = 0; i < 3; ++i) {
        for (int k = 0; k < 4; ++k) {
            P[i][0][k] = b[i * 13 + 3 - k];
            P[i][1][k] = b[i * 13 + 7 - k];
        }
        P[i][0][4] = P[i][1][4] = 0;
        for (int k = 0; k < 5; ++k) P[i][2][k] = b[i * 13 + 12 - k];
    }
    for (int k = 0; k < 11; ++k) c[k] = 0;
    static const int perms[6][3] = {{0, 1, 2}, {1, 2, 0}, {2, 0, 1}, {0, 2, 1}, {1, 0, 2}, {2, 1, 0}};
    static const double sgn[6] = {1, 1, 1, -1, -1, -1};
    for (int pi = 0; pi < 6; ++pi) {
        /* term = sgn * P[0][p0] * P[1][p1] * P[2][p2] */
        double t01[9], t012[13];
        const int p0 = perms[pi][0], p1 = perms[pi][1], p2 = perms[pi][2];
        poly_mul(P[0][p0], deg[p0], P[1][p1], deg[p1], t01);
        poly_mul(t01, deg[p0] + deg[p1], P[2][p2], deg[p2], t012);
        for (int k = 0; k <= 10; ++k) c[k] += sgn[pi] * t012[k];
    }
}

/* The 10 x 20 constraint matrix in the reference's column order (getCoeffMat, five-point.cpp:603-824), EE[b * 9 + k]. */
void oracle_coeff_matrix(const double *EE, double *A) { coeff_matrix(EE, A); }

static int run5point_rows(double *Q, int n, double *E_out, double *c_out, double *roots_out, double *xy1z_out);

static int run5point_impl(const double *q1, const double *q2, int n, double *E_out, double *c_out, double *roots_out,
                          double *xy1z_out) {
    if (n < 5) return 0;
    /* Q rows: [x1x2, y1x2, x2, x1y2, y1y2, y2, x1, y1, 1]  (five-point.cpp:374-383) */
    double *Q = (double *)malloc(sizeof(double) * (size_t)n * 9);
    for (int i = 0; i < n; ++i) {
        const double x1 = q1[2 * i], y1 = q1[2 * i + 1], x2 = q2[2 * i], y2 = q2[2 * i + 1];
        double *r = Q + (size_t)i * 9;
        r[0] = x1 * x2;
        r[1] = y1 * x2;
        r[2] = x2;
        r[3] = x1 * y2;
        r[4] = y1 * y2;
        r[5] = y2;
        r[6] = x1;
        r[7] = y1;
        r[8] = 1.0;
    }
    const int count = run5point_rows(Q, n, E_out, c_out, roots_out, xy1z_out);
    free(Q);
    return count;
}

/* The solver on an n x 9 system given row by row: row i = s_i * (u2_i (x) u1_i), entry 3 a + b = s_i u2_i[a] u1_i[b].  This is what
 * OpenGV's fivept_nister / fivept_stewenius build from unit bearing vectors for n >= 5 correspondences
 * (P/thirdparty/opengv/src/relative_pose/methods.cpp:183-268) and what the reference's weighted forms scale row by row
 * (P/source/usac/utils/weightingEssential.cpp:62-148); the essential matrices are those of the four right singular vectors of the
 * smallest singular values.  Q is overwritten. */
int oracle_run5point_rows(const double *rows, int n, double *E_out) {
    if (n < 5) return 0;
    double *Q = (double *)malloc(sizeof(double) * (size_t)n * 9);
    memcpy(Q, rows, sizeof(double) * (size_t)n * 9);
    const int count = run5point_rows(Q, n, E_out, NULL, NULL, NULL);
    free(Q);
    return count;
}

static int run5point_rows(double *Q, int n, double *E_out, double *c_out, double *roots_out, double *xy1z_out) {
    double w[9], V[81];
    jacobi_svd_impl(Q, n, 9, w, V, NULL);
    /* EE = columns 5..8 of V (five-point.cpp:386-388); EE[b*9 + k] = V[k][5+b] */
    double EE[36];
    for (int b = 0; b < 4; ++b)
        for (int k = 0; k < 9; ++k) EE[b * 9 + k] = V[k * 9 + 5 + b];

    double A[200];
    coeff_matrix(EE, A);
    if (!reduce_10x20(A)) return 0;

    /* B rows (five-point.cpp:396-414): B_i = A_row(2i+4) - z * A_row(2i+5), entries:
     *   [0..3] coefficient of x as cubic in z (z^3,z^2,z,1), [4..7] of y, [8..12] the quartic free term. */
    double b[39];
    for (int i = 0; i < 3; ++i) {
        const double *r1 = A + (2 * i + 4) * 20 + 10;
        const double *r2 = A + (2 * i + 5) * 20 + 10;
        double row1[13] = {0}, row2[13] = {0};
        for (int j = 0; j < 3; ++j) {
            row1[1 + j] = r1[j];
            row1[5 + j] = r1[3 + j];
            row2[0 + j] = r2[j];
            row2[4 + j] = r2[3 + j];
        }
        for (int j = 0; j < 4; ++j) {
            row1[9 + j] = r1[6 + j];
            row2[8 + j] = r2[6 + j];
        }
        for (int j = 0; j < 13; ++j) b[i * 13 + j] = row1[j] - row2[j];
    }
    double c[11];
    oracle_detpoly(b, c);

    double roots[2 * 10];
    oracle_solve_poly(c, 10, roots, 0);
    if (c_out) memcpy(c_out, c, sizeof(c));
    if (roots_out) memcpy(roots_out, roots, sizeof(roots));
    if (xy1z_out)
        for (int i = 0; i < 10; ++i) xy1z_out[i] = NAN;

    int count = 0;
    for (int i = 0; i < 10; ++i) {
        if (fabs(roots[2 * i + 1]) > 1e-10) continue; /* five-point.cpp:438 */
        const double z1 = roots[2 * i], z2 = z1 * z1, z3 = z2 * z1, z4 = z3 * z1;
        double bz[9];
        for (int j = 0; j < 3; ++j) {
            const double *br = b + j * 13;
            bz[j * 3 + 0] = br[0] * z3 + br[1] * z2 + br[2] * z1 + br[3];
            bz[j * 3 + 1] = br[4] * z3 + br[5] * z2 + br[6] * z1 + br[7];
            bz[j * 3 + 2] = br[8] * z4 + br[9] * z3 + br[10] * z2 + br[11] * z1 + br[12];
        }
        double w3[3], V3[9];
        jacobi_svd_impl(bz, 3, 3, w3, V3, NULL);
        const double xy1[3] = {V3[0 * 3 + 2], V3[1 * 3 + 2], V3[2 * 3 + 2]}; /* SVD::solveZ */
        if (xy1z_out) xy1z_out[i] = xy1[2];
        if (fabs(xy1[2]) < 1e-10) continue;                                  /* five-point.cpp:457 */
        const double x = xy1[0] / xy1[2], y = xy1[1] / xy1[2];
        double Ev[9], nrm = 0;
        for (int k = 0; k < 9; ++k) {
            Ev[k] = EE[0 * 9 + k] * x + EE[1 * 9 + k] * y + EE[2 * 9 + k] * z1 + EE[3 * 9 + k];
            nrm += Ev[k] * Ev[k];
        }
        nrm = sqrt(nrm);
        for (int k = 0; k < 9; ++k) E_out[count * 9 + k] = Ev[k] / nrm;
        count++;
    }
    return count;
}

int oracle_run5point(const double *q1, const double *q2, int n, double *E_out) {
    return run5point_impl(q1, q2, n, E_out, NULL, NULL, NULL);
}

/* Diagnostics for the parity tests: the degree-10 polynomial (ascending), solvePoly's roots (re, im) in root order and the
 * third component of the solveZ vector per root (NaN where the root was rejected as complex). */
int oracle_run5point_dbg(const double *q1, const double *q2, int n, double *E_out, double *c_out, double *roots_out,
                         double *xy1z_out) {
    return run5point_impl(q1, q2, n, E_out, c_out, roots_out, xy1z_out);
}

/* ------------------------------------------------------------------------------------------------------------
 * Sampson error / inliers
 * ---------------------------------------------------------------------------------------------------------- */
void oracle_sampson_err(const double *p1, const double *p2, int n, const double *E, float *err) {
    for (int i = 0; i < n; ++i) {
        const double x1 = p1[2 * i], y1 = p1[2 * i + 1], x2 = p2[2 * i], y2 = p2[2 * i + 1];
        /* Ex1 = E * (x1,y1,1): k-ordered sums as a 3x3 gemm */
        const double Ex1_0 = E[0] * x1 + E[1] * y1 + E[2] * 1.0;
        const double Ex1_1 = E[3] * x1 + E[4] * y1 + E[5] * 1.0;
        const double Ex1_2 = E[6] * x1 + E[7] * y1 + E[8] * 1.0;
        const double x2tEx1 = x2 * Ex1_0 + y2 * Ex1_1 + 1.0 * Ex1_2;
        /* Etx2 = E^T * (x2,y2,1) */
        const double Etx2_0 = E[0] * x2 + E[3] * y2 + E[6] * 1.0;
        const double Etx2_1 = E[1] * x2 + E[4] * y2 + E[7] * 1.0;
        const double a = Ex1_0 * Ex1_0, b = Ex1_1 * Ex1_1, c = Etx2_0 * Etx2_0, d = Etx2_1 * Etx2_1;
        err[i] = (float)(x2tEx1 * x2tEx1 / (a + b + c + d)); /* five-point.cpp:502 */
    }
}

int oracle_find_inliers(const double *p1, const double *p2, int n, const double *E, double thresh, float *err,
                        uint8_t *mask, double *err_sum) {
    oracle_sampson_err(p1, p2, n, E, err);
    const double t = thresh * thresh; /* modelest.cpp:79 */
    int good = 0;
    double s[4] = {0, 0, 0, 0};
    for (int i = 0; i < n; ++i) {
        mask[i] = (uint8_t)((double)err[i] <= t);
        good += mask[i];
        s[i & 3] += (double)err[i];
    }
    if (err_sum) *err_sum = (s[0] + s[2]) + (s[1] + s[3]);
    return good;
}

int oracle_ransac_update_num_iters(double p, double ep, int model_points, int max_iters) {
    p = p > 0. ? p : 0.;
    p = p < 1. ? p : 1.;
    ep = ep > 0. ? ep : 0.;
    ep = ep < 1. ? ep : 1.;
    double num = (1. - p) > DBL_MIN ? (1. - p) : DBL_MIN;
    double denom = 1. - pow(1. - ep, model_points);
    if (denom < DBL_MIN) return 0;
    num = log(num);
    denom = log(denom);
    return denom >= 0 || -num >= (double)max_iters * (-denom) ? max_iters : (int)round(num / denom);
}

/* checkSubset as written in the reference (modelest.cpp:613-650) with checkPartialSubsets == true. */
static int check_subset(const double *pts /* count x 2 */, int count) {
    int i, j, k;
    const int i0 = count - 1, i1 = count - 1;
    for (i = i0; i <= i1; i++) {
        for (j = 0; j < i; j++) {
            const double dx1 = pts[2 * j] - pts[2 * i];
            const double dy1 = pts[2 * j + 1] - pts[2 * i + 1];
            for (k = 0; k < j; k++) {
                const double dx2 = pts[2 * k] - pts[2 * i];
                const double dy2 = pts[2 * k + 1] - pts[2 * i + 1];
                if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) break;
            }
            if (k < j) break;
        }
        if (j < i) break;
    }
    return i >= i1; /* sic: always true */
}

int oracle_get_subset(oracle_glibc_rand *st, const double *p1, const double *p2, int n, int max_attempts, int *idx) {
    double ms1[10], ms2[10];
    int i = 0, j, iters = 0;
    for (; iters < max_attempts; iters++) {
        for (i = 0; i < 5 && iters < max_attempts;) {
            const int idx_i = oracle_rand(st) % n;
            idx[i] = idx_i;
            for (j = 0; j < i; j++)
                if (idx_i == idx[j]) break;
            if (j < i) continue;
            ms1[2 * i] = p1[2 * idx_i];
            ms1[2 * i + 1] = p1[2 * idx_i + 1];
            ms2[2 * i] = p2[2 * idx_i];
            ms2[2 * i + 1] = p2[2 * idx_i + 1];
            if (!check_subset(ms1, i + 1) || !check_subset(ms2, i + 1)) {
                iters++;
                continue;
            }
            i++;
        }
        break;
    }
    return i == 5 && iters < max_attempts;
}

int oracle_ransac_essential(const double *p1, const double *p2, int n, double thresh, double confidence, int max_iters,
                            int lesqu, unsigned seed, double *E, uint8_t *mask, int *n_inliers, int *iters_run,
                            oracle_ransac_trace *trace) {
    if (n_inliers) *n_inliers = 0;
    if (iters_run) *iters_run = 0;
    if (n < 5) return 0;
    oracle_glibc_rand st;
    oracle_srand(&st, seed);
    float *err = (float *)malloc(sizeof(float) * (size_t)n);
    uint8_t *tmask = (uint8_t *)malloc((size_t)n);
    uint8_t *best_mask = (uint8_t *)malloc((size_t)n);
    memset(best_mask, 0, (size_t)n);
    double models[90];
    double ms1[10], ms2[10];
    int niters = max_iters, maxGood = 0, iter;
    double errminsum = DBL_MAX;
    int result = 0;

    if (n == 5) niters = 1;
    for (iter = 0; iter < niters; iter++) {
        int idx[5] = {0, 1, 2, 3, 4};
        if (n > 5) {
            if (!oracle_get_subset(&st, p1, p2, n, 300, idx)) {
                if (iter == 0) goto done;
                break;
            }
        }
        for (int i = 0; i < 5; ++i) {
            ms1[2 * i] = p1[2 * idx[i]];
            ms1[2 * i + 1] = p1[2 * idx[i] + 1];
            ms2[2 * i] = p2[2 * idx[i]];
            ms2[2 * i + 1] = p2[2 * idx[i] + 1];
        }
        const int nmodels = oracle_run5point(ms1, ms2, 5, models);
        oracle_ransac_trace *tr = trace ? &trace[iter] : NULL;
        if (tr) {
            memset(tr, 0, sizeof(*tr));
            for (int i = 0; i < 5; ++i) tr->idx[i] = idx[i];
            tr->nmodels = nmodels;
            tr->best_taken = -1;
        }
        for (int i = 0; i < nmodels; ++i) {
            double esum;
            const int good = oracle_find_inliers(p1, p2, n, models + 9 * i, thresh, err, tmask, &esum);
            if (tr) {
                tr->good[i] = good;
                tr->err_sum[i] = esum;
            }
            if (good > (maxGood > 4 ? maxGood : 4)) { /* modelest.cpp:400 */
                memcpy(best_mask, tmask, (size_t)n);
                memcpy(E, models + 9 * i, sizeof(double) * 9);
                maxGood = good;
                niters = oracle_ransac_update_num_iters(confidence, (double)(n - good) / n, 5, niters);
                errminsum = esum;
                if (tr) tr->best_taken = i;
            } else if (good == (maxGood > 5 ? maxGood : 5) && errminsum < DBL_MAX && errminsum > esum) { /* :408 */
                memcpy(best_mask, tmask, (size_t)n);
                memcpy(E, models + 9 * i, sizeof(double) * 9);
                errminsum = esum;
                if (tr) tr->best_taken = i;
            }
        }
        if (tr) tr->niters_after = niters;
    }
    if (iters_run) *iters_run = iter;

    if (lesqu && maxGood > 0) { /* modelest.cpp:420-464 */
        double *s1 = (double *)malloc(sizeof(double) * 2 * (size_t)n), *s2 = (double *)malloc(sizeof(double) * 2 * (size_t)n);
        int m = 0;
        for (int i = 0; i < n; ++i)
            if (best_mask[i]) {
                s1[2 * m] = p1[2 * i];
                s1[2 * m + 1] = p1[2 * i + 1];
                s2[2 * m] = p2[2 * i];
                s2[2 * m + 1] = p2[2 * i + 1];
                ++m;
            }
        const int nmodels = oracle_run5point(s1, s2, m, models);
        free(s1);
        free(s2);
        if (nmodels <= 0) {
            result = 0; /* the reference returns `result` (false) here, modelest.cpp:442-443 */
            goto done;
        }
        for (int i = 0; i < nmodels; ++i) {
            double esum;
            const int good = oracle_find_inliers(p1, p2, n, models + 9 * i, thresh, err, tmask, &esum);
            if (good > (maxGood > 4 ? maxGood : 4)) {
                memcpy(best_mask, tmask, (size_t)n);
                memcpy(E, models + 9 * i, sizeof(double) * 9);
                maxGood = good;
                errminsum = esum;
            } else if (good == maxGood && errminsum < DBL_MAX && errminsum > esum) {
                memcpy(best_mask, tmask, (size_t)n);
                memcpy(E, models + 9 * i, sizeof(double) * 9);
                errminsum = esum;
            }
        }
    }
    if (maxGood > 0) {
        memcpy(mask, best_mask, (size_t)n);
        result = 1;
    }
    if (n_inliers) *n_inliers = maxGood;
done:
    free(err);
    free(tmask);
    free(best_mask);
    return result;
}

/* ------------------------------------------------------------------------------------------------------------
 * CvModelEstimator3::runLMeDS (modelest.cpp:483-564) as findEssentialMat drives it for method == LMEDS
 * (five-point.cpp:125-129: confidence = prob, maxIters = 2000; pose_estim.cpp:874-877 passes 0.999).
 * The error vector is sorted as int bit patterns (icvSortDistances on (int*)err.data, :541) -- for the non-negative
 * Sampson errors that is the float order; the median of an even count is the FLOAT sum of the two middle values times 0.5.
 * ---------------------------------------------------------------------------------------------------------- */
static int cmp_i32(const void *a, const void *b) {
    const int32_t x = *(const int32_t *)a, y = *(const int32_t *)b;
    return (x > y) - (x < y);
}

int oracle_lmeds_essential(const double *p1, const double *p2, int n, double confidence, int max_iters, unsigned seed,
                           double *E, uint8_t *mask, int *n_inliers, double *min_median) {
    if (n_inliers) *n_inliers = 0;
    if (n < 6) return 0;
    const double outlierRatio = 0.45;
    oracle_glibc_rand st;
    oracle_srand(&st, seed);
    float *err = (float *)malloc(sizeof(float) * (size_t)n);
    double models[90], ms1[10], ms2[10];
    double minMedian = DBL_MAX;
    int niters = (int)round(log(1. - confidence) / log(1. - pow(1. - outlierRatio, 5.0)));
    niters = niters > 3 ? niters : 3;
    niters = niters < max_iters ? niters : max_iters;
    for (int iter = 0; iter < niters; iter++) {
        int idx[5];
        if (!oracle_get_subset(&st, p1, p2, n, 300, idx)) {
            if (iter == 0) {
                free(err);
                return 0;
            }
            break;
        }
        for (int i = 0; i < 5; ++i) {
            ms1[2 * i] = p1[2 * idx[i]];
            ms1[2 * i + 1] = p1[2 * idx[i] + 1];
            ms2[2 * i] = p2[2 * idx[i]];
            ms2[2 * i + 1] = p2[2 * idx[i] + 1];
        }
        const int nmodels = oracle_run5point(ms1, ms2, 5, models);
        for (int i = 0; i < nmodels; ++i) {
            oracle_sampson_err(p1, p2, n, models + 9 * i, err);
            qsort(err, (size_t)n, sizeof(float), cmp_i32);
            const double median = (n % 2 != 0) ? (double)err[n / 2] : (double)(err[n / 2 - 1] + err[n / 2]) * 0.5;
            if (median < minMedian) {
                minMedian = median;
                memcpy(E, models + 9 * i, sizeof(double) * 9);
            }
        }
    }
    int result = 0;
    if (minMedian < DBL_MAX) {
        double sigma = 2.5 * 1.4826 * (1 + 5. / (n - 5)) * sqrt(minMedian);
        sigma = sigma > 0.001 ? sigma : 0.001;
        const int cnt = oracle_find_inliers(p1, p2, n, E, sigma, err, mask, NULL);
        if (n_inliers) *n_inliers = cnt;
        result = cnt >= 5;
    }
    if (min_median) *min_median = minMedian;
    free(err);
    return result;
}

/* ------------------------------------------------------------------------------------------------------------
 * Pose recovery
 * ---------------------------------------------------------------------------------------------------------- */
static double det3(const double *M) {
    return M[0] * (M[4] * M[8] - M[5] * M[7]) - M[1] * (M[3] * M[8] - M[5] * M[6]) + M[2] * (M[3] * M[7] - M[4] * M[6]);
}
static void mat3_mul(const double *A, const double *B, double *C) {
    for (int r = 0; r < 3; ++r)
        for (int c = 0; c < 3; ++c) C[r * 3 + c] = A[r * 3] * B[c] + A[r * 3 + 1] * B[3 + c] + A[r * 3 + 2] * B[6 + c];
}

/* SVD of a 3x3 E = U diag(w) Vt via one-sided Jacobi; U's columns for (near-)zero singular values are completed
 * with the cross product of the others so that U is orthogonal. */
static void svd3(const double *E, double *U, double *w, double *Vt) {
    double V[9], AV[9];
    jacobi_svd_impl(E, 3, 3, w, V, AV);
    for (int j = 0; j < 3; ++j)
        for (int i = 0; i < 3; ++i) U[i * 3 + j] = (w[j] > 0) ? AV[i * 3 + j] / w[j] : 0.0;
    if (!(w[2] > 1e-12 * w[0])) {
        /* u3 = u1 x u2 */
        U[0 * 3 + 2] = U[1 * 3 + 0] * U[2 * 3 + 1] - U[2 * 3 + 0] * U[1 * 3 + 1];
        U[1 * 3 + 2] = U[2 * 3 + 0] * U[0 * 3 + 1] - U[0 * 3 + 0] * U[2 * 3 + 1];
        U[2 * 3 + 2] = U[0 * 3 + 0] * U[1 * 3 + 1] - U[1 * 3 + 0] * U[0 * 3 + 1];
    }
    for (int i = 0; i < 3; ++i)
        for (int j = 0; j < 3; ++j) Vt[i * 3 + j] = V[j * 3 + i];
}

void oracle_decompose_essential(const double *E, double *R1, double *R2, double *t) {
    double U[9], w[3], Vt[9];
    svd3(E, U, w, Vt);
    if (det3(U) < 0)
        for (int i = 0; i < 9; ++i) U[i] = -U[i];
    if (det3(Vt) < 0)
        for (int i = 0; i < 9; ++i) Vt[i] = -Vt[i];
    const double W[9] = {0, 1, 0, -1, 0, 0, 0, 0, 1};
    const double Wt[9] = {0, -1, 0, 1, 0, 0, 0, 0, 1};
    double T[9];
    mat3_mul(U, W, T);
    mat3_mul(T, Vt, R1);
    mat3_mul(U, Wt, T);
    mat3_mul(T, Vt, R2);
    t[0] = U[2];
    t[1] = U[5];
    t[2] = U[8];
}

void oracle_triangulate_point(const double *P0, const double *P1, const double *x1, const double *x2, double *X4) {
    double A[16], w[4], V[16];
    for (int k = 0; k < 4; ++k) {
        A[0 * 4 + k] = x1[0] * P0[2 * 4 + k] - P0[0 * 4 + k];
        A[1 * 4 + k] = x1[1] * P0[2 * 4 + k] - P0[1 * 4 + k];
        A[2 * 4 + k] = x2[0] * P1[2 * 4 + k] - P1[0 * 4 + k];
        A[3 * 4 + k] = x2[1] * P1[2 * 4 + k] - P1[1 * 4 + k];
    }
    jacobi_svd_impl(A, 4, 4, w, V, NULL);
    for (int k = 0; k < 4; ++k) X4[k] = V[k * 4 + 3];
}

static int recover_pose_impl(const double *E, const double *t_only, const double *p1, const double *p2, int n, double dist,
                             double *R, double *t, double *Q, uint8_t *mask_inout);

int oracle_recover_pose(const double *E, const double *p1, const double *p2, int n, double dist, double *R, double *t,
                        double *Q, uint8_t *mask_inout) {
    return recover_pose_impl(E, NULL, p1, p2, n, dist, R, t, Q, mask_inout);
}

/* getTfromTransEssential (pose_helper.cpp:422-433) + recoverPose with t_only (five-point.cpp:178-193): R1 = I, only
 * [I|t] and [I|-t] are tested (good2 = good4 = 0). */
int oracle_recover_pose_translation(const double *Et, const double *p1, const double *p2, int n, double dist, double *R,
                                    double *t, double *Q, uint8_t *mask_inout) {
    double tv[3] = {Et[1 * 3 + 2], Et[2 * 3 + 0], Et[0 * 3 + 1]};
    const double nrm = sqrt(tv[0] * tv[0] + tv[1] * tv[1] + tv[2] * tv[2]);
    if (fabs(nrm - 1.0) > 1e-3)
        for (int i = 0; i < 3; ++i) tv[i] /= nrm;
    return recover_pose_impl(NULL, tv, p1, p2, n, dist, R, t, Q, mask_inout);
}

static int recover_pose_impl(const double *E, const double *t_only, const double *p1, const double *p2, int n, double dist,
                             double *R, double *t, double *Q, uint8_t *mask_inout) {
    double R1[9], R2[9], tv[3];
    if (E) {
        oracle_decompose_essential(E, R1, R2, tv);
    } else {
        const double I3[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        memcpy(R1, I3, sizeof(I3));
        memcpy(R2, I3, sizeof(I3));
        memcpy(tv, t_only, sizeof(tv));
    }
    const double P0[12] = {1, 0, 0, 0, 0, 1, 0, 0, 0, 0, 1, 0};
    double P[4][12];
    const double *Rs[4] = {R1, R2, R1, R2};
    const double ts[4] = {1, 1, -1, -1};
    for (int c = 0; c < 4; ++c)
        for (int r = 0; r < 3; ++r) {
            for (int k = 0; k < 3; ++k) P[c][r * 4 + k] = Rs[c][r * 3 + k];
            P[c][r * 4 + 3] = ts[c] * tv[r];
        }
    double *Qc = (double *)malloc(sizeof(double) * 4 * 3 * (size_t)n);
    uint8_t *mc = (uint8_t *)malloc(4 * (size_t)n);
    int good[4] = {0, 0, 0, 0};
    for (int c = 0; c < 4; ++c) {
        for (int i = 0; i < n; ++i) {
            double X[4];
            oracle_triangulate_point(P0, P[c], p1 + 2 * i, p2 + 2 * i, X);
            int m = (X[2] * X[3] > 0);
            const double qz = P[c][8] * X[0] + P[c][9] * X[1] + P[c][10] * X[2] + P[c][11] * X[3];
            m = m && (qz * X[3] > 0);
            const double qx = X[0] / X[3], qy = X[1] / X[3], qzz = X[2] / X[3];
            m = m && (qzz < dist);
            Qc[((size_t)c * n + i) * 3 + 0] = qx;
            Qc[((size_t)c * n + i) * 3 + 1] = qy;
            Qc[((size_t)c * n + i) * 3 + 2] = qzz;
            uint8_t mv = m ? 255 : 0;
            if (mask_inout) mv &= mask_inout[i];
            mc[(size_t)c * n + i] = mv;
            good[c] += (mv != 0);
        }
    }
    const int good1 = good[0], good2 = E ? good[1] : 0, good3 = good[2], good4 = E ? good[3] : 0;
    int pick = -1, ret;
    if (good1 >= good2 && good1 >= good3 && good1 >= good4) {
        pick = 0;
        ret = good1;
    } else if (good2 && good2 >= good1 && good2 >= good3 && good2 >= good4) {
        pick = 1;
        ret = good2;
    } else if (good3 >= good1 && good3 >= good2 && good3 >= good4) {
        pick = 2;
        ret = good3;
    } else {
        ret = good4;
        pick = good4 ? 3 : -1;
    }
    if (pick >= 0) {
        memcpy(R, Rs[pick], sizeof(double) * 9);
        for (int r = 0; r < 3; ++r) t[r] = ts[pick] * tv[r];
        memcpy(Q, Qc + (size_t)pick * n * 3, sizeof(double) * 3 * (size_t)n);
        if (mask_inout) memcpy(mask_inout, mc + (size_t)pick * n, (size_t)n);
    } else {
        const double I[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
        memcpy(R, I, sizeof(I));
        t[0] = t[1] = t[2] = 0;
        memset(Q, 0, sizeof(double) * 3 * (size_t)n);
        if (mask_inout) memset(mask_inout, 0, (size_t)n);
    }
    free(Qc);
    free(mc);
    return ret;
}
