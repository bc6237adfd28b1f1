/*
 * prepost_oracle.c -- CPU restatement of the pre/post steps either side of the robust estimation.
 * TEST INFRASTRUCTURE ONLY (see oracle.h).
 *
 * Follows /root/reference/matchinglib_poselib/source/poselib/source/pose_helper.cpp:
 *   :1100-1109  ImgToCamCoordTrans   (float result of a double operation)
 *   :1169-1223  Remove_LensDist      (drops correspondences whose undistortion does not re-distort onto the input,
 *                                     returns false when fewer than 16 remain, no-op when both coefficient sums are ~0)
 *   :1241-1279  LensDist_Oulu        (10 fixed-point iterations, 0.25 proof gate in float)
 *   :639-664    computeReprojError2  (Sampson error kept in double)
 *   :3030-3045  getInlierMask        (strict `error < th`)
 */
#include <math.h>
#include <stdlib.h>
#include <string.h>

#include "oracle.h"

void oracle_img_to_cam(float *pts /* n x 2, in place */, int n, const double K[4] /* fx fy cx cy */) {
    for (int i = 0; i < n; ++i) {
        pts[2 * i] = (float)(((double)pts[2 * i] - K[2]) / K[0]);
        pts[2 * i + 1] = (float)(((double)pts[2 * i + 1] - K[3]) / K[1]);
    }
}

static int lens_dist_oulu(const float *distorted, float *corrected, const double *dist, int iters) {
    const double k1 = dist[0], k2 = dist[1], p1 = dist[2], p2 = dist[3], k3 = dist[4], k4 = dist[5], k5 = dist[6], k6 = dist[7];
    double r2, _2xy, rad_corr, delta[2];
    for (int i = 0; i < iters; i++) {
        r2 = (double)corrected[0] * (double)corrected[0] + (double)corrected[1] * (double)corrected[1];
        _2xy = 2.0 * (double)corrected[0] * (double)corrected[1];
        rad_corr = (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1.0 + ((k6 * r2 + k5) * r2 + k4) * r2);
        delta[0] = p1 * _2xy + p2 * (r2 + 2.0 * (double)corrected[0] * (double)corrected[0]);
        delta[1] = p1 * (r2 + 2.0 * (double)corrected[1] * (double)corrected[1]) + p2 * _2xy;
        corrected[0] = (float)(((double)distorted[0] - delta[0]) / rad_corr);
        corrected[1] = (float)(((double)distorted[1] - delta[1]) / rad_corr);
    }
    float proof[2];
    r2 = (double)corrected[0] * (double)corrected[0] + (double)corrected[1] * (double)corrected[1];
    _2xy = 2.0 * (double)corrected[0] * (double)corrected[1];
    rad_corr = (1.0 + ((k3 * r2 + k2) * r2 + k1) * r2) / (1.0 + ((k6 * r2 + k5) * r2 + k4) * r2);
    delta[0] = p1 * _2xy + p2 * (r2 + 2.0 * (double)corrected[0] * (double)corrected[0]);
    delta[1] = p1 * (r2 + 2.0 * (double)corrected[1] * (double)corrected[1]) + p2 * _2xy;
    proof[0] = (float)((double)corrected[0] * rad_corr + delta[0] - (double)distorted[0]);
    proof[1] = (float)((double)corrected[1] * rad_corr + delta[1] - (double)distorted[1]);
    if (sqrtf(proof[0] * proof[0] + proof[1] * proof[1]) > 0.25f) return 0;
    return 1;
}

/* points: n x 2 floats each, rewritten in place (compacted); *n_out = remaining count.
 * Returns 1 (true) or 0 (false = fewer than 16 valid correspondences; points then hold the undistorted values of the
 * processed entries, like the reference, and *n_out = n). */
int oracle_remove_lens_dist(float *points1, float *points2, int n, const double dist1[8], const double dist2[8], int *n_out) {
    double s1 = 0, s2 = 0;
    for (int i = 0; i < 8; ++i) {
        s1 += dist1[i];
        s2 += dist2[i];
    }
    *n_out = n;
    if ((s1 < 1e-3 && s1 > -1e-3) && (s2 < 1e-3 && s2 > -1e-3)) return 1; /* nearZero(sum(dist)) : pose_helper.h:82-87 */
    unsigned char *mask = (unsigned char *)malloc((size_t)(n > 0 ? n : 1));
    int n1 = 0;
    for (int i = 0; i < n; ++i) {
        float d1[2] = {points1[2 * i], points1[2 * i + 1]}, d2[2] = {points2[2 * i], points2[2 * i + 1]};
        mask[i] = 1;
        if (!lens_dist_oulu(d1, points1 + 2 * i, dist1, 10)) {
            mask[i] = 0;
            continue;
        }
        if (!lens_dist_oulu(d2, points2 + 2 * i, dist2, 10)) mask[i] = 0;
    }
    for (int i = 0; i < n; ++i) n1 += mask[i];
    if (n1 < 16) {
        free(mask);
        return 0;
    }
    if (n1 < n) {
        int w = 0;
        for (int i = 0; i < n; ++i)
            if (mask[i]) {
                points1[2 * w] = points1[2 * i];
                points1[2 * w + 1] = points1[2 * i + 1];
                points2[2 * w] = points2[2 * i];
                points2[2 * w + 1] = points2[2 * i + 1];
                ++w;
            }
    }
    *n_out = n1;
    free(mask);
    return 1;
}

/* computeReprojError2 + getInlierMask: err[i] double Sampson error, mask[i] = err[i] < th2 (strict).  Returns #inliers. */
int oracle_get_inliers_strict(const double *p1, const double *p2, int n, const double *E, double th2, double *err,
                              unsigned char *mask) {
    int cnt = 0;
    for (int i = 0; i < n; ++i) {
        const double x1 = p1[2 * i], y1 = p1[2 * i + 1], x2 = p2[2 * i], y2 = p2[2 * i + 1];
        const double Ex1_0 = E[0] * x1 + E[1] * y1 + E[2] * 1.0;
        const double Ex1_1 = E[3] * x1 + E[4] * y1 + E[5] * 1.0;
        const double Ex1_2 = E[6] * x1 + E[7] * y1 + E[8] * 1.0;
        const double x2tEx1 = x2 * Ex1_0 + y2 * Ex1_1 + 1.0 * Ex1_2;
        const double Etx2_0 = E[0] * x2 + E[3] * y2 + E[6] * 1.0;
        const double Etx2_1 = E[1] * x2 + E[4] * y2 + E[7] * 1.0;
        const double a = Ex1_0 * Ex1_0, b = Ex1_1 * Ex1_1, c = Etx2_0 * Etx2_0, d = Etx2_1 * Etx2_1;
        err[i] = x2tEx1 * x2tEx1 / (a + b + c + d);
        mask[i] = (unsigned char)(err[i] < th2);
        cnt += mask[i];
    }
    return cnt;
}
