// opengv_5pt.cpp -- driver for the reference's vendored OpenGV fivept_nister
// (poselib/thirdparty/opengv/src/relative_pose/methods.cpp:239-268), the minimal solver the reference's USAC path
// uses for POSE_NISTER (poselib/include/usac/estimators/EssentialMatEstimator.h:395).
// TEST INFRASTRUCTURE: built only where /root/reference exists (oracle/Makefile), against the OpenGV + vendored
// Eigen 3.2.0 sources in place; pins oracle_run5point's E-sets (up to sign) and generates tests/golden vectors.
// This file is ours; it contains no reference source.
//
// usage: opengv_5pt in.bin out.bin
//   in.bin : int32 n_samples, int32 pts_per_sample(>=5) ; then n_samples*pts*4 doubles (x1,y1,x2,y2)
//   out.bin: per sample int32 count + 10*9 doubles (row-major E with x2^T E x1 = 0, Frobenius-normalised)
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <vector>

#include <opengv/relative_pose/CentralRelativeAdapter.hpp>
#include <opengv/relative_pose/methods.hpp>

int main(int argc, char **argv) {
    if (argc != 3) return 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t hdr[2];
    if (fread(hdr, 4, 2, f) != 2) return 2;
    const int ns = hdr[0], np = hdr[1];
    std::vector<double> pts((size_t)ns * np * 4);
    if (fread(pts.data(), 8, pts.size(), f) != pts.size()) return 2;
    fclose(f);
    FILE *out = fopen(argv[2], "wb");
    for (int s = 0; s < ns; ++s) {
        opengv::bearingVectors_t b1, b2;  // b1 <- view 2, b2 <- view 1 so that b1^T E b2 = x2^T E x1
        for (int i = 0; i < np; ++i) {
            const double *p = &pts[((size_t)s * np + i) * 4];
            opengv::bearingVector_t v1(p[0], p[1], 1.0), v2(p[2], p[3], 1.0);
            b2.push_back(v1 / v1.norm());
            b1.push_back(v2 / v2.norm());
        }
        opengv::relative_pose::CentralRelativeAdapter adapter(b1, b2);
        opengv::essentials_t es = opengv::relative_pose::fivept_nister(adapter);
        int32_t cnt = (int32_t)es.size();
        if (cnt > 10) cnt = 10;
        double buf[90] = {0};
        for (int k = 0; k < cnt; ++k) {
            double nrm = es[k].norm();
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) buf[k * 9 + r * 3 + c] = es[k](r, c) / nrm;
        }
        fwrite(&cnt, 4, 1, out);
        fwrite(buf, 8, 90, out);
    }
    fclose(out);
    return 0;
}
