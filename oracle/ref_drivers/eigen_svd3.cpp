// eigen_svd3.cpp -- driver for Eigen::JacobiSVD<Matrix3d> of the Eigen 3.2.0 the reference vendors
// (poselib/thirdparty/opengv/third_party_notuse/Eigen), the decomposition CvEMEstimator::ValidModel takes its epipole from
// (poselib/source/five-point-nister/five-point.cpp:553-566, V.col(2) of JacobiSVD(E^T)).
// TEST INFRASTRUCTURE: built only where /root/reference exists (oracle/Makefile), against the Eigen headers in place; pins
// oracle_eigen_svd3 (singular values, V including the SIGN of its columns).  This file is ours; it contains no reference source.
//
// usage: eigen_svd3 in.bin out.bin [eig]
//   in.bin : int32 count ; then count * 9 doubles (row-major 3x3)
//   out.bin: per matrix 3 singular values + 9 doubles U + 9 doubles V (row-major)
//   with `eig`: per matrix the real parts of Eigen::EigenSolver<Matrix3d>'s eigenvalues IN ITS ORDER (3 doubles) -- the order OpenGV's
//   eigensolver relies on (thirdparty/opengv/src/relative_pose/modules/main.cpp:646-659); pins dgm::eigen_diag_order3 and the oracle's
#include <cstdint>
#include <cstdio>
#include <vector>

#include <Eigen/Dense>
#include <Eigen/Eigenvalues>
#include <cstring>

int main(int argc, char **argv) {
    if (argc < 3) return 1;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t cnt;
    if (fread(&cnt, 4, 1, f) != 1) return 2;
    std::vector<double> m((size_t)cnt * 9);
    if (fread(m.data(), 8, m.size(), f) != m.size()) return 2;
    fclose(f);
    FILE *out = fopen(argv[2], "wb");
    const bool eig = argc > 3 && !std::strcmp(argv[3], "eig");
    for (int s = 0; s < cnt; ++s) {
        Eigen::Matrix3d M;
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) M(r, c) = m[(size_t)s * 9 + r * 3 + c];
        if (eig) {
            Eigen::EigenSolver<Eigen::Matrix3d> E(M, true);
            double o[3];
            for (int i = 0; i < 3; ++i) o[i] = E.eigenvalues()[i].real();
            fwrite(o, 8, 3, out);
            continue;
        }
        Eigen::JacobiSVD<Eigen::Matrix3d> svd(M, Eigen::ComputeFullU | Eigen::ComputeFullV);
        double o[21];
        for (int i = 0; i < 3; ++i) o[i] = svd.singularValues()(i);
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c) o[3 + r * 3 + c] = svd.matrixU()(r, c), o[12 + r * 3 + c] = svd.matrixV()(r, c);
        fwrite(o, 8, 21, out);
    }
    fclose(out);
    return 0;
}
