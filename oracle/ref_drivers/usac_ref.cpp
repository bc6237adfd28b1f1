// usac_ref.cpp -- drives the REFERENCE's own USAC control flow (poselib/include/usac/estimators/USAC.h, a header-only template with
// no OpenCV dependency: PROSAC sampling, SPRT verification, LO-RANSAC, stopping criteria) compiled WHERE IT LIES under /root/reference,
// together with the reference's usac/utils {MathFunctions,FundmatrixFunctions,PoseFunctions}.cpp and its vendored OpenGV fivept_nister
// (the minimal solver of PoseEstimator::POSE_NISTER, EssentialMatEstimator.h:395).
//
// TEST INFRASTRUCTURE: built only where /root/reference exists (oracle/Makefile target `ref`), output oracle/_ref/usac_ref.  It pins
// oracle/usac_oracle.cpp and the device path turn by turn (tests/golden/usac_trace.npz, generator tests/golden/make_golden.py).
//
// What is the reference's and what is ours.  The reference's problem class, EssentialMatEstimator (EssentialMatEstimator.h, 2447 lines),
// cannot be compiled here: it includes OpenCV (cv::Mat members, poselib/pose_estim.h).  The class below is a TRANSCRIPTION of the members
// the 5-point paths execute -- initProblem :189-352, generateMinimalSampleModels :384-398 + :456-520, generateRefinedModel
// REFINE_WEIGHTS :540-599 / REFINE_STEWENIUS(_WEIGHTS) :640-755 / REFINE_NISTER(_WEIGHTS) :757-850, validateSample :1043-1078,
// validateModel :1085-1104, evaluateModel :1110-1178, findWeights :2366-2428, storeModel :2436-2445 -- with the reference's statements
// and, in many places, its identifiers (adapter_denorm, inverseSolution, p_hom, reprojection1/2 ...), calling the reference's own FTools
// / MathTools / PoseTools / OpenGV functions wherever the original does.  With `check_degeneracy` (in.bin ih[5]) it also carries
// testSolutionDegeneracy :1334-1362, testSolutionDegeneracyRot :1511-1663, testSolutionDegeneracyNoMot :1838-1911,
// upgradeDegenerateModel's pose branches :2098-2361 and evaluateModelTrans :1264-1327 -- the configuration estimateEssentialMatUsac
// builds for a refinement other than the 8-point ones (enableHDegen = false, enableUpgradeDegenPose = true, 8000 upgrade samples,
// usac_estimations.cpp:443-456) -- on the reference's own opengv::relative_pose::{twopt_rotationOnly, rotationOnly, twopt,
// eigensolver}, opengv::triangulation::triangulate2 and PoseTools::{getRotError, getNoMotError}.  Everything USAC<> does with them
// (solve(), the samplers, designSPRTTest, the SPRT history, updateSPRTStopping, locallyOptimizeSolution, storeSolution,
// std::random_shuffle of the evaluation pool on the process-wide rand() stream) is the reference's code, unmodified.  So: USAC.h,
// usac/utils/*.cpp and OpenGV are genuinely the reference's; the problem class is a stand-in that pins the CONTROL FLOW, not itself
// pinned by anything.  It lives under oracle/ (test infrastructure, built only in the build container) and nothing in the product uses it.
// The weighted solvers fivept_nister_weight / fivept_stewenius_weight and computePseudoHuberWeight (P/source/usac/utils/
// weightingEssential.cpp:56-206; that file includes BA_driver.h, which needs SBA and a generated export header: unbuildable here) and
// poselib::costPseudoHuber (P/source/BA_driver.cpp:2639-2648) are transcribed below on the reference-built
// opengv::relative_pose::modules::fivept_{nister,stewenius}_main; Eigen::BDCSVD, which they name, does not exist in the vendored Eigen
// 3.2.0 -- JacobiSVD delivers the same four-dimensional right singular subspace, which is all the solvers read.
//
// ONE convention on top of OpenGV (shared with oracle/usac_oracle.cpp and the device path): the solutions of a minimal sample are
// ordered by ascending E(0,0) after scaling each to unit Frobenius norm with its largest-magnitude element positive.  OpenGV's own order
// is the breadth-first order of its Sturm bracketing in a null-space basis that Eigen's column-pivoted QR picks among five columns of
// EQUAL norm (unit bearing vectors): the pivot, hence the basis and the root order, is decided by the last bits of a vectorised sum whose
// association depends on the Eigen version and the alignment of the column -- rounding noise no restatement can reproduce.  The SET of
// solutions does not depend on it.  `--native-order` keeps OpenGV's order (diagnostics).
//
// usage: usac_ref in.bin out.bin [--native-order] [--solver-oracle] [--stewenius] [--eigvec-smallest]
//   in.bin : int32 n, seed, refine (poselib::RefineAlg: 0 = REF_WEIGHTS, 4 / 5 = REF_STEWENIUS(_WEIGHTS), 6 / 7 = REF_NISTER(_WEIGHTS)), prosac (0/1), max_hyp, check_degeneracy (0, 1, 3 = also after local optimisations), reserved[2];
//            double th, prosac_beta, sprt_delta, sprt_epsilon, sprt_mS, sprt_tM, conf, th_pixels / focal length;
//            n * 4 doubles (x1,y1,x2,y2); if prosac: n uint32 sorted indices
//   out.bin: int32 n_events; n_events * 16 doubles (event records, see emit()); then the final record (see main)
//   event types: 1 sample, 2 evaluation, 3 refined model, 4 stored model, 5 minimal model, 6 model rejected by validateModel,
//                7 degeneracy test [hyp, degenerate, upgrade, type, count_rot, count_noMot, best], 8 rotation found on all points
//                [hyp, pair, inliers of the 2-point rotation, inliers of the refit, kept], 9 upgrade [hyp, branch 1 = no motion -> t /
//                2 = R -> R + t, candidates tried, result], 10 upgrade candidate [hyp, branch, iteration, model (E, or t and 6 zeros)]
#include <algorithm>
#include <cfloat>
#include <cmath>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <iostream>
#include <fstream>
#include <string>
#include <memory>
#include <vector>

#include "usac/config/ConfigParamsEssentialMat.h"
#include "usac/utils/MathFunctions.h"
#include "usac/utils/FundmatrixFunctions.h"
#include "usac/utils/HomographyFunctions.h"
#include "usac/utils/PoseFunctions.h"
#include "usac/estimators/USAC.h"

#include <opengv/relative_pose/methods.hpp>
#include <opengv/triangulation/methods.hpp>
#include <opengv/relative_pose/CentralRelativeAdapter.hpp>
#include <opengv/relative_pose/modules/main.hpp>

static bool g_native_order = false;
// --solver-oracle: the minimal and the REFINE_NISTER models come from oracle_run5point (oracle/pose_oracle.c, linked in) instead of
// OpenGV, so that the trace of the reference's CONTROL FLOW can be compared decision by decision with oracle/usac_oracle.cpp, which uses
// the same solver.  (OpenGV's fivept_nister returns unconverged roots on a noticeable share of samples -- its Sturm brackets are
// bound / (10 roots) wide and get five Newton steps -- so with it the traces part at the first such sample; see tests/test_oracle_usac.py.)
static bool g_solver_oracle = false;
// --stewenius: the minimal models come from OpenGV's fivept_stewenius as ConfigUSAC's (and the harness') default estimator POSE_STEWENIUS
// takes them (EssentialMatEstimator.h:456-489) -- an eigenvalue solver, converged to rounding, unlike fivept_nister's bracketing; with the
// order convention on top, the trace of this REFERENCE-SOLVER run is what tests/test_oracle_usac.py holds the oracle's own solver to.
static bool g_stewenius = false;
// --eigvec-smallest: OpenGV's eigensolver takes its translation from column 0 of Eigen::EigenSolver's eigenvectors
// (modules/main.cpp:646-659) -- meant to be the eigenvector of the smallest eigenvalue, but EigenSolver orders nothing: on symmetric
// matrices of this kind the smallest eigenvalue sits at position 0 in about a third of the cases (probe: 6178 / 7806 / 6016 of 20000).
// With this flag the R -> R + t upgrade takes the eigenvector of the smallest eigenvalue from the same eigensolver output (a diagnostic:
// what the solver was meant to do).  Without it -- the fixtures -- OpenGV's own choice, which oracle/ and the device path reproduce by
// restating Eigen::RealSchur's eigenvalue order (oracle_eigen_order3, dgm::eigen_diag_order3; tests/golden/eigen_order3.npz).
static bool g_eigvec_smallest = false;
extern "C" int oracle_run5point(const double *q1, const double *q2, int n, double *E_out);
extern "C" int oracle_run5point_rows(const double *rows, int n, double *E_out);
static std::vector<double> g_events;  // 16 doubles per event

class RefEssential : public USAC<RefEssential> {
   public:
    std::vector<double> final_model_params_;
    int in_lo = 0;

    void emit(double type, const double *v, int nv) {
        double rec[16] = {0};
        rec[0] = type;
        for (int i = 0; i < nv && i < 15; ++i) rec[1 + i] = v[i];
        g_events.insert(g_events.end(), rec, rec + 16);
    }

    bool initProblem(const ConfigParamsEssential &cfg, double *pointData) {
        const unsigned n = cfg.common.numDataPoints;
        input_points_denorm_ = pointData;
        input_points_.assign((size_t)6 * n, 0.0);
        FTools::normalizePoints(input_points_denorm_, input_points_.data(), n, m_T1_, m_T2_);
        MathTools::mattr(m_T2_trans_, m_T2_, 3, 3);
        for (int i = 0; i < 3; i++)
            for (int j = 0; j < 3; j++) {
                m_T1_e(i, j) = m_T1_[3 * i + j];
                m_T2_e(i, j) = m_T2_[3 * i + j];
                m_T2_trans_e(i, j) = m_T2_trans_[3 * i + j];
            }
        m_T1_e_inv = m_T1_e.inverse();
        m_T2_e_inv = m_T2_e.inverse();
        m_T2_trans_e_inv = m_T2_trans_e.inverse();
        final_model_params_.assign(9, 0.0);
        models_.assign(usac_max_solns_per_sample_, std::vector<double>(9, 0.0));
        models_denorm_.assign(usac_max_solns_per_sample_, std::vector<double>(9, 0.0));
        opengv::bearingVectors_t &b1 = bearing1_, &b2 = bearing2_;  // the adapter keeps references to them
        for (unsigned i = 0; i < n; i++) {
            const double *p = input_points_denorm_ + 6 * i;
            opengv::point_t v1, v2;
            v1 << p[0], p[1], p[2];
            v2 << p[3], p[4], p[5];
            b1.push_back(v1 / v1.norm());
            b2.push_back(v2 / v2.norm());
        }
        adapter_denorm.reset(new opengv::relative_pose::CentralRelativeAdapter(b2, b1));
        essentials = opengv::essentials_t(usac_max_solns_per_sample_);
        essentials_denorm = opengv::essentials_t(usac_max_solns_per_sample_);
        data_matrix_.assign((size_t)9 * n, 0.0);
        FTools::computeDataMatrix(data_matrix_.data(), n, input_points_.data());
        refineMethod = cfg.fund.refineMethod;
        return true;
    }

    static double order_key(const opengv::essential_t &E) {
        double big = 0, nrm = E.norm();
        for (int r = 0; r < 3; ++r)
            for (int c = 0; c < 3; ++c)
                if (std::fabs(E(r, c)) > std::fabs(big)) big = E(r, c);
        return (big < 0 ? -E(0, 0) : E(0, 0)) / nrm;
    }
    opengv::essentials_t five_point(const std::vector<int> &indices) {
        if (g_stewenius) {  // ESTIM_STEWENIUS (:456-489): the solutions whose imaginary parts all pass nearZero(100 * imag), |.| < 1e-3
            opengv::complexEssentials_t Ec = opengv::relative_pose::fivept_stewenius(*adapter_denorm, indices);
            opengv::essentials_t out;
            for (auto &Ei : Ec) {
                bool imag = false;
                for (int r = 0; r < 3 && !imag; r++)
                    for (int c = 0; c < 3; c++) {
                        const double d = 100 * Ei(r, c).imag();
                        if (!(d < 1e-3 && d > -1e-3)) {
                            imag = true;
                            break;
                        }
                    }
                if (imag) continue;
                opengv::essential_t E;
                for (int r = 0; r < 3; r++)
                    for (int c = 0; c < 3; c++) E(r, c) = Ei(r, c).real();
                out.push_back(E);
            }
            return out;
        }
        if (!g_solver_oracle) return opengv::relative_pose::fivept_nister(*adapter_denorm, indices);
        std::vector<double> q1(2 * indices.size()), q2(2 * indices.size());
        for (size_t i = 0; i < indices.size(); ++i) {
            const double *p = input_points_denorm_ + 6 * indices[i];
            q1[2 * i] = p[0], q1[2 * i + 1] = p[1], q2[2 * i] = p[3], q2[2 * i + 1] = p[4];
        }
        double Es[90];
        const int ns = oracle_run5point(q1.data(), q2.data(), (int)indices.size(), Es);
        opengv::essentials_t out;
        for (int k = 0; k < ns; ++k) {
            opengv::essential_t E;
            for (int r = 0; r < 3; ++r)
                for (int c = 0; c < 3; ++c) E(r, c) = Es[9 * k + 3 * r + c];
            out.push_back(E);
        }
        return out;
    }

    static opengv::essentials_t real_solutions(const opengv::complexEssentials_t &Ec) {  // :671-696 (nearZero(100 * imag): |.| < 1e-3)
        opengv::essentials_t out;
        for (auto &Ei : Ec) {
            bool imag = false;
            for (int r = 0; r < 3 && !imag; r++)
                for (int c = 0; c < 3; c++) {
                    const double d = 100 * Ei(r, c).imag();
                    if (!(d < 1e-3 && d > -1e-3)) {
                        imag = true;
                        break;
                    }
                }
            if (imag) continue;
            opengv::essential_t E;
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) E(r, c) = Ei(r, c).real();
            out.push_back(E);
        }
        return out;
    }
    // the solutions generateRefinedModel chooses from: OpenGV's solver on all sample points (unit bearing vectors), for a re-weighted
    // step of the _WEIGHTS forms the rows scaled by w_i / |w| (fivept_nister_weight / fivept_stewenius_weight)
    opengv::essentials_t refine_solutions(const std::vector<unsigned int> &sample, unsigned numPoints, bool stewenius, bool use_weights,
                                          const double *weights) {
        Eigen::MatrixXd Q(numPoints, 9);
        double weightnorm = 0;
        if (use_weights) {
            for (unsigned i = 0; i < numPoints; i++) weightnorm += std::pow(weights[i], 2);
            weightnorm = std::sqrt(weightnorm);
        }
        for (unsigned i = 0; i < numPoints; i++) {
            // "computing the inverse transformation, so we simply invert the input here"
            opengv::bearingVector_t f = adapter_denorm->getBearingVector2(sample[i]);
            opengv::bearingVector_t fprime = adapter_denorm->getBearingVector1(sample[i]);
            Eigen::Matrix<double, 1, 9> row;
            row << f[0] * fprime[0], f[1] * fprime[0], f[2] * fprime[0], f[0] * fprime[1], f[1] * fprime[1], f[2] * fprime[1], f[0] * fprime[2],
                f[1] * fprime[2], f[2] * fprime[2];
            if (use_weights) row *= weights[i] / weightnorm;
            Q.row(i) = row;
        }
        opengv::essentials_t out;
        if (g_solver_oracle) {
            std::vector<double> rows((size_t)9 * numPoints);
            for (unsigned i = 0; i < numPoints; ++i)
                for (int k = 0; k < 9; ++k) rows[(size_t)9 * i + k] = Q(i, k);
            double Es[90];
            const int ns = oracle_run5point_rows(rows.data(), (int)numPoints, Es);
            for (int k = 0; k < ns; ++k) {
                opengv::essential_t E;
                for (int r = 0; r < 3; ++r)
                    for (int c = 0; c < 3; ++c) E(r, c) = Es[9 * k + 3 * r + c];
                out.push_back(E);
            }
            return out;
        }
        if (!use_weights) {  // the reference calls OpenGV's own entry points here (:669, :791)
            std::vector<int> indices;
            for (unsigned i = 0; i < numPoints; ++i) indices.push_back((int)sample[i]);
            if (stewenius) return real_solutions(opengv::relative_pose::fivept_stewenius(*adapter_denorm, indices));
            return opengv::relative_pose::fivept_nister(*adapter_denorm, indices);
        }
        Eigen::JacobiSVD<Eigen::MatrixXd> SVD(Q, Eigen::ComputeFullV);
        Eigen::Matrix<double, 9, 4> EE = SVD.matrixV().block(0, 5, 9, 4);
        if (stewenius) {
            opengv::complexEssentials_t complexEssentials;
            opengv::relative_pose::modules::fivept_stewenius_main(EE, complexEssentials);
            return real_solutions(complexEssentials);
        }
        opengv::relative_pose::modules::fivept_nister_main(EE, out);
        return out;
    }

    unsigned int generateMinimalSampleModels() override {
        std::vector<int> indices;
        for (unsigned i = 0; i < usac_min_sample_size_; ++i) indices.push_back((int)min_sample_[i]);
        essentials_denorm.clear();
        essentials_denorm = five_point(indices);
        unsigned nsols = (unsigned)essentials_denorm.size();
        {
            double v[8] = {(double)usac_results_.hyp_count_, (double)min_sample_[0], (double)min_sample_[1], (double)min_sample_[2],
                           (double)min_sample_[3], (double)min_sample_[4], (double)nsols, 0};
            emit(1, v, 7);
        }
        if (nsols > usac_max_solns_per_sample_) return 0;
        if (!g_native_order) {  // the order convention (see the header)
            std::vector<std::pair<double, int>> key(nsols);
            for (unsigned i = 0; i < nsols; ++i) key[i] = std::make_pair(order_key(essentials_denorm[i]), (int)i);
            std::stable_sort(key.begin(), key.end(), [](const std::pair<double, int> &a, const std::pair<double, int> &b) { return a.first < b.first; });
            opengv::essentials_t sorted;
            for (unsigned i = 0; i < nsols; ++i) sorted.push_back(essentials_denorm[key[i].second]);
            essentials_denorm = sorted;
        }
        essentials.clear();
        for (unsigned i = 0; i < nsols; ++i) {
            essentials.push_back(m_T2_trans_e_inv * essentials_denorm.at(i) * m_T1_e_inv);
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) {
                    models_[i][r * 3 + c] = essentials[i](r, c);
                    models_denorm_[i][r * 3 + c] = essentials_denorm[i](r, c);
                }
            double v[15];
            v[0] = usac_results_.hyp_count_, v[1] = i;
            for (int k = 0; k < 9; ++k) v[2 + k] = models_denorm_[i][k];
            emit(5, v, 11);
        }
        return nsols;
    }

    bool generateRefinedModel(std::vector<unsigned int> &sample, const unsigned int numPoints, bool weighted = false,
                              double *weights = nullptr) override {
        if (numPoints < usac_min_sample_size_) return false;
        bool ok = true;
        if (refineMethod == USACConfig::REFINE_WEIGHTS) {
            std::vector<double> A((size_t)numPoints * 9);
            double *dst = A.data();
            for (unsigned i = 0; i < numPoints; ++i) {
                const double *src = data_matrix_.data() + sample[i];
                for (unsigned j = 0; j < 9; ++j) {
                    *dst++ = weighted ? (*src) * weights[i] : *src;
                    src += usac_num_data_points_;
                }
            }
            double Cv[81], V[81], D[9];
            FTools::formCovMat(Cv, A.data(), numPoints, 9);
            MathTools::svdu1v(D, Cv, 9, V, 9);
            unsigned j = 0;
            for (unsigned i = 1; i < 9; ++i)
                if (D[i] < D[j]) j = i;
            for (unsigned i = 0; i < 9; ++i) models_[0][i] = V[9 * i + j];
            FTools::singulF(models_[0].data());
            double T2_F[9];
            MathTools::mmul(T2_F, m_T2_trans_, models_[0].data(), 3);
            MathTools::mmul(models_denorm_[0].data(), T2_F, m_T1_, 3);
            for (int r = 0; r < 3; r++)
                for (int c = 0; c < 3; c++) {
                    essentials[0](r, c) = models_[0][r * 3 + c];
                    essentials_denorm[0](r, c) = models_denorm_[0][r * 3 + c];
                }
        } else if (refineMethod == USACConfig::REFINE_NISTER || refineMethod == USACConfig::REFINE_NISTER_WEIGHTS ||
                   refineMethod == USACConfig::REFINE_STEWENIUS || refineMethod == USACConfig::REFINE_STEWENIUS_WEIGHTS) {
            const bool stew = refineMethod == USACConfig::REFINE_STEWENIUS || refineMethod == USACConfig::REFINE_STEWENIUS_WEIGHTS;
            const bool use_w = weighted && (refineMethod == USACConfig::REFINE_STEWENIUS_WEIGHTS || refineMethod == USACConfig::REFINE_NISTER_WEIGHTS);
            opengv::essentials_t Es = refine_solutions(sample, numPoints, stew, use_w, weights);
            if (!g_native_order && Es.size() > 1) {  // the order convention decides ties of the error sums only
                std::stable_sort(Es.begin(), Es.end(), [](const opengv::essential_t &a, const opengv::essential_t &b) { return order_key(a) < order_key(b); });
            }
            const size_t nsols = Es.size();
            opengv::essential_t E1;
            if (nsols > 1) {
                std::vector<double> errSums(nsols, 0);
                std::vector<std::vector<double>> pm(nsols, std::vector<double>(9));
                for (size_t j = 0; j < nsols; j++)
                    for (int r = 0; r < 3; r++)
                        for (int c = 0; c < 3; c++) pm[j][r * 3 + c] = Es[j](r, c);
                for (unsigned i = 0; i < usac_num_data_points_; ++i) {
                    if (usac_results_.inlier_flags_[i]) {
                        for (size_t j = 0; j < nsols; j++) errSums[j] += PoseTools::getSampsonError(pm[j], input_points_denorm_, i);
                        if ((i > 3) && (i % 4 == 0)) {
                            std::vector<double> t = errSums;
                            std::partial_sort(t.begin(), t.begin() + 2, t.end());
                            if (t[0] < 0.66 * t[1]) break;
                        }
                    }
                }
                E1 = Es[std::distance(errSums.begin(), std::min_element(errSums.begin(), errSums.end()))];
            } else if (nsols == 1)
                E1 = Es[0];
            else
                ok = false;
            if (ok) {
                essentials_denorm[0] = E1;
                essentials[0] = m_T2_trans_e_inv * E1 * m_T1_e_inv;
                for (int r = 0; r < 3; r++)
                    for (int c = 0; c < 3; c++) {
                        models_[0][r * 3 + c] = essentials[0](r, c);
                        models_denorm_[0][r * 3 + c] = essentials_denorm[0](r, c);
                    }
            }
        } else {
            return false;
        }
        double v[15];
        v[0] = usac_results_.hyp_count_, v[1] = numPoints, v[2] = weighted ? 1 : 0, v[3] = ok ? 1 : 0;
        for (int k = 0; k < 9; ++k) v[4 + k] = ok ? models_denorm_[0][k] : 0.0;
        emit(3, v, 13);
        return ok;
    }

    bool validateSample() override {
        int j, k, i;
        const double *ip = input_points_.data();
        for (i = 0; i < (int)usac_min_sample_size_; i++) {
            for (j = 0; j < i; j++) {
                const double pix = ip[min_sample_[i] * 6] / ip[min_sample_[i] * 6 + 2], piy = ip[min_sample_[i] * 6 + 1] / ip[min_sample_[i] * 6 + 2];
                const double pjx = ip[min_sample_[j] * 6] / ip[min_sample_[j] * 6 + 2], pjy = ip[min_sample_[j] * 6 + 1] / ip[min_sample_[j] * 6 + 2];
                const double dx1 = pjx - pix, dy1 = pjy - piy;
                for (k = 0; k < j; k++) {
                    const double pkx = ip[min_sample_[k] * 6] / ip[min_sample_[k] * 6 + 2], pky = ip[min_sample_[k] * 6 + 1] / ip[min_sample_[k] * 6 + 2];
                    const double dx2 = pkx - pix, dy2 = pky - piy;
                    if (fabs(dx2 * dy1 - dy2 * dx1) <= FLT_EPSILON * (fabs(dx1) + fabs(dy1) + fabs(dx2) + fabs(dy2))) break;
                }
                if (k < j) break;
            }
            if (j < i) break;
        }
        const bool ok = i >= (int)usac_min_sample_size_ - 1;
        if (!ok) {
            double v[7] = {(double)usac_results_.hyp_count_, (double)min_sample_[0], (double)min_sample_[1], (double)min_sample_[2],
                           (double)min_sample_[3], (double)min_sample_[4], -1.0};
            emit(1, v, 7);
        }
        return ok;
    }

    bool validateModel(unsigned int modelIndex) override {
        double e[3], sig1, sig2;
        bool ok = true;
        FTools::computeEpipole(e, models_[modelIndex].data());
        sig1 = FTools::getOriSign(models_[modelIndex].data(), e, input_points_.data() + 6 * min_sample_[0]);
        for (unsigned i = 1; i < min_sample_.size(); ++i) {
            sig2 = FTools::getOriSign(models_[modelIndex].data(), e, input_points_.data() + 6 * min_sample_[i]);
            if (sig1 * sig2 < 0) {
                ok = false;
                break;
            }
        }
        if (!ok) {
            double v[3] = {(double)usac_results_.hyp_count_, (double)modelIndex, 0};
            emit(6, v, 2);
        }
        return ok;
    }

    bool evaluateModel(unsigned int modelIndex, unsigned int *numInliers, unsigned int *numPointsTested) override {
        const double *model = models_denorm_[modelIndex].data();
        auto current_err_array = err_ptr_[0];
        bool good_flag = true;
        double lambdaj, lambdaj_1 = 1.0;
        *numInliers = 0;
        *numPointsTested = 0;
        const unsigned start_index = eval_pool_index_;
        unsigned i = 0;
        for (; i < usac_num_data_points_; ++i) {
            if (eval_pool_index_ > usac_num_data_points_ - 1) eval_pool_index_ = 0;
            const unsigned pt_index = evaluation_pool_[eval_pool_index_];
            ++eval_pool_index_;
            const double temp_err = PoseTools::getSampsonError(models_denorm_[modelIndex], input_points_denorm_, pt_index);
            *(current_err_array + pt_index) = temp_err;
            if (temp_err < usac_inlier_threshold_) ++(*numInliers);
            if (usac_verif_method_ == USACConfig::VERIF_SPRT) {
                if (temp_err < usac_inlier_threshold_)
                    lambdaj = lambdaj_1 * (sprt_delta_ / sprt_epsilon_);
                else
                    lambdaj = lambdaj_1 * ((1 - sprt_delta_) / (1 - sprt_epsilon_));
                if (lambdaj <= DBL_EPSILON) lambdaj = DBL_EPSILON * 10;
                if (lambdaj > decision_threshold_sprt_) {
                    good_flag = false;
                    *numPointsTested = i + 1;
                    break;
                } else
                    lambdaj_1 = lambdaj;
            }
        }
        if (good_flag) *numPointsTested = usac_num_data_points_;
        (void)model;
        double v[12] = {(double)usac_results_.hyp_count_, (double)modelIndex, (double)start_index, (double)*numInliers,
                        (double)*numPointsTested, good_flag ? 1.0 : 0.0, sprt_delta_, sprt_epsilon_, decision_threshold_sprt_,
                        usac_inlier_threshold_, (double)usac_results_.num_local_optimizations_, 0};
        emit(2, v, 11);
        return good_flag;
    }

    // ---- degeneracy tests and model upgrade (check_degeneracy; enableHDegen = false, enableUpgradeDegenPose = true) ------------------
    enum { DG_NOT_FOUND = 0x1, DG_H = 0x2, DG_ROT_TRANS = 0x4, DG_NO_MOT = 0x8, DG_UPGRADE = 0x10 };
    static bool near_zero(double d) { return (d < 1e-3) && (d > -1e-3); }  // poselib::nearZero (pose_helper.h:82-87)

    void testSolutionDegeneracy(bool *degenerateModel, bool *upgradeModel) override {
        *degenerateModel = false, *upgradeModel = false;
        degeneracyType = DG_H;  // :1346, taken whenever the homography test is off
        test_rotation(degenerateModel);
        if (degeneracyType & DG_UPGRADE) *upgradeModel = true;
        if (degeneracyType == (unsigned)(DG_ROT_TRANS | DG_UPGRADE)) test_no_motion(degenerateModel);
        double v[7] = {(double)usac_results_.hyp_count_, *degenerateModel ? 1.0 : 0.0, *upgradeModel ? 1.0 : 0.0, (double)degeneracyType,
                       (double)usac_results_.degen_inlier_count_rot, (double)usac_results_.degen_inlier_count_noMot,
                       (double)usac_results_.best_inlier_count_};
        emit(7, v, 7);
    }

    void test_rotation(bool *degenerateModel) {
        static const unsigned pair_of[20] = {0, 1, 0, 2, 0, 3, 0, 4, 1, 2, 1, 3, 1, 4, 2, 3, 2, 4, 3, 4};
        static const unsigned rest_of[30] = {2, 3, 4, 1, 3, 4, 1, 2, 4, 1, 2, 3, 0, 3, 4, 0, 2, 4, 0, 2, 3, 0, 1, 4, 0, 1, 3, 0, 1, 2};
        const unsigned n = usac_num_data_points_;
        opengv::rotation_t R_;
        std::vector<int> sample(5);
        std::vector<unsigned int> test(3);
        std::vector<double> errs;
        for (unsigned i = 0; i < 10; ++i) {
            for (unsigned j = 0; j < 2; ++j) sample[j] = (int)min_sample_[pair_of[2 * i + j]];
            R_ = opengv::relative_pose::twopt_rotationOnly(*adapter_denorm, sample);
            for (unsigned j = 0; j < 3; ++j) test[j] = min_sample_[rest_of[3 * i + j]];
            unsigned num_inliers = PoseTools::getRotError(test, 3, errs, adapter_denorm, R_, poseDegenTheshold);
            unsigned count1 = 2;
            for (unsigned j = 0; j < 3; ++j)
                if (errs[j] < poseDegenTheshold) sample[count1++] = test[j];
            if (num_inliers == 0) continue;
            num_inliers = PoseTools::getRotError(evaluation_pool_, n, errs, adapter_denorm, R_, poseDegenTheshold);
            const unsigned first_count = num_inliers;
            if (num_inliers < 2) continue;
            unsigned count = 0;
            std::vector<int> inlier_sample(num_inliers);
            for (unsigned j = 0; j < n; ++j)
                if (errs[j] < poseDegenTheshold) {
                    inlier_sample[count++] = (int)evaluation_pool_[j];
                    if (count1 < 5) sample[count1++] = (int)evaluation_pool_[j];
                }
            R_ = opengv::relative_pose::rotationOnly(*adapter_denorm, inlier_sample);
            num_inliers = PoseTools::getRotError(evaluation_pool_, n, errs, adapter_denorm, R_, poseDegenTheshold);
            double v[5] = {(double)usac_results_.hyp_count_, (double)i, (double)first_count, (double)num_inliers, 0};
            if (num_inliers < usac_results_.best_inlier_count_ / 5) {
                emit(8, v, 5);
                continue;
            }
            *degenerateModel = true;
            if (degeneracyType != (degeneracyType & (DG_UPGRADE | DG_ROT_TRANS))) degeneracyType = DG_ROT_TRANS;
            if (num_inliers > usac_results_.degen_inlier_count_rot) {
                degeneracyType |= DG_UPGRADE;
                count = 0;
                inlier_sample.resize(num_inliers);
                for (unsigned j = 0; j < n; ++j)
                    if (errs[j] < poseDegenTheshold) inlier_sample[count++] = evaluation_pool_[j];
                R_ = opengv::relative_pose::rotationOnly(*adapter_denorm, inlier_sample);
                usac_results_.degen_inlier_count_rot = num_inliers;
                for (unsigned r = 0; r < 3; ++r)
                    for (unsigned c = 0; c < 3; ++c) degen_final_model_params_rot[r * 3 + c] = R_(r, c);
                R_eigen_degen = R_;
                for (unsigned j = 0; j < n; ++j) {
                    const bool in = errs[j] < poseDegenTheshold;
                    usac_results_.degen_inlier_flags_rot[evaluation_pool_[j]] = in ? 1 : 0;
                    degen_outlier_flags_rot[evaluation_pool_[j]] = in ? 0 : 1;
                }
                degen_sample_rot = sample;
                v[4] = 1;
            }
            emit(8, v, 5);
        }
    }

    void test_no_motion(bool *degenerateModel) {
        if (usac_results_.degen_inlier_count_noMot > 0) return;
        const unsigned n = usac_num_data_points_;
        std::vector<unsigned int> test(5);
        std::vector<double> errs;
        for (unsigned j = 0; j < 5; ++j) test[j] = min_sample_[j];
        unsigned num_inliers = PoseTools::getNoMotError(test, 5, errs, adapter_denorm, poseDegenTheshold);
        degen_sample_noMot.clear();
        for (unsigned j = 0; j < 5; ++j)
            if (errs[j] < poseDegenTheshold) degen_sample_noMot.push_back((int)test[j]);
        if (num_inliers == 0) return;
        num_inliers = PoseTools::getNoMotError(evaluation_pool_, n, errs, adapter_denorm, poseDegenTheshold);
        if (num_inliers < usac_results_.best_inlier_count_ / 5) return;
        *degenerateModel = true;
        const bool dominant = (double)num_inliers > 0.7 * (double)usac_results_.degen_inlier_count_rot;
        if (dominant)
            if (degeneracyType != (degeneracyType & (DG_UPGRADE | DG_NO_MOT))) degeneracyType = DG_NO_MOT;
        if (num_inliers > usac_results_.degen_inlier_count_noMot) {
            if (dominant) degeneracyType |= DG_UPGRADE;
            usac_results_.degen_inlier_count_noMot = num_inliers;
            for (unsigned j = 0; j < n; ++j) {
                if (errs[j] < poseDegenTheshold) {
                    usac_results_.degen_inlier_flags_noMot[evaluation_pool_[j]] = 1;
                    degen_outlier_flags_noMot[evaluation_pool_[j]] = 0;
                    if (degen_sample_noMot.size() < 5) degen_sample_noMot.push_back((int)evaluation_pool_[j]);
                } else {
                    degen_outlier_flags_noMot[evaluation_pool_[j]] = 1;
                    usac_results_.degen_inlier_flags_noMot[evaluation_pool_[j]] = 0;
                }
            }
        }
    }

    // evaluateModelTrans (:1264-1327): reprojection error of the midpoint triangulation under (I, t), sequential test as evaluateModel
    bool evaluate_translation(const opengv::translation_t &model, unsigned int *numInliers, unsigned int *numPointsTested) {
        auto current_err_array = err_ptr_[0];
        bool good_flag = true;
        double lambdaj, lambdaj_1 = 1.0;
        *numInliers = 0, *numPointsTested = 0;
        const unsigned start_index = eval_pool_index_;
        opengv::rotation_t rotation = opengv::rotation_t::Identity();
        adapter_denorm->sett12(model);
        adapter_denorm->setR12(rotation);
        opengv::transformation_t inverseSolution;
        inverseSolution.block<3, 3>(0, 0) = rotation.transpose();
        inverseSolution.col(3) = -inverseSolution.block<3, 3>(0, 0) * model;
        Eigen::Matrix<double, 4, 1> p_hom;
        p_hom[3] = 1.0;
        for (unsigned i = 0; i < usac_num_data_points_; ++i) {
            if (eval_pool_index_ > usac_num_data_points_ - 1) eval_pool_index_ = 0;
            const unsigned pt_index = evaluation_pool_[eval_pool_index_];
            ++eval_pool_index_;
            p_hom.block<3, 1>(0, 0) = opengv::triangulation::triangulate2(*adapter_denorm, pt_index);
            opengv::bearingVector_t reprojection1 = p_hom.block<3, 1>(0, 0);
            opengv::bearingVector_t reprojection2 = inverseSolution * p_hom;
            reprojection1 = reprojection1 / reprojection1.norm();
            reprojection2 = reprojection2 / reprojection2.norm();
            opengv::bearingVector_t f1 = adapter_denorm->getBearingVector1(pt_index);
            opengv::bearingVector_t f2 = adapter_denorm->getBearingVector2(pt_index);
            const double reprojError1 = 1.0 - (f1.transpose() * reprojection1);
            const double reprojError2 = 1.0 - (f2.transpose() * reprojection2);
            const double temp_err = reprojError1 + reprojError2;
            *(current_err_array + pt_index) = temp_err;
            if (temp_err < poseDegenTheshold) ++(*numInliers);
            if (usac_verif_method_ == USACConfig::VERIF_SPRT) {
                if (temp_err < poseDegenTheshold)
                    lambdaj = lambdaj_1 * (sprt_delta_ / sprt_epsilon_);
                else
                    lambdaj = lambdaj_1 * ((1 - sprt_delta_) / (1 - sprt_epsilon_));
                if (lambdaj <= DBL_EPSILON) lambdaj = DBL_EPSILON * 10;
                if (lambdaj > decision_threshold_sprt_) {
                    good_flag = false;
                    *numPointsTested = i + 1;
                    break;
                } else
                    lambdaj_1 = lambdaj;
            }
        }
        if (good_flag) *numPointsTested = usac_num_data_points_;
        double v[12] = {(double)usac_results_.hyp_count_, -1.0, (double)start_index, (double)*numInliers, (double)*numPointsTested,
                        good_flag ? 1.0 : 0.0, sprt_delta_, sprt_epsilon_, decision_threshold_sprt_, poseDegenTheshold,
                        (double)usac_results_.num_local_optimizations_, 0};
        emit(2, v, 11);
        return good_flag;
    }

    // E = [t / |t|]_x R the way poselib::getEfromRT forms it (pose_helper.cpp:785-805: the vector times the reciprocal of its norm)
    static void e_from_rt(const opengv::rotation_t &R, const opengv::translation_t &t, double *E) {
        double nrm = 0;
        for (int i = 0; i < 3; ++i) nrm += t[i] * t[i];
        nrm = std::sqrt(nrm);
        const double s = 1.0 / nrm;
        const double a = t[0] * s, b = t[1] * s, c = t[2] * s;
        const double S[9] = {0, -c, b, c, 0, -a, -b, a, 0};
        for (int r = 0; r < 3; ++r)
            for (int k = 0; k < 3; ++k) {
                double acc = 0;
                for (int m = 0; m < 3; ++m) acc += S[3 * r + m] * R(m, k);
                E[3 * r + k] = acc;
            }
    }

    template <class It>
    unsigned count_on(It err, const std::vector<unsigned int> &idx, double limit, unsigned *untouched) const {
        unsigned c = 0, u = 0;
        for (auto j : idx) {
            if (err[j] < limit)
                ++c;
            else if (std::round(err[j] - DBL_MAX) == 0)
                ++u;
        }
        *untouched = u;
        return c;
    }

    unsigned int upgradeDegenerateModel() override {
        const unsigned n = usac_num_data_points_;
        unsigned best_upgrade_inliers = usac_results_.best_inlier_count_;
        unsigned best_upgrade_inliers_rot = usac_results_.degen_inlier_count_rot;
        unsigned best_upgrade_inliers_trans = usac_results_.degen_inlier_count_trans;
        unsigned num_outliers = n - usac_results_.degen_inlier_count_;
        if (num_outliers < 2) return 0;
        unsigned tried = 0, branch = 0;
        if (degeneracyType & DG_UPGRADE) {
            if (degeneracyType & DG_NO_MOT) {  // no motion -> translation only
                branch = 1;
                num_outliers = n - usac_results_.degen_inlier_count_noMot;
                if (num_outliers < 1) return 0;
                std::vector<unsigned int> outlier_indices(num_outliers);
                unsigned count = 0;
                for (unsigned i = 0; i < n; ++i)
                    if (degen_outlier_flags_noMot[i]) outlier_indices[count++] = i;
                std::vector<unsigned int> outlier_sample(1);
                std::fill(err_ptr_[0], err_ptr_[0] + n, DBL_MAX);
                auto current_err_array = err_ptr_[0];
                opengv::translation_t t_;
                const unsigned size_noMot = (unsigned)degen_sample_noMot.size();
                for (unsigned i = 0; i < degen_max_upgrade_samples_noMot_trans; ++i) {
                    ++tried;
                    generateUniformRandomSample(num_outliers, 1, &outlier_sample);
                    std::vector<int> index(2);
                    index[0] = (int)outlier_indices[outlier_sample[0]];
                    const unsigned pick = std::rand() % size_noMot;
                    index[1] = degen_sample_noMot[pick];
                    t_ = opengv::relative_pose::twopt(*adapter_denorm, false, index);
                    {
                        double v[12] = {(double)usac_results_.hyp_count_, 1.0, (double)i, t_[0], t_[1], t_[2], 0, 0, 0, 0, 0, 0};
                        emit(10, v, 12);
                    }
                    if (near_zero(t_.norm() * 100)) continue;
                    unsigned num_inliers, num_pts_tested;
                    evaluate_translation(t_, &num_inliers, &num_pts_tested);
                    if (num_inliers > best_upgrade_inliers_trans) {
                        if (num_inliers > best_upgrade_inliers ||
                            (near_zero(final_model_params_[0] * 100) && near_zero(final_model_params_[4] * 100) && near_zero(final_model_params_[8] * 100))) {
                            e_from_rt(opengv::rotation_t::Identity(), t_, models_denorm_[0].data());
                            storeSolution(0, num_inliers);
                            best_upgrade_inliers = num_inliers;
                            if (size_noMot > 3) {
                                unsigned k = 0;
                                for (size_t j = 0; j < 3; j++) {
                                    if (degen_sample_noMot[k] == index[1]) {
                                        j--;
                                        k++;
                                        continue;
                                    }
                                    min_sample_[j] = degen_sample_noMot[k];
                                    k++;
                                }
                                min_sample_[3] = index[1];
                                min_sample_[4] = index[0];
                            }
                            usac_results_.degen_inlier_count_trans = num_inliers;
                        }
                        best_upgrade_inliers_trans = num_inliers;
                        unsigned untouched = 0;
                        count = count_on(current_err_array, outlier_indices, poseDegenTheshold, &untouched);
                        const unsigned num_samples = updateStandardStopping(count, num_outliers - untouched, 1);
                        if (num_samples < degen_max_upgrade_samples_noMot_trans) degen_max_upgrade_samples_noMot_trans = num_samples;
                    }
                }
            } else {  // rotation -> rotation + translation
                branch = 2;
                num_outliers = n - usac_results_.degen_inlier_count_rot;
                if (num_outliers < 3) return 0;
                std::vector<unsigned int> outlier_indices(num_outliers);
                unsigned count = 0;
                for (unsigned i = 0; i < n; ++i)
                    if (degen_outlier_flags_rot[i]) outlier_indices[count++] = i;
                std::vector<unsigned int> outlier_sample(3);
                std::fill(err_ptr_[0], err_ptr_[0] + n, DBL_MAX);
                auto current_err_array = err_ptr_[0];
                opengv::rotation_t R_;
                opengv::translation_t t_;
                for (unsigned i = 0; i < degen_max_upgrade_samples_rot; ++i) {
                    ++tried;
                    generateUniformRandomSample(num_outliers, 3, &outlier_sample);
                    std::vector<int> index(5);
                    for (unsigned j = 0; j < 3; j++) index[j] = (int)outlier_indices[outlier_sample[j]];
                    index[3] = degen_sample_rot[0];
                    index[4] = degen_sample_rot[1];
                    opengv::eigensolverOutput_t eig_out;
                    adapter_denorm->setR12(R_eigen_degen);
                    eig_out.rotation = R_eigen_degen;
                    R_ = opengv::relative_pose::eigensolver(*adapter_denorm, index, eig_out);
                    t_ = eig_out.translation;
                    if (g_eigvec_smallest) {  // see the flag: the eigenvector of the SMALLEST eigenvalue, sign by eigensolver's own rule
                        int k = 0;
                        for (int q = 1; q < 3; ++q)
                            if (eig_out.eigenvalues[q] < eig_out.eigenvalues[k]) k = q;
                        const double a = eig_out.eigenvalues[(k + 1) % 3], b = eig_out.eigenvalues[(k + 2) % 3];
                        t_ = std::sqrt(a * a + b * b) * eig_out.eigenvectors.col(k);
                        opengv::bearingVector_t f1 = adapter_denorm->getBearingVector1(index[0]);
                        opengv::bearingVector_t f2 = R_ * adapter_denorm->getBearingVector2(index[0]);
                        if ((f1 - f2).dot(t_) < 0.0) t_ = -t_;
                    }
                    if (near_zero(t_.norm() * 100)) {
                        double v[12] = {(double)usac_results_.hyp_count_, 2.0, (double)i, 0, 0, 0, 0, 0, 0, 0, 0, 0};
                        emit(10, v, 12);
                        continue;
                    }
                    t_ /= t_.norm();
                    e_from_rt(R_, t_, models_denorm_[0].data());
                    {
                        double v[12] = {(double)usac_results_.hyp_count_, 2.0, (double)i};
                        for (int k = 0; k < 9; ++k) v[3 + k] = models_denorm_[0][k];
                        emit(10, v, 12);
                    }
                    unsigned num_inliers, num_pts_tested;
                    evaluateModel(0, &num_inliers, &num_pts_tested);
                    if (num_inliers > best_upgrade_inliers_rot) {
                        count = 0;
                        for (unsigned j = 2; j < 5; j++) degen_sample_rot[j] = index[count++];
                        if (num_inliers > best_upgrade_inliers ||
                            (near_zero(final_model_params_[0] * 100) && near_zero(final_model_params_[4] * 100) && near_zero(final_model_params_[8] * 100))) {
                            storeSolution(0, num_inliers);
                            best_upgrade_inliers = num_inliers;
                            for (size_t j = 0; j < 5; j++) min_sample_[j] = (unsigned int)degen_sample_rot[j];
                        }
                        best_upgrade_inliers_rot = num_inliers;
                        unsigned untouched = 0;
                        count = count_on(current_err_array, outlier_indices, usac_inlier_threshold_, &untouched);
                        const unsigned num_samples = updateStandardStopping(count, num_outliers - untouched, 1);
                        if (num_samples < degen_max_upgrade_samples_rot) degen_max_upgrade_samples_rot = num_samples;
                    }
                }
            }
        }
        double v[4] = {(double)usac_results_.hyp_count_, (double)branch, (double)tried, (double)best_upgrade_inliers};
        emit(9, v, 4);
        return best_upgrade_inliers;
    }

    void initDegeneracy(double rot_ratio) {
        const unsigned n = usac_num_data_points_;
        degeneracyType = DG_NOT_FOUND;
        degen_max_upgrade_samples_rot = degen_max_upgrade_samples_noMot_trans = 8000;  // usac_estimations.cpp:446
        degen_final_model_params_rot.assign(9, 0.0);
        degen_outlier_flags_rot.assign(n, 0), degen_outlier_flags_noMot.assign(n, 0);
        poseDegenTheshold = 1.0 - std::cos(std::atan(rot_ratio));  // EssentialMatEstimator.h:349
        R_eigen_degen = opengv::rotation_t::Identity();
    }
    unsigned degeneracyType = DG_NOT_FOUND;
    std::vector<double> degen_final_model_params_rot;

    void findWeights(unsigned int modelIndex, const std::vector<unsigned int> &inliers, unsigned int numInliers, double *weights) override {
        if (refineMethod == USACConfig::REFINE_STEWENIUS_WEIGHTS || refineMethod == USACConfig::REFINE_NISTER_WEIGHTS) {  // :2404-2428
            opengv::essential_t modele = essentials_denorm[modelIndex];
            double pseudohuberth = std::sqrt(usac_inlier_threshold_) / 50.0;
            for (unsigned i = 0; i < numInliers; ++i) {
                opengv::bearingVector_t f = adapter_denorm->getBearingVector2(inliers[i]);
                opengv::bearingVector_t fprime = adapter_denorm->getBearingVector1(inliers[i]);
                // computePseudoHuberWeight(f, fprime, modele, pseudohuberth): SampsonL1_Eigen, then costPseudoHuber
                Eigen::Vector3d xpE = fprime.transpose() * modele;
                const double num = xpE.dot(f);
                Eigen::Vector3d Ex1 = modele * f;
                const double a = Ex1(0) * Ex1(0), b = Ex1(1) * Ex1(1), c = xpE(0) * xpE(0), d = xpE(1) * xpE(1);
                const double denom1 = 1 / (std::sqrt(a + b + c + d) + 1e-8);
                const double thresh = pseudohuberth, dd = num * denom1;
                const double b_sq = thresh * thresh, d_abs = std::abs(dd) + 1e-12;
                const double q = d_abs / thresh;
                const double weight = std::sqrt(2 * b_sq * (std::sqrt(1 + q * q) - 1)) / d_abs;
                weights[i] = denom1 * weight;
            }
            return;
        }
        if (refineMethod != USACConfig::REFINE_WEIGHTS) return;
        const double *model = models_[modelIndex].data();
        for (unsigned i = 0; i < numInliers; ++i) {
            const double *pt = input_points_.data() + 6 * inliers[i];
            const double rxc = model[0] * pt[3] + model[3] * pt[4] + model[6];
            const double ryc = model[1] * pt[3] + model[4] * pt[4] + model[7];
            const double rx = model[0] * pt[0] + model[1] * pt[1] + model[2];
            const double ry = model[3] * pt[0] + model[4] * pt[1] + model[5];
            weights[i] = 1 / sqrt(rxc * rxc + ryc * ryc + rx * rx + ry * ry);
        }
    }

    void storeModel(unsigned int modelIndex, unsigned int numInliers) override {
        for (unsigned i = 0; i < 9; ++i) final_model_params_[i] = models_denorm_[modelIndex][i];
        double v[3] = {(double)usac_results_.hyp_count_, (double)modelIndex, (double)numInliers};
        emit(4, v, 3);
    }

    double sprtDelta() const { return sprt_delta_; }
    double sprtEpsilon() const { return sprt_epsilon_; }
    const std::vector<unsigned int> &pool() const { return evaluation_pool_; }

   private:
    double *input_points_denorm_ = nullptr;
    std::vector<double> input_points_, data_matrix_;
    double m_T1_[9], m_T2_[9], m_T2_trans_[9];
    Eigen::Matrix3d m_T1_e, m_T2_e, m_T2_trans_e, m_T1_e_inv, m_T2_e_inv, m_T2_trans_e_inv;
    opengv::essentials_t essentials, essentials_denorm;
    std::vector<std::vector<double>> models_, models_denorm_;
    opengv::bearingVectors_t bearing1_, bearing2_;
    std::shared_ptr<opengv::relative_pose::CentralRelativeAdapter> adapter_denorm;
    USACConfig::RefineAlgorithm refineMethod;
    double poseDegenTheshold = 0;
    unsigned degen_max_upgrade_samples_rot = 0, degen_max_upgrade_samples_noMot_trans = 0;
    std::vector<unsigned int> degen_outlier_flags_rot, degen_outlier_flags_noMot;
    std::vector<int> degen_sample_rot, degen_sample_noMot;
    opengv::rotation_t R_eigen_degen;
};

int main(int argc, char **argv) {
    if (argc < 3) return 1;
    for (int a = 3; a < argc; ++a)
        if (!std::strcmp(argv[a], "--native-order")) g_native_order = true;
        else if (!std::strcmp(argv[a], "--solver-oracle")) g_solver_oracle = true;
        else if (!std::strcmp(argv[a], "--stewenius")) g_stewenius = true;
        else if (!std::strcmp(argv[a], "--eigvec-smallest")) g_eigvec_smallest = true;
    FILE *f = fopen(argv[1], "rb");
    if (!f) return 2;
    int32_t ih[8];
    double dh[8];
    if (fread(ih, 4, 8, f) != 8 || fread(dh, 8, 8, f) != 8) return 2;
    const int n = ih[0];
    std::vector<double> pts((size_t)n * 4);
    if (fread(pts.data(), 8, pts.size(), f) != pts.size()) return 2;
    std::vector<unsigned int> sorted;
    if (ih[3]) {
        sorted.resize(n);
        if (fread(sorted.data(), 4, n, f) != (size_t)n) return 2;
    }
    fclose(f);

    // estimateEssentialMatUsac (poselib/source/usac/usac_estimations.cpp:283-470) with the seed exposed (the reference: time(nullptr))
    srand((unsigned)ih[1]);
    std::vector<double> pointData((size_t)6 * n);
    for (int i = 0; i < n; ++i) {
        pointData[6 * i] = pts[4 * i], pointData[6 * i + 1] = pts[4 * i + 1], pointData[6 * i + 2] = 1.0;
        pointData[6 * i + 3] = pts[4 * i + 2], pointData[6 * i + 4] = pts[4 * i + 3], pointData[6 * i + 5] = 1.0;
    }
    USACConfig::Common c_com;
    USACConfig::Losac c_lo;
    USACConfig::Prosac c_pro;
    USACConfig::Sprt c_sprt;
    USACConfig::EssMat c_ess;
    c_com.confThreshold = dh[6];
    c_com.minSampleSize = 5;
    c_com.inlierThreshold = dh[0];
    c_com.maxHypotheses = ih[4];
    c_com.maxSolutionsPerSample = 10;
    c_com.numDataPoints = n;
    c_com.prevalidateSample = true;
    c_com.prevalidateModel = true;
    c_com.testDegeneracy = ih[5] != 0;
    c_com.testDegeneracyLOSAC = (ih[5] & 2) != 0;  // the reference: with the 8-point refinements (usac_estimations.cpp:368-375)
    c_com.randomSamplingMethod = sorted.empty() ? USACConfig::SAMP_UNIFORM : USACConfig::SAMP_PROSAC;
    c_com.verifMethod = USACConfig::VERIF_SPRT;
    c_com.localOptMethod = USACConfig::LO_LOSAC;
    c_lo.innerRansacRepetitions = 5;
    c_lo.innerSampleSize = 14;
    c_lo.thresholdMultiplier = 2.0;
    c_lo.numStepsIterative = 4;
    if (!sorted.empty()) {
        c_pro.beta = dh[1];
        c_pro.maxSamples = 1000;
        c_pro.minStopLen = 20;
        c_pro.nonRandConf = 0.99;
        c_pro.sortedPointIndices = sorted.data();
    }
    c_sprt.delta = dh[2];
    c_sprt.epsilon = dh[3];
    c_sprt.mS = dh[4];
    c_sprt.tM = dh[5];
    c_ess.refineMethod = ih[2] == 6   ? USACConfig::REFINE_NISTER
                         : ih[2] == 7 ? USACConfig::REFINE_NISTER_WEIGHTS
                         : ih[2] == 4 ? USACConfig::REFINE_STEWENIUS
                         : ih[2] == 5 ? USACConfig::REFINE_STEWENIUS_WEIGHTS
                                      : USACConfig::REFINE_WEIGHTS;
    c_ess.used_estimator = USACConfig::ESTIM_NISTER;
    ConfigParamsEssential cfg(c_com, c_pro, c_sprt, c_lo, c_ess, false);
    std::unique_ptr<RefEssential> est(new RefEssential);
    est->initParamsUSAC(cfg);
    est->initDataUSAC(cfg);
    est->initProblem(cfg, pointData.data());
    if (ih[5]) est->initDegeneracy(dh[7]);
    const bool ok = est->solve();

    FILE *out = fopen(argv[2], "wb");
    const int32_t ne = (int32_t)(g_events.size() / 16);
    fwrite(&ne, 4, 1, out);
    fwrite(g_events.data(), 8, g_events.size(), out);
    // final: [ok, hyp_count, model_count, rejected_samples, rejected_models, best_inliers, points_verified, num_lo, sprt_delta_res,
    //         sprt_epsilon_res, delta_at_end, epsilon_at_end], E[9], n flags (as doubles), n pool entries
    const UsacResults &r = est->usac_results_;
    double fin[12] = {ok ? 1.0 : 0.0, (double)r.hyp_count_, (double)r.model_count_, (double)r.rejected_sample_count_,
                      (double)r.rejected_model_count_, (double)r.best_inlier_count_, (double)r.total_points_verified_,
                      (double)r.num_local_optimizations_, r.sprt_delta_, r.sprt_epsilon_, est->sprtDelta(), est->sprtEpsilon()};
    fwrite(fin, 8, 12, out);
    fwrite(est->final_model_params_.data(), 8, 9, out);
    std::vector<double> flags(n, 0.0), pool(n, 0.0);
    for (int i = 0; i < n && i < (int)r.inlier_flags_.size(); ++i) flags[i] = r.inlier_flags_[i];
    for (int i = 0; i < n && i < (int)est->pool().size(); ++i) pool[i] = est->pool()[i];
    fwrite(flags.data(), 8, n, out);
    fwrite(pool.data(), 8, n, out);
    if (ih[5]) {  // degeneracy results: [count_rot, count_noMot, type, 0], R_degenerate[9], n rotation flags, n no-motion flags
        double dg[4] = {(double)r.degen_inlier_count_rot, (double)r.degen_inlier_count_noMot, (double)est->degeneracyType, 0};
        fwrite(dg, 8, 4, out);
        fwrite(est->degen_final_model_params_rot.data(), 8, 9, out);
        std::vector<double> fr(n, 0.0), fn(n, 0.0);
        for (int i = 0; i < n; ++i) fr[i] = r.degen_inlier_flags_rot[i], fn[i] = r.degen_inlier_flags_noMot[i];
        fwrite(fr.data(), 8, n, out);
        fwrite(fn.data(), 8, n, out);
    }
    fclose(out);
    return 0;
}
