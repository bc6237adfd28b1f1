// pair_batch_impl.h -- a BATCH of image pairs through the whole hot path with the pair as a grid dimension (BASELINE config 5).
// Included by ransac_5pt.hip inside namespace mlpl.
//
// mlpl_pair_pose_dev runs one pair as ~14 launches and two host hops; its RANSAC half is latency-bound (1000 hypotheses do not fill a
// quarter of the chip), so a rank had to keep several pairs in flight from host threads.  Here B pairs share every launch:
//   matching          mlpl_match_hamming_dev(batch = B)                                          (one launch chain, as before)
//   hop 1             the B match counts (they size the sample tables: getSubset draws rand() % count, modelest.cpp:585)
//   gather + pack     blockIdx.y = pair
//   RANSAC passes     the first 324 iterations of EVERY pair as one pass, then the remaining iterations of the pairs the adaptive bound
//                     has not stopped (the reference's 0.999 / 1000 setting stops a 50 % inlier pair after ~220): solver and root kernels
//                     over all (pair, hypothesis) samples at once (global point indices: the solver does not know about pairs), a dense
//                     model list per pair, the counting kernel with blockIdx.z = pair, candidate / replay kernels with one workgroup per
//                     pair.  One host hop per pass (the B replay states: which pairs go on).
//   cheirality        decomposition, four triangulations and the reference's candidate choice per pair on the device
//   last hop          the B results
// Per pair the kernels are the single-pair kernels with per-slot arguments (PairSlot), so every pair's record is what
// mlpl_pair_pose_dev returns for it (tests/test_gpu_batch.py: byte-identical E, R, t, counts).

namespace {

constexpr int kBatchFirstPass = 324;   // multiple of kHypPerWave; above the stopping point of pairs with >= ~45 % inliers at 0.999
constexpr int kBatchPassMax = 1026;    // hypotheses per pair and pass (multiple of kHypPerWave)
constexpr int kBatchPairsPerCall = 256;  // pairs per internal batch (workspace ~2.6 MB per pair; 512 pairs in one call: 10.5 ms at 128, 9.8-9.9 at 256, 10.0-10.2 at 512)
constexpr int kSeqBatchPairsPerCall = 512;  // ... of the entries with a sequential estimator (USAC, ARRSAC): their cohorts of runs (usac_impl.h) want the
                                            // whole batch at hand (512 pairs: 24 ms at 256 per internal batch, 27 at 128); workspace ~0.7 MB per pair

// getSubset (modelest.cpp:567-610) on the device: one wave per slot turns the pair's raw rand() stream into the slot's samples --
// five draws `rand() % n` per sample, a draw that repeats an index of the sample is redrawn.  The stream position of sample i + 1
// depends on the redraws of sample i, so the wave speculates: lane l parses sample i0 + l at position pos0 + 5 l as if no earlier sample
// had redrawn anything; the lanes before the first one that meets a repeat are right and store their samples, that lane replays the
// reference's draw-by-draw loop for its own sample, and the wave restarts behind it.  A repeat costs one more round (~0.65 per 324 samples
// at n = 5000).  The host drew these tables in the two gaps of every internal batch (340 us per 128 pairs, a tenth of the C5 step).
// `raw`: [pairs][raw_cap] values of srand(seed) / rand(), written by the host while the device matches (pinned, device-mapped);
// pos[pair]: stream position, carried from pass to pass; ovf[pair] = 1 when the pair's stream would have to be longer than raw_cap
// (small n: many redraws) -- the host then redoes that pair through the single-pair entry.
__global__ __launch_bounds__(64) void draw_samples_kernel(const PairSlot *__restrict__ slots, const int32_t *__restrict__ raw, int raw_cap, int Hp, int NQ,
                                                          int32_t *__restrict__ smp, int32_t *__restrict__ pos, int32_t *__restrict__ ovf) {
    const PairSlot S = slots[blockIdx.x];
    const int lane = threadIdx.x;
    const int32_t *r = raw + (size_t)S.pair * raw_cap;
    int32_t *out = smp + (size_t)blockIdx.x * Hp * 5;
    const uint32_t n = (uint32_t)S.n;
    const int add = S.pair * NQ;  // global point index: the solver does not know about pairs
    int pos0 = pos[S.pair], i0 = 0;
    bool overflow = ovf[S.pair] != 0;
    while (i0 < S.cnt && !overflow) {
        const int i = i0 + lane, p = pos0 + 5 * lane;
        const bool valid = i < S.cnt, inb = p + 5 <= raw_cap;
        int v[5] = {0, 0, 0, 0, 0};
        bool dup = false;
        if (valid && inb) {
#pragma unroll
            for (int k = 0; k < 5; ++k) v[k] = (int)((uint32_t)r[p + k] % n);
            dup = v[0] == v[1] || v[0] == v[2] || v[0] == v[3] || v[0] == v[4] || v[1] == v[2] || v[1] == v[3] || v[1] == v[4] || v[2] == v[3] ||
                  v[2] == v[4] || v[3] == v[4];
        }
        const unsigned long long need = __ballot(valid && (dup || !inb));
        const int nvalid = min(64, S.cnt - i0);
        const int first = need ? __ffsll((long long)need) - 1 : nvalid;
        if (valid && lane < first) {
#pragma unroll
            for (int k = 0; k < 5; ++k) out[(size_t)i * 5 + k] = v[k] + add;
        }
        if (first >= nvalid) {
            i0 += nvalid, pos0 += 5 * nvalid;
            continue;
        }
        int consumed = 0, bad = 0;
        if (lane == first) {  // the reference's loop: one draw per pick, repeats redrawn
            int idx[5] = {0, 0, 0, 0, 0};
            int c = 0, q = p;
            while (c < 5) {
                if (q >= raw_cap) {
                    bad = 1;
                    break;
                }
                const int w = (int)((uint32_t)r[q++] % n);
                bool rep = false;
#pragma unroll
                for (int k = 0; k < 5; ++k) rep = rep || (k < c && idx[k] == w);
                if (rep) continue;
#pragma unroll
                for (int k = 0; k < 5; ++k) idx[k] = (k == c) ? w : idx[k];
                ++c;
            }
            if (!bad) {
#pragma unroll
                for (int k = 0; k < 5; ++k) out[(size_t)i * 5 + k] = idx[k] + add;
            }
            consumed = q - p;
        }
        consumed = __shfl(consumed, first), bad = __shfl(bad, first);
        if (bad) {
            overflow = true;
            break;
        }
        i0 += first + 1, pos0 += 5 * first + consumed;
    }
    if (overflow) {  // the pair is redone by the host; until then its remaining rows must hold valid point indices (the solver reads them)
        for (int i = i0 + lane; i < S.cnt; i += 64)
#pragma unroll
            for (int k = 0; k < 5; ++k) out[(size_t)i * 5 + k] = k + add;
    }
    if (lane == 0) {
        pos[S.pair] = pos0;
        if (overflow) ovf[S.pair] = 1;
    }
}

}  // namespace

// Correspondence mode (d_q == nullptr): the batch starts behind the matching -- problem b's correspondences are ext_p1 / ext_p2 + b * nq * 2
// (camera coordinates, device), ext_counts[b] of them (host); want_pose = 0 stops after the inlier masks; d_masks_ext: the caller's [B][nq]
// mask block or nullptr.  This is mlpl_ransac_essential_batch_dev.
int pair_pose_batch_dev(mlpl_ctx *ctx, int B, const uint8_t *d_q, int nq, const uint8_t *d_t, int nt, int nbytes, const float *d_kp1,
                        const float *d_kp2, const double K0[4], const double K1[4], double thresh, int max_iters, double confidence,
                        const uint32_t *seeds, double dist, mlpl_pair_result *out, mlpl_dmatch *d_matches_out, hipStream_t s,
                        const double *ext_p1 = nullptr, const double *ext_p2 = nullptr, const int32_t *ext_counts = nullptr, int want_pose = 1,
                        uint8_t *d_masks_ext = nullptr) {
    const bool points_mode = d_q == nullptr;
    const int min_count = points_mode ? 6 : 16;  // mlpl_ransac_essential's limit / the reference's working minimum for a matched pair
    // host-hop timeline of this call (diagnostics, mlpl_debug_hop_trace): microseconds since entry at which each wait on the stream returned
    const auto t_entry = std::chrono::steady_clock::now();
    ctx->hop_n = 0;
    auto hop = [&](int code) {
        if (ctx->hop_n < (int)(sizeof(ctx->hop_us) / sizeof(ctx->hop_us[0]))) {
            ctx->hop_us[ctx->hop_n] = (float)std::chrono::duration_cast<std::chrono::nanoseconds>(std::chrono::steady_clock::now() - t_entry).count() * 1e-3f;
            ctx->hop_code[ctx->hop_n++] = code;
        }
    };
    const int NQ = nq;
    const size_t n = (size_t)NQ;
    int rc;
    // ---- workspace ----
    const size_t pack_stride4 = (n * (sizeof(double4) + sizeof(double) + 5 * sizeof(float)) + sizeof(double4) - 1) / sizeof(double4);  // per pair, in double4 units
    const int Hs = kBatchPassMax;  // slot stride of the hypothesis tables
    const size_t off_idx = 0, off_dist = off_idx + (size_t)B * n * 8, off_match = off_dist + (size_t)B * n * 8,
                 off_p1 = off_match + (size_t)B * n * 16, off_p2 = off_p1 + (size_t)B * n * 16, off_mask = off_p2 + (size_t)B * n * 16,
                 off_cmask = (off_mask + (size_t)B * n + 255) / 256 * 256, off_small = (off_cmask + (size_t)B * 4 * n + 255) / 256 * 256;
    // small block: counts[B] | active[B] | cand_counts[B][4] | states[B] | P[B][69] | pose[B] | slots[B] | dense_total[B] | cand_count[B]
    const size_t sm_counts = 0, sm_active = sm_counts + (size_t)B * 4, sm_cc = sm_active + (size_t)B * 4, sm_st = (sm_cc + (size_t)B * 16 + 255) / 256 * 256,
                 sm_P = sm_st + (size_t)B * sizeof(ReplayState), sm_pose = sm_P + (size_t)B * 69 * 8, sm_slots = sm_pose + (size_t)B * sizeof(PairPoseDev),
                 sm_dt = sm_slots + (size_t)B * sizeof(PairSlot), sm_cnd = sm_dt + (size_t)B * 4, sm_end = sm_cnd + (size_t)B * 4;
    void *blk = nullptr;
    if ((rc = ws_get(ctx, WS_PIPE, off_small + sm_end + 256, &blk))) return rc;
    char *b0 = (char *)blk, *sm = b0 + off_small;
    mlpl_dmatch *d_m = d_matches_out ? d_matches_out : (mlpl_dmatch *)(b0 + off_match);  // the caller's [B][nq] block, or the workspace
    double *d_p1 = points_mode ? const_cast<double *>(ext_p1) : (double *)(b0 + off_p1);
    double *d_p2 = points_mode ? const_cast<double *>(ext_p2) : (double *)(b0 + off_p2);
    uint8_t *d_mask = d_masks_ext ? d_masks_ext : (uint8_t *)(b0 + off_mask), *d_cmask = (uint8_t *)(b0 + off_cmask);
    int32_t *d_counts = (int32_t *)(sm + sm_counts), *d_active = (int32_t *)(sm + sm_active), *d_cc = (int32_t *)(sm + sm_cc);
    ReplayState *d_st = (ReplayState *)(sm + sm_st);
    double *d_P = (double *)(sm + sm_P);
    PairPoseDev *d_pose = (PairPoseDev *)(sm + sm_pose);
    PairSlot *d_slots = (PairSlot *)(sm + sm_slots);
    int32_t *d_dense_total = (int32_t *)(sm + sm_dt), *d_cand_count = (int32_t *)(sm + sm_cnd);
    void *p;
    if ((rc = ws_get(ctx, WS_AUX3, (size_t)B * pack_stride4 * sizeof(double4), &p))) return rc;
    double4 *d_pack = (double4 *)p;
    const size_t hyps = (size_t)B * Hs;
    if ((rc = ws_get(ctx, WS_AUX5, hyps * 90 * 8, &p))) return rc;
    double *d_Etab = (double *)p;
    if ((rc = ws_get(ctx, WS_AUX6, hyps * 90 * 8, &p))) return rc;
    double *d_denseE = (double *)p;
    if ((rc = ws_get(ctx, WS_PARTIAL, hyps * sizeof(PolyRec), &p))) return rc;
    PolyRec *d_recs = (PolyRec *)p;
    // tables: esum[hyps*10] hsum[hyps] | n_models[hyps] dense_id[hyps*10] good[hyps*10] hgood[hyps] hslot[hyps] hmax[hyps] cand[hyps*10]
    if ((rc = ws_get(ctx, WS_AUX7, hyps * 11 * 8 + hyps * (1 + 10 + 10 + 1 + 1 + 1 + 10) * 4 + 256, &p))) return rc;
    double *d_esum = (double *)p, *d_hsum = d_esum + hyps * 10;
    int32_t *d_nm = (int32_t *)(d_hsum + hyps), *d_dense_id = d_nm + hyps, *d_good = d_dense_id + hyps * 10, *d_hgood = d_good + hyps * 10,
            *d_hslot = d_hgood + hyps, *d_hmax = d_hslot + hyps, *d_cand = d_hmax + hyps;
    // device: the sample tables of a pass | stream positions and overflow flags of the pairs
    if ((rc = ws_get(ctx, WS_BATCH_SMP, hyps * 20 + (size_t)B * 8 + 256, &p))) return rc;
    int32_t *d_smp = (int32_t *)p, *d_rng_pos = d_smp + hyps * 5 + 32, *d_rng_ovf = d_rng_pos + B;
    // raw rand() values kept per pair: every iteration's five draws, a quarter more for redraws (a pair that needs more -- tiny n -- is redone)
    const int raw_cap = ((ctx->opt_pair_batch_raw_cap > 0 ? ctx->opt_pair_batch_raw_cap
                                                          : (int)std::min<long long>((long long)max_iters * 5 + (long long)max_iters * 5 / 4 + 1024, 1 << 22)) + 30) / 31 * 31;
    // pinned: counts | states | pose | slots | overflow flags | raw streams
    const size_t pin_counts = 0, pin_st = (pin_counts + (size_t)B * 4 + 255) / 256 * 256, pin_pose = pin_st + (size_t)B * sizeof(ReplayState),
                 pin_slots = pin_pose + (size_t)B * sizeof(PairPoseDev), pin_act = pin_slots + (size_t)B * sizeof(PairSlot),
                 pin_ovf = pin_act + (size_t)B * 4, pin_smp = (pin_ovf + (size_t)B * 4 + 255) / 256 * 256, pin_end = pin_smp + (size_t)B * raw_cap * 4;
    void *pin;
    if ((rc = pinned_get(ctx, pin_end + 256, &pin))) return rc;
    char *hp = (char *)pin;
    int32_t *h_counts = (int32_t *)(hp + pin_counts), *h_active = (int32_t *)(hp + pin_act), *h_raw = (int32_t *)(hp + pin_smp), *h_ovf = (int32_t *)(hp + pin_ovf);
    ReplayState *h_st = (ReplayState *)(hp + pin_st);
    PairPoseDev *h_pose = (PairPoseDev *)(hp + pin_pose);
    PairSlot *h_slots = (PairSlot *)(hp + pin_slots);
    int32_t *d_raw_mapped = nullptr;
    MLPL_HIP_TRY(hipHostGetDevicePointer((void **)&d_raw_mapped, h_raw, 0));

    if (points_mode) {
        for (int b = 0; b < B; ++b) h_counts[b] = std::max(0, std::min(ext_counts[b], NQ));
        MLPL_HIP_TRY(hipMemcpyAsync(d_counts, h_counts, (size_t)B * 4, hipMemcpyHostToDevice, s));
    } else {
        // ---- matching, all pairs ----
        rc = mlpl_match_hamming_dev(ctx, d_q, nq, (size_t)nbytes, (size_t)nq * nbytes, d_t, nt, (size_t)nbytes, (size_t)nt * nbytes, nbytes, 1, 0.75f, B,
                                    (int32_t *)(b0 + off_idx), (int32_t *)(b0 + off_dist), d_m, d_counts, s);
        if (rc) return rc;
        MLPL_HIP_TRY(hipMemcpyAsync(h_counts, d_counts, (size_t)B * 4, hipMemcpyDeviceToHost, s));
    }
    MLPL_HIP_TRY(hipMemsetAsync(d_rng_pos, 0, (size_t)B * 8, s));  // positions and overflow flags
    // the raw glibc streams do not depend on the counts: the host generates them while the device matches, straight into the pinned block
    // the sampling kernel reads (the samples themselves -- rand() % count, repeats redrawn -- are drawn on the device, per pass)
    long long draw_us = 0;
    {
        const auto t_draw0 = std::chrono::steady_clock::now();
        GlibcRand g;
        for (int b = 0; b < B; ++b) {
            g.seed(seeds[b]);
            int32_t *row = h_raw + (size_t)b * raw_cap;
            for (int o = 0; o < raw_cap; o += 31) {
                g.refill();
                std::memcpy(row + o, g.out, 31 * sizeof(int32_t));
            }
        }
        draw_us = std::chrono::duration_cast<std::chrono::microseconds>(std::chrono::steady_clock::now() - t_draw0).count();
    }
    hop(1);   // workspace, launches of the matching step and the raw random streams are behind us
    if (!points_mode) MLPL_HIP_TRY(hipStreamSynchronize(s));  // hop 1: the match counts
    hop(2);
    std::vector<int> alive;
    for (int b = 0; b < B; ++b) {
        std::memset(&out[b], 0, sizeof(out[b]));
        out[b].n_matches = h_counts[b];
        h_active[b] = h_counts[b] >= min_count ? 1 : 0;  // (pairs: below 16 matches Remove_LensDist / StereoRefine refuse to work)
        if (h_active[b]) alive.push_back(b);
        else out[b].status = -1;
    }
    if (alive.empty()) return MLPL_OK;
    MLPL_HIP_TRY(hipMemcpyAsync(d_active, h_active, (size_t)B * 4, hipMemcpyHostToDevice, s));
    if (!points_mode && (rc = launch_gather_match_points_batch(d_m, d_counts, B, NQ, d_kp1, (size_t)nq * 2, d_kp2, (size_t)nt * 2, K0, K1, d_p1, d_p2, s)))
        return rc;
    hipLaunchKernelGGL(pack_points_kernel, dim3((NQ + 255) / 256, B), dim3(256), 0, s, (const double *)d_p1, (const double *)d_p2, 0, d_pack, d_st, max_iters,
                       (int32_t *)nullptr, 0, (const int32_t *)d_counts, NQ, pack_stride4);

    const double thresh2 = thresh * thresh, qmax = inlier_bound(thresh2);
    const double log_num = std::log(std::max(1. - std::min(std::max(confidence, 0.), 1.), DBL_MIN));
    const int ev_cap = ctx->opt_ransac_event_cap > 0 ? std::min(ctx->opt_ransac_event_cap, kMaxScanEvents) : kMaxScanEvents;
    std::vector<ReplayState> state((size_t)B);
    for (int b : alive) {
        std::memset(&state[b], 0, sizeof(ReplayState));
        state[b].niters = max_iters;
    }
    int max_n = 0;
    for (int b : alive) max_n = std::max(max_n, h_counts[b]);
    int base = 0;
    bool first_pass = true;
    long long passes = 0, slots_total = 0;
    while (!alive.empty() && base < max_iters) {
        const int A = (int)alive.size();
        const int H = std::min(first_pass ? kBatchFirstPass : kBatchPassMax, max_iters - base);
        const int Hp = (H + kHypPerWave - 1) / kHypPerWave * kHypPerWave;  // slot stride of this pass: hypotheses beyond a slot's count are padding (no models)
        if (!first_pass) MLPL_HIP_TRY(hipStreamSynchronize(s));           // the pinned sample / slot tables are rewritten
        for (int a = 0; a < A; ++a) {
            const int b = alive[a];
            const int cnt = std::min(H, std::min(max_iters, state[b].niters) - base);
            PairSlot &S = h_slots[a];
            S.pts = d_pack + (size_t)b * pack_stride4, S.n = h_counts[b], S.pair = b, S.cnt = cnt, S.iter_base = base;
            // (rows cnt .. Hp - 1 of a slot's sample table are padding: the solver writes "no models" for them without reading the row)
        }
        MLPL_HIP_TRY(hipMemcpyAsync(d_slots, h_slots, (size_t)A * sizeof(PairSlot), hipMemcpyHostToDevice, s));
        MLPL_HIP_TRY(hipMemsetAsync(d_dense_total, 0, (size_t)A * 4, s));
        const int total_hyps = A * Hp;
        hipLaunchKernelGGL(draw_samples_kernel, dim3(A), dim3(64), 0, s, (const PairSlot *)d_slots, (const int32_t *)d_raw_mapped, raw_cap, Hp, NQ, d_smp,
                           d_rng_pos, d_rng_ovf);
        prof_mark(ctx, MLPL_PROF_SOLVE_5PT, 0, s);
        launch_solve5pt(ctx, total_hyps, s, (const double *)d_p1, (const double *)d_p2, (const int32_t *)d_smp, 0, total_hyps, d_recs,
                        (const PairSlot *)d_slots, Hp);
        MLPL_LAUNCH_ROOTS(ctx->opt_solver_polish, dim3(total_hyps / kHypPerWave), s, (const PolyRec *)d_recs, 0, total_hyps, d_Etab, d_nm, d_denseE,
                          d_dense_id, d_dense_total, d_good, Hp);
        prof_mark(ctx, MLPL_PROF_SOLVE_5PT, 1, s);
        // inlier counts: blockIdx.z = slot, the slot's dense model list against the slot's pair
        prof_mark(ctx, MLPL_PROF_SCORE, 0, s);
        {
            const int ntiles_max = (max_n + kScoreTile - 1) / kScoreTile;
            const int point_splits = ctx->opt_ransac_count_tiles == 1 ? std::max(1, std::min(16, ntiles_max)) : std::max(1, std::min(8, ntiles_max / 2));
            if (ctx->opt_ransac_count_mpl == 2 && ctx->opt_ransac_count_defer && max_n < (1 << 23) && ctx->opt_ransac_count_threads == 256 && ctx->opt_ransac_count_wpe == 6) {
                const dim3 grid((Hp * 10 + 127) / 128, point_splits, A);
                hipLaunchKernelGGL((count_models_f32_kernel<256, kScoreTile, 2, true, 6>), grid, dim3(256), 0, s, (const double4 *)nullptr, 0,
                                   (const double *)d_denseE, (const int32_t *)d_dense_id, (const int32_t *)d_dense_total, 0, thresh2, qmax, d_good,
                                   (const PairSlot *)d_slots, Hp);
            } else if (ctx->opt_ransac_count_mpl == 2 && ctx->opt_ransac_count_defer && max_n < (1 << 23) && ctx->opt_ransac_count_threads == 256) {
                const dim3 grid((Hp * 10 + 127) / 128, point_splits, A);
                hipLaunchKernelGGL((count_models_f32_kernel<256, kScoreTile, 2, true>), grid, dim3(256), 0, s, (const double4 *)nullptr, 0,
                                   (const double *)d_denseE, (const int32_t *)d_dense_id, (const int32_t *)d_dense_total, 0, thresh2, qmax, d_good,
                                   (const PairSlot *)d_slots, Hp);
            } else if (ctx->opt_ransac_count_mpl == 2 && ctx->opt_ransac_count_defer && max_n < (1 << 23)) {   // (the queue entry holds 23 bits of correspondence index)
                const dim3 grid((Hp * 10 + 2 * kScoreModels - 1) / (2 * kScoreModels), point_splits, A);
                hipLaunchKernelGGL((count_models_f32_kernel<kScoreThreads, kScoreTile, 2, true>), grid, dim3(kScoreThreads), 0, s, (const double4 *)nullptr, 0,
                                   (const double *)d_denseE, (const int32_t *)d_dense_id, (const int32_t *)d_dense_total, 0, thresh2, qmax, d_good,
                                   (const PairSlot *)d_slots, Hp);
            } else if (ctx->opt_ransac_count_mpl == 2) {
                const dim3 grid((Hp * 10 + 2 * kScoreModels - 1) / (2 * kScoreModels), point_splits, A);
                hipLaunchKernelGGL((count_models_f32_kernel<kScoreThreads, kScoreTile, 2>), grid, dim3(kScoreThreads), 0, s, (const double4 *)nullptr, 0,
                                   (const double *)d_denseE, (const int32_t *)d_dense_id, (const int32_t *)d_dense_total, 0, thresh2, qmax, d_good,
                                   (const PairSlot *)d_slots, Hp);
            } else {
                const dim3 grid((Hp * 10 + kScoreModels - 1) / kScoreModels, point_splits, A);
                hipLaunchKernelGGL((count_models_f32_kernel<kScoreThreads, kScoreTile>), grid, dim3(kScoreThreads), 0, s, (const double4 *)nullptr, 0,
                                   (const double *)d_denseE, (const int32_t *)d_dense_id, (const int32_t *)d_dense_total, 0, thresh2, qmax, d_good,
                                   (const PairSlot *)d_slots, Hp);
            }
        }
        prof_mark(ctx, MLPL_PROF_SCORE, 1, s);
        hipLaunchKernelGGL(hyp_max_kernel, dim3((total_hyps + 255) / 256), dim3(256), 0, s, (const int32_t *)d_nm, (const int32_t *)d_good, total_hyps,
                           d_hmax);
        hipLaunchKernelGGL(candidate_kernel, dim3(A), dim3(1024), 0, s, (const int32_t *)d_nm, (const int32_t *)d_good, (const int32_t *)d_hmax, 0, 0,
                           d_cand, d_cand_count, ev_cap, (const PairSlot *)d_slots, Hp, (const ReplayState *)d_st);
        hipLaunchKernelGGL((score_models_block_kernel<false, true>), dim3(8, A), dim3(1024), (size_t)((max_n + 3) / 4 * 4) * sizeof(float), s,
                           (const double4 *)nullptr, 0, (const double *)d_Etab, (const int32_t *)nullptr, (const int32_t *)d_cand_count, 0, thresh2, qmax,
                           (int32_t *)nullptr, d_esum, (const int32_t *)d_cand, (const PairSlot *)d_slots, Hp);
        hipLaunchKernelGGL(hyp_best_kernel, dim3((total_hyps + 255) / 256), dim3(256), 0, s, (const int32_t *)d_nm, (const int32_t *)d_good,
                           (const double *)d_esum, total_hyps, d_hgood, d_hsum, d_hslot);
        hipLaunchKernelGGL(replay_kernel, dim3(A), dim3(1024), 0, s, (const int32_t *)d_hgood, (const double *)d_hsum, (const int32_t *)d_hslot,
                           (const double *)d_Etab, 0, (const int32_t *)nullptr, 0, 0ll, (const int32_t *)d_dense_total, d_st, log_num, ev_cap,
                           (const PairSlot *)d_slots, Hp);
        MLPL_HIP_TRY(hipGetLastError());
        MLPL_HIP_TRY(hipMemcpyAsync(h_st, d_st, (size_t)B * sizeof(ReplayState), hipMemcpyDeviceToHost, s));
        MLPL_HIP_TRY(hipMemcpyAsync(h_ovf, d_rng_ovf, (size_t)B * 4, hipMemcpyDeviceToHost, s));
        hop(3);
        MLPL_HIP_TRY(hipStreamSynchronize(s));  // one hop per pass: which pairs go on
        hop(4);
        passes++, slots_total += A;
        base += H;
        std::vector<int> next;
        for (int b : alive) {
            state[b] = h_st[b];
            if (h_ovf[b]) continue;  // its stream ran out: redone below, sample tables from the host
            if (!state[b].stop && base < std::min(max_iters, state[b].niters)) next.push_back(b);
        }
        alive.swap(next);
        first_pass = false;
    }
    // every device-evaluated iteration bound must be what the CPU path's libm gives (see mlpl_ransac_essential_dev); a pair whose bound
    // differs (probability ~1e-8 per value) is redone by the single-pair driver on a host table
    std::vector<int> redo;
    for (int b = 0; b < B; ++b) {
        if (!h_active[b]) continue;
        const ReplayState &fin = state[b];
        bool ok = fin.t_count <= kTUsedMax && !h_ovf[b];
        for (int i = 0; ok && i < fin.t_count; ++i)
            ok = fin.t_val[i] == update_num_iters(confidence, (double)(h_counts[b] - fin.t_g[i]) / h_counts[b], 5, INT32_MAX);
        if (!ok) redo.push_back(b);
    }
    // masks of the models held, cheirality, results
    {
        std::vector<int> act;
        for (int b = 0; b < B; ++b)
            if (h_active[b] && state[b].maxGood > 0) act.push_back(b);
        for (int b = 0; b < B; ++b)
            if (h_active[b] && state[b].maxGood <= 0) h_active[b] = 0, out[b].status = -2, out[b].iters = state[b].iter;
        if (!act.empty()) {
            MLPL_HIP_TRY(hipStreamSynchronize(s));
            for (size_t a = 0; a < act.size(); ++a) {
                const int b = act[a];
                PairSlot &S = h_slots[a];
                S.pts = d_pack + (size_t)b * pack_stride4, S.n = h_counts[b], S.pair = b, S.cnt = 0, S.iter_base = 0;
            }
            MLPL_HIP_TRY(hipMemcpyAsync(d_slots, h_slots, act.size() * sizeof(PairSlot), hipMemcpyHostToDevice, s));
            MLPL_HIP_TRY(hipMemcpyAsync(d_active, h_active, (size_t)B * 4, hipMemcpyHostToDevice, s));
            hipLaunchKernelGGL(inlier_mask_kernel, dim3((max_n + 255) / 256, (unsigned)act.size()), dim3(256), 0, s, (const double4 *)nullptr, 0,
                               (const double *)nullptr, thresh2, d_mask, (const PairSlot *)d_slots, (const ReplayState *)d_st, NQ);
            if (want_pose) {
                prof_mark(ctx, MLPL_PROF_RECOVER_POSE, 0, s);
                if ((rc = launch_recover_pose_batch((const char *)d_st + offsetof(ReplayState, E), sizeof(ReplayState), d_p1, d_p2, d_counts, d_active, B, NQ,
                                                    dist, d_mask, d_P, d_cmask, d_cc, d_pose, s)))
                    return rc;
                prof_mark(ctx, MLPL_PROF_RECOVER_POSE, 1, s);
                MLPL_HIP_TRY(hipMemcpyAsync(h_pose, d_pose, (size_t)B * sizeof(PairPoseDev), hipMemcpyDeviceToHost, s));
            }
            hop(5);
            MLPL_HIP_TRY(hipStreamSynchronize(s));  // last hop
            hop(6);
            for (int b : act) {
                mlpl_pair_result &o = out[b];
                o.status = 0, o.iters = state[b].iter, o.n_inliers = state[b].maxGood, o.n_good = want_pose ? h_pose[b].n_good : 0;
                std::memcpy(o.E, state[b].E, 72);
                if (want_pose) std::memcpy(o.R, h_pose[b].R, 72), std::memcpy(o.t, h_pose[b].t, 24);
            }
        }
    }
    ctx->last_batch_stats[0] = passes, ctx->last_batch_stats[1] = slots_total, ctx->last_batch_stats[2] = (long long)redo.size(), ctx->last_batch_stats[3] = draw_us;
    ctx->last_batch_stats[4] = ctx->last_batch_stats[5] = ctx->last_batch_stats[6] = 0;
    for (int b = 0; b < B; ++b)
        if (h_counts[b] >= min_count) {
            ctx->last_batch_stats[4] += state[b].models_scored;
            ctx->last_batch_stats[5] += state[b].models_scored * (long long)h_counts[b];
            ctx->last_batch_stats[6] += state[b].iter;
        }
    // the rare pairs whose iteration bound has to come from the host table: the single-pair pipeline on their inputs.  The nested entries
    // take the context's pinned block for their own records (their ReplayState lands where h_counts sits), so everything the loop needs
    // from that block is copied out before the first of them runs.
    const std::vector<int32_t> counts_copy(h_counts, h_counts + B);
    for (int b : redo) {
        if (points_mode) {  // the single-problem entries on this problem's correspondences
            mlpl_pair_result &o = out[b];
            const int nb = counts_copy[b];
            std::memset(&o, 0, sizeof(o));
            o.n_matches = nb;
            int ninl = 0, iters = 0;
            ctx->ransac_force_table = 1;
            rc = mlpl_ransac_essential_dev(ctx, d_p1 + (size_t)b * NQ * 2, d_p2 + (size_t)b * NQ * 2, nb, thresh, confidence, max_iters, 0, seeds[b], o.E,
                                           d_mask + (size_t)b * NQ, &ninl, &iters, s);
            ctx->ransac_force_table = 0;
            o.iters = iters;
            if (rc == MLPL_E_FAILED) {
                o.status = -2;
                continue;
            }
            if (rc) return rc;
            o.n_inliers = ninl;
            if (want_pose) {
                rc = mlpl_recover_pose_dev(ctx, o.E, d_p1 + (size_t)b * NQ * 2, d_p2 + (size_t)b * NQ * 2, nb, dist, o.R, o.t, nullptr, d_mask + (size_t)b * NQ, s);
                if (rc < 0) return rc;
                o.n_good = rc;
            }
            continue;
        }
        ctx->ransac_force_table = 1;
        rc = mlpl_pair_pose_dev(ctx, d_q + (size_t)b * nq * nbytes, nq, d_t + (size_t)b * nt * nbytes, nt, nbytes, d_kp1 + (size_t)b * nq * 2,
                                d_kp2 + (size_t)b * nt * 2, K0, K1, thresh, max_iters, confidence, 0, seeds[b], dist, &out[b], s);
        // (the caller's match block already holds this pair's matches: the same kernels produced them)
        ctx->ransac_force_table = 0;
        if (rc) return rc;
    }
    return MLPL_OK;
}
