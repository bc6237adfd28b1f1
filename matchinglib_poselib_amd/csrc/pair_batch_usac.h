// pair_batch_usac.h -- a batch of image pairs with USAC as the robust estimator (the harness default RobMethod, T/poselib-test/main.cpp:734).
// Included by ransac_5pt.hip inside namespace mlpl after usac_impl.h and pair_batch_impl.h.
//
//   matching          mlpl_match_hamming_dev(batch = B)                        as in pair_batch_impl.h
//   hop               the B match counts (and, for PROSAC, the matching costs: the order is getSortedMatchIdx' std::sort of them)
//   gather            blockIdx.y = pair: matched keypoints -> camera coordinates (ImgToCamCoordTrans)
//   USAC              usac_essential_batch_dev: every pair's sequential program on its own stack (a fiber), every launch merged over the pairs
//   cheirality        decomposition, four triangulations and the reference's candidate choice per pair on the device (launch_recover_pose_batch)
// Per pair the record is what the single-problem entries return for it: mlpl_match_hamming_dev + mlpl_gather_match_points_dev ->
// mlpl_usac_essential_dev (same parameters, seed and PROSAC order) -> mlpl_recover_pose_dev (tests/test_gpu_usac_batch.py).

// poselib::getSortedMatchIdx (pose_helper.cpp:2896-2923): std::sort of the matches by their distance, the indices in that order (the same
// std::sort on the same values: the same order, ties included)
void sorted_match_idx(const mlpl_dmatch *matches, int count, uint32_t *out) {
    struct Cost {
        float distance;
        uint32_t idx;
    };
    std::vector<Cost> c((size_t)count);
    for (int i = 0; i < count; ++i) c[i].distance = matches[i].distance, c[i].idx = (uint32_t)i;
    std::sort(c.begin(), c.end(), [](const Cost &x, const Cost &y) { return x.distance < y.distance; });
    for (int i = 0; i < count; ++i) out[i] = c[i].idx;
}

// the same on the bare costs (the batch entry fetches only those: 4 of a match's 16 bytes)
void sorted_cost_idx(const float *cost, int count, uint32_t *out) {
    struct Cost {
        float distance;
        uint32_t idx;
    };
    std::vector<Cost> c((size_t)count);
    for (int i = 0; i < count; ++i) c[i].distance = cost[i], c[i].idx = (uint32_t)i;
    std::sort(c.begin(), c.end(), [](const Cost &x, const Cost &y) { return x.distance < y.distance; });
    for (int i = 0; i < count; ++i) out[i] = c[i].idx;
}
__global__ __launch_bounds__(256) void match_cost_kernel(const mlpl_dmatch *__restrict__ m, const int32_t *__restrict__ counts, int stride, float *__restrict__ out) {
    const int b = blockIdx.y, i = blockIdx.x * 256 + threadIdx.x;
    if (i < counts[b]) out[(size_t)b * stride + i] = m[(size_t)b * stride + i].distance;
}

// tmpl != nullptr: USAC with these parameters (seeds[B], prosac); else ARRSAC (arr_thresh, arr_refine, arr_states[B][2] in / out)
int pair_pose_batch_usac_dev(mlpl_ctx *ctx, int B, const uint8_t *d_q, int nq, const uint8_t *d_t, int nt, int nbytes, const float *d_kp1,
                             const float *d_kp2, const double K0[4], const double K1[4], const mlpl_usac_params *tmpl, int prosac,
                             const uint32_t *seeds, double dist, mlpl_pair_result *out, mlpl_dmatch *d_matches_out, hipStream_t s,
                             double arr_thresh = 0, int arr_refine = 0, uint64_t *arr_states = nullptr) {
    if (!tmpl) prosac = 0;
    const int NQ = nq;
    const size_t n = (size_t)NQ;
    int rc;
    const size_t off_idx = 0, off_dist = off_idx + (size_t)B * n * 8, off_match = off_dist + (size_t)B * n * 8, off_p1 = off_match + (size_t)B * n * 16,
                 off_p2 = off_p1 + (size_t)B * n * 16, off_mask = off_p2 + (size_t)B * n * 16, off_cmask = (off_mask + (size_t)B * n + 255) / 256 * 256,
                 off_small = (off_cmask + (size_t)B * 4 * n + 255) / 256 * 256;
    // small block: counts[B] | active[B] | cand_counts[B][4] | E[B][9] | P[B][69] | pose[B]
    const size_t sm_counts = 0, sm_active = sm_counts + (size_t)B * 4, sm_cc = sm_active + (size_t)B * 4, sm_E = (sm_cc + (size_t)B * 16 + 255) / 256 * 256,
                 sm_P = sm_E + (size_t)B * 72, sm_pose = sm_P + (size_t)B * 69 * 8, sm_end = sm_pose + (size_t)B * sizeof(PairPoseDev);
    void *blk = nullptr;
    if ((rc = ws_get(ctx, WS_PIPE, off_small + sm_end + 256, &blk))) return rc;
    char *base = (char *)blk, *sm = base + off_small;
    mlpl_dmatch *d_m = d_matches_out ? d_matches_out : (mlpl_dmatch *)(base + off_match);
    double *d_p1 = (double *)(base + off_p1), *d_p2 = (double *)(base + off_p2);
    uint8_t *d_mask = (uint8_t *)(base + off_mask), *d_cmask = (uint8_t *)(base + off_cmask);
    int32_t *d_counts = (int32_t *)(sm + sm_counts), *d_active = (int32_t *)(sm + sm_active), *d_cc = (int32_t *)(sm + sm_cc);
    double *d_E = (double *)(sm + sm_E), *d_P = (double *)(sm + sm_P);
    PairPoseDev *d_pose = (PairPoseDev *)(sm + sm_pose);
    // pinned: counts | active | E | pose | (PROSAC) the matching costs
    const size_t pin_counts = 0, pin_act = (size_t)B * 4, pin_E = ((size_t)B * 8 + 255) / 256 * 256, pin_pose = pin_E + (size_t)B * 72,
                 pin_match = (pin_pose + (size_t)B * sizeof(PairPoseDev) + 255) / 256 * 256, pin_end = pin_match + (prosac ? (size_t)B * n * 4 : 0);
    // The pinned block belongs to this entry until it returns: pinned_get frees and reallocates when it grows, so no pointer into it may
    // live across a nested entry that asks for more (the matcher does not touch it at all; the batched estimators below run on
    // pinned_batch, their own block; checked again below).
    void *pin;
    if ((rc = pinned_get(ctx, pin_end + 256, &pin))) return rc;
    char *hp = (char *)pin;
    int32_t *h_counts = (int32_t *)(hp + pin_counts), *h_active = (int32_t *)(hp + pin_act);
    double *h_E = (double *)(hp + pin_E);
    PairPoseDev *h_pose = (PairPoseDev *)(hp + pin_pose);
    const float *h_cost = (const float *)(hp + pin_match);

    // ---- the producer side, cohort by cohort on the caller's stream, nothing waited for (round 5) ----------------------------------------
    // The estimators advance in cohorts of <= 128 pairs on up to four lanes.  Until round 4 all B pairs were matched first (3 ms of
    // matrix-core work during which no host thread had anything to do) and only then did the first estimator start (10+ ms of small
    // dependent launches that leave most of the chip idle).  Now cohort c's matching, count read-back, cost read-back (PROSAC) and
    // gather are enqueued with an event behind them, and a lane starts its cohort when THAT event has completed -- the matching of the
    // later cohorts runs beside the estimators of the earlier ones.
    int n_cohorts = 0, lanes_used = 0;
    int cohort = hub_cohort_size(ctx, B, tmpl ? kUsacBatchRuns : kArrBatchRuns, &n_cohorts, &lanes_used, tmpl ? hub_usac_lanes_default(tmpl->refine) : kHubLanesDefault);
    const bool use_feed = ctx->opt_pair_batch_feed != 0 && n_cohorts > 1;
    if (!use_feed) cohort = B, n_cohorts = 1;  // (option pair_batch_feed = 0, A/B: everything is matched first, then the estimators start -- round 4's order)
    std::vector<hipEvent_t> ready((size_t)n_cohorts, nullptr);
    struct EventsGuard {
        std::vector<hipEvent_t> &ev;
        ~EventsGuard() {
            for (hipEvent_t e : ev)
                if (e) (void)hipEventDestroy(e);
        }
    } ready_guard{ready};
    // on a stream of its own, ordered behind the caller's: lane 0 of the estimators serves on the caller's stream, which must not queue
    // behind the matching of the later cohorts
    HubStreams *hres = hub_resources(ctx);
    if (!hres->feed) MLPL_HIP_TRY(hipStreamCreateWithFlags(&hres->feed, hipStreamNonBlocking));
    if (!hres->feed_start) MLPL_HIP_TRY(hipEventCreateWithFlags(&hres->feed_start, hipEventDisableTiming));
    MLPL_HIP_TRY(hipEventRecord(hres->feed_start, s));
    MLPL_HIP_TRY(hipStreamWaitEvent(hres->feed, hres->feed_start, 0));
    const hipStream_t s_caller = s;
    s = hres->feed;
    // whatever way this entry is left, nothing of it is in flight on the feed stream afterwards: its copies land in the context's pinned
    // block and its kernels write the WS_PIPE block, which the next entry may regrow (free) at once.  (On success the stream is idle.)
    struct FeedDrain {
        hipStream_t f;
        ~FeedDrain() { (void)hipStreamSynchronize(f); }
    } feed_drain{hres->feed};
    for (int c = 0; c < n_cohorts; ++c) {
        const int b0 = c * cohort, nb = std::min(cohort, B - b0);
        rc = mlpl_match_hamming_dev(ctx, d_q + (size_t)b0 * nq * nbytes, nq, (size_t)nbytes, (size_t)nq * nbytes, d_t + (size_t)b0 * nt * nbytes, nt, (size_t)nbytes,
                                    (size_t)nt * nbytes, nbytes, 1, 0.75f, nb, (int32_t *)(base + off_idx) + (size_t)b0 * n * 2, (int32_t *)(base + off_dist) + (size_t)b0 * n * 2, d_m + (size_t)b0 * n,
                                    d_counts + b0, s);
        if (rc) return rc;
        MLPL_HIP_TRY(hipMemcpyAsync(h_counts + b0, d_counts + b0, (size_t)nb * 4, hipMemcpyDeviceToHost, s));
        if (prosac) {  // the costs of the matches, through the (now free) table of nearest-neighbour indices of this cohort
            float *d_cost = (float *)((int32_t *)(base + off_idx) + (size_t)b0 * n * 2);
            hipLaunchKernelGGL(match_cost_kernel, dim3((NQ + 255) / 256, nb), dim3(256), 0, s, (const mlpl_dmatch *)(d_m + (size_t)b0 * n),
                               (const int32_t *)(d_counts + b0), NQ, d_cost);
            MLPL_HIP_TRY(hipGetLastError());
            MLPL_HIP_TRY(hipMemcpyAsync(hp + pin_match + (size_t)b0 * n * 4, d_cost, (size_t)nb * n * 4, hipMemcpyDeviceToHost, s));
        }
        if ((rc = launch_gather_match_points_batch(d_m + (size_t)b0 * n, d_counts + b0, nb, NQ, d_kp1 + (size_t)b0 * nq * 2, (size_t)nq * 2, d_kp2 + (size_t)b0 * nt * 2,
                                                   (size_t)nt * 2, K0, K1, d_p1 + (size_t)b0 * n * 2, d_p2 + (size_t)b0 * n * 2, s)))
            return rc;
        MLPL_HIP_TRY(hipEventCreateWithFlags(&ready[(size_t)c], hipEventDisableTiming | hipEventBlockingSync));
        MLPL_HIP_TRY(hipEventRecord(ready[(size_t)c], s));
    }
    s = s_caller;
    // ---- the consumer side: what a lane does once its cohort's event has completed ---------------------------------------------------
    std::vector<int32_t> counts((size_t)B, 0);
    mlpl_usac_params none;
    std::memset(&none, 0, sizeof(none));
    std::vector<mlpl_usac_params> params((size_t)B, tmpl ? *tmpl : none);
    std::vector<std::vector<uint32_t>> orders(prosac ? (size_t)B : 0);
    for (int b = 0; b < B; ++b) params[b].seed = seeds ? seeds[b] : 0u, params[b].sorted_idx = nullptr;
    CohortFeed feed;
    feed.cohort = cohort, feed.n_cohorts = n_cohorts, feed.ready = ready.data();
    feed.on_ready = [&](int c, HubThreads &pool) -> int {
        const int b0 = c * cohort, nb = std::min(cohort, B - b0);
        for (int b = b0; b < b0 + nb; ++b) {
            std::memset(&out[b], 0, sizeof(out[b]));
            counts[b] = h_counts[b];
            out[b].n_matches = counts[b];
            h_active[b] = counts[b] >= 16 ? 1 : 0;  // below 16 matches Remove_LensDist / StereoRefine refuse to work
            if (!h_active[b]) out[b].status = -1, counts[b] = 0;
        }
        if (prosac) {  // the orders on the lane's (idle) run threads (a std::sort of 5000 costs takes 0.3 ms)
            const int T = std::min(nb, 64);
            pool.start(T, [&, b0, nb, T](int k) {
                for (int b = b0 + k; b < b0 + nb; b += T)
                    if (h_active[b]) {
                        orders[b].resize((size_t)counts[b]);
                        sorted_cost_idx(h_cost + (size_t)b * n, counts[b], orders[b].data());
                        params[b].sorted_idx = orders[b].data();
                    }
            }, 16);
            pool.wait();
        }
        return MLPL_OK;
    };
    std::vector<double> E((size_t)B * 9, 0.0), results((size_t)B * 12, 0.0);
    std::vector<int32_t> status((size_t)B, 0);
    const CohortFeed *feed_arg = &feed;
    if (!use_feed) {  // one hand-over for the whole batch, here; the estimators then see a batch with known counts
        MLPL_HIP_TRY(hipEventSynchronize(ready[0]));
        if ((rc = feed.on_ready(0, hres->threads))) return rc;
        feed_arg = nullptr;
    }
    if (tmpl) {
        if ((rc = usac_essential_batch_dev(ctx, B, d_p1, d_p2, NQ, counts.data(), params.data(), E.data(), d_mask, results.data(), status.data(), nullptr, nullptr, s, feed_arg)))
            return rc;
    } else {
        std::vector<int32_t> ninl((size_t)B, 0);
        if ((rc = arrsac_essential_batch_dev(ctx, B, d_p1, d_p2, NQ, counts.data(), arr_thresh, arr_refine, arr_states, E.data(), d_mask, ninl.data(), status.data(), s, feed_arg)))
            return rc;
        for (int b = 0; b < B; ++b) results[(size_t)b * 12 + 1] = 0, results[(size_t)b * 12 + 5] = ninl[b];
    }
    if (ctx->pinned != pin) {  // a nested entry must not have replaced the block these pointers look into
        set_error("pair batch (USAC / ARRSAC): the context's pinned block was reallocated by a nested call");
        return MLPL_E_INTERNAL;
    }
    for (int b = 0; b < B; ++b) {
        if (!h_active[b]) continue;
        if (status[b] == MLPL_E_FAILED) {
            h_active[b] = 0, out[b].status = -2;
            continue;
        }
        out[b].iters = (int32_t)results[(size_t)b * 12 + 1], out[b].n_inliers = (int32_t)results[(size_t)b * 12 + 5];
        std::memcpy(out[b].E, &E[(size_t)b * 9], 72);
        std::memcpy(h_E + (size_t)b * 9, &E[(size_t)b * 9], 72);
    }
    MLPL_HIP_TRY(hipMemcpyAsync(d_E, h_E, (size_t)B * 72, hipMemcpyHostToDevice, s));
    MLPL_HIP_TRY(hipMemcpyAsync(d_active, h_active, (size_t)B * 4, hipMemcpyHostToDevice, s));
    if ((rc = launch_recover_pose_batch((const char *)d_E, 72, d_p1, d_p2, d_counts, d_active, B, NQ, dist, d_mask, d_P, d_cmask, d_cc, d_pose, s))) return rc;
    MLPL_HIP_TRY(hipMemcpyAsync(h_pose, d_pose, (size_t)B * sizeof(PairPoseDev), hipMemcpyDeviceToHost, s));
    MLPL_HIP_TRY(hipStreamSynchronize(s));
    for (int b = 0; b < B; ++b) {
        if (!h_active[b]) continue;
        out[b].status = 0, out[b].n_good = h_pose[b].n_good;
        std::memcpy(out[b].R, h_pose[b].R, 72), std::memcpy(out[b].t, h_pose[b].t, 24);
    }
    return MLPL_OK;
}
