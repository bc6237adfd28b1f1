// mlpl_internal.h -- shared by the HIP translation units of libmlpl_hip.so (gfx950 only).
#pragma once

#include <hip/hip_runtime.h>

#include <cstdarg>
#include <cstddef>
#include <cstdint>
#include <cstdio>

#include "mlpl_c.h"
#include "mlpl_debug.h"

namespace mlpl {

void set_error(const char *fmt, ...);

#define MLPL_HIP_TRY(expr)                                                                                  \
    do {                                                                                                    \
        hipError_t e__ = (expr);                                                                            \
        if (e__ != hipSuccess) {                                                                            \
            ::mlpl::set_error("%s:%d: %s -> %s", __FILE__, __LINE__, #expr, hipGetErrorString(e__));        \
            return MLPL_E_HIP;                                                                              \
        }                                                                                                   \
    } while (0)

// Workspace slots: each grows monotonically and is reused across calls (no hipMalloc on the steady path).
enum WsSlot {
    WS_PACK_Q = 0,
    WS_PACK_T,
    WS_PARTIAL,
    WS_IDX,
    WS_DIST,
    WS_MATCH,
    WS_COUNT,
    WS_AUX0,
    WS_AUX1,
    WS_AUX2,
    WS_AUX3,
    WS_AUX4,
    WS_AUX5,
    WS_AUX6,
    WS_AUX7,
    WS_SCALARS,   // small host-readable outputs of the host-pointer entry points (match counts)
    WS_L2_FLAG,   // device gate flag of the L2 auto path
    WS_FRAG_Q,    // fp4 MFMA fragments of the query / train descriptors (knn_hamming_mfma.hip)
    WS_FRAG_T,
    WS_PIPE,      // per-pair pipeline block (mlpl_pair_pose_dev)
    WS_DEBUG,     // diagnostics (per-wave clock stamps)
    WS_COUNTERS,  // chunk counters of the dynamic-split Hamming kernel
    WS_SPLIT_TAB, // age-aware split table of the static LDS-ring Hamming kernel
    WS_ARR_E,     // ARRSAC: models of every sample solved in a call
    WS_ARR_F,     // ARRSAC: their inlier bit rows
    WS_F16_Q,     // fp16 L2 path (knn_l2_f16.hip): hi / lo fragments of the queries, of the train rows, row constants, partial U pairs, candidates
    WS_F16_T,
    WS_F16_CST,
    WS_F16_PART,
    WS_F16_CAND,
    WS_BATCH_SMP, // mlpl_pair_pose_batch_dev: sample tables of a pass (drawn on the device), stream positions
    WS_RAND_RAW,  // raw rand() stream of the last RANSAC seed on the device + the control block of the device-side sampling
    WS_BATCH_RUNS, // device blocks of the runs of a batched sequential estimator (USAC / ARRSAC), one slice per run
    WS_CLOCK,     // clock ring of the Hamming kernel (option hamming_stamps = 2): kClockRing records of 4 x u64
    WS_SCAN,      // pass counts of the merge workgroups when the merge kernel emits the matches itself (knn_hamming.hip MergeEmit)
    WS_TICKETS,   // ticket counters of the fused Hamming epilogue (zero between launches; knn_hamming_mfma.hip)
    WS_NUM_SLOTS
};

int launch_gather_match_points(const mlpl_dmatch *d_matches, int n, const float *d_kp1, const float *d_kp2, const double K0[4],
                               const double K1[4], double *d_p1, double *d_p2, hipStream_t s);

// Device gate of the L2 auto path: *flag == gen <=> the descriptors of call `gen` are not integer-valued in [0,255] (knn_l2_mfma.hip).
// flag == nullptr: no gate.
struct L2Gate {
    const int *flag;
    int gen;
};

bool knn_l2_f16_applicable(int dim, int nt, int k);

}  // namespace mlpl

struct mlpl_ctx {
    int device;
    hipStream_t stream;
    hipStream_t aux_stream[2];     // helper streams of the RANSAC driver: the root kernels of the slices run beside the elimination kernels and each other
    hipEvent_t aux_ev[8];          // fork/join events for them (timing disabled)
    void *ws[mlpl::WS_NUM_SLOTS];
    size_t ws_bytes[mlpl::WS_NUM_SLOTS];
    void *pinned;  // small pinned host scratch for async result readback
    size_t pinned_bytes;
    void *pinned_batch;  // pinned, device-mapped block of the batched sequential estimators (one slice per run; csrc/usac_batch.h)
    size_t pinned_batch_bytes;
    void *hub_streams;   // what the launch hubs keep between calls (HubStreams, csrc/batch_hub.h: streams, events, item tables, run threads; created on first use)
    int l2_mode;
    int opt_l2_mfma_waves;          // waves per workgroup of the L2 matrix-core kernel: 4, 8 or 0 = automatic
    int opt_l2_mfma_blocks_per_cu;  // its grid sizing target (0 = automatic)
    int l2_gen;         // call counter of the L2 auto path (see L2Gate)
    int opt_l2_float_mfma;  // non-integer float descriptors: 0 = exact fp32 kernel, 1 (default) = fp16 matrix-core candidates + exact re-rank once the
                            // previous call's data were seen to be non-integer (hint, no host hop), 2 = always enqueue that path
    int *l2_hint_host;      // pinned + device-mapped: 2 * generation + (not integer-valued) of the last auto call that has finished on the device
    int *l2_hint_dev;
    void *l2_flag_ptr;  // the flag buffer l2_gen counts for
    int num_cus;
    // tuning knobs (mlpl_set_option)
    int opt_hamming_variant;        // 3 = fp4 matrix-core kernel (default), 0 = LDS-tiled VALU, 1 = scalar-operand VALU, 2 = one wave per block
    int opt_hamming_qpl;            // queries per lane for variant 1 (1 or 2)
    int opt_hamming_blocks_per_cu;  // grid sizing target
    int opt_hamming_mfma_blocks_per_cu;  // grid sizing target of the matrix-core kernel (4-wave blocks)
    int opt_hamming_mfma_qt;             // query tiles per wave (0 = automatic, else 1, 2 or 4)
    int opt_hamming_mfma_lds;       // 32-byte descriptors: 2 (default) = LDS-ring kernel with dynamic train splits, 1 = LDS ring with static splits, 0 = register-prefetch kernel
    int opt_hamming_mfma_waves;     // waves per workgroup of the static LDS-ring kernel: 0 (default) = automatic (8 with four query tiles per wave, else 4), 4, 8
    int opt_hamming_split_rows;     // cap on the train rows of one split: 0 (default) = 8192, the exactness bound of the row fraction; 4096 = the cap up to round 4
    int opt_hamming_mfma_prefetch;  // prefetch distance of the ring in tiles with 8-wave workgroups: 0 / 2 (default), 4, 6
    int opt_hamming_mfma_prio;      // 1 = the LDS-ring kernel rotates wave priorities on a clock slice (equal finish times per SIMD; measured: no faster). Default 0
    int opt_hamming_mfma_weighted;  // 1 (default) = age-aware split sizes in the static LDS-ring kernel (4 workgroups per CU)
    long long split_tab_key;        // shape key of the split table currently in WS_COUNTERS
    void *split_tab_ptr;
    int opt_hamming_fused_merge;    // 1 (default) = the static LDS-ring kernel merges its splits / evaluates the ratio predicate itself (no merge launch)
    void *hamming_tickets_ptr;      // zeroed ticket counters of that merge (WS_TICKETS) ...
    size_t hamming_tickets_bytes;   // ... and how many bytes of them are known to be zero
    int opt_hamming_expand_fine;    // 1 (default) = small launches expand the train set with one thread per (tile, K-step, lane)
    int opt_hamming_stamps;         // diagnostics: 1 = the matrix-core kernel records per-wave clock stamps (mlpl_debug_hamming_stamps); 2 = one clock record per launch into a ring (mlpl_debug_hamming_clock)
    long long hamming_clock_launches;   // launches recorded into the clock ring so far
    int opt_hamming_merge_emit;     // 1 = one image pair per call: the merge kernel writes the DMatch rows itself (no ratio_write launch); default 0: measured, no faster
    void *hamming_scan_ptr;         // the WS_SCAN block the generation below counts for
    size_t hamming_scan_bytes;      // ... and its size when it was zeroed (a regrown block may return at the same address, never at the same size)
    uint32_t hamming_scan_gen;
    int opt_hamming_train01;        // 1 = {0, +1} train fragments in the static LDS-ring kernel (accumulator = pop(query) - distance), 0 = +-1
    int dbg_stamp_items;
    int opt_ransac_chunk;           // hypotheses per device pass (0 = 32768)
    int opt_ransac_event_cap;       // tests: capacity of the record-event list of candidate / replay kernels (0 = 1024); forces their serial fallback
    int opt_ransac_count_mpl;       // models per lane of the packed-fp32 counting kernel: 2 (default) or 1
    int opt_ransac_count_threads;   // threads per workgroup of the counting kernel: 256 (default since round 6: 4 waves, 96 VGPRs, five workgroups per CU) or 512 (rounds 2-5: 8 waves, two per CU)
    int opt_ransac_count_wpe;       // waves per SIMD the 256-thread counting kernel's registers are cut for: 5 (96 VGPRs) or 6 (80 VGPRs, a few spills outside the loop)
    int opt_ransac_count_tiles;     // 512-correspondence tiles one workgroup of the counting pass walks: 2 (rounds 3-5) or 1 (more, shorter workgroups: less idle tail)
    int opt_ransac_count_defer;     // 1 (default) = the packed-fp32 counting kernel queues its undecided evaluations in LDS and decides them workgroup-wide (same counts)
    int opt_ransac_f32_filter;      // 1 (default) = the count-only scoring kernels pre-filter in packed fp32 inside a rigorous error band (same counts)
    int opt_ransac_overlap;         // 1 (default) = large passes run their root kernels on the helper stream
    int opt_ransac_dev_split;      // device-drawn passes above 8192 hypotheses: per mille of the pass in the first of two solver slices (0 = one slice)
    int opt_ransac_lazy_sums;       // 1 (default) = division-free inlier counts + error sums only for models that can still win
    int opt_solver_polish;          // 1 (default) = every 5-point solution is finished by <= 4 Gauss-Newton steps on the cubic constraints (repairs the samples on which the DEVICE's elimination is ill conditioned); 0 = plain root path (A/B)
    int opt_solver_wave3;           // 1 (default) = solve5pt3_kernel (three hypotheses per wave, matrices in registers); 0 = one per wave
    int opt_ransac_host_table;      // 1 = always build the niters table on the host (default: evaluate on the device, verify)
    // cached table T[g] = cvRANSACUpdateNumIters1(conf, (n-g)/n, 5, inf) for the last (n, conf) (host libm values)
    int opt_rand_cache_max;         // tests: values of the rand() stream kept per context (0 = 4 Mi); beyond it a call generates privately
    void *rand_cache;               // raw rand() stream of the last RANSAC / LMedS seed (ransac_5pt.hip: RandCache)
    int32_t *ransac_T_host;
    int ransac_T_n;
    double ransac_T_conf;
    int ransac_force_table;        // 1 = build the host niters table (fallback / tests)
    int ransac_force_host_draw;    // 1 = draw the sample table on the host (fallback of the device-side sampling)
    int opt_ransac_device_draw;    // 1 (default) = passes of >= 4096 hypotheses draw their samples on the device from the cached raw stream
    void *rand_dev_ptr;            // the WS_RAND_RAW block the fields below describe
    unsigned rand_dev_seed;
    size_t rand_dev_len;           // values of srand(rand_dev_seed) / rand() present on the device
    long long ransac_draw_fallbacks;
    int last_ransac_dev_draw;      // the last mlpl_ransac_essential* call drew its samples on the device
    long long ransac_table_fallbacks;  // calls redone on the host table because a device-evaluated bound differed
    long long last_ransac_models, last_ransac_iters;  // statistics of the last mlpl_ransac_essential* call
    int32_t *arrsac_trace;                             // diagnostics: host buffer for the turn records of ARRSAC's first stage
    int arrsac_trace_cap, arrsac_trace_len;
    long long last_arrsac_stats[12];                   // ... of the last mlpl_arrsac_essential* call (mlpl_arrsac_last_stats)
    double *usac_trace;                                // diagnostics: host buffer for the decision records of USAC (16 doubles each)
    int usac_trace_cap, usac_trace_len;
    long long last_usac_stats[8];
    double last_usac_degen[16];                        // {tests on, inliers of the rotation, of "no motion", degeneracy type, R[9]}
    uint8_t *last_usac_flags;                          // malloc'ed, 2 * last_usac_flags_n bytes: inlier masks of the rotation / of "no motion"
    int last_usac_flags_n;
    unsigned *usac_prosac_tab;                         // calloc'ed, 1001 entries: PROSAC's non-randomness table of the last (beta, confidence) (usac_impl.h init_prosac)
    unsigned usac_prosac_tab_top;
    double usac_prosac_tab_beta, usac_prosac_tab_conf;
    float hop_us[48];                                  // diagnostics: microseconds since entry of the last mlpl_pair_pose_batch_dev call at its host hops ...
    int hop_code[48], hop_n;                           // ... 1 launches enqueued, 2 match counts back, 3 / 4 a pass enqueued / its states back, 5 / 6 pose enqueued / back
    long long ws_grows;                                // workspace / pinned blocks (re)allocated so far (a hipMalloc inside a call shows up here)
    long long last_batch_stats[8];                     // mlpl_pair_pose_batch_dev: {RANSAC passes, pair slots over all passes, pairs redone on a host table, 0}
    int opt_pair_batch;                                // pairs per internal batch of mlpl_pair_pose_batch_dev (0 = 256)
    int opt_pair_batch_feed;                           // 1 (default): the USAC / ARRSAC pair entries match cohort c + 1 while the estimators of cohort c run (pair_batch_usac.h)
    int opt_pair_batch_seq;                            // ... of mlpl_pair_pose_batch_usac_dev / _arrsac_dev (0 = 512)
    int opt_hub_lanes;                                 // cohorts in flight (0 = the estimator's own choice: six for USAC with REF_WEIGHTS, four otherwise; at most 8)
    int opt_eig_inverse_iteration;                     // 1 (default): the smallest eigenvector of the re-weighted 9 x 9 fits (USAC REF_WEIGHTS, robustEssentialRefine) by inverse iteration, Jacobi as the fallback
    int opt_hub_blocking_sync;                         // 1: a lane's thread sleeps on an event at the end of a round instead of spinning in hipStreamSynchronize
    int opt_hub_workers;                               // worker threads per cohort (0 = 16): the runs of a cohort are fibers on them
    int opt_hub_cohort;                                // runs of a batched USAC / ARRSAC call that advance together (0 = 128; two such cohorts are in flight)
    int opt_pair_batch_raw_cap;                        // tests: rand() values kept per pair for the device-side sampling (0 = 6.25 per iteration + 1024)
    int opt_arrsac_flag_points;                        // tests: correspondences every ARRSAC model is tested on up front (0 = 1024)
    int opt_arrsac_refine_warm_start;                  // 1 (default): robustEssentialRefine's rounds start their Jacobi iteration from the previous round's eigenvectors
    int opt_usac_first_batch;                          // samples in the first speculative batch of a USAC run (0 = as every batch: up to 128; measured, no robust gain from smaller ones)
    int opt_usac_lo5_fused_fit;                        // 1 (default): a fit of a usac5_* chain (solve -> roots -> choose) is one launch; 0 = three
    int opt_usac_lo_warm_start;                        // 1 (default): a refit of a local-optimisation chain starts its Jacobi iteration from the previous fit's eigenvectors
    int opt_usac_sprt_fast;         // 1 (default) = a sequential test that survives its first 128 steps is finished word by word on bounds (usac_impl.h sprt_walk)
    int opt_usac_lo_stepwise;                          // tests: every step of a local-optimisation chain goes through the resume path
    // optional per-kernel hipEvent bracketing (mlpl_profile_*)
    int prof_on;
    hipEvent_t *prof_ev[MLPL_PROF_NUM];  // pairs: [2*i] start, [2*i+1] stop
    int prof_n[MLPL_PROF_NUM];
    int prof_calls[MLPL_PROF_NUM];  // launches seen since the last reset (sampling: every prof_on-th is bracketed)
    int prof_take[MLPL_PROF_NUM];
};

namespace mlpl {

// Returns a device buffer of at least `bytes` for `slot`, growing it if needed (grow = sync + realloc).  A slot must not
// be requested twice with different sizes inside one entry point while the first pointer is still in use: every purpose
// that can coexist in a call has its own slot (see the WsSlot comments and the call sites).
int ws_get(mlpl_ctx *ctx, WsSlot slot, size_t bytes, void **out);
int pinned_get(mlpl_ctx *ctx, size_t bytes, void **out);
int pinned_batch_get(mlpl_ctx *ctx, size_t bytes, void **out);
void hub_streams_free(void *p);

constexpr int kProfMaxLaunches = 2048;
constexpr int kClockRing = 256;  // launches the Hamming clock ring keeps
// Records the start (phase 0) / stop (phase 1) event of one launch of kernel `id` on stream s when profiling is on.
void prof_mark(mlpl_ctx *ctx, int id, int phase, hipStream_t s);

// `stream` of the *_dev entry points is a hipStream_t; NULL is HIP's null (legacy default) stream, as everywhere
// in HIP.  The context's private stream is only used by the host-pointer entry points.
inline hipStream_t pick_stream(mlpl_ctx *, void *stream) { return reinterpret_cast<hipStream_t>(stream); }

// ---- kernel launchers (defined in the .hip files) ----
// emit_out (optional): the caller wants the DMatch rows (out [batch][nq], n_out [batch]); *emitted = 1 when the Hamming path wrote them
// itself (the merge kernel of the latency shape) and launch_ratio_compact is not needed.
struct HammingEmitOut {
    mlpl_dmatch *out;
    int32_t *n_out;
    int emitted;
};
int launch_knn_hamming(mlpl_ctx *ctx, const uint8_t *d_q, int nq, size_t q_stride, size_t q_bstride,
                       const uint8_t *d_t, int nt, size_t t_stride, size_t t_bstride, int nbytes, int k, int batch,
                       int32_t *d_idx, int32_t *d_dist, hipStream_t s, float ratio = 0.75f,
                       int32_t *d_group_counts = nullptr, HammingEmitOut *emit_out = nullptr);
int launch_knn_l2(mlpl_ctx *ctx, const float *d_q, int nq, size_t q_stride, size_t q_bstride, const float *d_t,
                  int nt, size_t t_stride, size_t t_bstride, int dim, int k, int batch, int32_t *d_idx,
                  float *d_dist, hipStream_t s, int nms_order = 0);
// group_counts_ready: d_group_counts (one int per kCountGroup queries per batch item, in the context workspace slot
// WS_COUNT) was already filled by knn_hamming_merge_kernel; otherwise a counting pass runs first.
constexpr int kRatioGroup = 256;  // queries per ratio_write block
constexpr int kCountGroup = 64;   // queries per entry of the pass-count table
int launch_ratio_compact(mlpl_ctx *ctx, const int32_t *d_idx, const void *d_dist, int dist_is_float, int nq, int k,
                         int batch, float ratio, mlpl_dmatch *d_out, int32_t *d_n_out, hipStream_t s,
                         int32_t *d_group_counts_ready = nullptr, int nms_emit = 0);

int launch_gather_match_points(const mlpl_dmatch *d_matches, int n, const float *d_kp1, const float *d_kp2, const double K0[4],
                               const double K1[4], double *d_p1, double *d_p2, hipStream_t s);
void free_rand_cache(void *p);

// cheirality for a batch of pairs (recover_pose.hip): decomposition, four triangulations, the reference's candidate choice and the mask
// of the chosen candidate, all on the device; d_out[pair] = {valid 3-D points, candidate (-1 none), R, t}
struct PairPoseDev {
    int32_t n_good, pick;
    double R[9], t[3];
};
int launch_recover_pose_batch(const char *d_E_base, size_t E_stride, const double *d_p1, const double *d_p2, const int32_t *d_counts,
                              const int32_t *d_active, int B, int pair_stride, double dist, uint8_t *d_mask, double *d_P,
                              uint8_t *d_cand_masks, int32_t *d_cand_counts, PairPoseDev *d_out, hipStream_t s);
int launch_gather_match_points_batch(const mlpl_dmatch *d_matches, const int32_t *d_counts, int B, int pair_stride, const float *d_kp1,
                                     size_t kp1_stride, const float *d_kp2, size_t kp2_stride, const double K0[4], const double K1[4],
                                     double *d_p1, double *d_p2, hipStream_t s);

}  // namespace mlpl
