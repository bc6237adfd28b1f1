/*
 * mlpl_debug.h -- diagnostics entry points of libmlpl_hip.so (traces, clock stamps, solver statistics, self-tests).
 *
 * Exported by the same shared library as the C ABI in mlpl_c.h, but NOT part of the drop-in surface: nothing in the reference corresponds
 * to them and no integration needs them.  The parity tests (tests/), the tools under tools/ and bench.py's clock passes use them.
 * They may change between versions.
 */
#ifndef MLPL_DEBUG_H
#define MLPL_DEBUG_H
#include "mlpl_c.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Diagnostics: the following mlpl_arrsac_essential* calls record the turns of their first stage into buf (20 ints per turn: k, inner-RANSAC
 * turn?, sample size, its first five indices, valid models, per model 1000 * accepted + inliers seen by the sequential test, sixth + 100 * seventh index); returns the
 * number of ints written since the previous call of this function.  buf = NULL switches the recording off. */
int mlpl_debug_arrsac_trace(mlpl_ctx *ctx, int32_t *buf, int cap);

/* Diagnostics: the following mlpl_usac_essential* calls record their decisions into buf, 16 doubles per record: [0] type -- 1 sample
 * {hypothesis, 5 indices, solutions (-1 = rejected by pre-validation)}, 2 evaluation {hypothesis, model, start position in the evaluation
 * order, inliers seen, correspondences tested, accepted, delta, epsilon, decision threshold, squared inlier threshold, local
 * optimisations so far}, 3 refined model {hypothesis, points, weighted, 1, model[9]}, 4 model stored {hypothesis, model, inliers},
 * 5 minimal model {hypothesis, index, model[9]}, 6 model rejected by the oriented constraint {hypothesis, index}, 7 degeneracy test
 * {hypothesis, degenerate, upgrade, type, inliers of the rotation, of "no motion", of the best model}, 8 rotation evaluated on all
 * correspondences {hypothesis, pair of the sample, inliers of the two-point rotation, of its refit, stored}, 9 upgrade {hypothesis,
 * 1 = no motion -> t / 2 = R -> R + t, candidates tried, best inlier count}, 10 upgrade candidate {hypothesis, branch, number, t[3] or
 * E[9]}; evaluations of translation candidates are type 2 with model -1 and the angular threshold.  Returns the number of
 * records produced since the previous call of this function (may exceed cap: only cap are written).  buf = NULL switches it off. */
int mlpl_debug_usac_trace(mlpl_ctx *ctx, double *buf, int cap_records);

/* Diagnostics of the device-side sampling of large RANSAC passes (option "ransac_device_draw", default 1: passes of >= 4096 hypotheses on
 * >= 64 correspondences draw their samples on the device from the cached raw rand() stream): {calls redone with the host drawing the table
 * because a window / list / the stream ran out, 1 if the last call's samples were drawn on the device}. */
int mlpl_debug_ransac_draw(mlpl_ctx *ctx, long long out[2]);

/* The run scheduler of the batched sequential estimators by itself (csrc/batch_hub.h: fibers on worker threads, futex hand-over), no GPU
 * and no context needed: n fibers on `workers` threads pass `rounds` times through the hand-over against a stand-in hub on the calling
 * thread.  Returns n * rounds, or a negative value on bad arguments / a fiber that was not released exactly once per round. */
long long mlpl_debug_fiber_selftest(int n, int workers, int rounds);

/* The two eigen-solvers of the re-weighted 9 x 9 fits on `count` symmetric matrices G[count][81] (tests): out12[count][12] = {steps of the
 * inverse iteration (0: it did not settle and the caller would take the Jacobi path), x^T G x, 0, x[9]}; jacobi10[count][10] = {smallest
 * eigenvalue, its eigenvector} of the full Jacobi decomposition; start[count][9] (may be NULL) = start vectors of the inverse iteration. */
int mlpl_debug_eig9(mlpl_ctx *ctx, const double *G, const double *start, int count, double *out12, double *jacobi10);

/* Diagnostics: root-iteration (Ehrlich-Aberth) sweep statistics of the solver since the last call: {sum, solves, max, (enabled), sample
 * index of the max, solves with <= 8, 12, 16, 24, 32, 64, 128, 256, < 400, = 400 sweeps, 0}; enable != 0 turns the (atomic) bookkeeping on.
 * Not for production use. */
int mlpl_debug_dk_stats(mlpl_ctx *ctx, int enable, int stats[16]);

/* Diagnostics: the four flag words of the float L2 paths after the calls enqueued so far have finished (synchronises the device):
 * {generation of the last call whose data were not integer-valued (int8 preparation), generation of the last fp16-path call with a row
 * outside that path's range, number of queries the fp16 path re-ranked against every train row since the flag block was created,
 * generation of the last fp16-path call whose data were not integer-valued}. */
int mlpl_debug_l2_flags(mlpl_ctx *ctx, int flags[4]);

/* Diagnostics: with option "hamming_stamps" = 1 every wave of the matrix-core Hamming kernel records {shader-clock cycles, 100 MHz
 * real-time ticks, 32x32 tiles processed, start tick}; this copies up to max_items records of 4 x u64 of the LAST launch to `out`
 * and returns their number.  In-kernel clock = cycles / ticks * 100 MHz.  Not for production use. */
int mlpl_debug_hamming_stamps(mlpl_ctx *ctx, unsigned long long *out, int max_items);

/* Diagnostics: with option "hamming_stamps" = 2 every launch of the static LDS-ring Hamming kernel leaves ONE record {shader-clock
 * cycles, 100 MHz ticks, start tick, launch number} of the lifetime of its first workgroup in a ring of 256 launches (one 32-byte store
 * per launch; nothing else changes).  Copies the records of the last min(max_items, 256, launches so far) launches, oldest first, and
 * returns their number (synchronises the device).  Shader clock of a launch = cycles / ticks * 100 MHz. */
int mlpl_debug_hamming_clock(mlpl_ctx *ctx, unsigned long long *out, int max_items);

/* Diagnostics: the host-hop timeline of the last mlpl_pair_pose_batch_dev / mlpl_ransac_essential_batch_dev call of this context:
 * us[i] = microseconds since the call's entry, codes[i] = 1 matching enqueued, 2 match counts back, 3 / 4 a RANSAC pass enqueued / its
 * states back, 5 / 6 pose step enqueued / back; *ws_grows = workspace blocks (re)allocated by this context so far.  Returns the number
 * of entries written (<= max_items). */
int mlpl_debug_hop_trace(mlpl_ctx *ctx, float *us, int *codes, int max_items, long long *ws_grows);

#ifdef __cplusplus
}
#endif
#endif /* MLPL_DEBUG_H */
