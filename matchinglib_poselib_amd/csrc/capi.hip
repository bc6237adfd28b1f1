// capi.hip -- context management and the host-pointer entry points of the C ABI (include/mlpl_c.h).
// Every compute path here ends in a HIP kernel launch; there is no CPU fallback.

#include <algorithm>
#include <cstdlib>
#include <cstring>
#include <new>

#include "mlpl_internal.h"

namespace mlpl {

static thread_local char g_err[512] = "";

void set_error(const char *fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof(g_err), fmt, ap);
    va_end(ap);
}

int ws_get(mlpl_ctx *ctx, WsSlot slot, size_t bytes, void **out) {
    if (bytes < 4096) bytes = 4096;  // small requests share one minimum size, so they never trigger a regrow
    if (ctx->ws_bytes[slot] < bytes) {
        // growing: make sure nothing in flight still uses the old buffer
        MLPL_HIP_TRY(hipDeviceSynchronize());
        if (ctx->ws[slot]) MLPL_HIP_TRY(hipFree(ctx->ws[slot]));
        ctx->ws[slot] = nullptr;
        ctx->ws_bytes[slot] = 0;
        ctx->ws_grows++;
        size_t want = bytes + bytes / 4;
        want = (want + 255) & ~size_t(255);
        hipError_t e = hipMalloc(&ctx->ws[slot], want);
        if (e != hipSuccess) {
            set_error("hipMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            return MLPL_E_NOMEM;
        }
        ctx->ws_bytes[slot] = want;
    }
    *out = ctx->ws[slot];
    return MLPL_OK;
}

int pinned_get(mlpl_ctx *ctx, size_t bytes, void **out) {
    if (ctx->pinned_bytes < bytes) {
        MLPL_HIP_TRY(hipDeviceSynchronize());
        if (ctx->pinned) MLPL_HIP_TRY(hipHostFree(ctx->pinned));
        ctx->pinned = nullptr;
        ctx->pinned_bytes = 0;
        ctx->ws_grows++;
        size_t want = std::max<size_t>(bytes * 2, 4096);
        hipError_t e = hipHostMalloc(&ctx->pinned, want, hipHostMallocMapped);
        if (e != hipSuccess) {
            set_error("hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            return MLPL_E_NOMEM;
        }
        ctx->pinned_bytes = want;
    }
    *out = ctx->pinned;
    return MLPL_OK;
}

int pinned_batch_get(mlpl_ctx *ctx, size_t bytes, void **out) {
    if (ctx->pinned_batch_bytes < bytes) {
        MLPL_HIP_TRY(hipDeviceSynchronize());
        if (ctx->pinned_batch) MLPL_HIP_TRY(hipHostFree(ctx->pinned_batch));
        ctx->pinned_batch = nullptr;
        ctx->pinned_batch_bytes = 0;
        const size_t want = bytes + bytes / 8 + 4096;
        hipError_t e = hipHostMalloc(&ctx->pinned_batch, want, hipHostMallocMapped);
        if (e != hipSuccess) {
            set_error("hipHostMalloc(%zu) failed: %s", want, hipGetErrorString(e));
            return MLPL_E_NOMEM;
        }
        ctx->pinned_batch_bytes = want;
    }
    *out = ctx->pinned_batch;
    return MLPL_OK;
}

void prof_mark(mlpl_ctx *ctx, int id, int phase, hipStream_t s) {
    if (!ctx->prof_on) return;
    if (!ctx->prof_ev[id]) return;  // pools are created by mlpl_profile_enable, outside any timed region
    if (ctx->prof_n[id] >= kProfMaxLaunches) return;
    // prof_on = N > 1: bracket every Nth launch only (two event records cost ~10 us of stream time per launch)
    if (phase == 0) ctx->prof_take[id] = (ctx->prof_calls[id]++ % ctx->prof_on) == 0;
    if (!ctx->prof_take[id]) return;
    (void)hipEventRecord(ctx->prof_ev[id][2 * ctx->prof_n[id] + phase], s);
    if (phase == 1) ctx->prof_n[id]++;
}

}  // namespace mlpl

using namespace mlpl;

extern "C" {

const char *mlpl_last_error(void) { return g_err; }
const char *mlpl_version(void) { return "mlpl-hip 0.1 (gfx950)"; }

int mlpl_device_count(void) {
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess) return 0;
    return n;
}

int mlpl_ctx_create(int device_ordinal, mlpl_ctx **out) {
    if (!out) return MLPL_E_BAD_INPUT;
    *out = nullptr;
    int n = 0;
    if (hipGetDeviceCount(&n) != hipSuccess || n <= 0) {
        set_error("no HIP device visible: this library has no CPU fallback");
        return MLPL_E_NO_DEVICE;
    }
    if (device_ordinal < 0 || device_ordinal >= n) {
        set_error("device ordinal %d out of range [0,%d)", device_ordinal, n);
        return MLPL_E_BAD_INPUT;
    }
    MLPL_HIP_TRY(hipSetDevice(device_ordinal));
    hipDeviceProp_t prop;
    MLPL_HIP_TRY(hipGetDeviceProperties(&prop, device_ordinal));
    if (std::strncmp(prop.gcnArchName, "gfx950", 6) != 0) {
        set_error("device %d is %s; this library carries gfx950 code objects only", device_ordinal, prop.gcnArchName);
        return MLPL_E_NO_DEVICE;
    }
    mlpl_ctx *ctx = new (std::nothrow) mlpl_ctx();
    if (!ctx) return MLPL_E_NOMEM;
    std::memset(ctx, 0, sizeof(*ctx));
    ctx->device = device_ordinal;
    ctx->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    ctx->opt_hamming_variant = 3;
    ctx->opt_hamming_qpl = 1;
    ctx->opt_hamming_blocks_per_cu = 32;
    ctx->opt_hamming_mfma_blocks_per_cu = 3;
    ctx->opt_hamming_mfma_qt = 0;
    ctx->opt_hamming_mfma_lds = 1;
    ctx->opt_hamming_mfma_weighted = 1;
    ctx->opt_hamming_mfma_prio = 0;
    ctx->opt_hamming_fused_merge = 1;
    ctx->opt_hamming_expand_fine = 1;
    ctx->opt_hamming_merge_emit = 0;  // measured (tools/single_pair_probe.py): the launch it saves is what the chained look-back costs -- 21.2-22.2 against 20.8-21.4 us
    ctx->opt_ransac_lazy_sums = 1;
    ctx->opt_ransac_overlap = 1;
    ctx->opt_ransac_f32_filter = 1;
    ctx->opt_ransac_count_mpl = 2;
    ctx->opt_ransac_count_tiles = 2;
    ctx->opt_ransac_count_wpe = 5;
    ctx->opt_ransac_count_threads = 256;   // round 6: 4-wave workgroups at 96 VGPRs, five per CU (C5 counting -6-8 %, C3 equal)
    ctx->opt_ransac_count_defer = 1;
    ctx->opt_pair_batch_feed = 1;
    ctx->opt_solver_polish = 1;   // the solver's accuracy safeguard stays on: measured CLOSER to the CPU path than the plain root path (tools/polish_default_ab.py)
    ctx->opt_solver_wave3 = 1;
    ctx->opt_ransac_device_draw = 1;
    ctx->opt_usac_lo_warm_start = 1;
    ctx->opt_usac_lo5_fused_fit = 1;
    ctx->opt_hub_blocking_sync = 1;
    ctx->opt_eig_inverse_iteration = 1;
    ctx->opt_arrsac_refine_warm_start = 1;
    ctx->opt_usac_sprt_fast = 1;
    ctx->opt_l2_float_mfma = 1;
    hipError_t e = hipStreamCreateWithFlags(&ctx->stream, hipStreamNonBlocking);
    if (e != hipSuccess) {
        set_error("hipStreamCreate failed: %s", hipGetErrorString(e));
        delete ctx;
        return MLPL_E_HIP;
    }
    e = hipStreamCreateWithFlags(&ctx->aux_stream[0], hipStreamNonBlocking);
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&ctx->aux_stream[1], hipStreamNonBlocking);
    for (int i = 0; e == hipSuccess && i < 8; ++i) e = hipEventCreateWithFlags(&ctx->aux_ev[i], hipEventDisableTiming);
    if (e != hipSuccess) {
        set_error("helper stream / events: %s", hipGetErrorString(e));
        mlpl_ctx_destroy(ctx);
        return MLPL_E_HIP;
    }
    *out = ctx;
    return MLPL_OK;
}

void mlpl_ctx_destroy(mlpl_ctx *ctx) {
    if (!ctx) return;
    (void)hipSetDevice(ctx->device);
    (void)hipDeviceSynchronize();
    for (int i = 0; i < WS_NUM_SLOTS; ++i)
        if (ctx->ws[i]) (void)hipFree(ctx->ws[i]);
    for (int k = 0; k < MLPL_PROF_NUM; ++k) {
        if (!ctx->prof_ev[k]) continue;
        for (int i = 0; i < 2 * kProfMaxLaunches; ++i) (void)hipEventDestroy(ctx->prof_ev[k][i]);
        delete[] ctx->prof_ev[k];
    }
    if (ctx->pinned) (void)hipHostFree(ctx->pinned);
    if (ctx->pinned_batch) (void)hipHostFree(ctx->pinned_batch);
    mlpl::hub_streams_free(ctx->hub_streams);
    if (ctx->l2_hint_host) (void)hipHostFree(ctx->l2_hint_host);
    delete[] ctx->ransac_T_host;
    std::free(ctx->last_usac_flags);
    std::free(ctx->usac_prosac_tab);
    mlpl::free_rand_cache(ctx->rand_cache);
    if (ctx->stream) (void)hipStreamDestroy(ctx->stream);
    for (int i = 0; i < 2; ++i)
        if (ctx->aux_stream[i]) (void)hipStreamDestroy(ctx->aux_stream[i]);
    for (int i = 0; i < 8; ++i)
        if (ctx->aux_ev[i]) (void)hipEventDestroy(ctx->aux_ev[i]);
    delete ctx;
}

void *mlpl_ctx_stream(mlpl_ctx *ctx) { return ctx ? (void *)ctx->stream : nullptr; }
int mlpl_ctx_device(mlpl_ctx *ctx) { return ctx ? ctx->device : -1; }
int mlpl_ctx_synchronize(mlpl_ctx *ctx) {
    if (!ctx) return MLPL_E_BAD_INPUT;
    MLPL_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MLPL_OK;
}

int mlpl_set_option(mlpl_ctx *ctx, const char *name, int value) {
    if (!ctx || !name) return MLPL_E_BAD_INPUT;
    if (!std::strcmp(name, "hamming_variant") && value >= 0 && value <= 3) ctx->opt_hamming_variant = value;
    else if (!std::strcmp(name, "hamming_mfma_qt") && (value == 0 || value == 1 || value == 2 || value == 4)) ctx->opt_hamming_mfma_qt = value;
    else if (!std::strcmp(name, "hamming_mfma_blocks_per_cu") && value >= 1 && value <= 64) ctx->opt_hamming_mfma_blocks_per_cu = value;
    else if (!std::strcmp(name, "hamming_mfma_lds") && (value >= 0 && value <= 2)) ctx->opt_hamming_mfma_lds = value;
    else if (!std::strcmp(name, "hamming_expand_fine") && (value == 0 || value == 1)) ctx->opt_hamming_expand_fine = value;
    else if (!std::strcmp(name, "hamming_mfma_prio") && (value >= 0 && value <= 3)) ctx->opt_hamming_mfma_prio = value;
    else if (!std::strcmp(name, "hamming_mfma_prefetch") && (value == 0 || value == 2 || value == 4 || value == 6)) ctx->opt_hamming_mfma_prefetch = value;
    else if (!std::strcmp(name, "hamming_split_rows") && (value == 0 || value == 4096 || value == 8192)) ctx->opt_hamming_split_rows = value;
    else if (!std::strcmp(name, "hamming_mfma_waves") && (value == 0 || value == 4 || value == 8 || value == 16)) ctx->opt_hamming_mfma_waves = value;
    else if (!std::strcmp(name, "hamming_mfma_weighted") && (value == 0 || value == 1)) ctx->opt_hamming_mfma_weighted = value;
    else if (!std::strcmp(name, "hamming_fused_merge") && (value == 0 || value == 1)) ctx->opt_hamming_fused_merge = value;
    else if (!std::strcmp(name, "hamming_stamps") && (value >= 0 && value <= 2)) ctx->opt_hamming_stamps = value;
    else if (!std::strcmp(name, "hamming_train01") && (value == 0 || value == 1)) ctx->opt_hamming_train01 = value;
    else if (!std::strcmp(name, "hamming_merge_emit") && (value == 0 || value == 1)) ctx->opt_hamming_merge_emit = value;
    else if (!std::strcmp(name, "l2_mfma_waves") && (value == 0 || value == 4 || value == 8)) ctx->opt_l2_mfma_waves = value;
    else if (!std::strcmp(name, "l2_mfma_blocks_per_cu") && value >= 0 && value <= 16) ctx->opt_l2_mfma_blocks_per_cu = value;
    else if (!std::strcmp(name, "hamming_qpl") && (value == 1 || value == 2)) ctx->opt_hamming_qpl = value;
    else if (!std::strcmp(name, "hamming_blocks_per_cu") && value >= 1 && value <= 64) ctx->opt_hamming_blocks_per_cu = value;
    else if (!std::strcmp(name, "ransac_lazy_sums") && (value == 0 || value == 1)) ctx->opt_ransac_lazy_sums = value;
    else if (!std::strcmp(name, "ransac_overlap") && (value == 0 || value == 1)) ctx->opt_ransac_overlap = value;
    else if (!std::strcmp(name, "ransac_dev_split") && value >= 0 && value <= 900) ctx->opt_ransac_dev_split = value;
    else if (!std::strcmp(name, "rand_cache_max") && value >= 0) ctx->opt_rand_cache_max = value;
    else if (!std::strcmp(name, "ransac_f32_filter") && (value == 0 || value == 1)) ctx->opt_ransac_f32_filter = value;
    else if (!std::strcmp(name, "arrsac_refine_warm_start") && (value == 0 || value == 1)) ctx->opt_arrsac_refine_warm_start = value;
    else if (!std::strcmp(name, "ransac_count_mpl") && (value == 1 || value == 2)) ctx->opt_ransac_count_mpl = value;
    else if (!std::strcmp(name, "ransac_count_tiles") && (value == 1 || value == 2)) ctx->opt_ransac_count_tiles = value;
    else if (!std::strcmp(name, "ransac_count_threads") && (value == 256 || value == 512)) ctx->opt_ransac_count_threads = value;
    else if (!std::strcmp(name, "ransac_count_wpe") && (value == 5 || value == 6)) ctx->opt_ransac_count_wpe = value;
    else if (!std::strcmp(name, "ransac_count_defer") && (value == 0 || value == 1)) ctx->opt_ransac_count_defer = value;
    else if (!std::strcmp(name, "ransac_event_cap") && value >= 0 && value <= 1024) ctx->opt_ransac_event_cap = value;
    else if (!std::strcmp(name, "solver_polish") && (value == 0 || value == 1)) ctx->opt_solver_polish = value;
    else if (!std::strcmp(name, "solver_wave3") && (value == 0 || value == 1)) ctx->opt_solver_wave3 = value;
    else if (!std::strcmp(name, "ransac_device_draw") && (value == 0 || value == 1)) ctx->opt_ransac_device_draw = value;
    else if (!std::strcmp(name, "l2_float_mfma") && value >= 0 && value <= 2) ctx->opt_l2_float_mfma = value;
    else if (!std::strcmp(name, "arrsac_flag_points") && (value == 0 || (value >= 128 && value <= 1024 && value % 64 == 0))) ctx->opt_arrsac_flag_points = value;
    else if (!std::strcmp(name, "pair_batch") && value >= 0 && value <= 1024) ctx->opt_pair_batch = value;
    else if (!std::strcmp(name, "hub_lanes") && value >= 0 && value <= 8) ctx->opt_hub_lanes = value;
    else if (!std::strcmp(name, "eig_inverse_iteration") && (value == 0 || value == 1)) ctx->opt_eig_inverse_iteration = value;
    else if (!std::strcmp(name, "hub_blocking_sync") && (value == 0 || value == 1)) ctx->opt_hub_blocking_sync = value;
    else if (!std::strcmp(name, "hub_workers") && value >= 0 && value <= 64) ctx->opt_hub_workers = value;
    else if (!std::strcmp(name, "hub_cohort") && (value == 0 || (value >= 8 && value <= 512))) ctx->opt_hub_cohort = value;
    else if (!std::strcmp(name, "pair_batch_seq") && value >= 0 && value <= 1024) ctx->opt_pair_batch_seq = value;
    else if (!std::strcmp(name, "pair_batch_feed") && (value == 0 || value == 1)) ctx->opt_pair_batch_feed = value;
    else if (!std::strcmp(name, "pair_batch_raw_cap") && (value == 0 || (value >= 64 && value <= (1 << 22)))) ctx->opt_pair_batch_raw_cap = value;
    else if (!std::strcmp(name, "usac_lo_stepwise") && (value == 0 || value == 1)) ctx->opt_usac_lo_stepwise = value;
    else if (!std::strcmp(name, "usac_lo_warm_start") && (value == 0 || value == 1)) ctx->opt_usac_lo_warm_start = value;
    else if (!std::strcmp(name, "usac_first_batch") && value >= 0 && value <= 128) ctx->opt_usac_first_batch = value;
    else if (!std::strcmp(name, "usac_lo5_fused_fit") && (value == 0 || value == 1)) ctx->opt_usac_lo5_fused_fit = value;
    else if (!std::strcmp(name, "usac_sprt_fast") && (value == 0 || value == 1)) ctx->opt_usac_sprt_fast = value;
    else if (!std::strcmp(name, "ransac_host_table") && (value == 0 || value == 1)) ctx->opt_ransac_host_table = value;
    else if (!std::strcmp(name, "ransac_chunk") && value >= 0 && value <= 32768) ctx->opt_ransac_chunk = value;
    else {
        set_error("mlpl_set_option: unknown option or bad value: %s=%d", name, value);
        return MLPL_E_BAD_INPUT;
    }
    return MLPL_OK;
}

int mlpl_get_option(mlpl_ctx *ctx, const char *name, int *value) {
    if (!ctx || !name || !value) return MLPL_E_BAD_INPUT;
    if (!std::strcmp(name, "hamming_variant")) *value = ctx->opt_hamming_variant;
    else if (!std::strcmp(name, "hamming_mfma_qt")) *value = ctx->opt_hamming_mfma_qt;
    else if (!std::strcmp(name, "hamming_mfma_lds")) *value = ctx->opt_hamming_mfma_lds;
    else if (!std::strcmp(name, "hamming_fused_merge")) *value = ctx->opt_hamming_fused_merge;
    else if (!std::strcmp(name, "hamming_train01")) *value = ctx->opt_hamming_train01;
    else if (!std::strcmp(name, "hamming_merge_emit")) *value = ctx->opt_hamming_merge_emit;
    else if (!std::strcmp(name, "hamming_stamps")) *value = ctx->opt_hamming_stamps;
    else if (!std::strcmp(name, "hamming_split_rows")) *value = ctx->opt_hamming_split_rows;
    else if (!std::strcmp(name, "hamming_mfma_waves")) *value = ctx->opt_hamming_mfma_waves;
    else if (!std::strcmp(name, "solver_polish")) *value = ctx->opt_solver_polish;
    else if (!std::strcmp(name, "ransac_count_mpl")) *value = ctx->opt_ransac_count_mpl;
    else if (!std::strcmp(name, "ransac_count_threads")) *value = ctx->opt_ransac_count_threads;
    else if (!std::strcmp(name, "ransac_count_tiles")) *value = ctx->opt_ransac_count_tiles;
    else if (!std::strcmp(name, "ransac_count_defer")) *value = ctx->opt_ransac_count_defer;
    else if (!std::strcmp(name, "hub_workers")) *value = ctx->opt_hub_workers;
    else if (!std::strcmp(name, "hub_lanes")) *value = ctx->opt_hub_lanes;
    else {
        set_error("mlpl_get_option: unknown option: %s", name);
        return MLPL_E_BAD_INPUT;
    }
    return MLPL_OK;
}

int mlpl_debug_hamming_stamps(mlpl_ctx *ctx, unsigned long long *out, int max_items) {
    if (!ctx || !out) return MLPL_E_BAD_INPUT;
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    MLPL_HIP_TRY(hipDeviceSynchronize());
    // max_items < 0: the per-tile trace that follows the records (48 u64 per wave), -max_items waves at most
    if (max_items < 0) {
        const int n = ctx->dbg_stamp_items < -max_items ? ctx->dbg_stamp_items : -max_items;
        if (n > 0 && ctx->ws[WS_DEBUG])
            MLPL_HIP_TRY(hipMemcpy(out, (char *)ctx->ws[WS_DEBUG] + (size_t)ctx->dbg_stamp_items * 32, (size_t)n * 48 * 8, hipMemcpyDeviceToHost));
        return n;
    }
    const int n = ctx->dbg_stamp_items < max_items ? ctx->dbg_stamp_items : max_items;
    if (n > 0 && ctx->ws[WS_DEBUG]) MLPL_HIP_TRY(hipMemcpy(out, ctx->ws[WS_DEBUG], (size_t)n * 32, hipMemcpyDeviceToHost));
    return n;
}

int mlpl_debug_hop_trace(mlpl_ctx *ctx, float *us, int *codes, int max_items, long long *ws_grows) {
    if (!ctx) return MLPL_E_BAD_INPUT;
    const int n = ctx->hop_n < max_items ? ctx->hop_n : max_items;
    for (int i = 0; i < n; ++i) {
        if (us) us[i] = ctx->hop_us[i];
        if (codes) codes[i] = ctx->hop_code[i];
    }
    if (ws_grows) *ws_grows = ctx->ws_grows;
    return n;
}

int mlpl_debug_hamming_clock(mlpl_ctx *ctx, unsigned long long *out, int max_items) {
    if (!ctx || !out || max_items < 0) return MLPL_E_BAD_INPUT;
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    MLPL_HIP_TRY(hipDeviceSynchronize());
    const long long have = ctx->hamming_clock_launches < kClockRing ? ctx->hamming_clock_launches : kClockRing;
    const int n = (int)(have < max_items ? have : max_items);
    if (n <= 0 || !ctx->ws[WS_CLOCK]) return 0;
    unsigned long long ring[kClockRing * 4];
    MLPL_HIP_TRY(hipMemcpy(ring, ctx->ws[WS_CLOCK], sizeof(ring), hipMemcpyDeviceToHost));
    for (int i = 0; i < n; ++i) {  // oldest first
        const long long launch = ctx->hamming_clock_launches - n + i;
        std::memcpy(out + (size_t)i * 4, ring + (size_t)(launch % kClockRing) * 4, 32);
    }
    return n;
}

int mlpl_profile_enable(mlpl_ctx *ctx, int on) {
    if (!ctx) return MLPL_E_BAD_INPUT;
    ctx->prof_on = on > 0 ? on : 0;  // N > 1: every Nth launch of a kernel is bracketed
    if (ctx->prof_on) {
        // event pools up front: creating them lazily put ~10^4 hipEventCreate calls into the caller's first timed launch
        for (int id = 0; id < MLPL_PROF_NUM; ++id) {
            if (ctx->prof_ev[id]) continue;
            ctx->prof_ev[id] = new hipEvent_t[2 * kProfMaxLaunches];
            for (int i = 0; i < 2 * kProfMaxLaunches; ++i) (void)hipEventCreate(&ctx->prof_ev[id][i]);
        }
    }
    return MLPL_OK;
}

int mlpl_profile_reset(mlpl_ctx *ctx) {
    if (!ctx) return MLPL_E_BAD_INPUT;
    MLPL_HIP_TRY(hipDeviceSynchronize());
    for (int k = 0; k < MLPL_PROF_NUM; ++k) ctx->prof_n[k] = ctx->prof_calls[k] = 0;
    return MLPL_OK;
}

int mlpl_profile_read(mlpl_ctx *ctx, int kernel_id, double *total_ms, int *launches) {
    if (!ctx || kernel_id < 0 || kernel_id >= MLPL_PROF_NUM || !total_ms || !launches) return MLPL_E_BAD_INPUT;
    double tot = 0.0;
    const int n = ctx->prof_n[kernel_id];
    for (int i = 0; i < n; ++i) {
        MLPL_HIP_TRY(hipEventSynchronize(ctx->prof_ev[kernel_id][2 * i + 1]));
        float ms = 0.f;
        MLPL_HIP_TRY(hipEventElapsedTime(&ms, ctx->prof_ev[kernel_id][2 * i], ctx->prof_ev[kernel_id][2 * i + 1]));
        tot += ms;
    }
    *total_ms = tot;
    *launches = n;
    return MLPL_OK;
}

int mlpl_debug_l2_flags(mlpl_ctx *ctx, int flags[4]) {
    if (!ctx || !flags) return MLPL_E_BAD_INPUT;
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    MLPL_HIP_TRY(hipDeviceSynchronize());
    flags[0] = flags[1] = flags[2] = flags[3] = 0;
    if (ctx->l2_flag_ptr) MLPL_HIP_TRY(hipMemcpy(flags, ctx->l2_flag_ptr, 16, hipMemcpyDeviceToHost));
    return MLPL_OK;
}

int mlpl_set_l2_path(mlpl_ctx *ctx, int mode) {
    if (!ctx || mode < 0 || mode > 3) return MLPL_E_BAD_INPUT;
    ctx->l2_mode = mode;
    return MLPL_OK;
}

// ---------------------------------------------------------------------------------------------------------
// device-pointer entry points
// ---------------------------------------------------------------------------------------------------------

int mlpl_knn2_hamming_dev(mlpl_ctx *ctx, const uint8_t *d_q, int nq, size_t q_stride, size_t q_batch_stride,
                          const uint8_t *d_t, int nt, size_t t_stride, size_t t_batch_stride, int nbytes, int k,
                          int batch, int32_t *d_idx, int32_t *d_dist, void *stream) {
    if (!ctx) return MLPL_E_BAD_INPUT;
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    return launch_knn_hamming(ctx, d_q, nq, q_stride, q_batch_stride, d_t, nt, t_stride, t_batch_stride, nbytes, k,
                              batch, d_idx, d_dist, pick_stream(ctx, stream));
}

int mlpl_knn2_l2sq_f32_dev(mlpl_ctx *ctx, const float *d_q, int nq, size_t q_stride, size_t q_batch_stride,
                           const float *d_t, int nt, size_t t_stride, size_t t_batch_stride, int dim, int k, int batch,
                           int32_t *d_idx, float *d_dist, void *stream) {
    if (!ctx) return MLPL_E_BAD_INPUT;
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    return launch_knn_l2(ctx, d_q, nq, q_stride, q_batch_stride, d_t, nt, t_stride, t_batch_stride, dim, k, batch,
                         d_idx, d_dist, pick_stream(ctx, stream));
}

int mlpl_ratio_compact_i32_dev(mlpl_ctx *ctx, const int32_t *d_idx, const int32_t *d_dist, int nq, int k, int batch,
                               float ratio, mlpl_dmatch *d_out, int32_t *d_n_out, void *stream) {
    if (!ctx) return MLPL_E_BAD_INPUT;
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    return launch_ratio_compact(ctx, d_idx, d_dist, 0, nq, k, batch, ratio, d_out, d_n_out, pick_stream(ctx, stream));
}

int mlpl_ratio_compact_f32_dev(mlpl_ctx *ctx, const int32_t *d_idx, const float *d_dist, int nq, int k, int batch,
                               float ratio, mlpl_dmatch *d_out, int32_t *d_n_out, void *stream) {
    if (!ctx) return MLPL_E_BAD_INPUT;
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    return launch_ratio_compact(ctx, d_idx, d_dist, 1, nq, k, batch, ratio, d_out, d_n_out, pick_stream(ctx, stream));
}

int mlpl_gather_match_points_dev(mlpl_ctx *ctx, const mlpl_dmatch *d_matches, int n, const float *d_kp1, const float *d_kp2,
                                 const double K0[4], const double K1[4], double *d_p1, double *d_p2, void *stream) {
    if (!ctx || !d_matches || !d_kp1 || !d_kp2 || !K0 || !K1 || !d_p1 || !d_p2 || n < 0) {
        set_error("mlpl_gather_match_points_dev: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    return launch_gather_match_points(d_matches, n, d_kp1, d_kp2, K0, K1, d_p1, d_p2, pick_stream(ctx, stream));
}

int mlpl_match_hamming_dev(mlpl_ctx *ctx, const uint8_t *d_q, int nq, size_t q_stride, size_t q_batch_stride,
                           const uint8_t *d_t, int nt, size_t t_stride, size_t t_batch_stride, int nbytes,
                           int ratio_test, float ratio, int batch, int32_t *d_idx, int32_t *d_dist, mlpl_dmatch *d_out,
                           int32_t *d_n_out, void *stream) {
    if (!ctx) return MLPL_E_BAD_INPUT;
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    hipStream_t s = pick_stream(ctx, stream);
    const int k = ratio_test ? 2 : 1;
    void *gc = nullptr;
    const int ncnt = (nq + kCountGroup - 1) / kCountGroup;
    int rc = ws_get(ctx, WS_COUNT, (size_t)batch * std::max(ncnt, 1) * sizeof(int32_t), &gc);
    if (rc) return rc;
    HammingEmitOut emit{d_out, d_n_out, 0};
    rc = launch_knn_hamming(ctx, d_q, nq, q_stride, q_batch_stride, d_t, nt, t_stride, t_batch_stride, nbytes, k, batch,
                            d_idx, d_dist, s, ratio, (int32_t *)gc, &emit);
    if (rc) return rc;
    if (emit.emitted) return MLPL_OK;  // (the latency shape: the merge kernel wrote the matches and their number)
    return launch_ratio_compact(ctx, d_idx, d_dist, 0, nq, k, batch, ratio, d_out, d_n_out, s, (int32_t *)gc);
}

// ---------------------------------------------------------------------------------------------------------
// host-pointer entry points: stage through the context workspace, run the device path, copy back
// ---------------------------------------------------------------------------------------------------------

namespace {

// Copies `rows` rows of `row_bytes` bytes (host stride `stride`) into a dense device buffer.
int upload_rows(mlpl_ctx *ctx, WsSlot slot, const void *host, int rows, size_t row_bytes, size_t stride, void **dev) {
    int rc = ws_get(ctx, slot, (size_t)rows * row_bytes, dev);
    if (rc) return rc;
    if (rows == 0) return MLPL_OK;
    if (stride == row_bytes)  // dense rows: one linear copy (a 2-D copy of 32-byte rows is an order of magnitude slower)
        MLPL_HIP_TRY(hipMemcpyAsync(*dev, host, (size_t)rows * row_bytes, hipMemcpyHostToDevice, ctx->stream));
    else
        MLPL_HIP_TRY(hipMemcpy2DAsync(*dev, row_bytes, host, stride, row_bytes, rows, hipMemcpyHostToDevice, ctx->stream));
    return MLPL_OK;
}

}  // namespace

int mlpl_knn2_hamming(mlpl_ctx *ctx, const uint8_t *q, int nq, size_t q_stride, const uint8_t *t, int nt, size_t t_stride,
                      int nbytes, int k, int32_t *idx, int32_t *dist) {
    if (!ctx || !q || !t || !idx || !dist || nq < 0 || nt < 0 || nbytes < 1 || q_stride < (size_t)nbytes ||
        t_stride < (size_t)nbytes || (k != 1 && k != 2) || nt < k) {
        set_error("mlpl_knn2_hamming: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    if (nq == 0) return MLPL_OK;
    void *dq, *dt, *didx, *ddist;
    int rc;
    if ((rc = upload_rows(ctx, WS_AUX0, q, nq, nbytes, q_stride, &dq))) return rc;
    if ((rc = upload_rows(ctx, WS_AUX1, t, nt, nbytes, t_stride, &dt))) return rc;
    if ((rc = ws_get(ctx, WS_IDX, (size_t)nq * k * 4, &didx))) return rc;
    if ((rc = ws_get(ctx, WS_DIST, (size_t)nq * k * 4, &ddist))) return rc;
    rc = launch_knn_hamming(ctx, (const uint8_t *)dq, nq, nbytes, 0, (const uint8_t *)dt, nt, nbytes, 0, nbytes, k, 1,
                            (int32_t *)didx, (int32_t *)ddist, ctx->stream);
    if (rc) return rc;
    MLPL_HIP_TRY(hipMemcpyAsync(idx, didx, (size_t)nq * k * 4, hipMemcpyDeviceToHost, ctx->stream));
    MLPL_HIP_TRY(hipMemcpyAsync(dist, ddist, (size_t)nq * k * 4, hipMemcpyDeviceToHost, ctx->stream));
    MLPL_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MLPL_OK;
}

int mlpl_knn2_l2sq_f32(mlpl_ctx *ctx, const float *q, int nq, size_t q_stride, const float *t, int nt, size_t t_stride,
                       int dim, int k, int32_t *idx, float *dist) {
    if (!ctx || !q || !t || !idx || !dist || nq < 0 || nt < 0 || dim < 1 || q_stride < (size_t)dim ||
        t_stride < (size_t)dim || (k != 1 && k != 2) || nt < k) {
        set_error("mlpl_knn2_l2sq_f32: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    if (nq == 0) return MLPL_OK;
    void *dq, *dt, *didx, *ddist;
    int rc;
    if ((rc = upload_rows(ctx, WS_AUX0, q, nq, (size_t)dim * 4, q_stride * 4, &dq))) return rc;
    if ((rc = upload_rows(ctx, WS_AUX1, t, nt, (size_t)dim * 4, t_stride * 4, &dt))) return rc;
    if ((rc = ws_get(ctx, WS_IDX, (size_t)nq * k * 4, &didx))) return rc;
    if ((rc = ws_get(ctx, WS_DIST, (size_t)nq * k * 4, &ddist))) return rc;
    rc = launch_knn_l2(ctx, (const float *)dq, nq, dim, 0, (const float *)dt, nt, dim, 0, dim, k, 1, (int32_t *)didx,
                       (float *)ddist, ctx->stream);
    if (rc) return rc;
    MLPL_HIP_TRY(hipMemcpyAsync(idx, didx, (size_t)nq * k * 4, hipMemcpyDeviceToHost, ctx->stream));
    MLPL_HIP_TRY(hipMemcpyAsync(dist, ddist, (size_t)nq * k * 4, hipMemcpyDeviceToHost, ctx->stream));
    MLPL_HIP_TRY(hipStreamSynchronize(ctx->stream));
    return MLPL_OK;
}

static int ratio_compact_host(mlpl_ctx *ctx, const int32_t *idx, const void *dist, int is_float, int nq, int k,
                              float ratio, mlpl_dmatch *out, int *n_out) {
    if (!ctx || !idx || !dist || !out || !n_out || nq < 0 || (k != 1 && k != 2)) {
        set_error("mlpl_ratio_compact: bad arguments");
        return MLPL_E_BAD_INPUT;
    }
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    *n_out = 0;
    if (nq == 0) return MLPL_OK;
    void *didx, *ddist, *dout, *dcnt;
    int rc;
    if ((rc = ws_get(ctx, WS_IDX, (size_t)nq * k * 4, &didx))) return rc;
    if ((rc = ws_get(ctx, WS_DIST, (size_t)nq * k * 4, &ddist))) return rc;
    if ((rc = ws_get(ctx, WS_MATCH, (size_t)nq * sizeof(mlpl_dmatch), &dout))) return rc;
    if ((rc = ws_get(ctx, WS_SCALARS, 64, &dcnt))) return rc;
    MLPL_HIP_TRY(hipMemcpyAsync(didx, idx, (size_t)nq * k * 4, hipMemcpyHostToDevice, ctx->stream));
    MLPL_HIP_TRY(hipMemcpyAsync(ddist, dist, (size_t)nq * k * 4, hipMemcpyHostToDevice, ctx->stream));
    rc = launch_ratio_compact(ctx, (const int32_t *)didx, ddist, is_float, nq, k, 1, ratio, (mlpl_dmatch *)dout,
                              (int32_t *)dcnt, ctx->stream);
    if (rc) return rc;
    int32_t cnt = 0;
    MLPL_HIP_TRY(hipMemcpyAsync(&cnt, dcnt, 4, hipMemcpyDeviceToHost, ctx->stream));
    MLPL_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (cnt > 0) MLPL_HIP_TRY(hipMemcpy(out, dout, (size_t)cnt * sizeof(mlpl_dmatch), hipMemcpyDeviceToHost));
    *n_out = cnt;
    return MLPL_OK;
}

int mlpl_ratio_compact_i32(mlpl_ctx *ctx, const int32_t *idx, const int32_t *dist, int nq, int k, float ratio,
                           mlpl_dmatch *out, int *n_out) {
    return ratio_compact_host(ctx, idx, dist, 0, nq, k, ratio, out, n_out);
}

int mlpl_ratio_compact_f32(mlpl_ctx *ctx, const int32_t *idx, const float *dist, int nq, int k, float ratio,
                           mlpl_dmatch *out, int *n_out) {
    return ratio_compact_host(ctx, idx, dist, 1, nq, k, ratio, out, n_out);
}

int mlpl_get_matches_linear(mlpl_ctx *ctx, int n_keypoints1, int n_keypoints2, const void *desc1, int rows1, size_t step1,
                            const void *desc2, int rows2, size_t step2, int cols, int desc_type, int ratio_test,
                            mlpl_dmatch *out, int *n_out) {
    if (!ctx || !n_out) return MLPL_E_BAD_INPUT;
    *n_out = 0;
    // reference matchers.cpp:123-133
    if (n_keypoints1 < 15 || n_keypoints2 < 15) return MLPL_E_FEW_KEYPOINTS;
    if (n_keypoints1 != rows1 || n_keypoints2 != rows2) return MLPL_E_BAD_INPUT;
    // reference matchers.cpp:540-547
    if (desc_type != 0 && desc_type != 5) return MLPL_E_BAD_INPUT;
    if (!desc1 || !desc2 || !out || cols < 1) return MLPL_E_BAD_INPUT;
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    const int k = ratio_test ? 2 : 1;  // matchers.cpp:529-538
    const size_t elem = desc_type == 0 ? 1 : 4;
    if (step1 < (size_t)cols * elem || step2 < (size_t)cols * elem) return MLPL_E_BAD_INPUT;
    void *dq, *dt, *didx, *ddist, *dout, *dcnt;
    int rc;
    if ((rc = upload_rows(ctx, WS_AUX0, desc1, rows1, (size_t)cols * elem, step1, &dq))) return rc;
    if ((rc = upload_rows(ctx, WS_AUX1, desc2, rows2, (size_t)cols * elem, step2, &dt))) return rc;
    if ((rc = ws_get(ctx, WS_IDX, (size_t)rows1 * k * 4, &didx))) return rc;
    if ((rc = ws_get(ctx, WS_DIST, (size_t)rows1 * k * 4, &ddist))) return rc;
    if ((rc = ws_get(ctx, WS_MATCH, (size_t)rows1 * sizeof(mlpl_dmatch), &dout))) return rc;
    if ((rc = ws_get(ctx, WS_SCALARS, 64, &dcnt))) return rc;
    if (desc_type == 0) {
        rc = launch_knn_hamming(ctx, (const uint8_t *)dq, rows1, cols, 0, (const uint8_t *)dt, rows2, cols, 0, cols, k, 1,
                                (int32_t *)didx, (int32_t *)ddist, ctx->stream);
    } else {
        rc = launch_knn_l2(ctx, (const float *)dq, rows1, cols, 0, (const float *)dt, rows2, cols, 0, cols, k, 1,
                           (int32_t *)didx, (float *)ddist, ctx->stream);
    }
    if (rc) return rc == MLPL_E_BAD_INPUT ? MLPL_E_BAD_INPUT : rc;
    rc = launch_ratio_compact(ctx, (const int32_t *)didx, ddist, desc_type == 5, rows1, k, 1, 0.75f, (mlpl_dmatch *)dout,
                              (int32_t *)dcnt, ctx->stream);
    if (rc) return rc;
    int32_t cnt = 0;
    MLPL_HIP_TRY(hipMemcpyAsync(&cnt, dcnt, 4, hipMemcpyDeviceToHost, ctx->stream));
    MLPL_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (cnt > 0) MLPL_HIP_TRY(hipMemcpy(out, dout, (size_t)cnt * sizeof(mlpl_dmatch), hipMemcpyDeviceToHost));
    *n_out = cnt;
    if (cnt < 2) return MLPL_E_FAILED;  // matchers.cpp:709-713 (MIN_FINAL_MATCHES = 2)
    return MLPL_OK;
}

int mlpl_get_matches_bruteforce_nms(mlpl_ctx *ctx, int n_keypoints1, int n_keypoints2, const void *desc1, int rows1,
                                    size_t step1, const void *desc2, int rows2, size_t step2, int cols, int desc_type,
                                    int ratio_test, mlpl_dmatch *out, int *n_out) {
    if (!ctx || !n_out) return MLPL_E_BAD_INPUT;
    *n_out = 0;
    if (n_keypoints1 < 15 || n_keypoints2 < 15) return MLPL_E_FEW_KEYPOINTS;                 // matchers.cpp:123-127
    if (n_keypoints1 != rows1 || n_keypoints2 != rows2) return MLPL_E_BAD_INPUT;             // matchers.cpp:129-133
    if (desc_type != 0 && desc_type != 5) {
        set_error("BRUTEFORCENMS: CV_8U and CV_32F descriptors are built (the reference also takes CV_64F)");
        return desc_type == 6 ? MLPL_E_UNSUPPORTED : MLPL_E_BAD_INPUT;                       // matchers.cpp:514-518
    }
    if (!desc1 || !desc2 || !out || cols < 1 || rows2 < 2) return MLPL_E_BAD_INPUT;
    MLPL_HIP_TRY(hipSetDevice(ctx->device));
    const size_t elem = desc_type == 0 ? 1 : 4;
    if (step1 < (size_t)cols * elem || step2 < (size_t)cols * elem) return MLPL_E_BAD_INPUT;
    const int k = 2;  // K = 2 always (nmslib_matchers.h:182)
    void *dq, *dt, *didx, *ddist, *dout, *dcnt;
    int rc;
    if ((rc = upload_rows(ctx, WS_AUX0, desc1, rows1, (size_t)cols * elem, step1, &dq))) return rc;
    if ((rc = upload_rows(ctx, WS_AUX1, desc2, rows2, (size_t)cols * elem, step2, &dt))) return rc;
    if ((rc = ws_get(ctx, WS_IDX, (size_t)rows1 * k * 4, &didx))) return rc;
    if ((rc = ws_get(ctx, WS_DIST, (size_t)rows1 * k * 4, &ddist))) return rc;
    if ((rc = ws_get(ctx, WS_MATCH, (size_t)rows1 * sizeof(mlpl_dmatch), &dout))) return rc;
    if ((rc = ws_get(ctx, WS_SCALARS, 64, &dcnt))) return rc;
    if (desc_type == 0) {
        // two bytes per int, last int dropped by SpaceBitHamming::HiddenDistance: the last 2 bytes (1 for odd widths) are ignored
        const int eff = (cols % 2 == 0) ? cols - 2 : cols - 1;
        if (eff < 1) return MLPL_E_BAD_INPUT;
        rc = launch_knn_hamming(ctx, (const uint8_t *)dq, rows1, cols, 0, (const uint8_t *)dt, rows2, cols, 0, eff, k, 1,
                                (int32_t *)didx, (int32_t *)ddist, ctx->stream);
    } else {
        rc = launch_knn_l2(ctx, (const float *)dq, rows1, cols, 0, (const float *)dt, rows2, cols, 0, cols, k, 1,
                           (int32_t *)didx, (float *)ddist, ctx->stream, /*nms_order=*/1);
    }
    if (rc) return rc;
    rc = launch_ratio_compact(ctx, (const int32_t *)didx, ddist, desc_type == 5, rows1, k, 1, 0.75f, (mlpl_dmatch *)dout,
                              (int32_t *)dcnt, ctx->stream, nullptr, ratio_test ? 1 : 2);
    if (rc) return rc;
    int32_t cnt = 0;
    MLPL_HIP_TRY(hipMemcpyAsync(&cnt, dcnt, 4, hipMemcpyDeviceToHost, ctx->stream));
    MLPL_HIP_TRY(hipStreamSynchronize(ctx->stream));
    if (cnt > 0) MLPL_HIP_TRY(hipMemcpy(out, dout, (size_t)cnt * sizeof(mlpl_dmatch), hipMemcpyDeviceToHost));
    *n_out = cnt;
    return MLPL_OK;  // no minimum-match check on this branch of getMatches
}

}  // extern "C"
