// hub_kernels.h -- kernel ids of the launches a BatchHub can merge, and the Args forms of the solver kernels shared with the RANSAC path.
// Included by ransac_5pt.hip inside namespace mlpl after batch_hub.h (the kernels' bodies are defined above it in that file).
#pragma once

namespace {

enum HubKernelId {
    HK_SOLVE3 = 0,     // solve5pt3_body: three hypotheses per wave
    HK_SOLVE1,         // solve5pt_body: one per wave (option solver_wave3 = 0)
    HK_ROOTS_POLISH,   // roots_body<true>
    HK_ROOTS_PLAIN,    // roots_body<false>
    HK_REFIT_SOLVE,    // refit_solve_body
    HK_COPY_BYTES,     // host-mapped -> device byte copy
    HK_USAC_POOL_PACK,
    HK_USAC_CHECK,
    HK_USAC_LO,
    HK_USAC_DG_ROWS,
    HK_USAC5_BEGIN,
    HK_USAC5_CHOOSE,
    HK_USAC5_FIT,      // refit_solve_body + roots_body + usac5_choose_body of one chain in one wave
    HK_USAC5_EVAL,
    HK_USAC5_GRAM,
    HK_ARR_SAMPLE,
    HK_ARR_VALID,
    HK_ARR_CHECK,
    HK_ARR_GATHER,
    HK_ARR_EXTEND,
    HK_ARR_MASK_COUNT,
    HK_ARR_REFINE,
    HK_PACK_POINTS,
    HK_NUM
};
static_assert(HK_NUM <= kHubMaxKernels, "kernel table too small");

struct SolveArgs {
    KHdr hdr;
    const double *p1, *p2;
    const int32_t *samples;
    int n_samples;
    PolyRec *recs;
};
__device__ __forceinline__ void hub_solve3_body(const SolveArgs &a, int bx, int) { solve5pt3_body(a.p1, a.p2, a.samples, 0, a.n_samples, a.recs, nullptr, 0, bx); }
__device__ __forceinline__ void hub_solve1_body(const SolveArgs &a, int bx, int) { solve5pt_body(a.p1, a.p2, a.samples, 0, a.n_samples, a.recs, nullptr, 0, bx); }
MLPL_HUB_KERNEL(HK_SOLVE3, SolveArgs, hub_solve3_body, 64);
MLPL_HUB_KERNEL(HK_SOLVE1, SolveArgs, hub_solve1_body, 64);

struct RootsArgs {
    KHdr hdr;
    const PolyRec *recs;
    int n_samples;
    double *E_tab;
    int32_t *n_models;
};
template <bool kPolish>
__device__ __forceinline__ void hub_roots_body(const RootsArgs &a, int bx, int) {
    roots_body<kPolish>(a.recs, 0, a.n_samples, a.E_tab, a.n_models, nullptr, nullptr, nullptr, nullptr, 0, bx);
}
MLPL_HUB_KERNEL(HK_ROOTS_POLISH, RootsArgs, hub_roots_body<true>, 64);
MLPL_HUB_KERNEL(HK_ROOTS_PLAIN, RootsArgs, hub_roots_body<false>, 64);

struct RefitSolveArgs {
    KHdr hdr;
    const double *gram_part;
    int nparts;
    PolyRec *rec;
    size_t part_stride;
    const char *gate;
    size_t gate_stride;
    char *warm;          // optional: 82 doubles per system, warm_stride bytes apart (refit_solve_body)
    size_t warm_stride;
};
__device__ __forceinline__ void hub_refit_solve_body(const RefitSolveArgs &a, int bx, int) {
    refit_solve_body(a.gram_part, a.nparts, a.rec, a.part_stride, a.gate, a.gate_stride, bx, a.warm, a.warm_stride);
}
MLPL_HUB_KERNEL(HK_REFIT_SOLVE, RefitSolveArgs, hub_refit_solve_body, 64);

struct CopyBytesArgs {  // n bytes from (device-visible, 16-byte aligned, padded) src to dst of ANY alignment; 256 threads x 16 bytes per block
    KHdr hdr;
    const uint4 *src;
    uint8_t *dst;
    int n;
};
__device__ __forceinline__ void hub_copy_bytes_body(const CopyBytesArgs &a, int bx, int) {
    const int i = bx * 256 + threadIdx.x;
    if (i * 16 >= a.n) return;
    const uint4 v = a.src[i];
    uint8_t *d = a.dst + (size_t)i * 16;
    if ((reinterpret_cast<uintptr_t>(d) & 15) == 0 && i * 16 + 16 <= a.n) {
        *reinterpret_cast<uint4 *>(d) = v;
        return;
    }
    const uint32_t w[4] = {v.x, v.y, v.z, v.w};
    for (int k = 0; k < 16 && i * 16 + k < a.n; ++k) d[k] = (uint8_t)(w[k >> 2] >> (8 * (k & 3)));
}
MLPL_HUB_KERNEL(HK_COPY_BYTES, CopyBytesArgs, hub_copy_bytes_body, 256);

struct PackPtsArgs {  // (x1, y1, x2, y2) rows of the correspondences: what ARRSAC's kernels read
    KHdr hdr;
    const double *p1, *p2;
    int n;
    double4 *pts;
};
__device__ __forceinline__ void hub_pack_pts_body(const PackPtsArgs &a, int bx, int) {
    const int i = bx * 256 + threadIdx.x;
    if (i < a.n) a.pts[i] = make_double4(a.p1[2 * i], a.p1[2 * i + 1], a.p2[2 * i], a.p2[2 * i + 1]);
}
MLPL_HUB_KERNEL(HK_PACK_POINTS, PackPtsArgs, hub_pack_pts_body, 256);

struct MaskCountArgs {  // inlier_mask_count_kernel: findInliers with one model; *count must be zero before (count_zero: block 0 of a launch BEFORE zeroes it)
    KHdr hdr;
    const double4 *pts;
    int n;
    const double *E;
    double thresh2;
    uint8_t *mask;
    int32_t *count;
    double *E_copy;  // optional: the model, for the host (9 doubles)
};
__device__ __forceinline__ void hub_mask_count_body(const MaskCountArgs &a, int bx, int) {
    const int i = bx * 256 + threadIdx.x;
    if (a.E_copy && bx == 0 && threadIdx.x < 9) a.E_copy[threadIdx.x] = a.E[threadIdx.x];
    bool in = false;
    if (i < a.n) {
        double e[9];
#pragma unroll
        for (int k = 0; k < 9; ++k) e[k] = a.E[k];
        const double4 p = a.pts[i];
        in = (double)sampson_err_f32(e, p.x, p.y, p.z, p.w) <= a.thresh2;
        a.mask[i] = in ? 1 : 0;
    }
    const unsigned long long b = __ballot(in);
    if ((threadIdx.x & 63) == 0 && b) atomicAdd(a.count, __popcll(b));
}
MLPL_HUB_KERNEL(HK_ARR_MASK_COUNT, MaskCountArgs, hub_mask_count_body, 256);

// the five-point solver + root kernels for `count` samples through a Launcher
inline void hub_launch_solver(Launcher &L, mlpl_ctx *ctx, int count, const double *p1, const double *p2, const int32_t *samples, PolyRec *recs,
                              double *E_tab, int32_t *n_models) {
    SolveArgs sa{{0, 1}, p1, p2, samples, count, recs};
    if (ctx->opt_solver_wave3) {
        sa.hdr.gx = (count + kHypPerSolveWave - 1) / kHypPerSolveWave;
        L.launch(HK_SOLVE3, sa);
    } else {
        sa.hdr.gx = count;
        L.launch(HK_SOLVE1, sa);
    }
    RootsArgs ra{{(count + kHypPerWave - 1) / kHypPerWave, 1}, recs, count, E_tab, n_models};
    L.launch(ctx->opt_solver_polish ? HK_ROOTS_POLISH : HK_ROOTS_PLAIN, ra);
}
inline void hub_copy_bytes(Launcher &L, const void *src_dev_visible, void *dst, size_t bytes) {
    const int n16 = (int)((bytes + 15) / 16);
    CopyBytesArgs ca{{(n16 + 255) / 256, 1}, (const uint4 *)src_dev_visible, (uint8_t *)dst, (int)bytes};
    L.launch(HK_COPY_BYTES, ca);
}

}  // namespace
