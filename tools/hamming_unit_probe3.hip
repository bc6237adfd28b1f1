// Unit-cost probe with CONTROLLED occupancy: every kernel is compiled for exactly WPE waves per SIMD (amdgpu_waves_per_eu) and
// launched with WPE 4-wave blocks per CU; spills would show as scratch in the resource remarks (build with -Rpass-analysis=...).
// unit = one 32 x 32 distance tile (K = 256): 4 x v_mfma_scale_f32_32x32x64 (fp4) + the 22-op running top-2 per query tile.
// Build: hipcc -w -O3 --offload-arch=gfx950 -fno-honor-nans -o /tmp/probe3 tools/hamming_unit_probe3.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>
#include <algorithm>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v16f mf(uint4 a, uint4 b, v16f c) {
    const v8i av = {(int)a.x, (int)a.y, (int)a.z, (int)a.w, 0, 0, 0, 0};
    const v8i bv = {(int)b.x, (int)b.y, (int)b.z, (int)b.w, 0, 0, 0, 0};
#ifdef PROBE_SCALED   // rounds 1-4: the scaled form with unit scales in a register (a two-part instruction); default since round 5: the unscaled opcode
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
#else
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 4, 4, 0, 0, 0, 0);
#endif
}
__device__ __forceinline__ void update(float &m1r, float &m2r, const v16f &acc) {
    float m1 = m1r + 0.001953125f, m2 = m2r + 0.001953125f;
#pragma unroll
    for (int reg = 0; reg < 16; reg += 4) {
        const float s0 = __builtin_amdgcn_fmed3f(m1, acc[reg], acc[reg + 1]);
        const float t0 = __builtin_fmaxf(__builtin_fmaxf(m1, acc[reg]), acc[reg + 1]);
        const float s1 = __builtin_amdgcn_fmed3f(t0, acc[reg + 2], acc[reg + 3]);
        m1 = __builtin_fmaxf(__builtin_fmaxf(t0, acc[reg + 2]), acc[reg + 3]);
        m2 = __builtin_fmaxf(__builtin_fmaxf(m2, s0), s1);
    }
    m1r = m1, m2r = m2;
}

// QT query tiles per wave, ILV interleaved accumulator chains (ILV divides QT), DB = 1: next tile's A fragments prefetched into a
// second register set (4 loads per tile), 0: loaded into the same registers right before use (exposes the latency to the other waves)
template <int QT, int ILV, int WPE, int DB, int VALU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void probe(const uint4 *__restrict__ src, float *out,
                                                                                             int iters, long long *cyc) {
    __shared__ uint4 lds[2][256];
    for (int i = threadIdx.x; i < 512; i += 256) lds[i >> 8][i & 255] = src[i & 255];
    __syncthreads();
    const int l = threadIdx.x & 63;
    uint4 a[4], nx[4], b[QT][4];
    for (int s = 0; s < 4; ++s) a[s] = nx[s] = src[l + 64 * s];
    for (int t = 0; t < QT; ++t)
        for (int s = 0; s < 4; ++s) b[t][s] = src[l + 64 * (4 + 4 * t + s)];
    v16f cinit;
    for (int i = 0; i < 16; ++i) cinit[i] = -(float)i * (1.0f / 16384.0f);
    float m1[QT], m2[QT];
    for (int t = 0; t < QT; ++t) m1[t] = m2[t] = -1e30f;
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        const uint4 *p = src + l + 64 * (((i & 7) + 1) * 4);
        if (DB == 2) {  // A fragments from LDS (4 x ds_read_b128 per tile), nothing from L1
            asm volatile("" ::: "memory");
#pragma unroll
            for (int s = 0; s < 4; ++s) a[s] = lds[i & 1][s * 64 + l];
        } else if (DB) {
#pragma unroll
            for (int s = 0; s < 4; ++s) nx[s] = p[64 * s];
        } else {
#pragma unroll
            for (int s = 0; s < 4; ++s) a[s] = p[64 * s];
        }
#pragma unroll
        for (int g = 0; g < QT; g += ILV) {
            v16f acc[ILV];
#pragma unroll
            for (int u = 0; u < ILV; ++u) acc[u] = mf(a[0], b[g + u][0], cinit);
#pragma unroll
            for (int s = 1; s < 4; ++s)
#pragma unroll
                for (int u = 0; u < ILV; ++u) acc[u] = mf(a[s], b[g + u][s], acc[u]);
#pragma unroll
            for (int u = 0; u < ILV; ++u) {
                if (VALU) update(m1[g + u], m2[g + u], acc[u]);
                else m1[g + u] = __builtin_fmaxf(m1[g + u], acc[u][0]);
            }
        }
        if (DB == 1) {
#pragma unroll
            for (int s = 0; s < 4; ++s) a[s] = nx[s];
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        cyc[4 * w] = t1 - t0;
        cyc[4 * w + 1] = r1 - r0;
        cyc[4 * w + 2] = r0;
        cyc[4 * w + 3] = r1;
    }
    float s = 0.f;
    for (int t = 0; t < QT; ++t) s += m1[t] + m2[t];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 16383] = s;
}

template <int QT, int ILV, int WPE, int DB, int VALU>
void run(const uint4 *src, float *out) {
    long long *dc;
    const int blocks = 256 * WPE, waves = blocks * 4, iters = 3000;
    hipMalloc(&dc, waves * 32);
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, probe<QT, ILV, WPE, DB, VALU>, 256, 0);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe<QT, ILV, WPE, DB, VALU>), dim3(blocks), dim3(256), 0, 0, src, out, iters, dc);
    std::vector<long long> h(waves * 4);
    hipMemcpy(h.data(), dc, waves * 32, hipMemcpyDeviceToHost);
    double cs = 0, rs = 0;
    long long rmin = h[2], rmax = h[3];
    for (int i = 0; i < waves; ++i) {
        cs += h[4 * i], rs += h[4 * i + 1];
        rmin = std::min(rmin, h[4 * i + 2]);
        rmax = std::max(rmax, h[4 * i + 3]);
    }
    const double span_cycles = (double)(rmax - rmin) * (cs / rs);  // realtime ticks (100 MHz) -> shader cycles
    const double per_wave_unit = cs / waves / (iters * (double)QT);
    printf("QT=%d ILV=%d DB=%d VALU=%d waves/SIMD=%d (occupancy API: %d blocks/CU): %.3f GHz, %6.1f cycles/unit/wave -> %6.1f cycles per unit per SIMD\n",
           QT, ILV, DB, VALU, WPE, nb, cs / rs * 0.1, per_wave_unit, per_wave_unit / WPE);
    printf("      launch span: %.0f cycles -> %.1f cycles per unit per SIMD (AGGREGATE: all units / wall time)\n", span_cycles, span_cycles / (iters * (double)QT * WPE));
    hipFree(dc);
}

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f mf16(uint4 a, uint4 b, v4f c) {
    const v8i av = {(int)a.x, (int)a.y, (int)a.z, (int)a.w, 0, 0, 0, 0};
    const v8i bv = {(int)b.x, (int)b.y, (int)b.z, (int)b.w, 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}
// the same unit on v_mfma_scale_f32_16x16x128: 8 MFMAs (2 row halves x 2 query halves x 2 K halves), LDS-fed A, WPE waves per SIMD
template <int QT, int WPE, int VALU>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void probe16(const uint4 *__restrict__ src, float *out,
                                                                                               int iters, long long *cyc) {
    __shared__ uint4 lds[2][256];
    for (int i = threadIdx.x; i < 512; i += 256) lds[i >> 8][i & 255] = src[i & 255];
    __syncthreads();
    const int l = threadIdx.x & 63;
    uint4 a[4], b[QT][4];
    for (int t = 0; t < QT; ++t)
        for (int s = 0; s < 4; ++s) b[t][s] = src[l + 64 * (4 + 4 * t + s)];
    v4f c4[2];
    for (int i = 0; i < 4; ++i) c4[0][i] = -(float)i * (1.0f / 16384.0f), c4[1][i] = -(float)(16 + i) * (1.0f / 16384.0f);
    float m[QT][4];
    for (int t = 0; t < QT; ++t)
        for (int k = 0; k < 4; ++k) m[t][k] = -1e30f;
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        asm volatile("" ::: "memory");
#pragma unroll
        for (int s = 0; s < 4; ++s) a[s] = lds[i & 1][s * 64 + l];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            v4f acc[2][2];   // a[rh * 2 + kh], b[t][qh * 2 + kh]
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                for (int qh = 0; qh < 2; ++qh) acc[rh][qh] = mf16(a[rh * 2], b[t][qh * 2], c4[rh]);
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                for (int qh = 0; qh < 2; ++qh) acc[rh][qh] = mf16(a[rh * 2 + 1], b[t][qh * 2 + 1], acc[rh][qh]);
            if (VALU) {
#pragma unroll
                for (int qh = 0; qh < 2; ++qh) {
                    float m1 = m[t][2 * qh] + 0.001953125f, m2 = m[t][2 * qh + 1] + 0.001953125f;
#pragma unroll
                    for (int rh = 0; rh < 2; ++rh) {
                        const v4f &v = acc[rh][qh];
                        const float s0 = __builtin_amdgcn_fmed3f(m1, v[0], v[1]);
                        const float t0 = __builtin_fmaxf(__builtin_fmaxf(m1, v[0]), v[1]);
                        const float s1 = __builtin_amdgcn_fmed3f(t0, v[2], v[3]);
                        m1 = __builtin_fmaxf(__builtin_fmaxf(t0, v[2]), v[3]);
                        m2 = __builtin_fmaxf(__builtin_fmaxf(m2, s0), s1);
                    }
                    m[t][2 * qh] = m1, m[t][2 * qh + 1] = m2;
                }
            } else {
                m[t][0] = __builtin_fmaxf(m[t][0], acc[0][0][0] + acc[0][1][0] + acc[1][0][0] + acc[1][1][0]);
            }
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        cyc[4 * w] = t1 - t0;
        cyc[4 * w + 1] = r1 - r0;
        cyc[4 * w + 2] = r0;
        cyc[4 * w + 3] = r1;
    }
    float s = 0.f;
    for (int t = 0; t < QT; ++t)
        for (int k = 0; k < 4; ++k) s += m[t][k];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 16383] = s;
}
template <int QT, int WPE, int VALU>
void run16(const uint4 *src, float *out) {
    long long *dc;
    const int blocks = 256 * WPE, waves = blocks * 4, iters = 3000;
    hipMalloc(&dc, waves * 32);
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, probe16<QT, WPE, VALU>, 256, 0);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((probe16<QT, WPE, VALU>), dim3(blocks), dim3(256), 0, 0, src, out, iters, dc);
    std::vector<long long> h(waves * 4);
    hipMemcpy(h.data(), dc, waves * 32, hipMemcpyDeviceToHost);
    double cs = 0, rs = 0;
    long long rmin = h[2], rmax = h[3];
    for (int i = 0; i < waves; ++i) {
        cs += h[4 * i], rs += h[4 * i + 1];
        rmin = std::min(rmin, h[4 * i + 2]);
        rmax = std::max(rmax, h[4 * i + 3]);
    }
    const double span_cycles = (double)(rmax - rmin) * (cs / rs);  // realtime ticks (100 MHz) -> shader cycles
    const double per_wave_unit = cs / waves / (iters * (double)QT);
    printf("16x16x128: QT=%d VALU=%d waves/SIMD=%d (occupancy API: %d blocks/CU): %.3f GHz, %6.1f cycles/unit/wave -> %6.1f cycles per unit per SIMD\n",
           QT, VALU, WPE, nb, cs / rs * 0.1, per_wave_unit, per_wave_unit / WPE);
    printf("      launch span: %.0f cycles -> %.1f cycles per unit per SIMD (AGGREGATE)\n", span_cycles, span_cycles / (iters * (double)QT * WPE));
    hipFree(dc);
}

// The LDS ring of the real kernel: 4 waves share every train tile; wave w copies K-step w with global_load_lds_dwordx4.
// S tiles per stage (one s_barrier per stage), NBS stages in the ring, prefetch distance 2 stages.  SYNC = 1: s_barrier,
// 0: no synchronisation at all (wrong results; shows the price of the barrier alone), 2: barrier but no DMA (tile stays in LDS).
template <int QT, int WPE, int S, int SYNC>
__global__ __launch_bounds__(256) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void ringk(const uint4 *__restrict__ src, float *out,
                                                                                            int iters, long long *cyc) {
    constexpr int NBS = 4;
    __shared__ __attribute__((aligned(16))) uint4 ring[NBS * S][256];
    for (int i = threadIdx.x; i < NBS * S * 256; i += 256) ring[i >> 8][i & 255] = src[i & 255];
    __syncthreads();
    const int l = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint4 b[QT][4];
    for (int t = 0; t < QT; ++t)
        for (int s = 0; s < 4; ++s) b[t][s] = src[l + 64 * (4 + 4 * t + s)];
    v16f cinit;
    for (int i = 0; i < 16; ++i) cinit[i] = -(float)i * (1.0f / 16384.0f);
    float m1[QT], m2[QT];
    for (int t = 0; t < QT; ++t) m1[t] = m2[t] = -1e30f;
    const uint4 *tbase = src + (size_t)w * 64;
    auto copy_stage = [&](int st) {
        if (SYNC == 2) return;
#pragma unroll
        for (int k = 0; k < S; ++k)
            __builtin_amdgcn_global_load_lds((const void *)(tbase + (size_t)(((st * S + k) & 7) * 256) + l),
                                             (__attribute__((address_space(3))) void *)&ring[(st & (NBS - 1)) * S + k][w * 64], 16, 0, 0);
    };
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4 *)&ring[0][0] + (uint32_t)l * 16u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    copy_stage(0);
    copy_stage(1);
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const int nst = iters / S;
    for (int st = 0; st < nst; ++st) {
        copy_stage(st + 2);
        if (SYNC != 2) {
            if (S == 1) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
            else if (S == 2) asm volatile("s_waitcnt vmcnt(4)" ::: "memory");
            else asm volatile("s_waitcnt vmcnt(8)" ::: "memory");
        }
        if (SYNC) __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int k = 0; k < S; ++k) {
            u32x4 r[4];
            const uint32_t addr = ring_lds + (uint32_t)((st & (NBS - 1)) * S + k) * 4096u;
            asm volatile(
                "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
                : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
                : "v"(addr)
                : "memory");
            asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(r[0]));
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(r[1]));
            asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(r[2]));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[3]));
            uint4 a[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) a[s] = make_uint4(r[s].x, r[s].y, r[s].z, r[s].w);
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                v16f acc = mf(a[0], b[t][0], cinit);
#pragma unroll
                for (int s = 1; s < 4; ++s) acc = mf(a[s], b[t][s], acc);
                update(m1[t], m2[t], acc);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const int wv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        cyc[4 * wv] = t1 - t0;
        cyc[4 * wv + 1] = r1 - r0;
        cyc[4 * wv + 2] = r0;
        cyc[4 * wv + 3] = r1;
    }
    float sres = 0.f;
    for (int t = 0; t < QT; ++t) sres += m1[t] + m2[t];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 16383] = sres;
}
template <int QT, int WPE, int S, int SYNC>
void run_ring(const uint4 *src, float *out) {
    long long *dc;
    const int blocks = 256 * WPE, waves = blocks * 4, iters = 3000;
    hipMalloc(&dc, waves * 32);
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ringk<QT, WPE, S, SYNC>, 256, 0);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((ringk<QT, WPE, S, SYNC>), dim3(blocks), dim3(256), 0, 0, src, out, iters, dc);
    std::vector<long long> h(waves * 4);
    hipMemcpy(h.data(), dc, waves * 32, hipMemcpyDeviceToHost);
    double cs = 0, rs = 0;
    long long rmin = h[2], rmax = h[3];
    for (int i = 0; i < waves; ++i) {
        cs += h[4 * i], rs += h[4 * i + 1];
        rmin = std::min(rmin, h[4 * i + 2]);
        rmax = std::max(rmax, h[4 * i + 3]);
    }
    const double span_cycles = (double)(rmax - rmin) * (cs / rs);  // realtime ticks (100 MHz) -> shader cycles
    const double per_wave_unit = cs / waves / ((iters / S * S) * (double)QT);
    printf("ring: QT=%d S=%d SYNC=%d waves/SIMD=%d (occupancy API: %d blocks/CU): %.3f GHz, %6.1f cycles/unit/wave -> %6.1f cycles per unit per SIMD\n",
           QT, S, SYNC, WPE, nb, cs / rs * 0.1, per_wave_unit, per_wave_unit / WPE);
    printf("      launch span: %.0f cycles -> %.1f cycles per unit per SIMD (AGGREGATE)\n", span_cycles, span_cycles / ((iters / S * S) * (double)QT * WPE));
    hipFree(dc);
}

// 16-wave workgroups (1024 threads, ONE per CU): the four waves of every SIMD belong to the same workgroup and advance together
// (one barrier per stage of S tiles), so there is no spread of finish times.  Wave w copies K-step (w & 3) of the tiles t with
// (t & 3) == (w >> 2) of each group of four tiles; S must be a multiple of 4.
template <int QT, int S>
__global__ __launch_bounds__(1024) __attribute__((amdgpu_waves_per_eu(4, 4))) void bigk(const uint4 *__restrict__ src, float *out,
                                                                                        int iters, long long *cyc) {
    constexpr int NBS = 3;
    __shared__ __attribute__((aligned(16))) uint4 ring[NBS * S][256];
    for (int i = threadIdx.x; i < NBS * S * 256; i += 1024) ring[i >> 8][i & 255] = src[i & 255];
    __syncthreads();
    const int l = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint4 b[QT][4];
    for (int t = 0; t < QT; ++t)
        for (int s = 0; s < 4; ++s) b[t][s] = src[l + 64 * (4 + 4 * t + s)];
    v16f cinit;
    for (int i = 0; i < 16; ++i) cinit[i] = -(float)i * (1.0f / 16384.0f);
    float m1[QT], m2[QT];
    for (int t = 0; t < QT; ++t) m1[t] = m2[t] = -1e30f;
    const uint4 *tbase = src + (size_t)(w & 3) * 64;
    auto copy_stage = [&](int st) {
#pragma unroll
        for (int k = (w >> 2); k < S; k += 4)
            __builtin_amdgcn_global_load_lds((const void *)(tbase + (size_t)(((st * S + k) & 7) * 256) + l),
                                             (__attribute__((address_space(3))) void *)&ring[(st % NBS) * S + k][(w & 3) * 64], 16, 0, 0);
    };
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4 *)&ring[0][0] + (uint32_t)l * 16u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    copy_stage(0);
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const int nst = iters / S;
    for (int st = 0; st < nst; ++st) {
        copy_stage(st + 1);
        if (S == 4) asm volatile("s_waitcnt vmcnt(1)" ::: "memory");
        else asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __builtin_amdgcn_s_barrier();
#pragma unroll
        for (int k = 0; k < S; ++k) {
            u32x4 r[4];
            const uint32_t addr = ring_lds + (uint32_t)((st % NBS) * S + k) * 4096u;
            asm volatile(
                "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
                : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3])
                : "v"(addr)
                : "memory");
            asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(r[0]));
            asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(r[1]));
            asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(r[2]));
            asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[3]));
            uint4 a[4];
#pragma unroll
            for (int s = 0; s < 4; ++s) a[s] = make_uint4(r[s].x, r[s].y, r[s].z, r[s].w);
#pragma unroll
            for (int t = 0; t < QT; ++t) {
                v16f acc = mf(a[0], b[t][0], cinit);
#pragma unroll
                for (int s = 1; s < 4; ++s) acc = mf(a[s], b[t][s], acc);
                update(m1[t], m2[t], acc);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const int wv = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        cyc[4 * wv] = t1 - t0;
        cyc[4 * wv + 1] = r1 - r0;
        cyc[4 * wv + 2] = r0;
        cyc[4 * wv + 3] = r1;
    }
    float sres = 0.f;
    for (int t = 0; t < QT; ++t) sres += m1[t] + m2[t];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 16383] = sres;
}
template <int QT, int S>
void run_big(const uint4 *src, float *out) {
    long long *dc;
    const int blocks = 256, waves = blocks * 16, iters = 3000;
    hipMalloc(&dc, waves * 32);
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, bigk<QT, S>, 1024, 0);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((bigk<QT, S>), dim3(blocks), dim3(1024), 0, 0, src, out, iters, dc);
    std::vector<long long> h(waves * 4);
    hipMemcpy(h.data(), dc, waves * 32, hipMemcpyDeviceToHost);
    double cs = 0, rs = 0, mx = 0, mn = 1e30;
    long long rmin = h[2], rmax = h[3];
    for (int i = 0; i < waves; ++i) {
        cs += h[4 * i], rs += h[4 * i + 1];
        mx = std::max(mx, (double)h[4 * i]);
        mn = std::min(mn, (double)h[4 * i]);
        rmin = std::min(rmin, h[4 * i + 2]);
        rmax = std::max(rmax, h[4 * i + 3]);
    }
    const double span_cycles = (double)(rmax - rmin) * (cs / rs);
    const double per_wave_unit = cs / waves / ((iters / S * S) * (double)QT);
    printf("16-wave blocks: QT=%d S=%d (occupancy API: %d blocks/CU): %.3f GHz, %6.1f cycles/unit/wave -> %6.1f cycles per unit per SIMD (min/max wave %.0f / %.0f kcycles)\n",
           QT, S, nb, cs / rs * 0.1, per_wave_unit, per_wave_unit / 4, mn / 1e3, mx / 1e3);
    printf("      launch span: %.0f cycles -> %.1f cycles per unit per SIMD (AGGREGATE)\n", span_cycles, span_cycles / ((iters / S * S) * (double)QT * 4));
    hipFree(dc);
}


// Round 5: the ring with NW waves per workgroup (waves 0..3 copy one K-step each, all NW read the tile; one barrier per tile) at WPE waves per
// SIMD: what a workgroup of 10 waves with two query tiles per wave (5 waves per SIMD) or of 8 / 16 waves would cost per unit.
template <int QT, int NW, int WPE>
__global__ __launch_bounds__(64 * NW) __attribute__((amdgpu_waves_per_eu(WPE, WPE))) void ringnw(const uint4 *__restrict__ src, float *out, int iters, long long *cyc) {
    constexpr int NBS = 4;
    __shared__ __attribute__((aligned(16))) uint4 ring[NBS][256];
    for (int i = threadIdx.x; i < NBS * 256; i += 64 * NW) ring[i >> 8][i & 255] = src[i & 255];
    __syncthreads();
    const int l = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
    uint4 b[QT][4];
    for (int t = 0; t < QT; ++t)
        for (int s = 0; s < 4; ++s) b[t][s] = src[l + 64 * (4 + 4 * t + s)];
    v16f cinit;
    for (int i = 0; i < 16; ++i) cinit[i] = -(float)i * (1.0f / 16384.0f);
    float m1[QT], m2[QT];
    for (int t = 0; t < QT; ++t) m1[t] = m2[t] = -1e30f;
    const uint4 *tbase = src + (size_t)(w & 3) * 64;
    auto copy_stage = [&](int st) {
        if (w < 4)
            __builtin_amdgcn_global_load_lds((const void *)(tbase + (size_t)((st & 7) * 256) + l),
                                             (__attribute__((address_space(3))) void *)&ring[st & (NBS - 1)][w * 64], 16, 0, 0);
    };
    typedef uint32_t u32x4 __attribute__((ext_vector_type(4)));
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4 *)&ring[0][0] + (uint32_t)l * 16u;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    copy_stage(0);
    copy_stage(1);
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int st = 0; st < iters; ++st) {
        copy_stage(st + 2);
        if (w < 4) asm volatile("s_waitcnt vmcnt(2)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        u32x4 r[4];
        const uint32_t addr = ring_lds + (uint32_t)(st & (NBS - 1)) * 4096u;
        asm volatile("ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\tds_read_b128 %3, %4 offset:3072"
                     : "=&v"(r[0]), "=&v"(r[1]), "=&v"(r[2]), "=&v"(r[3]) : "v"(addr) : "memory");
        asm volatile("s_waitcnt lgkmcnt(3)" : "+v"(r[0]));
        asm volatile("s_waitcnt lgkmcnt(2)" : "+v"(r[1]));
        asm volatile("s_waitcnt lgkmcnt(1)" : "+v"(r[2]));
        asm volatile("s_waitcnt lgkmcnt(0)" : "+v"(r[3]));
        uint4 a[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) a[s] = make_uint4(r[s].x, r[s].y, r[s].z, r[s].w);
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            v16f acc = mf(a[0], b[t][0], cinit);
#pragma unroll
            for (int s = 1; s < 4; ++s) acc = mf(a[s], b[t][s], acc);
            update(m1[t], m2[t], acc);
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const int wv = blockIdx.x * NW + (threadIdx.x >> 6);
        cyc[4 * wv] = t1 - t0, cyc[4 * wv + 1] = r1 - r0, cyc[4 * wv + 2] = r0, cyc[4 * wv + 3] = r1;
    }
    float sres = 0.f;
    for (int t = 0; t < QT; ++t) sres += m1[t] + m2[t];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 16383] = sres;
}
template <int QT, int NW, int WPE>
void run_ringnw(const uint4 *src, float *out) {
    long long *dc;
    const int blocks = 256 * WPE * 4 / NW, waves = blocks * NW, iters = 3000;
    hipMalloc(&dc, waves * 32);
    int nb = 0;
    hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, ringnw<QT, NW, WPE>, 64 * NW, 0);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((ringnw<QT, NW, WPE>), dim3(blocks), dim3(64 * NW), 0, 0, src, out, iters, dc);
    std::vector<long long> h(waves * 4);
    hipMemcpy(h.data(), dc, waves * 32, hipMemcpyDeviceToHost);
    double cs = 0, rs = 0;
    long long rmin = h[2], rmax = h[3];
    for (int i = 0; i < waves; ++i) {
        cs += h[4 * i], rs += h[4 * i + 1];
        rmin = std::min(rmin, h[4 * i + 2]);
        rmax = std::max(rmax, h[4 * i + 3]);
    }
    const double span_cycles = (double)(rmax - rmin) * (cs / rs);
    printf("ringnw: QT=%d waves/workgroup=%d waves/SIMD=%d (occupancy API: %d blocks/CU): %.3f GHz -> %.1f cycles per unit per SIMD (AGGREGATE)\n", QT, NW, WPE, nb,
           cs / rs * 0.1, span_cycles / (iters * (double)QT * WPE));
    hipFree(dc);
}

int main() {
    uint4 *src;
    float *out;
    hipMalloc(&src, 64 * 64 * 16);
    hipMalloc(&out, 16384 * 4);
    std::vector<uint32_t> h(64 * 64 * 4);
    uint32_t x = 12345;
    for (auto &v : h) {
        uint32_t w = 0;
        for (int k = 0; k < 8; ++k) {
            x = x * 1664525u + 1013904223u;
            w |= ((x >> 16) & 1 ? 0x2u : 0xAu) << (4 * k);
        }
        v = w;
    }
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    run<4, 1, 3, 1, 1>(src, out);   // shipped r1 shape
    run<4, 1, 4, 2, 1>(src, out);   // LDS-fed, 4 waves
    run<4, 1, 4, 2, 0>(src, out);   // MFMA only
    run<2, 1, 5, 2, 1>(src, out);
    run<1, 1, 8, 2, 1>(src, out);
    run<1, 1, 8, 2, 0>(src, out);
    run<4, 2, 4, 2, 0>(src, out);   // two units' MFMA chains interleaved on two accumulators: MFMA only
    run<4, 2, 4, 2, 1>(src, out);   // ... with the top-2 update
    run<4, 2, 3, 2, 1>(src, out);   // ... at 3 waves per SIMD (168 registers)
    run_ring<4, 4, 1, 1>(src, out);
    run_big<4, 4>(src, out);
    run_ringnw<4, 4, 4>(src, out);    // = the shipped shape
    run_ringnw<4, 8, 4>(src, out);
    run_ringnw<4, 16, 4>(src, out);
    run_ringnw<2, 4, 5>(src, out);
    run_ringnw<2, 10, 5>(src, out);
    run_ringnw<2, 8, 4>(src, out);
    run_ringnw<3, 4, 4>(src, out);
    run_ringnw<3, 8, 4>(src, out);
    run_ringnw<1, 8, 8>(src, out);
    run_ringnw<1, 16, 8>(src, out);
    hipDeviceSynchronize();
    return 0;
}
