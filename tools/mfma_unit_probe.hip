// What does one (4 x fp4 MFMA + 22 VALU top-2 update) unit cost per SIMD with W waves resident and NO memory traffic?
// Build: hipcc -w -O3 --offload-arch=gfx950 -fno-honor-nans -o /tmp/u tools/mfma_unit_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ v16f mf(uint4 a, uint4 b, v16f c) {
    const v8i av = {(int)a.x, (int)a.y, (int)a.z, (int)a.w, 0, 0, 0, 0};
    const v8i bv = {(int)b.x, (int)b.y, (int)b.z, (int)b.w, 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}

__device__ __forceinline__ void update(float &m1, float &m2, const v16f &acc) {
    m1 += 0.001953125f;
    m2 += 0.001953125f;
#pragma unroll
    for (int reg = 0; reg < 16; reg += 4) {
        const float s0 = __builtin_amdgcn_fmed3f(m1, acc[reg], acc[reg + 1]);
        const float t0 = __builtin_fmaxf(__builtin_fmaxf(m1, acc[reg]), acc[reg + 1]);
        const float s1 = __builtin_amdgcn_fmed3f(t0, acc[reg + 2], acc[reg + 3]);
        m1 = __builtin_fmaxf(__builtin_fmaxf(t0, acc[reg + 2]), acc[reg + 3]);
        m2 = __builtin_fmaxf(__builtin_fmaxf(m2, s0), s1);
    }
}

template <int PIPE, int MEM>
__global__ __launch_bounds__(1024) void unitk(const uint4 *src, float *out, int iters, long long *cyc) {
    __shared__ uint4 lbuf[64 * 4];
    if (threadIdx.x < 256) lbuf[threadIdx.x] = src[threadIdx.x];
    __syncthreads();
    const int l = threadIdx.x & 63;
    uint4 a[4], b[4][4];
    for (int s = 0; s < 4; ++s) a[s] = src[l + 64 * s];
    for (int t = 0; t < 4; ++t)
        for (int s = 0; s < 4; ++s) b[t][s] = src[l + 64 * (4 + 4 * t + s)];
    v16f cinit;
    for (int i = 0; i < 16; ++i) cinit[i] = -(float)i * (1.0f / 16384.0f);
    float m1[4], m2[4];
    for (int t = 0; t < 4; ++t) m1[t] = m2[t] = -1e30f;
    auto chain = [&](int t) {
        v16f acc = mf(a[0], b[t][0], cinit);
#pragma unroll
        for (int s = 1; s < 4; ++s) acc = mf(a[s], b[t][s], acc);
        return acc;
    };
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (PIPE == 0) {
        for (int i = 0; i < iters; ++i) {
            if (MEM == 1) {  // next tile's fragments: 4 x 1 KiB coalesced loads (L1/L2 hits), consumed one iteration later
                const uint4 *p = src + l + 64 * ((i & 3) * 4);
#pragma unroll
                for (int s = 0; s < 4; ++s) a[s] = p[64 * s];
            } else if (MEM == 2) {
#pragma unroll
                for (int s = 0; s < 4; ++s) a[s] = lbuf[l + 64 * s];
            }
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                v16f acc = chain(t);
                update(m1[t], m2[t], acc);
            }
        }
    } else {
        v16f cur = chain(0);
        for (int i = 0; i < iters; ++i) {
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                const v16f nxt = chain((t + 1) & 3);
                update(m1[t], m2[t], cur);
                if (PIPE == 1) {
#pragma unroll
                    for (int s = 0; s < 4; ++s) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 6, 0);
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < 3; ++s) {
                        __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                        __builtin_amdgcn_sched_group_barrier(0x002, 4, 0);
                    }
                    __builtin_amdgcn_sched_group_barrier(0x008, 1, 0);
                    __builtin_amdgcn_sched_group_barrier(0x002, 10, 0);
                }
                cur = nxt;
            }
        }
        m1[0] += cur[0];
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        cyc[2 * w] = t1 - t0;
        cyc[2 * w + 1] = r1 - r0;
    }
    float s = 0;
    for (int t = 0; t < 4; ++t) s += m1[t] + m2[t];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 16383] = s;
}

template <int PIPE, int MEM = 0>
void run(const uint4 *src, float *out, int wpb, int blocks) {
    long long *dc;
    const int waves = blocks * wpb, iters = 4000;
    hipMalloc(&dc, waves * 16);
    hipLaunchKernelGGL((unitk<PIPE, MEM>), dim3(blocks), dim3(64 * wpb), 0, 0, src, out, iters, dc);
    long long *h = new long long[waves * 2];
    hipMemcpy(h, dc, waves * 16, hipMemcpyDeviceToHost);
    double cs = 0, rs = 0;
    for (int i = 0; i < waves; ++i) cs += h[2 * i], rs += h[2 * i + 1];
    const double per_wave_unit = cs / waves / (iters * 4.0);
    const double wps = (double)waves / 1024.0;
    printf("mem=%d pipe=%d waves/SIMD=%.0f: %.3f GHz, %.1f cycles per unit per wave -> %.1f cycles per unit per SIMD\n", MEM, PIPE, wps, cs / rs * 0.1,
           per_wave_unit, per_wave_unit / wps);
    hipFree(dc);
    delete[] h;
}

// 4-wave workgroups sharing each train tile through LDS (ring of NB tiles), one barrier per tile.
//   SH = 1: staging by global_load_lds_dwordx4 (wave w moves K-step w), counted vmcnt, raw s_barrier
//   SH = 2: staging by one ordinary global_load_dwordx4 per wave + ds_write_b128, __syncthreads-free (raw barrier too)
template <int SH>
__global__ __launch_bounds__(256) void sharek(const uint4 *src, float *out, int iters, long long *cyc) {
    constexpr int NB = 4;
    __shared__ uint4 ring[NB * 256];
    const int l = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint4 b[4][4];
    for (int t = 0; t < 4; ++t)
        for (int s = 0; s < 4; ++s) b[t][s] = src[l + 64 * (4 + 4 * t + s)];
    v16f cinit;
    for (int i = 0; i < 16; ++i) cinit[i] = -(float)i * (1.0f / 16384.0f);
    float m1[4], m2[4];
    for (int t = 0; t < 4; ++t) m1[t] = m2[t] = -1e30f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4 *)ring + (uint32_t)l * 16u;
    auto stage = [&](int j) {
        const uint4 *g = src + ((j & 3) * 256) + w * 64 + l;
        uint4 *d = ring + (j & (NB - 1)) * 256 + w * 64;
        if (SH == 1) {
            __builtin_amdgcn_global_load_lds((const void *)g, (__attribute__((address_space(3))) void *)d, 16, 0, 0);
        }
    };
    uint4 staged;
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    if (SH == 1) {
        for (int j = 0; j < NB - 1; ++j) stage(j);
        for (int it = 0; it < iters; ++it) {
            asm volatile("s_waitcnt vmcnt(%0)" ::"n"(NB - 2) : "memory");
            __builtin_amdgcn_s_barrier();
            stage(it + NB - 1);
            uint4 a[4];
            const uint32_t addr = ring_lds + (uint32_t)((it & (NB - 1)) * 4096);
            asm volatile(
                "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3])
                : "v"(addr)
                : "memory");
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                v16f acc = mf(a[0], b[t][0], cinit);
#pragma unroll
                for (int s = 1; s < 4; ++s) acc = mf(a[s], b[t][s], acc);
                update(m1[t], m2[t], acc);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    } else {
        // register-staged: the load for tile it+2 is issued, tile it+1's data is written to LDS, tile it is consumed
        staged = src[w * 64 + l];
        for (int it = 0; it < iters; ++it) {
            // write my piece of tile it (loaded one iteration ago) into slot it & 1, then sync
            const uint32_t waddr = ring_lds + (uint32_t)((it & 1) * 4096 + w * 1024);
            typedef unsigned u32x4 __attribute__((ext_vector_type(4)));
            const u32x4 sv = {staged.x, staged.y, staged.z, staged.w};
            asm volatile("s_waitcnt vmcnt(0)\n\tds_write_b128 %0, %1\n\ts_waitcnt lgkmcnt(0)" ::"v"(waddr), "v"(sv) : "memory");
            __builtin_amdgcn_s_barrier();
            staged = src[(((it + 1) & 3) * 256) + w * 64 + l];
            uint4 a[4];
            const uint32_t addr = ring_lds + (uint32_t)((it & 1) * 4096);
            asm volatile(
                "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3])
                : "v"(addr)
                : "memory");
            __builtin_amdgcn_s_barrier();  // slot (it & 1) ^ 1 may be overwritten next iteration only after everyone has read... (2 slots)
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                v16f acc = mf(a[0], b[t][0], cinit);
#pragma unroll
                for (int s = 1; s < 4; ++s) acc = mf(a[s], b[t][s], acc);
                update(m1[t], m2[t], acc);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const int wi = blockIdx.x * 4 + (threadIdx.x >> 6);
        cyc[2 * wi] = t1 - t0;
        cyc[2 * wi + 1] = r1 - r0;
    }
    float sres = 0;
    for (int t = 0; t < 4; ++t) sres += m1[t] + m2[t];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 16383] = sres + staged.x;
}

template <int SH>
void run_share(const uint4 *src, float *out, int wps) {
    long long *dc;
    const int blocks = 256 * wps, waves = blocks * 4, iters = 1000;
    hipMalloc(&dc, waves * 16);
    hipLaunchKernelGGL((sharek<SH>), dim3(blocks), dim3(256), 0, 0, src, out, iters, dc);
    long long *h = new long long[waves * 2];
    hipMemcpy(h, dc, waves * 16, hipMemcpyDeviceToHost);
    double cs = 0, rs = 0;
    for (int i = 0; i < waves; ++i) cs += h[2 * i], rs += h[2 * i + 1];
    const double per_wave_unit = cs / waves / (iters * 4.0);
    printf("LDS-shared tiles, staging=%s, %d blocks of 4 waves per CU: %.3f GHz, %.1f cycles per unit per wave -> %.1f per SIMD\n",
           SH == 1 ? "global_load_lds" : "load+ds_write", wps, cs / rs * 0.1, per_wave_unit, per_wave_unit / wps);
    hipFree(dc);
    delete[] h;
}

// Group staging: G tiles per barrier (two LDS groups of G tiles, glds prefetch of the next group while the current one is
// consumed), 4-wave workgroups, A fragments single-buffered from LDS.
template <int G>
__global__ __launch_bounds__(256) void groupk(const uint4 *src, float *out, int iters, long long *cyc) {
    __shared__ uint4 ring[2 * G * 256];
    const int l = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    uint4 b[4][4];
    for (int t = 0; t < 4; ++t)
        for (int s = 0; s < 4; ++s) b[t][s] = src[l + 64 * (4 + 4 * t + s)];
    v16f cinit;
    for (int i = 0; i < 16; ++i) cinit[i] = -(float)i * (1.0f / 16384.0f);
    float m1[4], m2[4];
    for (int t = 0; t < 4; ++t) m1[t] = m2[t] = -1e30f;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    const uint32_t ring_lds = (uint32_t)(uintptr_t)(__attribute__((address_space(3))) uint4 *)ring + (uint32_t)l * 16u;
    auto stage_group = [&](int grp) {  // wave w moves K-step w of each of the G tiles: G glds per wave
#pragma unroll
        for (int j = 0; j < G; ++j) {
            const uint4 *g = src + (((grp * G + j) & 3) * 256) + w * 64 + l;
            uint4 *d = ring + ((grp & 1) * G + j) * 256 + w * 64;
            __builtin_amdgcn_global_load_lds((const void *)g, (__attribute__((address_space(3))) void *)d, 16, 0, 0);
        }
    };
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    stage_group(0);
    for (int grp = 0; grp < iters / G; ++grp) {
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");  // my pieces of this group have landed
        __builtin_amdgcn_s_barrier();                     // everyone's have, and everyone is done with the other group
        stage_group(grp + 1);
        for (int j = 0; j < G; ++j) {
            uint4 a[4];
            const uint32_t addr = ring_lds + (uint32_t)(((grp & 1) * G + j) * 4096);
            asm volatile(
                "ds_read_b128 %0, %4\n\tds_read_b128 %1, %4 offset:1024\n\tds_read_b128 %2, %4 offset:2048\n\t"
                "ds_read_b128 %3, %4 offset:3072\n\ts_waitcnt lgkmcnt(0)"
                : "=&v"(a[0]), "=&v"(a[1]), "=&v"(a[2]), "=&v"(a[3])
                : "v"(addr)
                : "memory");
#pragma unroll
            for (int t = 0; t < 4; ++t) {
                v16f acc = mf(a[0], b[t][0], cinit);
#pragma unroll
                for (int s = 1; s < 4; ++s) acc = mf(a[s], b[t][s], acc);
                update(m1[t], m2[t], acc);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const int wi = blockIdx.x * 4 + (threadIdx.x >> 6);
        cyc[2 * wi] = t1 - t0;
        cyc[2 * wi + 1] = r1 - r0;
    }
    float sres = 0;
    for (int t = 0; t < 4; ++t) sres += m1[t] + m2[t];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 16383] = sres;
}

template <int G>
void run_group(const uint4 *src, float *out, int wps) {
    long long *dc;
    const int blocks = 256 * wps, waves = blocks * 4, iters = 1024;
    hipMalloc(&dc, waves * 16);
    hipLaunchKernelGGL((groupk<G>), dim3(blocks), dim3(256), 0, 0, src, out, iters, dc);
    long long *h = new long long[waves * 2];
    hipMemcpy(h, dc, waves * 16, hipMemcpyDeviceToHost);
    double cs = 0, rs = 0;
    for (int i = 0; i < waves; ++i) cs += h[2 * i], rs += h[2 * i + 1];
    const double per_wave_unit = cs / waves / (iters * 4.0);
    printf("LDS group staging, %d tiles per barrier, %d blocks of 4 waves per CU: %.3f GHz, %.1f cycles per unit per wave -> %.1f per SIMD\n", G,
           wps, cs / rs * 0.1, per_wave_unit, per_wave_unit / wps);
    hipFree(dc);
    delete[] h;
}

// Barrier-free LDS sharing: each of the 4 waves of a workgroup stages ITS K-step piece of every tile (one ordinary
// global_load_dwordx4, prefetched one tile ahead, then ds_write_b128) into a ring of NB slots; per-slot LDS counters say
// "all four pieces of tile i have landed" (ready) and "all four waves have read tile i" (done).  Waves drift up to NB - 1 tiles
// apart; nobody waits at a barrier.  Spins are bounded (a protocol error must not hang the GPU).
template <int QT>
__global__ __launch_bounds__(256, 4) void flagk(const uint4 *src, float *out, int iters, long long *cyc) {
    constexpr int NB = 4;
    __shared__ uint4 ring[NB * 256];
    __shared__ unsigned ready[NB], done[NB];
    const int l = threadIdx.x & 63;
    const int w = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    if (threadIdx.x < NB) ready[threadIdx.x] = 0, done[threadIdx.x] = 0;
    __syncthreads();
    uint4 b[QT][4];
    for (int t = 0; t < QT; ++t)
        for (int s = 0; s < 4; ++s) b[t][s] = src[l + 64 * (4 + 4 * t + s)];
    v16f cinit;
    for (int i = 0; i < 16; ++i) cinit[i] = -(float)i * (1.0f / 16384.0f);
    float m1[QT], m2[QT];
    for (int t = 0; t < QT; ++t) m1[t] = m2[t] = -1e30f;
    volatile unsigned *vready = ready, *vdone = done;
    int bad = 0;
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    uint4 staged = src[w * 64 + l];  // my piece of tile 0
    for (int it = 0; it < iters; ++it) {
        const int slot = it & (NB - 1);
        const unsigned round = (unsigned)(it / NB);
        // WAR: slot is free once all four waves have read the tile that lived there NB tiles ago
        if (it >= NB) {
            int spins = 0;
            while (vdone[slot] < 4u * round && ++spins < (1 << 16)) {}
            bad |= spins >= (1 << 16);
        }
        ring[slot * 256 + w * 64 + l] = staged;                       // ds_write_b128 of my piece
        __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup");
        if (l == 0) atomicAdd(&ready[slot], 1u);
        staged = src[(((it + 1) & 3) * 256) + w * 64 + l];            // prefetch my piece of the next tile
        {
            int spins = 0;
            while (vready[slot] < 4u * (round + 1) && ++spins < (1 << 16)) {}
            bad |= spins >= (1 << 16);
        }
        __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "workgroup");
        uint4 a[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) a[s] = ring[slot * 256 + s * 64 + l];
#pragma unroll
        for (int t = 0; t < QT; ++t) {
            v16f acc = mf(a[0], b[t][0], cinit);
#pragma unroll
            for (int s = 1; s < 4; ++s) acc = mf(a[s], b[t][s], acc);
            update(m1[t], m2[t], acc);
        }
        if (l == 0) atomicAdd(&done[slot], 1u);  // my reads of this slot were consumed by the MFMAs above
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const int wi = blockIdx.x * 4 + (threadIdx.x >> 6);
        cyc[2 * wi] = t1 - t0;
        cyc[2 * wi + 1] = r1 - r0;
    }
    float sres = (float)bad;
    for (int t = 0; t < QT; ++t) sres += m1[t] + m2[t];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 16383] = sres + staged.x;
}

template <int QT>
void run_flag(const uint4 *src, float *out, int wps) {
    long long *dc;
    const int blocks = 256 * wps, waves = blocks * 4, iters = 1024;
    hipMalloc(&dc, waves * 16);
    hipLaunchKernelGGL((flagk<QT>), dim3(blocks), dim3(256), 0, 0, src, out, iters, dc);
    long long *h = new long long[waves * 2];
    hipMemcpy(h, dc, waves * 16, hipMemcpyDeviceToHost);
    double cs = 0, rs = 0;
    for (int i = 0; i < waves; ++i) cs += h[2 * i], rs += h[2 * i + 1];
    const double per_wave_unit = cs / waves / (iters * (double)QT);
    printf("LDS sharing with ready/done counters (no barrier), QT=%d, %d blocks of 4 waves per CU: %.3f GHz, %.1f cycles per unit per wave -> %.1f per SIMD\n",
           QT, wps, cs / rs * 0.1, per_wave_unit, per_wave_unit / wps);
    hipFree(dc);
    delete[] h;
}

int main() {
    uint4 *src;
    float *out;
    hipMalloc(&src, 64 * 40 * 16);
    hipMalloc(&out, 1 << 16);
    uint32_t *h = new uint32_t[64 * 40 * 4];
    uint32_t x = 12345;
    for (int i = 0; i < 64 * 40 * 4; ++i) {
        uint32_t w = 0;
        for (int n = 0; n < 8; ++n) {
            x = x * 1664525u + 1013904223u;
            w |= ((x >> 16) & 1 ? 0x2u : 0xAu) << (4 * n);
        }
        h[i] = w;
    }
    hipMemcpy(src, h, 64 * 40 * 16, hipMemcpyHostToDevice);
    for (int wpb : {4, 8, 12, 16}) {  // 256 blocks = one per CU -> wpb/4 waves per SIMD
        run<0>(src, out, wpb, 256);
        run<0, 1>(src, out, wpb, 256);
        run<0, 2>(src, out, wpb, 256);
    }
    for (int wps : {2, 3, 4}) {
        run_share<1>(src, out, wps);
        run_share<2>(src, out, wps);
    }
    for (int wps : {3, 4}) {
        run_group<2>(src, out, wps);
        run_group<4>(src, out, wps);
    }
    run_group<8>(src, out, 2);
    for (int wps : {3, 4}) {
        run_flag<4>(src, out, wps);
        run_flag<3>(src, out, wps);
    }
    return 0;
}
