// What bounds one "unit" (32x32 distance tile = 4 fp4 MFMAs + the 22-op running top-2) of knn_hamming_mfma_kernel on a SIMD?
// Every variant defeats loop-invariant hoisting with an empty asm that "modifies" the A fragment, so the MFMAs really run.
// Build: hipcc -w -O3 --offload-arch=gfx950 -fno-honor-nans -o /tmp/probe2 tools/hamming_unit_probe2.hip
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
    return __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}
__device__ __forceinline__ void touch(uint4 &a) { asm volatile("" : "+v"(a.x), "+v"(a.y), "+v"(a.z), "+v"(a.w)); }

// VALU = 0: none; 1: the kernel's single-chain update (20 + 2 ops); 2: two independent chains (regs 0-7 / 8-15)
template <int VALU>
__device__ __forceinline__ void update(float (&m)[4], const v16f &acc) {
    if (VALU == 0) {
        m[0] = __builtin_fmaxf(m[0], acc[0]);   // keep the result alive: 1 op
        return;
    }
    if (VALU == 1) {
        float m1 = m[0] + 0.001953125f, m2 = m[1] + 0.001953125f;
#pragma unroll
        for (int reg = 0; reg < 16; reg += 4) {
            const float s0 = __builtin_amdgcn_fmed3f(m1, acc[reg], acc[reg + 1]);
            const float t0 = __builtin_fmaxf(__builtin_fmaxf(m1, acc[reg]), acc[reg + 1]);
            const float s1 = __builtin_amdgcn_fmed3f(t0, acc[reg + 2], acc[reg + 3]);
            m1 = __builtin_fmaxf(__builtin_fmaxf(t0, acc[reg + 2]), acc[reg + 3]);
            m2 = __builtin_fmaxf(__builtin_fmaxf(m2, s0), s1);
        }
        m[0] = m1, m[1] = m2;
        return;
    }
#pragma unroll
    for (int c = 0; c < 2; ++c) {
        float m1 = m[2 * c] + 0.001953125f, m2 = m[2 * c + 1] + 0.001953125f;
#pragma unroll
        for (int reg = 8 * c; reg < 8 * c + 8; reg += 4) {
            const float s0 = __builtin_amdgcn_fmed3f(m1, acc[reg], acc[reg + 1]);
            const float t0 = __builtin_fmaxf(__builtin_fmaxf(m1, acc[reg]), acc[reg + 1]);
            const float s1 = __builtin_amdgcn_fmed3f(t0, acc[reg + 2], acc[reg + 3]);
            m1 = __builtin_fmaxf(__builtin_fmaxf(t0, acc[reg + 2]), acc[reg + 3]);
            m2 = __builtin_fmaxf(__builtin_fmaxf(m2, s0), s1);
        }
        m[2 * c] = m1, m[2 * c + 1] = m2;
    }
}

// MEM = 0: operands stay in registers; 1: 4 x global_load_dwordx4 per tile (4 units), double-buffered (consumed one tile later);
//       2: 4 x ds_read_b128 per tile from LDS (volatile); 3: as 1 but ONE load per tile (what a 4-wave LDS share would leave per wave)
// NMFMA: MFMAs per unit (4 = the real thing, 0 = VALU only);  ACC2: 1 = software-pipelined with two accumulators
template <int VALU, int MEM, int NMFMA, int ACC2>
__global__ __launch_bounds__(256) void unitk(const uint4 *__restrict__ src, float *out, int iters, long long *cyc) {
    __shared__ uint4 lbuf[2][64 * 4];
    for (int i = threadIdx.x; i < 512; i += blockDim.x) lbuf[i >> 8][i & 255] = src[i & 255];
    __syncthreads();
    const int l = threadIdx.x & 63;
    uint4 a[4], nx[4], b[4][4];
    for (int s = 0; s < 4; ++s) a[s] = nx[s] = src[l + 64 * s];
    for (int t = 0; t < 4; ++t)
        for (int s = 0; s < 4; ++s) b[t][s] = src[l + 64 * (4 + 4 * t + s)];
    v16f cinit;
    for (int i = 0; i < 16; ++i) cinit[i] = -(float)i * (1.0f / 16384.0f);
    float m[4][4];
    for (int t = 0; t < 4; ++t)
        for (int k = 0; k < 4; ++k) m[t][k] = -1e30f;
    auto chain = [&](int t) {
        v16f acc = cinit;
        if (NMFMA >= 1) acc = mf(a[0], b[t][0], cinit);
#pragma unroll
        for (int s = 1; s < NMFMA; ++s) acc = mf(a[s], b[t][s], acc);
        return acc;
    };
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    v16f cur;
    if (ACC2) cur = chain(0);
    for (int i = 0; i < iters; ++i) {
        if (MEM == 1 || MEM == 3) {
            const uint4 *p = src + l + 64 * (((i & 7) + 1) * 4);
#pragma unroll
            for (int s = 0; s < (MEM == 1 ? 4 : 1); ++s) nx[s] = p[64 * s];
        } else if (MEM == 2) {
            const volatile uint4 *p = &lbuf[i & 1][l];
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                nx[s].x = p[64 * s].x, nx[s].y = p[64 * s].y, nx[s].z = p[64 * s].z, nx[s].w = p[64 * s].w;
            }
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            touch(a[t]);
            if (ACC2) {
                const v16f nxt = chain((t + 1) & 3);
                update<VALU>(m[t], cur);
                cur = nxt;
            } else {
                const v16f acc = chain(t);
                update<VALU>(m[t], acc);
            }
        }
        if (MEM != 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) a[s] = nx[s];
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        cyc[2 * w] = t1 - t0;
        cyc[2 * w + 1] = r1 - r0;
    }
    float s = ACC2 ? cur[3] : 0.f;
    for (int t = 0; t < 4; ++t)
        for (int k = 0; k < 4; ++k) s += m[t][k];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 16383] = s;
}

template <int VALU, int MEM, int NMFMA, int ACC2>
void run(const uint4 *src, float *out, int blocks_per_cu, const char *what) {
    long long *dc;
    const int blocks = 256 * blocks_per_cu, waves = blocks * 4, iters = 3000;
    hipMalloc(&dc, waves * 16);
    for (int rep = 0; rep < 2; ++rep)
        hipLaunchKernelGGL((unitk<VALU, MEM, NMFMA, ACC2>), dim3(blocks), dim3(256), 0, 0, src, out, iters, dc);
    std::vector<long long> h(waves * 2);
    hipMemcpy(h.data(), dc, waves * 16, hipMemcpyDeviceToHost);
    double cs = 0, rs = 0;
    for (int i = 0; i < waves; ++i) cs += h[2 * i], rs += h[2 * i + 1];
    const double per_wave_unit = cs / waves / (iters * 4.0);
    printf("%-46s waves/SIMD=%d: %.3f GHz, %6.1f cycles/unit/wave -> %6.1f cycles per unit per SIMD\n", what, blocks_per_cu, cs / rs * 0.1,
           per_wave_unit, per_wave_unit / blocks_per_cu);
    hipFree(dc);
}

// ILV interleaved accumulator chains: the K-steps of ILV units alternate (u0 k0, u1 k0, u0 k1, u1 k1, ...), so no MFMA waits for the
// one issued just before it; the ILV top-2 updates follow.  MEM as above (1 = 4 global loads per tile, double-buffered).
template <int ILV, int MEM>
__global__ __launch_bounds__(256) void ilvk(const uint4 *__restrict__ src, float *out, int iters, long long *cyc) {
    const int l = threadIdx.x & 63;
    uint4 a[4], nx[4], b[4][4];
    for (int s = 0; s < 4; ++s) a[s] = nx[s] = src[l + 64 * s];
    for (int t = 0; t < 4; ++t)
        for (int s = 0; s < 4; ++s) b[t][s] = src[l + 64 * (4 + 4 * t + s)];
    v16f cinit;
    for (int i = 0; i < 16; ++i) cinit[i] = -(float)i * (1.0f / 16384.0f);
    float m[4][4];
    for (int t = 0; t < 4; ++t)
        for (int k = 0; k < 4; ++k) m[t][k] = -1e30f;
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (MEM == 1) {
            const uint4 *p = src + l + 64 * (((i & 7) + 1) * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s) nx[s] = p[64 * s];
        }
#pragma unroll
        for (int g = 0; g < 4; g += ILV) {
            touch(a[g]);
            v16f acc[ILV];
#pragma unroll
            for (int u = 0; u < ILV; ++u) acc[u] = mf(a[0], b[g + u][0], cinit);
#pragma unroll
            for (int s = 1; s < 4; ++s)
#pragma unroll
                for (int u = 0; u < ILV; ++u) acc[u] = mf(a[s], b[g + u][s], acc[u]);
#pragma unroll
            for (int u = 0; u < ILV; ++u) update<1>(m[g + u], acc[u]);
        }
        if (MEM != 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) a[s] = nx[s];
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        cyc[2 * w] = t1 - t0;
        cyc[2 * w + 1] = r1 - r0;
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int k = 0; k < 4; ++k) s += m[t][k];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 16383] = s;
}

template <int ILV, int MEM>
void run_ilv(const uint4 *src, float *out, int blocks_per_cu, const char *what) {
    long long *dc;
    const int blocks = 256 * blocks_per_cu, waves = blocks * 4, iters = 3000;
    hipMalloc(&dc, waves * 16);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((ilvk<ILV, MEM>), dim3(blocks), dim3(256), 0, 0, src, out, iters, dc);
    std::vector<long long> h(waves * 2);
    hipMemcpy(h.data(), dc, waves * 16, hipMemcpyDeviceToHost);
    double cs = 0, rs = 0;
    for (int i = 0; i < waves; ++i) cs += h[2 * i], rs += h[2 * i + 1];
    const double per_wave_unit = cs / waves / (iters * 4.0);
    printf("%-46s waves/SIMD=%d: %.3f GHz, %6.1f cycles/unit/wave -> %6.1f cycles per unit per SIMD\n", what, blocks_per_cu, cs / rs * 0.1,
           per_wave_unit, per_wave_unit / blocks_per_cu);
    hipFree(dc);
}

typedef float v4f __attribute__((ext_vector_type(4)));
__device__ __forceinline__ v4f mf16(uint4 a, uint4 b, v4f c) {
    const v8i av = {(int)a.x, (int)a.y, (int)a.z, (int)a.w, 0, 0, 0, 0};
    const v8i bv = {(int)b.x, (int)b.y, (int)b.z, (int)b.w, 0, 0, 0, 0};
    return __builtin_amdgcn_mfma_scale_f32_16x16x128_f8f6f4(av, bv, c, 4, 4, 0, 0x7F7F7F7F, 0, 0x7F7F7F7F);
}
// The same unit with v_mfma_scale_f32_16x16x128: 8 MFMAs (2 row halves x 2 query halves x 2 K halves), four 4-register accumulators.
template <int VALU, int MEM>
__global__ __launch_bounds__(256) void k16(const uint4 *__restrict__ src, float *out, int iters, long long *cyc) {
    const int l = threadIdx.x & 63;
    uint4 a[4], nx[4], b[4][4];
    for (int s = 0; s < 4; ++s) a[s] = nx[s] = src[l + 64 * s];
    for (int t = 0; t < 4; ++t)
        for (int s = 0; s < 4; ++s) b[t][s] = src[l + 64 * (4 + 4 * t + s)];
    v4f c4[2];
    for (int i = 0; i < 4; ++i) c4[0][i] = -(float)i * (1.0f / 16384.0f), c4[1][i] = -(float)(16 + i) * (1.0f / 16384.0f);
    float m[4][4];
    for (int t = 0; t < 4; ++t)
        for (int k = 0; k < 4; ++k) m[t][k] = -1e30f;
    const long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
        if (MEM == 1) {
            const uint4 *p = src + l + 64 * (((i & 7) + 1) * 4);
#pragma unroll
            for (int s = 0; s < 4; ++s) nx[s] = p[64 * s];
        }
#pragma unroll
        for (int t = 0; t < 4; ++t) {
            touch(a[t]);
            // a[rh*2+kh], b[t][qh*2+kh]
            v4f acc[2][2];
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                for (int qh = 0; qh < 2; ++qh) acc[rh][qh] = mf16(a[rh * 2], b[t][qh * 2], c4[rh]);
#pragma unroll
            for (int rh = 0; rh < 2; ++rh)
#pragma unroll
                for (int qh = 0; qh < 2; ++qh) acc[rh][qh] = mf16(a[rh * 2 + 1], b[t][qh * 2 + 1], acc[rh][qh]);
            if (VALU) {
                // per lane: query half qh has its own running pair (m[t][2 qh], m[t][2 qh + 1]); 8 candidates each
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
        if (MEM != 0) {
#pragma unroll
            for (int s = 0; s < 4; ++s) a[s] = nx[s];
        }
    }
    const long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if ((threadIdx.x & 63) == 0) {
        const int w = blockIdx.x * (blockDim.x >> 6) + (threadIdx.x >> 6);
        cyc[2 * w] = t1 - t0;
        cyc[2 * w + 1] = r1 - r0;
    }
    float s = 0.f;
    for (int t = 0; t < 4; ++t)
        for (int k = 0; k < 4; ++k) s += m[t][k];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 16383] = s;
}
template <int VALU, int MEM>
void run16(const uint4 *src, float *out, int blocks_per_cu, const char *what) {
    long long *dc;
    const int blocks = 256 * blocks_per_cu, waves = blocks * 4, iters = 3000;
    hipMalloc(&dc, waves * 16);
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k16<VALU, MEM>), dim3(blocks), dim3(256), 0, 0, src, out, iters, dc);
    std::vector<long long> h(waves * 2);
    hipMemcpy(h.data(), dc, waves * 16, hipMemcpyDeviceToHost);
    double cs = 0, rs = 0;
    for (int i = 0; i < waves; ++i) cs += h[2 * i], rs += h[2 * i + 1];
    const double per_wave_unit = cs / waves / (iters * 4.0);
    printf("%-46s waves/SIMD=%d: %.3f GHz, %6.1f cycles/unit/wave -> %6.1f cycles per unit per SIMD\n", what, blocks_per_cu, cs / rs * 0.1,
           per_wave_unit, per_wave_unit / blocks_per_cu);
    hipFree(dc);
}

int main() {
    uint4 *src;
    float *out;
    hipMalloc(&src, 64 * 64 * 16);
    hipMalloc(&out, 16384 * 4);
    std::vector<uint32_t> h(64 * 64 * 4);
    uint32_t x = 12345;
    for (auto &v : h) {  // random +-1 nibbles (0x2 / 0xA)
        uint32_t w = 0;
        for (int k = 0; k < 8; ++k) {
            x = x * 1664525u + 1013904223u;
            w |= ((x >> 16) & 1 ? 0x2u : 0xAu) << (4 * k);
        }
        v = w;
    }
    hipMemcpy(src, h.data(), h.size() * 4, hipMemcpyHostToDevice);
    for (int w = 1; w <= 4; ++w) run<0, 0, 4, 0>(src, out, w, "4 MFMA only");
    for (int w = 1; w <= 4; ++w) run<1, 0, 0, 0>(src, out, w, "22 VALU only (1 chain)");
    for (int w = 1; w <= 4; ++w) run<1, 0, 4, 0>(src, out, w, "4 MFMA + 22 VALU, regs");
    for (int w = 1; w <= 4; ++w) run<1, 0, 4, 1>(src, out, w, "4 MFMA + 22 VALU, regs, 2 accumulators");
    for (int w = 2; w <= 4; ++w) run<1, 1, 4, 0>(src, out, w, "4 MFMA + 22 VALU + 4 global_load / tile");
    for (int w = 2; w <= 4; ++w) run<0, 1, 4, 0>(src, out, w, "4 MFMA + 4 global_load / tile (no VALU)");
    for (int w = 1; w <= 4; ++w) run_ilv<2, 0>(src, out, w, "2 interleaved chains + 2 x 22 VALU, regs");
    for (int w = 1; w <= 4; ++w) run_ilv<4, 0>(src, out, w, "4 interleaved chains + 4 x 22 VALU, regs");
    for (int w = 1; w <= 4; ++w) run_ilv<2, 1>(src, out, w, "2 interleaved chains + VALU + 4 loads / tile");
    for (int w = 1; w <= 4; ++w) run_ilv<4, 1>(src, out, w, "4 interleaved chains + VALU + 4 loads / tile");
    for (int w = 1; w <= 4; ++w) run16<0, 0>(src, out, w, "8 x 16x16x128 MFMA only");
    for (int w = 1; w <= 4; ++w) run16<1, 0>(src, out, w, "8 x 16x16x128 MFMA + 24 VALU, regs");
    for (int w = 1; w <= 4; ++w) run16<1, 1>(src, out, w, "8 x 16x16x128 MFMA + 24 VALU + 4 loads / tile");
    hipDeviceSynchronize();
    return 0;
}
