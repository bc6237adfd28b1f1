// Measures whether VALU issue overlaps an executing MFMA on gfx950: loop of {1 MFMA + N independent v_max3_f32}, one wave per SIMD.
// Build: hipcc -w -O3 --offload-arch=gfx950 -o /tmp/p tools/mfma_overlap_probe.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef short v8s __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

template <int N, int KIND>
__global__ void loopk(float *out, int iters, long long *cyc) {
    const int l = threadIdx.x;
    v8i av = {l, l + 1, l + 2, l + 3, 0, 0, 0, 0}, bv = {l * 3, l * 5, l * 7, l * 11, 0, 0, 0, 0};
    v8s ah = {(short)l, 1, 2, 3, 4, 5, 6, 7}, bh = {(short)(l * 3), 1, 2, 3, 4, 5, 6, 7};
    v16f c = {0};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(l + i);
    float y = (float)l * 0.5f, z = (float)l * 0.25f;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            if (KIND == 0) c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
            if (KIND == 1) c = __builtin_amdgcn_mfma_f32_32x32x16_bf16(ah, bh, c, 0, 0, 0);
#pragma unroll
            for (int k = 0; k < N; ++k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(y), "v"(z));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (l == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c[i];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[blockIdx.x * 64 + l] = s;
}

template <int N, int KIND>
void run(float *out, long long *dc, const char *name) {
    hipLaunchKernelGGL((loopk<N, KIND>), dim3(1), dim3(64), 0, 0, out, 2000, dc);
    long long c;
    hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    printf("%s N=%d cycles per {MFMA + N VALU} = %.1f\n", name, N, (double)c / 16000.0);
}

// chip-wide clock under the fp4-MFMA + VALU mix: s_memtime (shader cycles) against s_memrealtime (100 MHz)
template <int N>
__global__ void clockk(float *out, int iters, long long *cyc) {
    const int l = threadIdx.x & 63;
    v8i av = {l, l + 1, l + 2, l + 3, 0, 0, 0, 0}, bv = {l * 3, l * 5, l * 7, l * 11, 0, 0, 0, 0};
    v16f c = {0};
    float x[8];
    for (int i = 0; i < 8; ++i) x[i] = (float)(l + i);
    float y = (float)l * 0.5f, z = (float)l * 0.25f;
    long long t0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    for (int i = 0; i < iters; ++i) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            c = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
#pragma unroll
            for (int k = 0; k < N; ++k) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(x[k]) : "v"(y), "v"(z));
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    if (threadIdx.x == 0) {
        cyc[2 * blockIdx.x] = t1 - t0;
        cyc[2 * blockIdx.x + 1] = r1 - r0;
    }
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c[i];
    for (int i = 0; i < 8; ++i) s += x[i];
    out[(blockIdx.x * blockDim.x + threadIdx.x) & 16383] = s;
}

template <int N>
void run_clock(float *out, int wpb) {
    long long *dc;
    const int blocks = 512;
    hipMalloc(&dc, blocks * 16);
    hipLaunchKernelGGL((clockk<N>), dim3(blocks), dim3(64 * wpb), 0, 0, out, 20000, dc);
    long long *h = new long long[blocks * 2];
    hipMemcpy(h, dc, blocks * 16, hipMemcpyDeviceToHost);
    double cs = 0, rs = 0;
    for (int i = 0; i < blocks; ++i) cs += h[2 * i], rs += h[2 * i + 1];
    printf("chip-wide fp4 MFMA + %d VALU, %d waves/block x %d blocks: %.3f GHz, %.1f cycles per {MFMA+N VALU} per wave, %.2f ms\n", N, wpb, blocks,
           cs / rs * 0.1, cs / blocks / 160000.0, rs / blocks / 1e5);
    hipFree(dc);
    delete[] h;
}

int main() {
    float *out;
    long long *dc;
    hipMalloc(&out, 1 << 16);
    hipMalloc(&dc, 8);
    run<1, 2>(out, dc, "valu-only");
    run<4, 2>(out, dc, "valu-only");
    run<8, 2>(out, dc, "valu-only");
    run<0, 0>(out, dc, "fp4 32x32x64");
    run<2, 0>(out, dc, "fp4 32x32x64");
    run<4, 0>(out, dc, "fp4 32x32x64");
    run<6, 0>(out, dc, "fp4 32x32x64");
    run<8, 0>(out, dc, "fp4 32x32x64");
    run<0, 1>(out, dc, "bf16 32x32x16");
    run<2, 1>(out, dc, "bf16 32x32x16");
    run<4, 1>(out, dc, "bf16 32x32x16");
    run<6, 1>(out, dc, "bf16 32x32x16");
    run<8, 1>(out, dc, "bf16 32x32x16");
    run_clock<0>(out, 4);
    run_clock<5>(out, 4);
    run_clock<5>(out, 8);
    run_clock<0>(out, 8);
    return 0;
}
