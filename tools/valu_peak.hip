// valu_peak.hip -- measures the int32 VALU issue rate on gfx950 (v_xor_b32 / v_bcnt_u32_b32 chains), to price the
// Hamming kernel against the real peak.  hipcc --offload-arch=gfx950 -O3 tools/valu_peak.hip -o /tmp/valu_peak
#include <hip/hip_runtime.h>
#include <cstdio>

template <int MODE>
__global__ __launch_bounds__(256) void k(unsigned *out, int iters) {
    unsigned a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 2654435761u + i;
    unsigned acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (MODE == 0) {  // xor chain (8 independent)
                asm volatile("v_xor_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i]));
            } else if (MODE == 1) {  // bcnt accumulate
                asm volatile("v_bcnt_u32_b32 %0, %1, %0" : "+v"(acc[i]) : "v"(a[i]));
            } else if (MODE == 2) {  // fma f32
                float f = __uint_as_float(acc[i]);
                asm volatile("v_fma_f32 %0, %1, %1, %0" : "+v"(f) : "v"(__uint_as_float(a[i])));
                acc[i] = __float_as_uint(f);
            } else {  // med3
                asm volatile("v_med3_u32 %0, %1, %0, %2" : "+v"(acc[i]) : "v"(a[i]), "v"(a[(i + 1) & 7]));
            }
        }
    }
    unsigned r = 0;
    for (int i = 0; i < 8; ++i) r ^= acc[i];
    out[blockIdx.x * 256 + threadIdx.x] = r;
}

template <int MODE>
void run(const char *name, int blocks) {
    unsigned *d;
    hipMalloc(&d, blocks * 256 * 4);
    const int iters = 4096;
    hipEvent_t e0, e1;
    hipEventCreate(&e0);
    hipEventCreate(&e1);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipDeviceSynchronize();
    hipEventRecord(e0);
    hipLaunchKernelGGL(k<MODE>, dim3(blocks), dim3(256), 0, 0, d, iters);
    hipEventRecord(e1);
    hipEventSynchronize(e1);
    float ms;
    hipEventElapsedTime(&ms, e0, e1);
    double winstr = (double)blocks * 4 * iters * 8;  // wave-instructions
    double per_simd = winstr / 1024.0;
    printf("%-6s blocks=%5d: %.3f ms  -> %.2f Tlane-ops/s, %.2f ns per wave-instr per SIMD (= %.2f cyc @2.4GHz)\n", name, blocks,
           ms, winstr * 64 / ms / 1e9, ms * 1e6 / per_simd, ms * 1e6 / per_simd * 2.4);
    hipFree(d);
}

int main() {
    for (int blocks : {256, 512, 1024, 2048}) {
        run<0>("xor", blocks);
        run<1>("bcnt", blocks);
        run<2>("fma", blocks);
        run<3>("med3", blocks);
    }
    return 0;
}
