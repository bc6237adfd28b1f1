#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#include <cmath>
typedef int v8i __attribute__((ext_vector_type(8)));
typedef float v16f __attribute__((ext_vector_type(16)));

// A: 32 rows x 64 k (fp4), B: 64 k x 32 cols. lane (r = l&31, h = l>>5) holds 32 nibbles: k = 32h + j
__global__ void probe(const uint4 *A, const uint4 *B, const float *Cin, float *D, int scale) {
    const int l = threadIdx.x;
    uint4 a = A[l], b = B[l];
    v8i av = {(int)a.x, (int)a.y, (int)a.z, (int)a.w, 0, 0, 0, 0};
    v8i bv = {(int)b.x, (int)b.y, (int)b.z, (int)b.w, 0, 0, 0, 0};
    v16f c;
    for (int i = 0; i < 16; ++i) c[i] = Cin[i * 64 + l];
    v16f d = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c, 4, 4, 0, scale, 0, scale);
    for (int i = 0; i < 16; ++i) D[i * 64 + l] = d[i];
}

__global__ void timing(const uint4 *A, const uint4 *B, float *D, int iters, long long *cyc) {
    const int l = threadIdx.x & 63;
    uint4 a = A[l], b = B[l];
    v8i av = {(int)a.x, (int)a.y, (int)a.z, (int)a.w, 0, 0, 0, 0};
    v8i bv = {(int)b.x, (int)b.y, (int)b.z, (int)b.w, 0, 0, 0, 0};
    v16f c0 = {0}, c1 = {0}, c2 = {0}, c3 = {0};
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int i = 0; i < iters; ++i) {
        c0 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c0, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        c1 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c1, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        c2 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c2, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
        c3 = __builtin_amdgcn_mfma_scale_f32_32x32x64_f8f6f4(av, bv, c3, 4, 4, 0, 0x7f7f7f7f, 0, 0x7f7f7f7f);
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    if (threadIdx.x == 0 && blockIdx.x == 0) cyc[0] = t1 - t0;
    float s = 0;
    for (int i = 0; i < 16; ++i) s += c0[i] + c1[i] + c2[i] + c3[i];
    D[blockIdx.x * blockDim.x + threadIdx.x] = s;
}

int main() {
    // random +-1 matrices, with a few zeros
    std::vector<int> a(32 * 64), b(64 * 32);
    srand(1);
    for (auto &x : a) { int r = rand() % 10; x = r == 0 ? 0 : (r & 1 ? 1 : -1); }
    for (auto &x : b) { int r = rand() % 10; x = r == 0 ? 0 : (r & 1 ? 1 : -1); }
    auto nib = [](int v) { return v == 0 ? 0x0u : (v > 0 ? 0x2u : 0xAu); };
    std::vector<uint32_t> Af(64 * 4, 0), Bf(64 * 4, 0);
    for (int l = 0; l < 64; ++l) {
        int r = l & 31, h = l >> 5;
        for (int j = 0; j < 32; ++j) {
            int k = 32 * h + j;
            Af[l * 4 + j / 8] |= nib(a[r * 64 + k]) << (4 * (j % 8));
            Bf[l * 4 + j / 8] |= nib(b[k * 32 + r]) << (4 * (j % 8));
        }
    }
    std::vector<float> Cin(16 * 64), D(16 * 64);
    const float eps = 1.0f / 16384.0f;
    for (int i = 0; i < 16; ++i)
        for (int l = 0; l < 64; ++l) Cin[i * 64 + l] = 192.0f - (float)((i * 64 + l) * 7 % 4096) * eps;  // integer + fraction
    uint4 *dA, *dB; float *dC, *dD; long long *dcyc;
    hipMalloc(&dA, 1024); hipMalloc(&dB, 1024); hipMalloc(&dC, 4096); hipMalloc(&dD, 1 << 20); hipMalloc(&dcyc, 8);
    hipMemcpy(dA, Af.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(dB, Bf.data(), 1024, hipMemcpyHostToDevice);
    hipMemcpy(dC, Cin.data(), 4096, hipMemcpyHostToDevice);
    for (int scale : {0x7f7f7f7f, 0}) {
        hipLaunchKernelGGL(probe, dim3(1), dim3(64), 0, 0, dA, dB, dC, dD, scale);
        hipMemcpy(D.data(), dD, 4096, hipMemcpyDeviceToHost);
        int bad = 0;
        for (int reg = 0; reg < 16; ++reg)
            for (int l = 0; l < 64; ++l) {
                int col = l & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (l >> 5);
                int dot = 0;
                for (int k = 0; k < 64; ++k) dot += a[row * 64 + k] * b[k * 32 + col];
                float want = (float)((double)Cin[reg * 64 + l] + dot);
                if (D[reg * 64 + l] != want) { if (bad < 5) printf("scale %x mismatch reg %d lane %d got %.8f want %.8f\n", scale, reg, l, D[reg*64+l], want); ++bad; }
            }
        printf("scale=%08x mismatches=%d\n", scale, bad);
    }
    for (int waves : {1, 2, 4, 8}) {
        hipLaunchKernelGGL(timing, dim3(1), dim3(64 * waves), 0, 0, dA, dB, dD, 1000, dcyc);
        long long c; hipMemcpy(&c, dcyc, 8, hipMemcpyDeviceToHost);
        printf("waves/block=%d cycles per MFMA (wave 0 view) = %.2f\n", waves, (double)c / 4000.0);
    }
    return 0;
}
