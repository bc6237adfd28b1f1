// count_mfma_probe.hip -- can the bilinear forms of the Sampson counting kernel run on the matrix cores BESIDE the packed-fp32 vector
// work?  Measures, at 4 waves per SIMD on every CU, per 1024 (model, correspondence) evaluations = one 32 x 32 tile:
//   V : the counting kernel's own instruction stream -- 21 packed (v_pk_fma/mul_f32) + 8 other vector ops per PAIR of evaluations per lane
//       (8 wave-iterations per 1024 evaluations);
//   M : the forms on v_mfma_f32_32x32x2_f32: E x1 (3 forms, K = 3 -> 4) and E^T x2 (2 forms, K = 3 -> 4) = 10 MFMAs per tile;
//   MV: the 10 MFMAs + the vector work that is left (x2^T E x1 from the three forms, squares, band, comparisons: 11 packed + 8 others per
//       pair of accumulator registers) in ONE wave's stream;
//   M|V: half of the resident waves run M, the other half run the left-over vector stream (two different waves of a SIMD: do the pipes overlap?).
// In-kernel clocks (s_memtime / s_memrealtime) give cycles, so the figures do not depend on the clock the chip holds.
//   hipcc --offload-arch=gfx950 -O3 tools/count_mfma_probe.hip -o /tmp/count_mfma_probe && /tmp/count_mfma_probe
#include <hip/hip_runtime.h>

#include <cstdio>
#include <vector>

typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef float v16f __attribute__((ext_vector_type(16)));

__device__ __forceinline__ f32x2 pkfma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }

// the vector work of one PAIR of evaluations in the present kernel: 21 packed + 8 other instructions
__device__ __forceinline__ int valu_full(const f32x2 *E, f32x2 X1, f32x2 Y1, f32x2 X2, f32x2 Y2, f32x2 KM, f32x2 KP, f32x2 Q, f32x2 C6, int cnt) {
    const f32x2 A = pkfma(E[0], X1, pkfma(E[1], Y1, E[2]));
    const f32x2 B = pkfma(E[3], X1, pkfma(E[4], Y1, E[5]));
    const f32x2 C = pkfma(E[6], X1, pkfma(E[7], Y1, E[8]));
    const f32x2 S = pkfma(X2, A, pkfma(Y2, B, C));
    const f32x2 A2 = pkfma(E[0], X2, pkfma(E[3], Y2, E[6]));
    const f32x2 B2 = pkfma(E[1], X2, pkfma(E[4], Y2, E[7]));
    const f32x2 D = pkfma(A, A, pkfma(B, B, pkfma(A2, A2, B2 * B2)));
    const f32x2 N = S * S;
    const f32x2 diff = pkfma(-Q, D, N);
    const f32x2 H = pkfma(C6, pkfma(Q, D, N), KM * KP);
    const bool in0 = diff.x < -H.x, in1 = diff.y < -H.y;
    const bool c0 = in0 || diff.x > H.x, c1 = in1 || diff.y > H.y;
    cnt += in0 ? 1 : 0;
    cnt += in1 ? 1 : 0;
    if (__builtin_expect(!(c0 && c1), 0)) cnt += 1000;
    return cnt;
}

// what is left of it when A, B, C, A2, B2 come out of the matrix cores: 11 packed + 8 others per pair
__device__ __forceinline__ int valu_rest(f32x2 A, f32x2 B, f32x2 C, f32x2 A2, f32x2 B2, f32x2 X2, f32x2 Y2, f32x2 KM, f32x2 KP, f32x2 Q, f32x2 C6, int cnt) {
    const f32x2 S = pkfma(X2, A, pkfma(Y2, B, C));
    const f32x2 D = pkfma(A, A, pkfma(B, B, pkfma(A2, A2, B2 * B2)));
    const f32x2 N = S * S;
    const f32x2 diff = pkfma(-Q, D, N);
    const f32x2 H = pkfma(C6, pkfma(Q, D, N), KM * KP);
    const bool in0 = diff.x < -H.x, in1 = diff.y < -H.y;
    const bool c0 = in0 || diff.x > H.x, c1 = in1 || diff.y > H.y;
    cnt += in0 ? 1 : 0;
    cnt += in1 ? 1 : 0;
    if (__builtin_expect(!(c0 && c1), 0)) cnt += 1000;
    return cnt;
}

// MODE 0 = V, 1 = M, 2 = MV, 3 = M|V (waves 0, 1 of a workgroup: M; waves 2, 3: the left-over vector stream)
template <int MODE>
__global__ __launch_bounds__(256, 4) void probe(const float *__restrict__ in, int *__restrict__ out, unsigned long long *__restrict__ stamps, int tiles) {
    const int l = threadIdx.x & 63, w = threadIdx.x >> 6;
    __shared__ __attribute__((aligned(16))) float tile_xy[256 * 8];
    __shared__ __attribute__((aligned(8))) float tile_k[256 * 2];
    for (int i = threadIdx.x; i < 256 * 8; i += 256) tile_xy[i] = in[i & 511] * 3.f;
    for (int i = threadIdx.x; i < 256 * 2; i += 256) tile_k[i] = in[(i * 7) & 511] * in[(i * 7) & 511];
    __syncthreads();
    f32x2 E[9];
#pragma unroll
    for (int k = 0; k < 9; ++k) E[k] = f32x2{in[k + l], in[k + 16 + l]};
    f32x2 X1 = {in[l + 32], in[l + 33]}, Y1 = {in[l + 34], in[l + 35]}, X2 = {in[l + 36], in[l + 37]}, Y2 = {in[l + 38], in[l + 39]};
    const f32x2 KM = {1e-9f, 1e-9f}, KP = {in[l + 40], in[l + 41]}, Q = {1e-6f, 1e-6f}, C6 = {0x1p-6f, 0x1p-6f};
    const float a0 = in[l + 50], a1 = in[l + 51], b0 = in[l + 52], b1 = in[l + 53];
    int cnt = 0;
    v16f acc[5];
#pragma unroll
    for (int f = 0; f < 5; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[f][r] = 0.f;
    const unsigned long long c0 = __builtin_amdgcn_s_memtime(), r0 = __builtin_amdgcn_s_memrealtime();
    const bool mfma_wave = MODE == 1 || MODE == 2 || (MODE == 3 && w < 2);
    const bool valu_wave = MODE == 0 || MODE == 2 || (MODE == 3 && w >= 2);
    for (int t = 0; t < tiles; ++t) {
        if (MODE == 0) {
#pragma unroll
            for (int it = 0; it < 8; ++it) {  // 8 wave-iterations x 64 lanes x 2 = 1024 evaluations; operands from LDS as in the kernel
                const int slot = ((t * 8 + it) * 4 + (l & 3)) & 255;
                const float4 v0 = *reinterpret_cast<const float4 *>(tile_xy + slot * 8);
                const float4 v1 = *reinterpret_cast<const float4 *>(tile_xy + slot * 8 + 4);
                const float2 kk = *reinterpret_cast<const float2 *>(tile_k + slot * 2);
                cnt = valu_full(E, f32x2{v0.x, v0.y}, f32x2{v0.z, v0.w}, f32x2{v1.x, v1.y}, f32x2{v1.z, v1.w}, KM, f32x2{kk.x, kk.y}, Q, C6, cnt);
            }
        } else {
            if (mfma_wave) {
#pragma unroll
                for (int f = 0; f < 5; ++f) {  // each form: K = 4 = two 32x32x2 MFMAs, C = 0 at the first
                    v16f z = acc[f];
                    if (MODE != 1) {
#pragma unroll
                        for (int r = 0; r < 16; ++r) z[r] = 0.f;
                    }
                    z = __builtin_amdgcn_mfma_f32_32x32x2f32(a0 + (float)f, b0, z, 0, 0, 0);
                    z = __builtin_amdgcn_mfma_f32_32x32x2f32(a1, b1 + (float)t, z, 0, 0, 0);
                    acc[f] = z;
                }
            }
            if (valu_wave) {
                const int slot = (t * 32 + (l & 31)) & 255;  // the lane's correspondence (column) of this tile: x2, y2, band constant
                const float4 v1 = *reinterpret_cast<const float4 *>(tile_xy + slot * 8 + 4);
                const float kp = tile_k[slot * 2];
                const f32x2 x2 = {v1.x, v1.x}, y2 = {v1.z, v1.z}, kpp = {kp, kp};
#pragma unroll
                for (int r = 0; r < 16; r += 2) {  // 8 register pairs = 16 evaluations per lane = 1024 per wave; KM per model (register)
                    const f32x2 A = {acc[0][r], acc[0][r + 1]}, B = {acc[1][r], acc[1][r + 1]}, C = {acc[2][r], acc[2][r + 1]};
                    const f32x2 A2 = {acc[3][r], acc[3][r + 1]}, B2 = {acc[4][r], acc[4][r + 1]};
                    cnt = valu_rest(A, B, C, A2, B2, x2, y2, E[r >> 1], kpp, Q, C6, cnt);
                }
            }
        }
    }
    const unsigned long long c1 = __builtin_amdgcn_s_memtime(), r1 = __builtin_amdgcn_s_memrealtime();
    float sum = 0;
#pragma unroll
    for (int f = 0; f < 5; ++f)
#pragma unroll
        for (int r = 0; r < 16; ++r) sum += acc[f][r];
    out[blockIdx.x * 256 + threadIdx.x] = cnt + (int)sum;
    if (l == 0) {
        unsigned long long *o = stamps + ((size_t)blockIdx.x * 4 + w) * 4;
        o[0] = c1 - c0, o[1] = r1 - r0, o[2] = r0, o[3] = r1;
    }
}

template <int MODE>
void run(const char *name, int tiles) {
    const int blocks = 256 * 4;  // 4 workgroups of 4 waves per CU: 4 waves per SIMD
    float *din;
    int *dout;
    unsigned long long *dst;
    hipMalloc(&din, 4096);
    std::vector<float> h(1024);
    for (int i = 0; i < 1024; ++i) h[i] = 0.001f * (float)((i * 37) % 101) - 0.05f;
    hipMemcpy(din, h.data(), 4096, hipMemcpyHostToDevice);
    hipMalloc(&dout, blocks * 256 * 4);
    hipMalloc(&dst, (size_t)blocks * 4 * 32);
    for (int rep = 0; rep < 3; ++rep) hipLaunchKernelGGL(probe<MODE>, dim3(blocks), dim3(256), 0, 0, din, dout, dst, tiles);
    hipDeviceSynchronize();
    std::vector<unsigned long long> st((size_t)blocks * 16);
    hipMemcpy(st.data(), dst, st.size() * 8, hipMemcpyDeviceToHost);
    unsigned long long rmin = ~0ull, rmax = 0;
    double cyc = 0, ticks = 0;
    for (int i = 0; i < blocks * 4; ++i) {
        cyc += (double)st[i * 4], ticks += (double)st[i * 4 + 1];
        if (st[i * 4 + 2] < rmin) rmin = st[i * 4 + 2];
        if (st[i * 4 + 3] > rmax) rmax = st[i * 4 + 3];
    }
    const double ghz = cyc / ticks * 0.1;
    const double span_us = (double)(rmax - rmin) / 100.0;
    // evaluations: V, M, MV: every wave does `tiles` tiles; M|V: a tile needs one M wave-tile AND one V wave-tile -> half the waves' tile count
    const double wave_tiles = (double)blocks * 4 * tiles * (MODE == 3 ? 0.5 : 1.0);
    const double cyc_per_tile_per_simd = span_us * 1e-6 * ghz * 1e9 / (wave_tiles / 1024.0);
    printf("%-4s tiles/wave %5d: span %8.1f us, clock %.3f GHz, SIMD-cycles per 1024 evaluations %7.1f  (%.3f per evaluation)\n", name, tiles, span_us, ghz,
           cyc_per_tile_per_simd, cyc_per_tile_per_simd / 1024.0);
    hipFree(din), hipFree(dout), hipFree(dst);
}

int main() {
    for (int tiles : {200, 800}) {
        run<0>("V", tiles);
        run<1>("M", tiles);
        run<2>("MV", tiles);
        run<3>("M|V", tiles);
    }
    return 0;
}
