// count_sgpr_probe.hip -- compile-only probe (round 5): the counting predicate with the POINTS in vector registers (two per lane, packed) and the
// MODELS streamed through scalar registers.  `hipcc --offload-arch=gfx950 -O3 -ffp-contract=off -save-temps -c` shows what the compiler makes of it:
// v_pk_fma_f32 takes an SGPR pair with op_sel broadcasting (one scalar operand per instruction: the constant bus), so the five model entries
// that enter as ADDENDS need three v_mov_b64 per model; per model and 128 points 21 packed + 4 compares + 3 moves + 1 v_writelane (counts by
// s_bcnt1 on the scalar unit) = 29 vector instructions -- what the LDS-tiled kernel issues per 128 evaluations today; two / four point pairs per
// lane amortise the moves to 27 / 26.  Not built: DESIGN 4.4.
#include <hip/hip_runtime.h>
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pkfma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
// models: 12 floats per model {e0..e8, km, pad, pad}; uniform loads
__global__ __launch_bounds__(512) void k(const float *__restrict__ models, int m_begin, int m_count, const float *__restrict__ pts, int n, float qf, int *__restrict__ out) {
    const int lane = threadIdx.x & 63;
    const int p0 = (blockIdx.x * 8 + (threadIdx.x >> 6)) * 128 + lane * 2;
    f32x2 X1 = {pts[p0 * 5], pts[p0 * 5 + 5]}, Y1 = {pts[p0 * 5 + 1], pts[p0 * 5 + 6]}, X2 = {pts[p0 * 5 + 2], pts[p0 * 5 + 7]}, Y2 = {pts[p0 * 5 + 3], pts[p0 * 5 + 8]},
          KP = {pts[p0 * 5 + 4], pts[p0 * 5 + 9]};
    const f32x2 Q = {qf, qf}, C6 = {0x1p-6f * 1.02f, 0x1p-6f * 1.02f};
    int vcnt = 0;
    int und = 0;
    for (int m = 0; m < m_count; ++m) {
        const float *e = models + (size_t)(m_begin + m) * 12;   // uniform address
        f32x2 E[9];
#pragma unroll
        for (int q = 0; q < 9; ++q) E[q] = f32x2{e[q], e[q]};
        const f32x2 KM = {e[9], e[9]};
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
        const int c = __popcll(__ballot(in0)) + __popcll(__ballot(in1));
        asm volatile("s_mov_b32 m0, %2\n\tv_writelane_b32 %0, %1, m0" : "+v"(vcnt) : "s"(c), "s"(m & 63) : "m0");
        if (__ballot(!(c0 && c1))) und += 1;
        if ((m & 63) == 63) {
            atomicAdd(&out[m_begin + m - 63 + lane], vcnt);
            vcnt = 0;
        }
    }
    if (und) out[0] += und;
}
