// knn_l2_common.h -- pieces shared by the squared-L2 kernels (knn_l2.hip, knn_l2_f16.hip).
#pragma once
#include "mlpl_internal.h"

namespace mlpl {

// lexicographic (distance bits, train row) running top-2 on 64-bit keys: cvflann's KNNUniqueResultSet order
__device__ __forceinline__ void l2_top2_update(unsigned long long &k0, unsigned long long &k1, unsigned long long key) {
    const bool lt0 = key < k0;
    const bool lt1 = key < k1;
    k1 = lt0 ? k0 : (lt1 ? key : k1);
    k0 = lt0 ? key : k0;
}

// cvflann::L2<float>::operator() restated for one 4-group: result += ((d0*d0 + d1*d1) + d2*d2) + d3*d3 with d = a - b, every operation
// rounded to nearest on its own (the reference is built without FMA contraction).
__device__ __forceinline__ float l2_group4(float result, float4 a, float4 b) {
    const float d0 = __fsub_rn(a.x, b.x), d1 = __fsub_rn(a.y, b.y), d2 = __fsub_rn(a.z, b.z), d3 = __fsub_rn(a.w, b.w);
    float s = __fadd_rn(__fmul_rn(d0, d0), __fmul_rn(d1, d1));
    s = __fadd_rn(s, __fmul_rn(d2, d2));
    s = __fadd_rn(s, __fmul_rn(d3, d3));
    return __fadd_rn(result, s);
}

}  // namespace mlpl
