// The arithmetic of an inference batch-norm and of the element-wise tail of a residual block, one
// value at a time, shared by the kernels that apply it (bnact.hip on a stored tensor, conv1x1_mfma.hip
// on the product it has just computed) so that both give the same value for the same input:
//   bn:   inv = v_rsq_f32(|var + eps|);  t = (x - mean) * inv;  o = fma(weight, t, bias)   (fp32)
//         rounded to the tensor's dtype (for fp16: to nearest even, as a separate step)
//   add:  the two rounded operands added in fp32, rounded again
//   relu: max(v, 0), NaN kept
#pragma once

#include "common.h"

namespace spice {

struct BnChannel {
    float mean, inv, weight, bias;
};

__device__ __forceinline__ BnChannel bn_channel_of(const float *mean, const float *var, const float *weight,
                                                   const float *bias, float eps, unsigned ch) {
    BnChannel p;
    p.mean = mean[ch];
    p.inv = __builtin_amdgcn_rsqf(fabsf(var[ch] + eps));
    p.weight = weight[ch];
    p.bias = bias[ch];
    return p;
}

// round to the storage type and come back, so that later steps see what a separate kernel would read
template <typename T>
__device__ __forceinline__ float round_trip(float v);
template <>
__device__ __forceinline__ float round_trip<float>(float v) { return v; }
template <>
__device__ __forceinline__ float round_trip<_Float16>(float v) {
    asm volatile("" : "+v"(v));   // the conversion stays a step of its own (no mixed-precision multiply-add)
    return static_cast<float>(static_cast<_Float16>(v));
}

// relu?( bn_p(x) [+ r | + bn_q(r)] ) for one element; x and r are values of the storage type T
template <typename T, bool HAS_RES, bool RES_BN>
__device__ __forceinline__ float bn_tail(float x, float r, const BnChannel &p, const BnChannel &q, int relu) {
    const float t = (x - p.mean) * p.inv;
    float v = round_trip<T>(__builtin_fmaf(p.weight, t, p.bias));
    if constexpr (HAS_RES) {
        if constexpr (RES_BN)   // the shortcut's batch-norm, stored in T by its own kernel before the add
            r = round_trip<T>(__builtin_fmaf(q.weight, (r - q.mean) * q.inv, q.bias));
        v = round_trip<T>(v + r);
    }
    if (relu) v = v != v ? v : fmaxf(v, 0.0f);
    return v;
}

}  // namespace spice
