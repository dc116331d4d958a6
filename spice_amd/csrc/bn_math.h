// The arithmetic of an inference batch-norm and of the element-wise tail of a residual block, one
// value at a time, shared by the kernels that apply it (bnact.hip on a stored tensor, conv1x1_mfma.hip
// on the product it has just computed) so that both give the same value for the same input:
//   bn:   inv = v_rsq_f32(|var + eps|);  t = (x - mean) * inv;  o = fma(weight, t, bias)   (fp32)
//         rounded to the tensor's dtype (for fp16: to nearest even, as a separate step)
//   add:  the two rounded operands added in fp32, rounded again
//   relu: max(v, 0), NaN kept
//   silu (tail code 2, the EfficientNet blocks): v / (1 + exp(-v)) in fp32 on the rounded value, rounded again
//         (torch's kernel for the dtype does the same; the exponential and the reciprocal here are the hardware's
//         1-ulp instructions, so a result can differ from torch's in its last fp32 bits and - rarely - by one step of
//         the storage type)
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

__device__ __forceinline__ float silu_f32(float v) {
    return v * __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * v));
}

// Tail codes of the entry points' `relu` argument: 0 none, 1 ReLU, 2 SiLU (any other non-zero value: ReLU).
__host__ __device__ __forceinline__ int tail_code(int relu) { return relu == 2 ? 2 : (relu ? 1 : 0); }

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
    if (relu == 1) v = v != v ? v : fmaxf(v, 0.0f);
    else if (relu == 2) v = silu_f32(v);
    return v;
}

typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

// bn_tail of bn_math.h for two neighbouring pixels of one channel at once, value for value the same: the fp32 steps
// as packed operations, every rounding to fp16 a conversion of its own (never folded into the multiply-add before
// it), the clamp on the rounded halves (rounding is monotonic and keeps 0 and NaN, so max-then-round and
// round-then-max agree).
template <bool HAS_RES, bool RES_BN>
__device__ __forceinline__ half2_t bn_tail_pair(half2_t x, half2_t r, const BnChannel &p, const BnChannel &q, int relu) {
    const float2_t xf = {static_cast<float>(x[0]), static_cast<float>(x[1])};
    const float2_t t = (xf - float2_t{p.mean, p.mean}) * float2_t{p.inv, p.inv};
    float2_t o = __builtin_elementwise_fma(float2_t{p.weight, p.weight}, t, float2_t{p.bias, p.bias});
    asm volatile("" : "+v"(o));
    half2_t v = {static_cast<_Float16>(o[0]), static_cast<_Float16>(o[1])};
    if constexpr (HAS_RES) {
        float2_t rf = {static_cast<float>(r[0]), static_cast<float>(r[1])};
        if constexpr (RES_BN) {
            const float2_t rt = (rf - float2_t{q.mean, q.mean}) * float2_t{q.inv, q.inv};
            float2_t ro = __builtin_elementwise_fma(float2_t{q.weight, q.weight}, rt, float2_t{q.bias, q.bias});
            asm volatile("" : "+v"(ro));
            const half2_t rh = {static_cast<_Float16>(ro[0]), static_cast<_Float16>(ro[1])};
            rf = float2_t{static_cast<float>(rh[0]), static_cast<float>(rh[1])};
        }
        float2_t sum = float2_t{static_cast<float>(v[0]), static_cast<float>(v[1])} + rf;
        asm volatile("" : "+v"(sum));
        v = half2_t{static_cast<_Float16>(sum[0]), static_cast<_Float16>(sum[1])};
    }
    if (relu == 1) v = __builtin_elementwise_maximum(v, half2_t{0, 0});   // NaN-propagating maximum
    else if (relu == 2) {   // SiLU on the rounded halves, in fp32, rounded again (bn_math.h)
        const float2_t f = {silu_f32(static_cast<float>(v[0])), silu_f32(static_cast<float>(v[1]))};
        v = half2_t{static_cast<_Float16>(f[0]), static_cast<_Float16>(f[1])};
    }
    return v;
}

// The same for tail codes 0 and 1 without a branch on the code: the clamp as a lower bound, {0, 0} for ReLU and
// {-inf, -inf} for none (the NaN-propagating maximum of a value and -inf is the value).
__device__ __forceinline__ half2_t bn_lower_bound(int relu) {
    const _Float16 lo = relu == 1 ? static_cast<_Float16>(0.0f) : -static_cast<_Float16>(__builtin_inff());
    return half2_t{lo, lo};
}
template <bool HAS_RES, bool RES_BN>
__device__ __forceinline__ half2_t bn_tail_pair_lb(half2_t x, half2_t r, const BnChannel &p, const BnChannel &q, half2_t lower) {
    return __builtin_elementwise_maximum(bn_tail_pair<HAS_RES, RES_BN>(x, r, p, q, 0), lower);
}

}  // namespace spice
