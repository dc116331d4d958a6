// Inference batch-norm with the element-wise tail of a residual block fused in:
//   y = relu?( bn(x) [+ residual] )      or, for a block with a projection shortcut,
//   y = relu?( bn(x) + bn_r(residual) )  (the shortcut's own batch-norm, rounded where its kernel would store)
// for NCHW tensors, replacing three launches of the forward of the models of
// pruning/models.py:35-39 (MIOpen's batch-norm, torch's add, torch's clamp) by one that reads x (and the
// residual) once and writes y once.
//
// The arithmetic (bn_math.h) repeats what those three kernels do, value for value.
// For fp16 tensors this is what MIOpen's kernel computes when H*W is even; the host refuses odd
// plane sizes for fp16 (MIOpen's code for those fuses the rounding into the multiply-add).
#include "bn_math.h"

#include <hip/hip_fp16.h>

namespace spice {

constexpr int kBnThreads = 256;
constexpr int kBnVecPerLane = 2;

struct BnArgs {
    const void *x;
    const void *res;
    void *y;
    const float *mean, *var, *weight, *bias;
    const float *r_mean, *r_var, *r_weight, *r_bias;   // batch-norm of the residual (RES_BN), else unused
    float eps, r_eps;
    unsigned total;   // elements
    unsigned n_vec;   // total / VEC
    unsigned hw, c;
    int relu;
};

__device__ __forceinline__ BnChannel bn_channel(const BnArgs &a, unsigned ch) {
    return bn_channel_of(a.mean, a.var, a.weight, a.bias, a.eps, ch);
}

__device__ __forceinline__ BnChannel bn_channel_res(const BnArgs &a, unsigned ch) {
    return bn_channel_of(a.r_mean, a.r_var, a.r_weight, a.r_bias, a.r_eps, ch);
}

template <typename T, int VEC>
struct Pack {
    typedef T type __attribute__((ext_vector_type(VEC)));
};

template <typename T, int VEC, bool HAS_RES, bool RES_BN>
__device__ __forceinline__ void bn_vector(const BnArgs &a, typename Pack<T, VEC>::type xv,
                                          typename Pack<T, VEC>::type rv, unsigned idx,
                                          typename Pack<T, VEC>::type *out) {
    const unsigned plane = idx / a.hw;
    const unsigned rem = idx - plane * a.hw;
    const unsigned ch0 = plane % a.c;
    const unsigned ch1 = ch0 + 1 == a.c ? 0 : ch0 + 1;
    const BnChannel p0 = bn_channel(a, ch0);
    const BnChannel p1 = bn_channel(a, ch1);   // a vector spans at most two planes (hw >= VEC)
    BnChannel q0 = p0, q1 = p1;
    if constexpr (RES_BN) {
        q0 = bn_channel_res(a, ch0);
        q1 = bn_channel_res(a, ch1);
    }
    typename Pack<T, VEC>::type o;
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const bool next = rem + e >= a.hw;
        const float v = bn_tail<T, HAS_RES, RES_BN>(static_cast<float>(xv[e]), static_cast<float>(rv[e]), next ? p1 : p0,
                                                    next ? q1 : q0, a.relu);
        o[e] = static_cast<T>(v);
    }
    *out = o;
}

template <typename T, int VEC, bool HAS_RES, bool RES_BN>
__global__ __launch_bounds__(kBnThreads) void bn_act_kernel(BnArgs a) {
    typedef typename Pack<T, VEC>::type vec_t;
    const vec_t *x = static_cast<const vec_t *>(a.x);
    const vec_t *r = static_cast<const vec_t *>(a.res);
    vec_t *y = static_cast<vec_t *>(a.y);
    const unsigned first = blockIdx.x * (kBnThreads * kBnVecPerLane) + threadIdx.x;
    const bool full = first - threadIdx.x + kBnThreads * kBnVecPerLane <= a.n_vec;   // uniform
    if (full) {
        vec_t xv[kBnVecPerLane], rv[kBnVecPerLane];
#pragma unroll
        for (int u = 0; u < kBnVecPerLane; ++u) {
            xv[u] = x[first + u * kBnThreads];
            if constexpr (HAS_RES) rv[u] = r[first + u * kBnThreads];
            else rv[u] = xv[u];
        }
#pragma unroll
        for (int u = 0; u < kBnVecPerLane; ++u) {
            const unsigned v = first + u * kBnThreads;
            bn_vector<T, VEC, HAS_RES, RES_BN>(a, xv[u], rv[u], v * VEC, y + v);
        }
    } else {
        for (int u = 0; u < kBnVecPerLane; ++u) {
            const unsigned v = first + u * kBnThreads;
            if (v < a.n_vec) {
                vec_t rv = x[v];
                if constexpr (HAS_RES) rv = r[v];
                bn_vector<T, VEC, HAS_RES, RES_BN>(a, x[v], rv, v * VEC, y + v);
            }
        }
    }
}

template <typename T, int VEC>
static int launch_bn(const BnArgs &a, hipStream_t st) {
    const unsigned per_block = kBnThreads * kBnVecPerLane;
    const dim3 grid((a.n_vec + per_block - 1) / per_block);
    if (a.res != nullptr && a.r_mean != nullptr)
        hipLaunchKernelGGL((bn_act_kernel<T, VEC, true, true>), grid, dim3(kBnThreads), 0, st, a);
    else if (a.res != nullptr)
        hipLaunchKernelGGL((bn_act_kernel<T, VEC, true, false>), grid, dim3(kBnThreads), 0, st, a);
    else
        hipLaunchKernelGGL((bn_act_kernel<T, VEC, false, false>), grid, dim3(kBnThreads), 0, st, a);
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}

}  // namespace spice

static int bn_act_launch(const void *x, const void *residual, void *y, int32_t dtype, const float *mean,
                         const float *var, const float *weight, const float *bias, float eps, const float *r_mean,
                         const float *r_var, const float *r_weight, const float *r_bias, float r_eps, int64_t n,
                         int64_t c, int64_t hw, int32_t relu, void *stream) {
    using namespace spice;
    if (n < 0 || c <= 0 || hw <= 0) return SPICE_ERR_ARG;
    if (n == 0) return SPICE_OK;
    if (x == nullptr || y == nullptr || mean == nullptr || var == nullptr || weight == nullptr || bias == nullptr)
        return SPICE_ERR_ARG;
    if (dtype != 0 && dtype != 1) return SPICE_ERR_UNSUPPORTED;
    if (dtype == 1 && (hw & 1)) return SPICE_ERR_UNSUPPORTED;   // see the header of this file
    if (n > INT32_MAX / c || n * c > INT32_MAX / hw) return SPICE_ERR_UNSUPPORTED;   // 32-bit element indices
    const int64_t total = n * c * hw;
    const int vec = dtype == 1 ? 8 : 4;
    const bool aligned = aligned16(x) && aligned16(y) && (residual == nullptr || aligned16(residual));
    const bool wide = aligned && total % vec == 0 && hw >= vec;
    BnArgs a;
    a.x = x;
    a.res = residual;
    a.y = y;
    a.mean = mean;
    a.var = var;
    a.weight = weight;
    a.bias = bias;
    a.eps = eps;
    a.r_mean = r_mean;
    a.r_var = r_var;
    a.r_weight = r_weight;
    a.r_bias = r_bias;
    a.r_eps = r_eps;
    a.total = static_cast<unsigned>(total);
    a.n_vec = static_cast<unsigned>(wide ? total / vec : total);
    a.hw = static_cast<unsigned>(hw);
    a.c = static_cast<unsigned>(c);
    a.relu = tail_code(relu);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (dtype == 1) return wide ? launch_bn<_Float16, 8>(a, st) : launch_bn<_Float16, 1>(a, st);
    return wide ? launch_bn<float, 4>(a, st) : launch_bn<float, 1>(a, st);
}

extern "C" int spice_bn_act(const void *x, const void *residual, void *y, int32_t dtype, const float *mean,
                            const float *var, const float *weight, const float *bias, float eps, int64_t n,
                            int64_t c, int64_t hw, int32_t relu, void *stream) {
    return bn_act_launch(x, residual, y, dtype, mean, var, weight, bias, eps, nullptr, nullptr, nullptr, nullptr, 0.0f, n,
                         c, hw, relu, stream);
}

extern "C" int spice_bn_act_dual(const void *x, const void *residual, void *y, int32_t dtype, const float *mean,
                                 const float *var, const float *weight, const float *bias, float eps,
                                 const float *r_mean, const float *r_var, const float *r_weight, const float *r_bias,
                                 float r_eps, int64_t n, int64_t c, int64_t hw, int32_t relu, void *stream) {
    if (residual == nullptr || r_mean == nullptr || r_var == nullptr || r_weight == nullptr || r_bias == nullptr)
        return SPICE_ERR_ARG;
    return bn_act_launch(x, residual, y, dtype, mean, var, weight, bias, eps, r_mean, r_var, r_weight, r_bias, r_eps, n, c,
                         hw, relu, stream);
}
