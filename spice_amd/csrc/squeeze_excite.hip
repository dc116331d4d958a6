// Squeeze-and-excite of the EfficientNet blocks (the models of pruning/models.py:35-39 built on
// "timm-efficientnet-b3"), fp16 NCHW, inference:
//     s = mean over the plane of x          [N, C]
//     h = silu(W1 s + b1)                   [N, R]
//     g = sigmoid(W2 h + b2)                [N, C]
//     y = x * g
// torch runs this as a reduction, two 1x1 convolutions on 1 x 1 planes through MIOpen (whose solver look-up a sweep
// candidate with a new channel count triggers), three element-wise launches and the product: eight-odd launches per
// block, 26 blocks per forward.  Here it is four short ones: the plane means (a wave per plane), the hidden values (a
// wave per value), the gate (a lane per channel), the product with x.  Every intermediate is rounded to fp16 where torch's
// modules would have stored it under fp16 autocast (the mean, each convolution's result, each activation), sums in
// fp32: within a few fp16 steps of torch's gate, which multiplies x.
#include "common.h"
#include "bn_math.h"

#include <hip/hip_fp16.h>

namespace spice {

constexpr int kSeThreads = 256;
constexpr int kSeMaxChannels = 4096;   // the gate kernel holds an image's means in LDS

typedef _Float16 se_f16x8_t __attribute__((ext_vector_type(8)));

__device__ __forceinline__ float se_round_half(float v) {
    asm volatile("" : "+v"(v));
    return static_cast<float>(static_cast<_Float16>(v));
}

// mean[plane] = mean of x[plane, :], a wave per plane, fp32 sum, rounded to fp16 (as torch's mean of a half tensor)
__global__ __launch_bounds__(kSeThreads) void se_mean_kernel(const _Float16 *__restrict__ x, float *__restrict__ mean,
                                                             int planes, int hw) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int plane = blockIdx.x * (kSeThreads / 64) + wave;
    if (plane >= planes) return;
    const _Float16 *src = x + static_cast<int64_t>(plane) * hw;
    float sum = 0.0f;
    if ((hw & 7) == 0 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
        const se_f16x8_t *v = reinterpret_cast<const se_f16x8_t *>(src);
        for (int i = lane; i < (hw >> 3); i += 64) {
            const se_f16x8_t p = v[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += static_cast<float>(p[e]);
        }
    } else {
        for (int i = lane; i < hw; i += 64) sum += static_cast<float>(src[i]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if (lane == 0) mean[plane] = se_round_half(sum / static_cast<float>(hw));
}

// The same with a workgroup per plane, for planes long enough to keep 256 lanes busy (a wave per plane leaves a
// [8,40,256,256] tensor to 320 waves).
__global__ __launch_bounds__(kSeThreads) void se_mean_block_kernel(const _Float16 *__restrict__ x, float *__restrict__ mean,
                                                                   int hw) {
    __shared__ float part[kSeThreads / 64];
    const int plane = blockIdx.x, tid = threadIdx.x;
    const _Float16 *src = x + static_cast<int64_t>(plane) * hw;
    float sum = 0.0f;
    if ((hw & 7) == 0 && (reinterpret_cast<uintptr_t>(src) & 15u) == 0) {
        const se_f16x8_t *v = reinterpret_cast<const se_f16x8_t *>(src);
        for (int i = tid; i < (hw >> 3); i += kSeThreads) {
            const se_f16x8_t p = v[i];
#pragma unroll
            for (int e = 0; e < 8; ++e) sum += static_cast<float>(p[e]);
        }
    } else {
        for (int i = tid; i < hw; i += kSeThreads) sum += static_cast<float>(src[i]);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    if ((tid & 63) == 0) part[tid >> 6] = sum;
    __syncthreads();
    if (tid == 0) {
        float total = 0.0f;
#pragma unroll
        for (int k = 0; k < kSeThreads / 64; ++k) total += part[k];
        mean[plane] = se_round_half(total / static_cast<float>(hw));
    }
}

// hidden[n, j] = silu(W1[j, :] . mean[n, :] + b1[j]): a wave per (image, hidden value), a lane's loads of the row
// all in flight before the first sum (the sums are short: the kernel is latency, not arithmetic).  Weights rounded
// to fp16 as autocast casts them.
__global__ __launch_bounds__(kSeThreads) void se_hidden_kernel(const float *__restrict__ mean, const float *__restrict__ w1,
                                                               const float *__restrict__ b1, float *__restrict__ hidden,
                                                               int n_images, int c, int r) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int item = blockIdx.x * (kSeThreads / 64) + wave;
    if (item >= n_images * r) return;
    const int n = item / r, j = item - n * r;
    const float *row = w1 + static_cast<int64_t>(j) * c;
    const float *m = mean + static_cast<int64_t>(n) * c;
    float acc[4] = {0.f, 0.f, 0.f, 0.f};
    int i = lane;
    for (; i + 192 < c; i += 256) {
        float wv[4], mv[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            wv[u] = row[i + 64 * u];
            mv[u] = m[i + 64 * u];
        }
#pragma unroll
        for (int u = 0; u < 4; ++u) acc[u] = __builtin_fmaf(se_round_half(wv[u]), mv[u], acc[u]);
    }
    for (; i < c; i += 64) acc[0] = __builtin_fmaf(se_round_half(row[i]), m[i], acc[0]);
    float total = (acc[0] + acc[1]) + (acc[2] + acc[3]);
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) total += __shfl_xor(total, off);
    if (lane == 0) {
        const float pre = se_round_half(total + se_round_half(b1[j]));   // the convolution's fp16 result
        hidden[item] = se_round_half(silu_f32(pre));
    }
}

// gate[n, i] = sigmoid(W2[i, :] . hidden[n, :] + b2[i]): a lane per channel
__global__ __launch_bounds__(kSeThreads) void se_gate_kernel(const float *__restrict__ hidden, const float *__restrict__ w2,
                                                             const float *__restrict__ b2, _Float16 *__restrict__ gate,
                                                             int c, int r, int slabs) {
    __shared__ float s_hidden[kSeMaxChannels / 4];
    const int n = blockIdx.x / slabs, slab = blockIdx.x - n * slabs, tid = threadIdx.x;
    for (int j = tid; j < r; j += kSeThreads) s_hidden[j] = hidden[static_cast<int64_t>(n) * r + j];
    __syncthreads();
    const int i = slab * kSeThreads + tid;
    if (i >= c) return;
    const float *row = w2 + static_cast<int64_t>(i) * r;
    float acc4[4] = {0.f, 0.f, 0.f, 0.f};
    int j = 0;
    for (; j + 8 <= r; j += 8) {
        float wv[8];
#pragma unroll
        for (int u = 0; u < 8; ++u) wv[u] = row[j + u];
#pragma unroll
        for (int u = 0; u < 8; ++u) acc4[u & 3] = __builtin_fmaf(se_round_half(wv[u]), s_hidden[j + u], acc4[u & 3]);
    }
    for (; j < r; ++j) acc4[0] = __builtin_fmaf(se_round_half(row[j]), s_hidden[j], acc4[0]);
    const float pre = se_round_half((acc4[0] + acc4[1]) + (acc4[2] + acc4[3]) + se_round_half(b2[i]));
    const float g = __builtin_amdgcn_rcpf(1.0f + __builtin_amdgcn_exp2f(-1.44269504088896341f * pre));
    gate[static_cast<int64_t>(n) * c + i] = static_cast<_Float16>(g);
}

// y[plane, :] = x[plane, :] * gate[plane]
__global__ __launch_bounds__(kSeThreads) void se_scale_kernel(const _Float16 *__restrict__ x, const _Float16 *__restrict__ gate,
                                                              _Float16 *__restrict__ y, int64_t total, int hw, int vec) {
    const int64_t step = static_cast<int64_t>(gridDim.x) * kSeThreads;
    if (vec) {
        const se_f16x8_t *xv = reinterpret_cast<const se_f16x8_t *>(x);
        se_f16x8_t *yv = reinterpret_cast<se_f16x8_t *>(y);
        for (int64_t i = static_cast<int64_t>(blockIdx.x) * kSeThreads + threadIdx.x; i < (total >> 3); i += step) {
            const float g = static_cast<float>(gate[(i << 3) / hw]);   // hw % 8 == 0: a vector lies inside one plane
            const se_f16x8_t p = xv[i];
            se_f16x8_t o;
#pragma unroll
            for (int e = 0; e < 8; ++e) o[e] = static_cast<_Float16>(static_cast<float>(p[e]) * g);
            yv[i] = o;
        }
    } else {
        for (int64_t i = static_cast<int64_t>(blockIdx.x) * kSeThreads + threadIdx.x; i < total; i += step)
            y[i] = static_cast<_Float16>(static_cast<float>(x[i]) * static_cast<float>(gate[i / hw]));
    }
}

// The image-pooling branch of an ASPP (pruning/models.py:35-39's DeepLabV3+ decoders): plane means -> 1x1 convolution
// -> batch-norm -> ReLU -> the 1 x 1 result spread over the plane (bilinear up-sampling of a single pixel is that pixel).
// A wave per (image, output channel): the dot product of the channel's fp16 weights with the image's means in fp32,
// rounded to fp16 where the convolution stores it, the batch-norm tail of bn_math.h, and the plane written out.
__global__ __launch_bounds__(kSeThreads) void pooled_branch_kernel(const float *__restrict__ mean, const _Float16 *__restrict__ w,
                                                                   _Float16 *__restrict__ y, int items, int cin, int cout,
                                                                   int hw, int64_t y_image_stride, const float *bn_mean,
                                                                   const float *bn_var, const float *bn_weight,
                                                                   const float *bn_bias, float eps, int relu) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int item = blockIdx.x * (kSeThreads / 64) + wave;
    if (item >= items) return;
    const int n = item / cout, co = item - n * cout;
    const float *m = mean + static_cast<int64_t>(n) * cin;
    const _Float16 *wr = w + static_cast<int64_t>(co) * cin;
    float sum = 0.0f;
    if ((cin & 7) == 0 && (reinterpret_cast<uintptr_t>(w) & 15u) == 0) {
        // 16 bytes of the weight row and 32 of the means per lane and step, the steps' loads in flight together
        const se_f16x8_t *w8 = reinterpret_cast<const se_f16x8_t *>(wr);
        const float4 *m4 = reinterpret_cast<const float4 *>(m);
        const int pieces = cin >> 3;
#pragma unroll 4
        for (int k = lane; k < pieces; k += 64) {
            const se_f16x8_t a = w8[k];
            const float4 b0 = m4[2 * k], b1 = m4[2 * k + 1];
            sum = __builtin_fmaf(static_cast<float>(a[0]), b0.x, sum);
            sum = __builtin_fmaf(static_cast<float>(a[1]), b0.y, sum);
            sum = __builtin_fmaf(static_cast<float>(a[2]), b0.z, sum);
            sum = __builtin_fmaf(static_cast<float>(a[3]), b0.w, sum);
            sum = __builtin_fmaf(static_cast<float>(a[4]), b1.x, sum);
            sum = __builtin_fmaf(static_cast<float>(a[5]), b1.y, sum);
            sum = __builtin_fmaf(static_cast<float>(a[6]), b1.z, sum);
            sum = __builtin_fmaf(static_cast<float>(a[7]), b1.w, sum);
        }
    } else {
#pragma unroll 4
        for (int k = lane; k < cin; k += 64) sum = __builtin_fmaf(static_cast<float>(wr[k]), m[k], sum);
    }
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) sum += __shfl_xor(sum, off);
    float v = se_round_half(sum);
    if (bn_mean != nullptr) {
        const BnChannel p = bn_channel_of(bn_mean, bn_var, bn_weight, bn_bias, eps, static_cast<unsigned>(co));
        v = bn_tail<_Float16, false, false>(v, 0.0f, p, p, relu);
    }
    const _Float16 hv = static_cast<_Float16>(v);
    _Float16 *dst = y + n * y_image_stride + static_cast<int64_t>(co) * hw;
    if ((hw & 7) == 0 && (reinterpret_cast<uintptr_t>(dst) & 15u) == 0) {
        const se_f16x8_t v8 = {hv, hv, hv, hv, hv, hv, hv, hv};
        for (int i = lane; i < (hw >> 3); i += 64) reinterpret_cast<se_f16x8_t *>(dst)[i] = v8;
    } else {
        for (int i = lane; i < hw; i += 64) dst[i] = hv;
    }
}

}  // namespace spice

extern "C" size_t spice_pooled_branch_workspace_bytes(int64_t n, int64_t cin) {
    if (n < 0 || cin < 0) return 0;
    return static_cast<size_t>(n * cin) * 4 + 256;   // the means, fp32
}

extern "C" int spice_pooled_branch_f16(const void *x, const void *weight, void *y, int64_t n, int64_t cin, int64_t hw,
                                       int64_t cout, int64_t y_image_stride, const float *mean, const float *var,
                                       const float *bn_weight, const float *bn_bias, float eps, int32_t relu,
                                       void *workspace, size_t workspace_bytes, void *stream) {
    using namespace spice;
    if (n < 0 || cin <= 0 || hw <= 0 || cout <= 0) return SPICE_ERR_ARG;
    if (n == 0) return SPICE_OK;
    if (!x || !weight || !y || !workspace) return SPICE_ERR_ARG;
    if (mean != nullptr && (!var || !bn_weight || !bn_bias)) return SPICE_ERR_ARG;
    if (y_image_stride == 0) y_image_stride = cout * hw;   // dense
    if (y_image_stride < cout * hw) return SPICE_ERR_ARG;
    if (hw > INT32_MAX || n * cin > INT32_MAX || n * cout > INT32_MAX) return SPICE_ERR_UNSUPPORTED;
    if (workspace_bytes < spice_pooled_branch_workspace_bytes(n, cin)) return SPICE_ERR_WORKSPACE;
    float *means = static_cast<float *>(workspace);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int planes = static_cast<int>(n * cin);
    const _Float16 *xh = static_cast<const _Float16 *>(x);
    if (hw >= 4096)
        hipLaunchKernelGGL(se_mean_block_kernel, dim3(planes), dim3(kSeThreads), 0, st, xh, means, static_cast<int>(hw));
    else
        hipLaunchKernelGGL(se_mean_kernel, dim3((planes + 3) / 4), dim3(kSeThreads), 0, st, xh, means, planes, static_cast<int>(hw));
    const int items = static_cast<int>(n * cout);
    hipLaunchKernelGGL(pooled_branch_kernel, dim3((items + 3) / 4), dim3(kSeThreads), 0, st, means,
                       static_cast<const _Float16 *>(weight), static_cast<_Float16 *>(y), items, static_cast<int>(cin),
                       static_cast<int>(cout), static_cast<int>(hw), y_image_stride, mean, var, bn_weight, bn_bias, eps,
                       tail_code(relu));
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}

extern "C" size_t spice_squeeze_excite_workspace_bytes(int64_t n, int64_t c) {
    if (n < 0 || c < 0) return 0;
    // means (fp32), gate (fp16), hidden values (fp32, at most 1,024 per image)
    return static_cast<size_t>(n * c) * 4 + static_cast<size_t>(n * c) * 2 + static_cast<size_t>(n) * 4096 + 512;
}

extern "C" int spice_squeeze_excite_f16(const void *x, void *y, int64_t n, int64_t c, int64_t hw, const float *w1,
                                        const float *b1, int64_t r, const float *w2, const float *b2, void *workspace,
                                        size_t workspace_bytes, void *stream) {
    using namespace spice;
    if (n < 0 || c <= 0 || hw <= 0 || r <= 0) return SPICE_ERR_ARG;
    if (n == 0) return SPICE_OK;
    if (!x || !y || !w1 || !b1 || !w2 || !b2 || !workspace) return SPICE_ERR_ARG;
    if (c > kSeMaxChannels || r > kSeMaxChannels / 4 || hw > INT32_MAX || n * c > INT32_MAX) return SPICE_ERR_UNSUPPORTED;
    if (workspace_bytes < spice_squeeze_excite_workspace_bytes(n, c)) return SPICE_ERR_WORKSPACE;
    if ((reinterpret_cast<uintptr_t>(workspace) & 255u) != 0) return SPICE_ERR_ARG;
    float *mean = static_cast<float *>(workspace);
    char *base = static_cast<char *>(workspace);
    const size_t gate_at = (static_cast<size_t>(n * c) * 4 + 255) & ~size_t{255};
    const size_t hidden_at = (gate_at + static_cast<size_t>(n * c) * 2 + 255) & ~size_t{255};
    _Float16 *gate = reinterpret_cast<_Float16 *>(base + gate_at);
    float *hidden = reinterpret_cast<float *>(base + hidden_at);
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int planes = static_cast<int>(n * c);
    const _Float16 *xh = static_cast<const _Float16 *>(x);
    if (hw >= 4096)
        hipLaunchKernelGGL(se_mean_block_kernel, dim3(planes), dim3(kSeThreads), 0, st, xh, mean, static_cast<int>(hw));
    else
        hipLaunchKernelGGL(se_mean_kernel, dim3((planes + 3) / 4), dim3(kSeThreads), 0, st, xh, mean, planes, static_cast<int>(hw));
    const int slabs = static_cast<int>((c + kSeThreads - 1) / kSeThreads);
    const int items = static_cast<int>(n * r);
    hipLaunchKernelGGL(se_hidden_kernel, dim3((items + 3) / 4), dim3(kSeThreads), 0, st, mean, w1, b1, hidden,
                       static_cast<int>(n), static_cast<int>(c), static_cast<int>(r));
    hipLaunchKernelGGL(se_gate_kernel, dim3(static_cast<unsigned>(n * slabs)), dim3(kSeThreads), 0, st, hidden, w2, b2, gate,
                       static_cast<int>(c), static_cast<int>(r), slabs);
    const int64_t total = n * c * hw;
    const int vec = ((hw & 7) == 0 && aligned16(x) && aligned16(y)) ? 1 : 0;
    const int64_t work = vec ? (total >> 3) : total;
    int64_t blocks = (work + kSeThreads - 1) / kSeThreads;
    if (blocks > 16384) blocks = 16384;
    hipLaunchKernelGGL(se_scale_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kSeThreads), 0, st, xh, gate,
                       static_cast<_Float16 *>(y), total, static_cast<int>(hw), vec);
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}
