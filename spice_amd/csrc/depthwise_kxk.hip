// Depthwise k x k convolution (k = 3 or 5, stride 1 or 2, dilation 1 or 2, padding (k / 2) * dilation, no bias), fp16
// NCHW: the depthwise convolutions of the EfficientNet encoders (the models of pruning/models.py:35-39 built on
// "timm-efficientnet-b3").  MIOpen serves most of these calls image by image with an unfold pass and a matrix
// product per image (144 pairs of launches per forward of the B3 encoder at [8,3,512,512], 15 of its 21 ms); torch's
// own depthwise kernel takes the rest at a tenth of the memory system's rate.  Same arithmetic as the 3x3 kernels of
// depthwise.hip: the k * k products (exact in fp32) added to an fp32 sum in tap order, rows first, with fused
// multiply-adds, one rounding to fp16 - what torch's own depthwise kernel computes.
//
// A workgroup takes a tile of 1,024 outputs of one plane (16 x 64, 32 x 32 or 64 x 16 by the plane's width), stages
// the input window under it in LDS (32-bit loads from the even column at or left of the window; what lies outside the
// plane is zero), and every lane sums four neighbouring outputs of one row from it.
#include "common.h"
#include "bn_math.h"

#include <hip/hip_fp16.h>

namespace spice {

constexpr int kDwkThreads = 256;
constexpr int kDwkLdsWords = 3328;   // 13 KB: the largest window (stride 2, 5 x 5, dilation 2 under 16 x 64 outputs) and its pitch

struct DwkArgs {
    const _Float16 *x;
    const void *w;     // [C, 1, k, k], fp16 or fp32 (rounded to fp16 here, as autocast's cast does)
    _Float16 *y;
    int c, h, wd, oh, ow;
    int txc_log2;      // lanes along a tile row: 16, 8 or 4 (tile 64, 32 or 16 outputs wide)
    int tiles_x, tiles_y;
    int w_is_f32;
    // the batch-norm (+ activation) behind the convolution, applied to the rounded result before it is stored; null: none
    const float *mean, *var, *weight, *bias;
    float eps;
    int act;   // tail code of bn_math.h: 0 none, 1 ReLU, 2 SiLU
};

template <int K, int S, int D>
__global__ __launch_bounds__(kDwkThreads) void depthwise_kxk_kernel(DwkArgs a) {
    __shared__ unsigned lds[kDwkLdsWords];
    constexpr int PAD = (K / 2) * D;
    constexpr int SPAN = 3 * S + (K - 1) * D + 1;   // input columns under a lane's four outputs
    const int tid = threadIdx.x;
    int t = blockIdx.x;
    const int tile_x = t % a.tiles_x;
    t /= a.tiles_x;
    const int tile_y = t % a.tiles_y;
    const int plane = t / a.tiles_y;
    const int ch = plane % a.c;
    const int TXC = 1 << a.txc_log2, TW = 4 * TXC, TH = kDwkThreads >> a.txc_log2;
    const int oy0 = tile_y * TH, ox0 = tile_x * TW;
    const int th = min(TH, a.oh - oy0);                 // output rows of this tile that exist
    const int iy0 = oy0 * S - PAD, ix0 = ox0 * S - PAD;
    const int ix0e = ix0 & ~1;                          // the even column at or left of the window (also for -1)
    const int shift = ix0 - ix0e;
    const int ih = (th - 1) * S + (K - 1) * D + 1;
    const int iw_words = ((TW - 1) * S + (K - 1) * D + 1 + shift + 1) >> 1;
    const int pitch = iw_words | 1;                      // words; odd, so that rows start on different banks

    // the taps first: in flight together with the window's loads
    unsigned wraw[K * K];
    if (a.w_is_f32) {
        const unsigned *wp = static_cast<const unsigned *>(a.w) + ch * (K * K);
#pragma unroll
        for (int k = 0; k < K * K; ++k) wraw[k] = wp[k];
    } else {
        const unsigned short *wp = static_cast<const unsigned short *>(a.w) + ch * (K * K);
#pragma unroll
        for (int k = 0; k < K * K; ++k) wraw[k] = wp[k];
    }

    const _Float16 *src = a.x + static_cast<int64_t>(plane) * a.h * a.wd;
    const float inv_words = 1.0f / static_cast<float>(iw_words);
    for (int i = tid; i < ih * iw_words; i += kDwkThreads) {
        const int r = static_cast<int>((static_cast<float>(i) + 0.5f) * inv_words);   // exact: i stays far below 2^22
        const int cw = i - r * iw_words;
        const int gy = iy0 + r, gx = ix0e + 2 * cw;
        unsigned word = 0;
        // gx and the plane's width are even: a word lies inside the row or outside it, never across its end
        if (gy >= 0 && gy < a.h && gx >= 0 && gx < a.wd)
            word = *reinterpret_cast<const unsigned *>(src + static_cast<int64_t>(gy) * a.wd + gx);
        lds[r * pitch + cw] = word;
    }
    __syncthreads();

    const int ty = tid >> a.txc_log2, tx = tid & (TXC - 1);
    const int oy = oy0 + ty, ox = ox0 + 4 * tx;
    if (ty >= th || ox >= a.ow) return;
    float wt[K * K];
#pragma unroll
    for (int k = 0; k < K * K; ++k) {
        const _Float16 from_f32 = static_cast<_Float16>(__builtin_bit_cast(float, wraw[k]));
        const _Float16 from_f16 = __builtin_bit_cast(_Float16, static_cast<unsigned short>(wraw[k]));
        wt[k] = static_cast<float>(a.w_is_f32 ? from_f32 : from_f16);
    }
    const _Float16 *tile = reinterpret_cast<const _Float16 *>(lds);
    float acc[4] = {0.0f, 0.0f, 0.0f, 0.0f};
#pragma unroll
    for (int ky = 0; ky < K; ++ky) {   // tap order: rows first, left to right (a tap outside the plane adds w * 0)
        const _Float16 *row = tile + (ty * S + ky * D) * (2 * pitch) + shift + 4 * tx * S;
        float v[SPAN];
#pragma unroll
        for (int j = 0; j < SPAN; ++j) v[j] = static_cast<float>(row[j]);
#pragma unroll
        for (int kx = 0; kx < K; ++kx)
#pragma unroll
            for (int e = 0; e < 4; ++e) acc[e] = __builtin_fmaf(wt[ky * K + kx], v[e * S + kx * D], acc[e]);
    }
    _Float16 out[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float o = acc[e];
        asm volatile("" : "+v"(o));   // fp32 sum first, conversion as a step of its own
        out[e] = static_cast<_Float16>(o);
    }
    if (a.mean != nullptr) {   // what a batch-norm launch behind this one would compute from the stored halves
        const BnChannel p = bn_channel_of(a.mean, a.var, a.weight, a.bias, a.eps, static_cast<unsigned>(ch));
#pragma unroll
        for (int e = 0; e < 4; ++e)
            out[e] = static_cast<_Float16>(bn_tail<_Float16, false, false>(static_cast<float>(out[e]), 0.0f, p, p, a.act));
    }
    _Float16 *dst = a.y + (static_cast<int64_t>(plane) * a.oh + oy) * a.ow + ox;
    if ((a.ow & 3) == 0) {   // four outputs exist and lie on an 8-byte boundary
        typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
        *reinterpret_cast<f16x4_t *>(dst) = f16x4_t{out[0], out[1], out[2], out[3]};
    } else {
#pragma unroll
        for (int e = 0; e < 4; ++e)
            if (ox + e < a.ow) dst[e] = out[e];
    }
}

template <int K, int S>
static void dwk_launch_d(int dilation, unsigned grid, hipStream_t st, const DwkArgs &a) {
    if (dilation == 1)
        hipLaunchKernelGGL((depthwise_kxk_kernel<K, S, 1>), dim3(grid), dim3(kDwkThreads), 0, st, a);
    else
        hipLaunchKernelGGL((depthwise_kxk_kernel<K, S, 2>), dim3(grid), dim3(kDwkThreads), 0, st, a);
}

}  // namespace spice

extern "C" int spice_depthwise_conv_f16(const void *x, const void *weight, int32_t weight_dtype, void *y, int64_t n,
                                        int64_t c, int32_t h, int32_t w, int32_t kernel, int32_t stride,
                                        int32_t dilation, const float *mean, const float *var, const float *bn_weight,
                                        const float *bn_bias, float eps, int32_t act, void *stream) {
    using namespace spice;
    if (n < 0 || c <= 0 || h <= 0 || w <= 0) return SPICE_ERR_ARG;
    if ((kernel != 3 && kernel != 5) || (stride != 1 && stride != 2) || (dilation != 1 && dilation != 2))
        return SPICE_ERR_UNSUPPORTED;
    if (n == 0) return SPICE_OK;
    if (x == nullptr || weight == nullptr || y == nullptr) return SPICE_ERR_ARG;
    if (weight_dtype != 0 && weight_dtype != 1) return SPICE_ERR_UNSUPPORTED;
    if (mean != nullptr && (var == nullptr || bn_weight == nullptr || bn_bias == nullptr)) return SPICE_ERR_ARG;
    // 32-bit loads of column pairs, 8-byte stores of four outputs: an even width, planes that start on those boundaries
    if ((w & 1) != 0 || !aligned16(x) || !aligned16(y)) return SPICE_ERR_UNSUPPORTED;
    const int oh = (h - 1) / stride + 1, ow = (w - 1) / stride + 1;
    DwkArgs a;
    a.x = static_cast<const _Float16 *>(x);
    a.w = weight;
    a.y = static_cast<_Float16 *>(y);
    a.c = static_cast<int>(c);
    a.h = h;
    a.wd = w;
    a.oh = oh;
    a.ow = ow;
    a.txc_log2 = ow > 32 ? 4 : (ow > 16 ? 3 : 2);
    const int tw = 4 << a.txc_log2, th = kDwkThreads >> a.txc_log2;
    a.tiles_x = static_cast<int>(ceil_div(ow, tw));
    a.tiles_y = static_cast<int>(ceil_div(oh, th));
    a.w_is_f32 = weight_dtype == 0;
    a.mean = mean;
    a.var = var;
    a.weight = bn_weight;
    a.bias = bn_bias;
    a.eps = eps;
    a.act = tail_code(act);
    {   // the largest window of a tile must fit the LDS array
        const int pad_span = (kernel - 1) * dilation + 1;
        const int ih = (th - 1) * stride + pad_span;
        const int iw_words = ((tw - 1) * stride + pad_span + 1 + 1) >> 1;
        if (static_cast<int64_t>(ih) * (iw_words | 1) > kDwkLdsWords) return SPICE_ERR_UNSUPPORTED;
    }
    const int64_t blocks = n * c * a.tiles_x * a.tiles_y;
    if (blocks > INT32_MAX || n * c > INT32_MAX) return SPICE_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const unsigned grid = static_cast<unsigned>(blocks);
    if (kernel == 3 && stride == 1) dwk_launch_d<3, 1>(dilation, grid, st, a);
    else if (kernel == 3) dwk_launch_d<3, 2>(dilation, grid, st, a);
    else if (stride == 1) dwk_launch_d<5, 1>(dilation, grid, st, a);
    else dwk_launch_d<5, 2>(dilation, grid, st, a);
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}
