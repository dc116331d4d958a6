// The stem of a ResNet encoder after its 7x7 convolution, in one pass over the convolution's result:
//   full   = relu?( bn(x) )                                   (the encoder's first feature map; optional)
//   pooled = max_pool2d(full, kernel 3, stride 2, padding 1)   (what the first stage consumes)
// replacing the batch-norm launch and torch's max-pooling launch of pruning/models.py:35-39's ResNet encoders:
// x is read once, `full` is written once if a decoder wants it, `pooled` once; the two launches read x, write
// `full`, and read it again.  fp16 NCHW.  The batch-norm is bn_math.h's arithmetic (what spice_bn_act stores),
// the maximum is exact and keeps NaN, as torch's kernel does, so both outputs equal the two launches' bit for bit.
//
// A lane produces 8 consecutive outputs of one pooled row from input rows 2y-1 .. 2y+1, columns 2x-1 .. 2x+15:
// three pairs of 16-byte loads and three single halfs.  It owns (and stores into `full`) rows 2y and 2y+1 of its
// 16 columns; row 2y-1 and column 2x-1 are its neighbours' and are only normalised again for the maximum.
#include "bn_math.h"

namespace spice {

constexpr int kPoolThreads = 256;

typedef _Float16 h8_t __attribute__((ext_vector_type(8)));
typedef _Float16 h2_t __attribute__((ext_vector_type(2)));
typedef float f2_t __attribute__((ext_vector_type(2)));

struct PoolArgs {
    const _Float16 *x;
    _Float16 *full;     // or null
    _Float16 *pooled;
    const float *mean, *var, *weight, *bias;
    float eps;
    int c, h, w, ho, wo;
    int lanes_per_row;      // wo / 8
    int rows_per_block;
    int blocks_per_plane;
    int relu;
};

// relu?( bn(x) ) for two neighbouring values, as bn_tail<_Float16, false, false> gives them
__device__ __forceinline__ h2_t pool_bn_pair(h2_t x, const BnChannel &p, int relu) {
    const f2_t xf = {static_cast<float>(x[0]), static_cast<float>(x[1])};
    const f2_t t = (xf - f2_t{p.mean, p.mean}) * f2_t{p.inv, p.inv};
    f2_t o = __builtin_elementwise_fma(f2_t{p.weight, p.weight}, t, f2_t{p.bias, p.bias});
    asm volatile("" : "+v"(o));   // the rounding to fp16 stays a step of its own
    h2_t v = {static_cast<_Float16>(o[0]), static_cast<_Float16>(o[1])};
    if (relu) v = __builtin_elementwise_maximum(v, h2_t{0, 0});   // NaN-propagating
    return v;
}

__device__ __forceinline__ h8_t pool_bn8(h8_t x, const BnChannel &p, int relu) {
    h8_t o;
#pragma unroll
    for (int e = 0; e < 8; e += 2) {
        const h2_t v = pool_bn_pair(h2_t{x[e], x[e + 1]}, p, relu);
        o[e] = v[0];
        o[e + 1] = v[1];
    }
    return o;
}

__global__ __launch_bounds__(kPoolThreads) void bn_act_maxpool_kernel(PoolArgs a) {
    const int plane = blockIdx.x / a.blocks_per_plane;
    const int band = blockIdx.x - plane * a.blocks_per_plane;
    const int ly = threadIdx.x / a.lanes_per_row;
    const int lx = threadIdx.x - ly * a.lanes_per_row;
    const int yo = band * a.rows_per_block + ly;
    if (ly >= a.rows_per_block || yo >= a.ho) return;   // no barrier in this kernel
    const BnChannel p = bn_channel_of(a.mean, a.var, a.weight, a.bias, a.eps, plane % a.c);
    const int x0 = lx * 16;
    const _Float16 *src = a.x + static_cast<size_t>(plane) * a.h * a.w;
    const _Float16 ninf = static_cast<_Float16>(-__builtin_inff());

    h8_t lo[3], hi[3];
    _Float16 left[3];
    bool row_in[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {   // all loads first
        const int yi = 2 * yo - 1 + r;
        row_in[r] = yi >= 0 && yi < a.h;
        const _Float16 *row = src + static_cast<size_t>(min(max(yi, 0), a.h - 1)) * a.w;
        lo[r] = *reinterpret_cast<const h8_t *>(row + x0);
        hi[r] = *reinterpret_cast<const h8_t *>(row + x0 + 8);
        left[r] = row[max(x0 - 1, 0)];
    }
    h8_t best;
#pragma unroll
    for (int e = 0; e < 8; ++e) best[e] = ninf;
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        if (!row_in[r]) continue;   // padding: -inf, never the maximum of a window that holds one real value
        const h8_t a0 = pool_bn8(lo[r], p, a.relu), a1 = pool_bn8(hi[r], p, a.relu);
        if (r >= 1 && a.full != nullptr) {
            _Float16 *dst = a.full + (static_cast<size_t>(plane) * a.h + (2 * yo - 1 + r)) * a.w + x0;
            *reinterpret_cast<h8_t *>(dst) = a0;
            *reinterpret_cast<h8_t *>(dst + 8) = a1;
        }
        const h2_t lv = pool_bn_pair(h2_t{left[r], left[r]}, p, a.relu);
        const _Float16 before = x0 > 0 ? lv[0] : ninf;
        // window of output j: columns 2j-1, 2j, 2j+1 of this lane's 16 (column -1 = `before`)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const _Float16 c0 = j == 0 ? before : (2 * j - 1 < 8 ? a0[2 * j - 1] : a1[2 * j - 9]);
            const _Float16 c1 = 2 * j < 8 ? a0[2 * j] : a1[2 * j - 8];
            const _Float16 c2 = 2 * j + 1 < 8 ? a0[2 * j + 1] : a1[2 * j - 7];
            _Float16 m = __builtin_elementwise_maximum(c0, c1);
            m = __builtin_elementwise_maximum(m, c2);
            best[j] = __builtin_elementwise_maximum(best[j], m);
        }
    }
    *reinterpret_cast<h8_t *>(a.pooled + (static_cast<size_t>(plane) * a.ho + yo) * a.wo + lx * 8) = best;
}

}  // namespace spice

extern "C" int spice_bn_act_maxpool_f16(const void *x, void *full, void *pooled, const float *mean, const float *var,
                                        const float *weight, const float *bias, float eps, int64_t n, int64_t c, int32_t h,
                                        int32_t w, int32_t relu, void *stream) {
    using namespace spice;
    if (n < 0 || c <= 0 || h <= 0 || w <= 0) return SPICE_ERR_ARG;
    if (n == 0) return SPICE_OK;
    if (x == nullptr || pooled == nullptr || mean == nullptr || var == nullptr || weight == nullptr || bias == nullptr)
        return SPICE_ERR_ARG;
    // 16 input columns per lane, whole windows: even height, width a multiple of 16
    if ((h & 1) || (w & 15) || w / 16 > kPoolThreads) return SPICE_ERR_UNSUPPORTED;
    if (full == x || pooled == x) return SPICE_ERR_ARG;   // a lane reads rows its neighbours own: never in place
    if (!aligned16(x) || !aligned16(pooled) || (full != nullptr && !aligned16(full))) return SPICE_ERR_UNSUPPORTED;
    if (n * c > INT32_MAX / 4096) return SPICE_ERR_UNSUPPORTED;
    PoolArgs a;
    a.x = static_cast<const _Float16 *>(x);
    a.full = static_cast<_Float16 *>(full);
    a.pooled = static_cast<_Float16 *>(pooled);
    a.mean = mean;
    a.var = var;
    a.weight = weight;
    a.bias = bias;
    a.eps = eps;
    a.c = static_cast<int>(c);
    a.h = h;
    a.w = w;
    a.ho = h / 2;
    a.wo = w / 2;
    a.lanes_per_row = a.wo / 8;
    a.rows_per_block = kPoolThreads / a.lanes_per_row;
    if (a.rows_per_block > a.ho) a.rows_per_block = a.ho;
    a.blocks_per_plane = static_cast<int>(ceil_div(a.ho, a.rows_per_block));
    a.relu = relu ? 1 : 0;
    const int64_t blocks = n * c * a.blocks_per_plane;
    if (blocks > INT32_MAX) return SPICE_ERR_UNSUPPORTED;
    hipLaunchKernelGGL(bn_act_maxpool_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kPoolThreads), 0,
                       static_cast<hipStream_t>(stream), a);
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}
