// Bilinear up-sampling of NCHW planes: the x4 / x2 steps of the segmentation decoders and heads
// (the models of pruning/models.py:35-39), which torch runs in fp32 under autocast.
//
// HBM-bound on the WRITE: the output is scale^2 times the input (176 MB of fp32 logits from 5.5 MB of
// fp16 head output at [8,21,512,512]).  One workgroup produces a band of whole output rows of one
// plane: it interpolates the few source rows the band touches horizontally into LDS, then every
// lane combines two of those rows for four consecutive outputs and writes them with one
// streaming store.
//
// Arithmetic follows torch's upsample_bilinear2d device kernel operation for operation (source
// index = scale * dst or scale * (dst + 0.5) - 0.5 clamped at 0, lambdas from the truncated index,
// the two-level weighted sum with the same fused multiply-adds), all in fp32.
#include "common.h"
#include "bilinear_math.h"

#include <hip/hip_fp16.h>
#include <stdlib.h>

namespace spice {

constexpr int kUpThreads = 256;
constexpr int kUpTargetOutputs = 4096;   // outputs per workgroup (whole rows)
constexpr int kUpMaxLdsBytes = 64 * 1024;

typedef unsigned u32x2_t __attribute__((ext_vector_type(2)));

struct UpArgs {
    const void *src;
    void *dst;
    int in_h, in_w, out_h, out_w;
    int rows_per_block;   // output rows per workgroup
    int blocks_per_plane;
    int lds_rows;         // source rows a band can touch (upper bound)
    int align_corners;
    float scale_h, scale_w;
    float inv_per_row;    // 1 / (lane work items per output row)
    int planes_per_group;        // destination: planes of a group lie one behind the other,
    int64_t dst_group_stride;    // groups this many elements apart (a channel slice of a larger tensor)
};

template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) { return v; }
template <>
__device__ __forceinline__ float to_f32<__half>(__half v) { return __half2float(v); }

// The weighted sum is separable exactly as torch writes it:
//   out = h0 * (w0 * a + w1 * b) + h1 * (w0 * c + w1 * d)
// and the two inner sums depend on the SOURCE row only.  Phase 1 computes them once per source row
// of the band and output column (fp32, into LDS); phase 2 combines two of those rows per output.
// At x4 every horizontally interpolated row serves about four output rows.
template <typename TIn, typename TOut, int VEC>
__global__ __launch_bounds__(kUpThreads) void upsample_bilinear_kernel(UpArgs a) {
    extern __shared__ __align__(16) float hrows[];   // [source rows of the band][out_w]
    const int plane = blockIdx.x / a.blocks_per_plane;
    const int band = blockIdx.x - plane * a.blocks_per_plane;
    const int oy0 = band * a.rows_per_block;
    const int n_rows = min(a.rows_per_block, a.out_h - oy0);

    // source rows of the band: from the upper row of its first output row to the lower row of its last
    const int y_lo = static_cast<int>(src_index(a.scale_h, oy0, a.align_corners));
    const int y_last = static_cast<int>(src_index(a.scale_h, oy0 + n_rows - 1, a.align_corners));
    // the host sized the LDS tile for lds_rows rows, an upper bound of what a band touches; the clamp keeps a
    // violated bound from becoming an out-of-range LDS access
    const int y_hi = min(min(y_last + 1, a.in_h - 1), y_lo + a.lds_rows - 1);
    const int n_src_rows = y_hi - y_lo + 1;
    const TIn *sp = static_cast<const TIn *>(a.src) + (static_cast<int64_t>(plane) * a.in_h + y_lo) * a.in_w;
    const int per_row = a.out_w / VEC;

    for (int i = threadIdx.x; i < n_src_rows * per_row; i += kUpThreads) {
        const int r = static_cast<int>((static_cast<float>(i) + 0.5f) * a.inv_per_row);
        const int ox = (i - r * per_row) * VEC;
        const TIn *row = sp + r * a.in_w;
        float v[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float sx = src_index(a.scale_w, ox + e, a.align_corners);
            const int x0 = min(static_cast<int>(sx), a.in_w - 1);   // never past the row, whatever the rounding of sx
            const int x1 = min(x0 + 1, a.in_w - 1);
            const float w1 = sx - static_cast<float>(x0);
            const float w0 = 1.0f - w1;
            v[e] = __builtin_fmaf(w0, to_f32<TIn>(row[x0]), w1 * to_f32<TIn>(row[x1]));
        }
        float *h = hrows + r * a.out_w + ox;
        if constexpr (VEC == 4) {
            *reinterpret_cast<float4 *>(h) = make_float4(v[0], v[1], v[2], v[3]);
        } else {
            h[0] = v[0];
        }
    }
    __syncthreads();

    const int group = plane / a.planes_per_group;
    TOut *dp = static_cast<TOut *>(a.dst) + group * a.dst_group_stride +
               (static_cast<int64_t>(plane - group * a.planes_per_group) * a.out_h + oy0) * a.out_w;
    for (int q = threadIdx.x; q < n_rows * per_row; q += kUpThreads) {
        const int ry = static_cast<int>((static_cast<float>(q) + 0.5f) * a.inv_per_row);
        const int ox = (q - ry * per_row) * VEC;
        const float sy = src_index(a.scale_h, oy0 + ry, a.align_corners);
        const int y0 = min(static_cast<int>(sy), y_hi);           // inside the rows staged above
        const int y1 = min(y0 + 1, y_hi);
        const float h1 = sy - static_cast<float>(y0);
        const float h0 = 1.0f - h1;
        const float *r0 = hrows + (y0 - y_lo) * a.out_w + ox;
        const float *r1 = hrows + (y1 - y_lo) * a.out_w + ox;
        float v[VEC];
        if constexpr (VEC == 4) {
            const float4 t = *reinterpret_cast<const float4 *>(r0);
            const float4 b = *reinterpret_cast<const float4 *>(r1);
            v[0] = __builtin_fmaf(h0, t.x, h1 * b.x);
            v[1] = __builtin_fmaf(h0, t.y, h1 * b.y);
            v[2] = __builtin_fmaf(h0, t.z, h1 * b.z);
            v[3] = __builtin_fmaf(h0, t.w, h1 * b.w);
        } else {
            v[0] = __builtin_fmaf(h0, r0[0], h1 * r1[0]);
        }
        TOut *o = dp + static_cast<int64_t>(ry) * a.out_w + ox;
        if constexpr (sizeof(TOut) == 2) {
            // keep the fp32 result a value of its own: otherwise the last multiply-add and the
            // conversion become one mixed-precision instruction that rounds once, to fp16, and
            // differs from "fp32 result, then cast" wherever the fp32 result is a tie
#pragma unroll
            for (int e = 0; e < VEC; ++e) asm volatile("" : "+v"(v[e]));
        }
        if constexpr (VEC == 4 && sizeof(TOut) == 4) {
            st_stream16(reinterpret_cast<float4 *>(o), make_float4(v[0], v[1], v[2], v[3]));
        } else if constexpr (VEC == 4) {
            typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
            f16x4_t pack;
            pack.x = static_cast<_Float16>(v[0]);
            pack.y = static_cast<_Float16>(v[1]);
            pack.z = static_cast<_Float16>(v[2]);
            pack.w = static_cast<_Float16>(v[3]);
            __builtin_nontemporal_store(__builtin_bit_cast(u32x2_t, pack), reinterpret_cast<u32x2_t *>(o));
        } else if constexpr (sizeof(TOut) == 4) {
            o[0] = v[0];
        } else {
            *reinterpret_cast<_Float16 *>(o) = static_cast<_Float16>(v[0]);
        }
    }
}

template <typename TIn, typename TOut>
static int launch_up(const UpArgs &a, int64_t planes, bool vec, size_t lds, hipStream_t st) {
    const dim3 grid(static_cast<unsigned>(planes * a.blocks_per_plane));
    if (vec)
        hipLaunchKernelGGL((upsample_bilinear_kernel<TIn, TOut, 4>), grid, dim3(kUpThreads), lds, st, a);
    else
        hipLaunchKernelGGL((upsample_bilinear_kernel<TIn, TOut, 1>), grid, dim3(kUpThreads), lds, st, a);
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}

}  // namespace spice

SPICE_ENV_KNOB(knob_up_target_outputs, "SPICE_UP_TARGET_OUTPUTS", 0)  // tuning knob, read once

extern "C" int spice_upsample_bilinear(const void *src, int32_t src_dtype, void *dst, int32_t dst_dtype,
                                       int64_t planes, int32_t in_h, int32_t in_w, int32_t out_h,
                                       int32_t out_w, int32_t align_corners, float scale_h, float scale_w,
                                       void *stream) {
    return spice_upsample_bilinear_into(src, src_dtype, dst, dst_dtype, planes, in_h, in_w, out_h, out_w, align_corners,
                                        scale_h, scale_w, planes > 0 ? planes : 1, 0, stream);
}

extern "C" int spice_upsample_bilinear_into(const void *src, int32_t src_dtype, void *dst, int32_t dst_dtype,
                                            int64_t planes, int32_t in_h, int32_t in_w, int32_t out_h,
                                            int32_t out_w, int32_t align_corners, float scale_h, float scale_w,
                                            int64_t planes_per_group, int64_t dst_group_stride, void *stream) {
    using namespace spice;
    if (planes < 0 || in_h <= 0 || in_w <= 0 || out_h <= 0 || out_w <= 0) return SPICE_ERR_ARG;
    if (planes == 0) return SPICE_OK;
    if (src == nullptr || dst == nullptr) return SPICE_ERR_ARG;
    if (planes_per_group < 1 || planes_per_group > INT32_MAX || planes % planes_per_group != 0) return SPICE_ERR_ARG;
    // groups must not overlap (one group needs no stride)
    if (planes > planes_per_group && dst_group_stride < planes_per_group * static_cast<int64_t>(out_h) * out_w)
        return SPICE_ERR_ARG;
    if ((src_dtype != 0 && src_dtype != 1) || (dst_dtype != 0 && dst_dtype != 1)) return SPICE_ERR_UNSUPPORTED;
    if (out_h < in_h || out_w < in_w) return SPICE_ERR_UNSUPPORTED;   // up-sampling only
    UpArgs a;
    a.src = src;
    a.dst = dst;
    a.in_h = in_h;
    a.in_w = in_w;
    a.out_h = out_h;
    a.out_w = out_w;
    a.align_corners = align_corners ? 1 : 0;
    a.planes_per_group = static_cast<int>(planes_per_group);
    a.dst_group_stride = dst_group_stride;
    // torch's area_pixel_compute_scale: (in-1)/(out-1) with aligned corners (0 for one output),
    // otherwise 1/scale_factor when the caller passed one (> 0 here) or in/out
    if (a.align_corners) {
        a.scale_h = out_h > 1 ? static_cast<float>(in_h - 1) / static_cast<float>(out_h - 1) : 0.0f;
        a.scale_w = out_w > 1 ? static_cast<float>(in_w - 1) / static_cast<float>(out_w - 1) : 0.0f;
    } else {
        a.scale_h = scale_h > 0.0f ? 1.0f / scale_h : static_cast<float>(in_h) / static_cast<float>(out_h);
        a.scale_w = scale_w > 0.0f ? 1.0f / scale_w : static_cast<float>(in_w) / static_cast<float>(out_w);
    }
    if (a.scale_h > 1.0f || a.scale_w > 1.0f) return SPICE_ERR_UNSUPPORTED;
    const int target = knob_up_target_outputs() > 0 ? (int)knob_up_target_outputs() : kUpTargetOutputs;
    a.rows_per_block = static_cast<int>(target / out_w);
    if (a.rows_per_block < 1) a.rows_per_block = 1;
    if (a.rows_per_block > out_h) a.rows_per_block = out_h;
    a.blocks_per_plane = static_cast<int>(ceil_div(out_h, a.rows_per_block));
    if (planes * a.blocks_per_plane > INT32_MAX) return SPICE_ERR_UNSUPPORTED;
    // a band of R output rows spans at most (R-1)*scale source rows between its first and last upper
    // rows (+1 for where the truncations fall), plus the lower neighbour of the last, plus one row
    // of slack for fp32 rounding of the product
    a.lds_rows = static_cast<int>(static_cast<float>(a.rows_per_block - 1) * a.scale_h) + 4;
    if (a.lds_rows > in_h) a.lds_rows = in_h;
    const size_t out_elem = dst_dtype == 0 ? 4 : 2;
    const bool vec = out_w % 4 == 0 && (reinterpret_cast<uintptr_t>(dst) % (4 * out_elem)) == 0 && dst_group_stride % 4 == 0;
    a.inv_per_row = 1.0f / static_cast<float>(vec ? out_w / 4 : out_w);
    // the float reciprocal splits a work-item number into (row, column) exactly while the numbers
    // stay far below 2^22
    if (static_cast<int64_t>(a.lds_rows > a.rows_per_block ? a.lds_rows : a.rows_per_block) * out_w > (1 << 20))
        return SPICE_ERR_UNSUPPORTED;
    const size_t lds = static_cast<size_t>(a.lds_rows) * out_w * sizeof(float);
    if (lds > static_cast<size_t>(kUpMaxLdsBytes)) return SPICE_ERR_UNSUPPORTED;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (src_dtype == 0 && dst_dtype == 0) return launch_up<float, float>(a, planes, vec, lds, st);
    if (src_dtype == 1 && dst_dtype == 0) return launch_up<__half, float>(a, planes, vec, lds, st);
    if (src_dtype == 1 && dst_dtype == 1) return launch_up<__half, __half>(a, planes, vec, lds, st);
    return launch_up<float, __half>(a, planes, vec, lds, st);
}
