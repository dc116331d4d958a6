// Depthwise 3x3 convolution, stride 1, padding 1, fp16 NCHW: the first half of the separable
// convolutions of the DeepLabV3+ decoder (the models of pruning/models.py:35-39).  MIOpen runs these
// shapes on its reference kernel (one thread per output, fp64 accumulation), ten times slower than
// the memory traffic asks for.  This kernel streams the planes and sums as torch's OWN depthwise
// kernel does (the one torch runs for these calls with MIOpen switched off, and for the dilated
// calls below in any case): the nine products, exact in fp32, added to an fp32 sum in tap order with
// fused multiply-adds, one rounding to fp16 at the end.  Within one fp16 step of MIOpen's exact sum.
// (A first version summed in fp64 to equal MIOpen's kernel bit for bit; the 72 fp64 multiply-adds
// and conversions per lane made it bound by arithmetic at 2.7 times the time the traffic needs.)
//
// HBM-bound: reads the input once (neighbouring rows and the two edge columns of a lane's window
// come from L1/L2) and writes the output once.  A lane produces 8 consecutive outputs of one row:
// three 16-byte row loads plus the two neighbours left and right of each, 72 multiply-adds,
// one 16-byte store.  Planes large enough take the four-rows-per-lane form further down (six row
// loads for 32 outputs, edge columns from the neighbouring lanes' registers).
#include "common.h"

#include <hip/hip_fp16.h>

namespace spice {

constexpr int kDwThreads = 256;

typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

struct DwArgs {
    const _Float16 *x;
    const void *w;     // [C, 1, 3, 3], fp16 or fp32 (rounded to fp16 here, as autocast's cast does)
    _Float16 *y;
    int c, h, wd;
    int lanes_per_row;   // wd / 8
    int rows_per_block;
    int blocks_per_plane;
    int w_is_f32;
};

// The nine taps of channel `ch` as raw bits (fp32 words or fp16 halfs): loads only, so that they are in flight together
// with the loads of x; dw_taps_to_float turns them into the fp16 values (rounded from fp32 where the weights are fp32,
// as autocast casts them) in fp32 registers.
__device__ __forceinline__ void dw_load_taps(const DwArgs &a, int ch, unsigned (&raw)[9]) {
    if (a.w_is_f32) {
        const unsigned *wp = static_cast<const unsigned *>(a.w) + ch * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) raw[k] = wp[k];
    } else {
        const unsigned short *wp = static_cast<const unsigned short *>(a.w) + ch * 9;
#pragma unroll
        for (int k = 0; k < 9; ++k) raw[k] = wp[k];
    }
}

__device__ __forceinline__ void dw_taps_to_float(const DwArgs &a, const unsigned (&raw)[9], float (&wt)[9]) {
#pragma unroll
    for (int k = 0; k < 9; ++k) {
        const _Float16 from_f32 = static_cast<_Float16>(__builtin_bit_cast(float, raw[k]));
        const _Float16 from_f16 = __builtin_bit_cast(_Float16, static_cast<unsigned short>(raw[k]));
        wt[k] = static_cast<float>(a.w_is_f32 ? from_f32 : from_f16);
    }
}

__global__ __launch_bounds__(kDwThreads) void depthwise3x3_kernel(DwArgs a) {
    const int plane = blockIdx.x / a.blocks_per_plane;
    const int band = blockIdx.x - plane * a.blocks_per_plane;
    const int ch = plane % a.c;
    const int ly = threadIdx.x / a.lanes_per_row;
    const int lx = threadIdx.x - ly * a.lanes_per_row;
    const int y = band * a.rows_per_block + ly;
    const bool active = ly < a.rows_per_block && y < a.h;
    const int yc = min(y, a.h - 1);
    const int x0 = lx * 8;

    unsigned wraw[9];   // converted behind the loads of x: nothing waits for a tap before those are out
    dw_load_taps(a, ch, wraw);

    const _Float16 *src = a.x + static_cast<int64_t>(plane) * a.h * a.wd;
    // all loads first, from clamped (always valid) addresses; what lies outside the plane is zeroed after
    f16x8_t mid[3];
    _Float16 left[3], right[3];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int yy = min(max(yc + r - 1, 0), a.h - 1);
        const _Float16 *row = src + static_cast<int64_t>(yy) * a.wd;
        mid[r] = *reinterpret_cast<const f16x8_t *>(row + x0);
        left[r] = row[max(x0 - 1, 0)];
        right[r] = row[min(x0 + 8, a.wd - 1)];
    }
    float v[3][10];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int yy = yc + r - 1;
        const bool row_in = yy >= 0 && yy < a.h;
        v[r][0] = row_in && x0 > 0 ? static_cast<float>(left[r]) : 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[r][e + 1] = row_in ? static_cast<float>(mid[r][e]) : 0.0f;
        v[r][9] = row_in && x0 + 8 < a.wd ? static_cast<float>(right[r]) : 0.0f;
    }
    float wt[9];
    dw_taps_to_float(a, wraw, wt);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
#pragma unroll
    for (int r = 0; r < 3; ++r)   // tap order: rows first, left to right (a tap outside the plane adds w * 0)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf(wt[r * 3 + j], v[r][e + j], acc[e]);
    f16x8_t out;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float o = acc[e];
        asm volatile("" : "+v"(o));   // fp32 sum first, conversion as a step of its own
        out[e] = static_cast<_Float16>(o);
    }
    if (active) *reinterpret_cast<f16x8_t *>(a.y + (static_cast<int64_t>(plane) * a.h + y) * a.wd + x0) = out;
}

// Four output rows per lane: six row loads serve 32 outputs (the one-row form above loads three rows and six edge
// halfs for 8), and the two edge columns of a lane's window come from its neighbours' registers (the lanes of a plane
// row sit side by side in one wave: lanes_per_row divides 64), so a lane has 96 bytes in flight from six loads and
// nothing else to wait for.  Same sums, in the same order, as the one-row form.
constexpr int kDwRowsPerLane = 4;
typedef unsigned dw_u32x4_t __attribute__((ext_vector_type(4)));

__global__ __launch_bounds__(kDwThreads) void depthwise3x3_rows4_kernel(DwArgs a) {
    const int plane = blockIdx.x / a.blocks_per_plane;
    const int band = blockIdx.x - plane * a.blocks_per_plane;
    const int ch = plane % a.c;
    const int ly = threadIdx.x / a.lanes_per_row;
    const int lx = threadIdx.x - ly * a.lanes_per_row;
    const int y0 = (band * a.rows_per_block + ly) * kDwRowsPerLane;   // rows_per_block: lane rows per workgroup
    const bool lane_row_in = ly < a.rows_per_block;
    const int x0 = lx * 8;

    unsigned wraw[9];
    dw_load_taps(a, ch, wraw);

    const _Float16 *src = a.x + static_cast<int64_t>(plane) * a.h * a.wd;
    // all loads first, from clamped (always valid) addresses; what lies outside the plane is zeroed after
    dw_u32x4_t raw[kDwRowsPerLane + 2];
#pragma unroll
    for (int r = 0; r < kDwRowsPerLane + 2; ++r) {
        const int yy = min(max(y0 + r - 1, 0), a.h - 1);
        raw[r] = *reinterpret_cast<const dw_u32x4_t *>(src + static_cast<int64_t>(yy) * a.wd + x0);
    }
    float v[kDwRowsPerLane + 2][10];
#pragma unroll
    for (int r = 0; r < kDwRowsPerLane + 2; ++r) {
        const int yy = y0 + r - 1;
        const bool row_in = yy >= 0 && yy < a.h;
        // the neighbours' last and first column (every lane of the wave takes part; a row's first and last lane
        // get their own value back and are zeroed: the plane's edge)
        const unsigned lw = __shfl_up(raw[r].w, 1, a.lanes_per_row);
        const unsigned rw = __shfl_down(raw[r].x, 1, a.lanes_per_row);
        const _Float16 left = __builtin_bit_cast(_Float16, static_cast<unsigned short>(lw >> 16));
        const _Float16 right = __builtin_bit_cast(_Float16, static_cast<unsigned short>(rw & 0xffffu));
        const f16x8_t mid = __builtin_bit_cast(f16x8_t, raw[r]);
        v[r][0] = row_in && lx > 0 ? static_cast<float>(left) : 0.0f;
#pragma unroll
        for (int e = 0; e < 8; ++e) v[r][e + 1] = row_in ? static_cast<float>(mid[e]) : 0.0f;
        v[r][9] = row_in && lx + 1 < a.lanes_per_row ? static_cast<float>(right) : 0.0f;
    }
    float wt[9];
    dw_taps_to_float(a, wraw, wt);
#pragma unroll
    for (int k = 0; k < kDwRowsPerLane; ++k) {
        float acc[8];
#pragma unroll
        for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
#pragma unroll
        for (int r = 0; r < 3; ++r)   // tap order: rows first, left to right (a tap outside the plane adds w * 0)
#pragma unroll
            for (int j = 0; j < 3; ++j)
#pragma unroll
                for (int e = 0; e < 8; ++e) acc[e] = __builtin_fmaf(wt[r * 3 + j], v[k + r][e + j], acc[e]);
        f16x8_t out;
#pragma unroll
        for (int e = 0; e < 8; ++e) {
            float o = acc[e];
            asm volatile("" : "+v"(o));   // fp32 sum first, conversion as a step of its own
            out[e] = static_cast<_Float16>(o);
        }
        if (lane_row_in && y0 + k < a.h)
            *reinterpret_cast<f16x8_t *>(a.y + (static_cast<int64_t>(plane) * a.h + y0 + k) * a.wd + x0) = out;
    }
}

// Dilated variant (dilation a multiple of 4, padding = dilation): the three depthwise branches of the
// ASPP block (dilation 12 / 24 / 36 on 32x32 planes).  torch runs these on its own depthwise kernel,
// which accumulates the nine products in fp32 in tap order (row by row, left to right); so does this
// one, with the same fused multiply-adds (a tap outside the plane contributes w * 0, which leaves the
// sum as it is).  A lane's 8 outputs read, per tap, two groups of four consecutive inputs; with the
// dilation and the width multiples of 4 a group lies entirely inside or entirely outside the plane.
typedef _Float16 f16x4_t __attribute__((ext_vector_type(4)));
typedef unsigned u32x4_t __attribute__((ext_vector_type(4)));

// A plane smaller than the workgroup (32x32: 128 lanes) shares it with the next plane(s): `planes_per_block`.
__global__ __launch_bounds__(kDwThreads) void depthwise3x3_dilated_kernel(DwArgs a, int dil, int planes_per_block,
                                                                         int n_planes) {
    const int lanes_per_plane = a.lanes_per_row * a.rows_per_block;
    const int sub = planes_per_block > 1 ? static_cast<int>(threadIdx.x) / lanes_per_plane : 0;
    const int t = static_cast<int>(threadIdx.x) - sub * lanes_per_plane;
    const int plane_raw = planes_per_block > 1 ? static_cast<int>(blockIdx.x) * planes_per_block + sub
                                               : static_cast<int>(blockIdx.x) / a.blocks_per_plane;
    const int band = planes_per_block > 1 ? 0 : static_cast<int>(blockIdx.x) - plane_raw * a.blocks_per_plane;
    const int plane = min(plane_raw, n_planes - 1);
    const int ch = plane % a.c;
    const int ly = t / a.lanes_per_row;
    const int lx = t - ly * a.lanes_per_row;
    const int y = band * a.rows_per_block + ly;
    const bool active = plane_raw < n_planes && sub < planes_per_block && ly < a.rows_per_block && y < a.h;
    const int yc = min(y, a.h - 1);
    const int x0 = lx * 8;

    unsigned wraw[9];   // converted behind the loads of x: nothing waits for a tap before those are out
    dw_load_taps(a, ch, wraw);
    // One 16-byte load per tap through a buffer descriptor over the plane (8-byte aligned: the dilation is a
    // multiple of 4), nine loads with no branch between them - the address unit takes a wave's load in the same
    // sixteen clocks whether a lane asks for 8 bytes or 16, and eighteen 8-byte loads per lane were what bounded this
    // kernel.  A row above or below the plane is read from an offset outside the buffer (zeros, no memory access); a
    // group of four to the right of the row comes back with the next row's values and is zeroed; a load that would
    // start four columns left of the row starts at the row instead.
    const int first = planes_per_block > 1 ? static_cast<int>(blockIdx.x) * planes_per_block
                                           : static_cast<int>(blockIdx.x) / a.blocks_per_plane;   // uniform
    const int here = max(min(planes_per_block, n_planes - first), 0);   // planes of this workgroup inside the tensor
    const __amdgpu_buffer_rsrc_t plane_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16 *>(a.x + static_cast<int64_t>(min(first, n_planes - 1)) * a.h * a.wd), 0, here * a.h * a.wd * 2,
        0x00020000);
    const int plane_off = sub * a.h * a.wd;   // halfs, from the first plane of this workgroup
    f16x4_t g[3][3][2];
#pragma unroll
    for (int r = 0; r < 3; ++r) {
        const int yy = yc + (r - 1) * dil;
        const bool row_in = active && yy >= 0 && yy < a.h;
#pragma unroll
        for (int j = 0; j < 3; ++j) {
            const int s = x0 + (j - 1) * dil;   // a multiple of 4: the groups s .. s+3 and s+4 .. s+7 are each inside the row or outside it
            const bool neg = s < 0;             // then only s == -4 has a group inside: load from the row's start
            const bool left = row_in && !neg && s + 4 <= a.wd, right = row_in && s + 4 >= 0 && s + 8 <= a.wd;
            const int off = (left || right) ? (plane_off + yy * a.wd + (neg ? s + 4 : s)) * 2 : 0x40000000;
            const u32x4_t raw = __builtin_amdgcn_raw_buffer_load_b128(plane_rsrc, off, 0, 0);
            const unsigned r0 = left ? raw.x : 0u, r1 = left ? raw.y : 0u;
            const unsigned r2 = right ? (neg ? raw.x : raw.z) : 0u, r3 = right ? (neg ? raw.y : raw.w) : 0u;
            g[r][j][0] = __builtin_bit_cast(f16x4_t, uint2{r0, r1});
            g[r][j][1] = __builtin_bit_cast(f16x4_t, uint2{r2, r3});
        }
    }
    float wt[9];
    dw_taps_to_float(a, wraw, wt);
    float acc[8];
#pragma unroll
    for (int e = 0; e < 8; ++e) acc[e] = 0.0f;
#pragma unroll
    for (int r = 0; r < 3; ++r)
#pragma unroll
        for (int j = 0; j < 3; ++j)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                const float v = static_cast<float>(g[r][j][e >> 2][e & 3]);   // 0 outside the plane: w * 0 leaves the sum
                acc[e] = __builtin_fmaf(wt[r * 3 + j], v, acc[e]);
            }
    f16x8_t out;
#pragma unroll
    for (int e = 0; e < 8; ++e) {
        float v = acc[e];
        asm volatile("" : "+v"(v));   // fp32 sum first, conversion as a step of its own
        out[e] = static_cast<_Float16>(v);
    }
    if (active) *reinterpret_cast<f16x8_t *>(a.y + (static_cast<int64_t>(plane) * a.h + y) * a.wd + x0) = out;
}

}  // namespace spice

SPICE_ENV_KNOB(knob_dw_rows4, "SPICE_DW_ROWS4", 1)   // 0: always one output row per lane (A/B measurements)

static int spice_depthwise_common(const void *x, const void *weight, int32_t weight_dtype, void *y, int64_t n,
                                  int64_t c, int32_t h, int32_t w, int32_t dilation, void *stream);

extern "C" int spice_depthwise_conv3x3_dilated(const void *x, const void *weight, int32_t weight_dtype, void *y,
                                               int64_t n, int64_t c, int32_t h, int32_t w, int32_t dilation,
                                               void *stream) {
    if (dilation < 4 || dilation % 4 != 0) return SPICE_ERR_UNSUPPORTED;
    return spice_depthwise_common(x, weight, weight_dtype, y, n, c, h, w, dilation, stream);
}

extern "C" int spice_depthwise_conv3x3(const void *x, const void *weight, int32_t weight_dtype, void *y,
                                       int64_t n, int64_t c, int32_t h, int32_t w, void *stream) {
    return spice_depthwise_common(x, weight, weight_dtype, y, n, c, h, w, 1, stream);
}

static int spice_depthwise_common(const void *x, const void *weight, int32_t weight_dtype, void *y, int64_t n,
                                  int64_t c, int32_t h, int32_t w, int32_t dilation, void *stream) {
    using namespace spice;
    if (n < 0 || c <= 0 || h <= 0 || w <= 0) return SPICE_ERR_ARG;
    if (n == 0) return SPICE_OK;
    if (x == nullptr || weight == nullptr || y == nullptr) return SPICE_ERR_ARG;
    if (weight_dtype != 0 && weight_dtype != 1) return SPICE_ERR_UNSUPPORTED;
    if (w % 8 != 0 || w / 8 > kDwThreads || !aligned16(x) || !aligned16(y)) return SPICE_ERR_UNSUPPORTED;
    if (n * c > INT32_MAX / 64) return SPICE_ERR_UNSUPPORTED;
    DwArgs a;
    a.x = static_cast<const _Float16 *>(x);
    a.w = weight;
    a.y = static_cast<_Float16 *>(y);
    a.c = static_cast<int>(c);
    a.h = h;
    a.wd = w;
    a.lanes_per_row = w / 8;
    a.rows_per_block = kDwThreads / a.lanes_per_row;
    if (a.rows_per_block > h) a.rows_per_block = h;
    a.blocks_per_plane = static_cast<int>(ceil_div(h, a.rows_per_block));
    a.w_is_f32 = weight_dtype == 0;
    const int64_t blocks = n * c * a.blocks_per_plane;
    if (blocks > INT32_MAX) return SPICE_ERR_UNSUPPORTED;
    const int lane_rows = static_cast<int>(ceil_div(h, kDwRowsPerLane));   // rows of lanes a plane needs at four rows per lane
    if (dilation == 1 && knob_dw_rows4() != 0 && 64 % a.lanes_per_row == 0 && lane_rows * a.lanes_per_row >= kDwThreads / 2) {
        // four rows per lane where a plane keeps at least half a workgroup busy that way
        a.rows_per_block = kDwThreads / a.lanes_per_row;
        if (a.rows_per_block > lane_rows) a.rows_per_block = lane_rows;
        a.blocks_per_plane = static_cast<int>(ceil_div(lane_rows, a.rows_per_block));
        hipLaunchKernelGGL(depthwise3x3_rows4_kernel, dim3(static_cast<unsigned>(n * c * a.blocks_per_plane)),
                           dim3(kDwThreads), 0, static_cast<hipStream_t>(stream), a);
    } else if (dilation == 1)
        hipLaunchKernelGGL(depthwise3x3_kernel, dim3(static_cast<unsigned>(blocks)), dim3(kDwThreads), 0,
                           static_cast<hipStream_t>(stream), a);
    else {
        // whole small planes side by side in one workgroup
        const int lanes_per_plane = a.lanes_per_row * a.rows_per_block;
        const int ppb = a.blocks_per_plane == 1 ? kDwThreads / lanes_per_plane : 1;
        const int64_t grid = ppb > 1 ? ceil_div(n * c, ppb) : blocks;
        // the kernel addresses the planes of a workgroup with 32-bit byte offsets
        if (static_cast<int64_t>(h) * w * 2 * ppb >= (int64_t{1} << 30)) return SPICE_ERR_UNSUPPORTED;
        hipLaunchKernelGGL(depthwise3x3_dilated_kernel, dim3(static_cast<unsigned>(grid)), dim3(kDwThreads), 0,
                           static_cast<hipStream_t>(stream), a, dilation, ppb, static_cast<int>(n * c));
    }
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}
