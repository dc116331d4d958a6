// 3x3 convolution (stride 1, dilation 1, padding 1) to AT MOST 32 output channels, fp16 NCHW, on the matrix cores, with
// the inference batch-norm (+ ReLU / SiLU) applied to the product before it is stored: the last decoder blocks and the
// segmentation head of the U-Net models (pruning/models.py:35-39: 128 -> 32 and 32 -> 32 channels on 256 x 256 planes,
// 32 -> 16, 16 -> 16 and 16 -> classes on 512 x 512).  In the general kernel of conv1x1_mfma.hip such a layer fills a
// quarter or a half of a 64-channel tile and restages x for every tap; its bytes ask for a fifth of the time.
//
// A workgroup (four waves) takes 4 rows x 128 columns of one image's output, all channels:
//   * per block of CK (16 or 32) input channels the patch of x under the tile - 6 x 130 pixels, zeros where it leaves
//     the image - is staged ONCE, pixel-major: patch[pixel][CK + 8] halfs.  Eight neighbouring lanes load the same 8
//     pixels of 8 consecutive channels (16 bytes each, a wave's request is 8 channels x 128 contiguous bytes),
//     transpose their 8 x 8 halfs among themselves and write one pixel's 8 channels each (one ds_write_b128);
//   * a tap's view of the patch is the same image moved by whole pixels, so the A operand of
//     v_mfma_f32_32x32x16_f16 (32 pixels x 16 channels) is one aligned ds_read_b128 per lane for every tap;
//   * the weights ([Cout][9][Cin], tap-major as for the general kernel) do not touch LDS: a lane's B operand (its
//     channel's row, 8 consecutive K) is a 16-byte buffer load, all of a block's taps requested before its products;
//   * wave v owns row v of the tile (four 32-pixel blocks: 64 accumulator registers);
//   * epilogue: accumulators -> fp16 -> LDS image [channel][pixel] -> 16-byte row pieces -> batch-norm tail -> memory.
// What lies outside an operand is requested outside its buffer descriptor and reads as zeros: no load behind a branch.
#include "bn_math.h"

namespace spice {

typedef _Float16 nw_half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 nw_half4_t __attribute__((ext_vector_type(4)));
typedef float nw_float16_t __attribute__((ext_vector_type(16)));

constexpr int kNwThreads = 256;
constexpr int kNwRows = 4, kNwCols = 128;                       // output pixels of a tile
constexpr int kNwPRows = kNwRows + 2, kNwPCols = kNwCols + 2;   // the patch under it
constexpr int kNwPieces = kNwCols / 8 + 2;                      // 8-pixel pieces per patch row: one more on either side for the edge columns
constexpr int kNwCPitch = kNwRows * kNwCols + 8;                // halfs per channel of the C image
constexpr int kNwOutside = 0x40000000;   // a byte offset outside W and outside an image of x (the host keeps both below 2^30 - 2^20)

struct NwArgs {
    const _Float16 *x, *w;
    _Float16 *y;
    const float *mean, *var, *weight, *bias;   // null mean: the plain product
    float eps;
    int relu;
    int M, cin, h, w_px;
    int tiles_x, tiles_y;
    unsigned total_tiles;
};

constexpr int nw_lds_halfs(int ck) {
    const int patch = kNwPRows * kNwPCols * (ck + 8), c = 32 * kNwCPitch;
    return patch > c ? patch : c;
}

// 8 x 8 halfs among eight neighbouring lanes: lane i comes with 8 pixels of channel i and leaves with the 8
// channels of ONE pixel, pixel 4 * bit1(i) + 2 * bit2(i) + bit0(i).
__device__ __forceinline__ uint4 nw_transpose8(uint4 v, int lane) {
    const unsigned sel16 = (lane & 1) ? 0x03020706u : 0x05040100u;
    unsigned d0 = v.x, d1 = v.y, d2 = v.z, d3 = v.w;
    // lanes l, l ^ 1: 2 x 2 blocks of halfs
    d0 = __builtin_amdgcn_perm(__builtin_amdgcn_mov_dpp(d0, 0xB1, 0xF, 0xF, true), d0, sel16);
    d1 = __builtin_amdgcn_perm(__builtin_amdgcn_mov_dpp(d1, 0xB1, 0xF, 0xF, true), d1, sel16);
    d2 = __builtin_amdgcn_perm(__builtin_amdgcn_mov_dpp(d2, 0xB1, 0xF, 0xF, true), d2, sel16);
    d3 = __builtin_amdgcn_perm(__builtin_amdgcn_mov_dpp(d3, 0xB1, 0xF, 0xF, true), d3, sel16);
    {   // lanes l, l ^ 2: 2 x 2 blocks of dwords
        const bool hi = (lane & 2) != 0;
        const unsigned r0 = __builtin_amdgcn_mov_dpp(hi ? d0 : d2, 0x4E, 0xF, 0xF, true);
        const unsigned r1 = __builtin_amdgcn_mov_dpp(hi ? d1 : d3, 0x4E, 0xF, 0xF, true);
        const unsigned n0 = hi ? r0 : d0, n1 = hi ? d2 : r0, n2 = hi ? r1 : d1, n3 = hi ? d3 : r1;
        d0 = n0, d1 = n1, d2 = n2, d3 = n3;
    }
    {   // lanes l, l ^ 4: 2 x 2 blocks of dword pairs (four lanes up for banks 0 and 2 of a row of 16, four down for 1 and 3)
        const bool hi = (lane & 4) != 0;
        const unsigned s0 = hi ? d0 : d2, s1 = hi ? d1 : d3;
        unsigned r0 = __builtin_amdgcn_update_dpp(0u, s0, 0x104 /* row_shl:4 */, 0xF, 0x5, false);
        r0 = __builtin_amdgcn_update_dpp(r0, s0, 0x114 /* row_shr:4 */, 0xF, 0xA, false);
        unsigned r1 = __builtin_amdgcn_update_dpp(0u, s1, 0x104, 0xF, 0x5, false);
        r1 = __builtin_amdgcn_update_dpp(r1, s1, 0x114, 0xF, 0xA, false);
        const unsigned n0 = hi ? r0 : d0, n1 = hi ? r1 : d1, n2 = hi ? d2 : r0, n3 = hi ? d3 : r1;
        d0 = n0, d1 = n1, d2 = n2, d3 = n3;
    }
    return make_uint4(d0, d1, d2, d3);
}

// The tile's C image [channel][pixel] -> 16-byte row pieces -> batch-norm tail -> memory.  ACT: -1 the plain product,
// else the tail code (a constant here: no branch per pair of values).  Every store is issued - outside the image or
// beyond the last channel it goes outside the descriptor and is dropped - so the pieces' reads, tails and stores
// of a lane are one straight run of instructions.
template <int ACT>
__device__ __forceinline__ void nw_store_tile(const _Float16 *Cs, const float4 *bn_lds, const NwArgs &a, size_t img, int x0,
                                              int y0, int tid) {
    const int W = a.w_px, plane = a.h * W;
    const __amdgpu_buffer_rsrc_t y_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(a.y + img * static_cast<size_t>(a.M) * plane, 0, a.M * plane * 2, 0x00020000);
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int q = tid + kNwThreads * i;
        const int m = q >> 6, pp = q & 63;
        const int row = pp >> 4, col = (pp & 15) * 8;
        const bool in = m < a.M && y0 + row < a.h && x0 + col < W;
        const uint4 c = *reinterpret_cast<const uint4 *>(Cs + m * kNwCPitch + row * kNwCols + col);
        unsigned d[4] = {c.x, c.y, c.z, c.w};
        if constexpr (ACT >= 0) {
            const float4 c4 = bn_lds[m];   // (beyond the last channel: whatever the array holds; the store is dropped)
            const BnChannel pch = {c4.x, c4.y, c4.z, c4.w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const half2_t in2 = __builtin_bit_cast(half2_t, d[e]);
                const half2_t out2 = bn_tail_pair<false, false>(in2, half2_t{0, 0}, pch, pch, ACT);
                d[e] = __builtin_bit_cast(unsigned, out2);
            }
        }
        const u32x4_t raw = {d[0], d[1], d[2], d[3]};
        __builtin_amdgcn_raw_buffer_store_b128(raw, y_rsrc, in ? (m * plane + (y0 + row) * W + x0 + col) * 2 : kNwOutside, 0, 0);
    }
}

__device__ __forceinline__ void nw_store_tile_any(const _Float16 *Cs, const float4 *bn_lds, const NwArgs &a, size_t img,
                                                  int x0, int y0, int tid) {
    if (a.mean == nullptr) nw_store_tile<-1>(Cs, bn_lds, a, img, x0, y0, tid);
    else if (a.relu == 0) nw_store_tile<0>(Cs, bn_lds, a, img, x0, y0, tid);
    else if (a.relu == 1) nw_store_tile<1>(Cs, bn_lds, a, img, x0, y0, tid);
    else nw_store_tile<2>(Cs, bn_lds, a, img, x0, y0, tid);
}

// The channels' batch-norm constants into LDS (read behind at least one barrier).
__device__ __forceinline__ void nw_stage_bn(float4 *bn_lds, const NwArgs &a, int tid) {
    if (a.mean != nullptr && tid < a.M) {
        const BnChannel p = bn_channel_of(a.mean, a.var, a.weight, a.bias, a.eps, static_cast<unsigned>(tid));
        bn_lds[tid] = make_float4(p.mean, p.inv, p.weight, p.bias);
    }
}

template <int CK, int OCC>
__global__ __launch_bounds__(kNwThreads, OCC) void conv3x3_narrow_kernel(NwArgs a) {
    constexpr int NO = CK / 8, NKS = CK / 16, XP = CK + 8;
    constexpr int GROUPS = kNwPRows * kNwPieces * NO;   // groups of eight lanes' pieces per block of channels
    constexpr int ROUNDS = (GROUPS + 31) / 32;
    __shared__ __attribute__((aligned(16))) _Float16 lds[nw_lds_halfs(CK) + 256];
    float4 *bn_lds = reinterpret_cast<float4 *>(lds + nw_lds_halfs(CK));
    nw_stage_bn(bn_lds, a, threadIdx.x);
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    // workgroups are dealt to the eight XCDs round-robin: neighbouring tiles (which share patch rows) go to ONE XCD's L2
    unsigned tile = blockIdx.x;
    if (a.total_tiles % 8u == 0) tile = (tile & 7u) * (a.total_tiles >> 3) + (tile >> 3);
    const int tx = static_cast<int>(tile % static_cast<unsigned>(a.tiles_x));
    const unsigned rest = tile / static_cast<unsigned>(a.tiles_x);
    const int ty = static_cast<int>(rest % static_cast<unsigned>(a.tiles_y));
    const size_t img = rest / static_cast<unsigned>(a.tiles_y);
    const int x0 = tx * kNwCols, y0 = ty * kNwRows;
    const int W = a.w_px, plane = a.h * W;

    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(a.w), 0, a.M * 9 * a.cin * 2, 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16 *>(a.x + img * static_cast<size_t>(a.cin) * plane), 0, a.cin * plane * 2, 0x00020000);
    auto as_uint4 = [](u32x4_t v) { return make_uint4(v.x, v.y, v.z, v.w); };

    const int px_of_lane = 4 * ((lane >> 1) & 1) + 2 * ((lane >> 2) & 1) + (lane & 1);
    // the lane's row of W: channel lane & 31, K from 8 h on
    const int w_row = (lane & 31) < a.M ? ((lane & 31) * 9 * a.cin + 8 * h) * 2 : kNwOutside;
    // the lane's pixels: row `wave` of the tile, pixel 32 t + (lane & 31); their place in the patch at tap (0, 0)
    const int px_base = (wave * kNwPCols + (lane & 31)) * XP + 8 * h;

    nw_float16_t acc[4];
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int r = 0; r < 16; ++r) acc[t][r] = 0.0f;

    const int n_chunks = a.cin / CK;
    for (int cc = 0; cc < n_chunks; ++cc) {
        if (cc > 0) __syncthreads();   // the products of the block before have read the patch
        uint4 wq[9][NKS];
#pragma unroll
        for (int tap = 0; tap < 9; ++tap)
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks)
                wq[tap][ks] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_row + (tap * a.cin + cc * CK + 16 * ks) * 2, 0, 0));
        {
            uint4 v[ROUNDS];
            int dst[ROUNDS];
#pragma unroll
            for (int j = 0; j < ROUNDS; ++j) {
                const int gi = (tid >> 3) + 32 * j;
                const int g = gi % kNwPieces, t1 = gi / kNwPieces;
                const int o = t1 % NO, r = t1 / NO;
                const int c = cc * CK + 8 * o + (tid & 7);
                const int gy = y0 - 1 + r, gx = x0 - 8 + 8 * g;
                // (the width is a multiple of 8: a piece lies inside a row or outside it)
                const bool in = gi < GROUPS && static_cast<unsigned>(gy) < static_cast<unsigned>(a.h) &&
                                static_cast<unsigned>(gx) < static_cast<unsigned>(W);
                const int off = in ? (c * plane + gy * W + gx) * 2 : kNwOutside;
                const int pc = 8 * g + px_of_lane - 7;   // patch column of the pixel the transposition leaves with this lane
                dst[j] = (gi < GROUPS && pc >= 0 && pc < kNwPCols) ? (r * kNwPCols + pc) * XP + 8 * o : -1;
                v[j] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(x_rsrc, off, 0, 0));
            }
#pragma unroll
            for (int j = 0; j < ROUNDS; ++j) {
                const uint4 tv = nw_transpose8(v[j], lane);
                if (dst[j] >= 0) *reinterpret_cast<uint4 *>(lds + dst[j]) = tv;
            }
        }
        __syncthreads();
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            const int toff = ((tap / 3) * kNwPCols + (tap % 3)) * XP;
#pragma unroll
            for (int ks = 0; ks < NKS; ++ks) {
                const nw_half8_t wh = __builtin_bit_cast(nw_half8_t, wq[tap][ks]);
#pragma unroll
                for (int t = 0; t < 4; ++t) {
                    const nw_half8_t fx = *reinterpret_cast<const nw_half8_t *>(lds + px_base + 32 * t * XP + toff + 16 * ks);
                    acc[t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fx, wh, acc[t], 0, 0, 0);
                }
            }
        }
    }
    __syncthreads();

    // accumulators -> fp16 -> C image [channel][pixel of the tile] -> 16-byte row pieces -> batch-norm tail -> memory
    // (a lane holds channel lane & 31, pixels 8 (r >> 2) + 4 h + (r & 3) of its 32-pixel block in register r)
    _Float16 *Cs = lds;
#pragma unroll
    for (int t = 0; t < 4; ++t)
#pragma unroll
        for (int rq = 0; rq < 4; ++rq) {
            nw_half4_t v;
#pragma unroll
            for (int e = 0; e < 4; ++e) v[e] = static_cast<_Float16>(acc[t][4 * rq + e]);
            *reinterpret_cast<nw_half4_t *>(Cs + (lane & 31) * kNwCPitch + wave * kNwCols + 32 * t + 8 * rq + 4 * h) = v;
        }
    __syncthreads();
    nw_store_tile_any(Cs, bn_lds, a, img, x0, y0, tid);
}

SPICE_ENV_KNOB(nw_knob, "SPICE_CONV3X3_NARROW", 1)   // 0 = layers of at most 32 output channels stay with the general kernel; 2 = this form wherever it applies

// The narrow form where it applies, SPICE_ERR_UNSUPPORTED otherwise (the caller then takes the general kernel).
int conv3x3_narrow_launch(const void *x, const void *weight9, void *y, int64_t n, int64_t cin, int32_t h, int32_t w,
                          int64_t cout, const float *mean, const float *var, const float *bn_weight,
                          const float *bn_bias, float eps, int32_t relu, void *stream) {
    if (nw_knob() == 0) return SPICE_ERR_UNSUPPORTED;
    if (n <= 0 || cin <= 0 || cout <= 0 || cout > 32 || h <= 0 || w <= 0) return SPICE_ERR_UNSUPPORTED;
    if (x == nullptr || weight9 == nullptr || y == nullptr) return SPICE_ERR_UNSUPPORTED;
    if (mean != nullptr && (var == nullptr || bn_weight == nullptr || bn_bias == nullptr)) return SPICE_ERR_UNSUPPORTED;
    // blocks of 16 input channels (no block ends inside a K step; rows of W that are whole 16-byte pieces); the sums
    // of deeper layers belong on the general kernel's larger tiles
    if ((cin & 15) != 0 || cin > 512 || (w & 7) != 0) return SPICE_ERR_UNSUPPORTED;
    if (!aligned16(x) || !aligned16(y) || !aligned16(weight9)) return SPICE_ERR_UNSUPPORTED;
    const int64_t limit = (int64_t{1} << 30) - (int64_t{1} << 20);
    if (cin * static_cast<int64_t>(h) * w * 2 > limit || cout * static_cast<int64_t>(h) * w * 2 > limit) return SPICE_ERR_UNSUPPORTED;
    NwArgs a;
    a.x = static_cast<const _Float16 *>(x);
    a.w = static_cast<const _Float16 *>(weight9);
    a.y = static_cast<_Float16 *>(y);
    a.mean = mean;
    a.var = var;
    a.weight = bn_weight;
    a.bias = bn_bias;
    a.eps = eps;
    a.relu = tail_code(relu);
    a.M = static_cast<int>(cout);
    a.cin = static_cast<int>(cin);
    a.h = h;
    a.w_px = w;
    a.tiles_x = static_cast<int>(ceil_div(w, kNwCols));
    a.tiles_y = static_cast<int>(ceil_div(h, kNwRows));
    const int64_t total = static_cast<int64_t>(a.tiles_x) * a.tiles_y * n;
    if (total > INT32_MAX) return SPICE_ERR_UNSUPPORTED;
    // deep sums on planes that give fewer tiles than there are compute units stay with the general kernel's K-deep
    // pipeline (MI355X, batch 8: 512 -> 21 on 64 x 64, 128 tiles: 74 us here against 53; 256 -> 32 on 128 x 128, 256
    // tiles: 51 against 86)
    if (cin > 64 && total < device_cu_count() && nw_knob() != 2) return SPICE_ERR_UNSUPPORTED;
    a.total_tiles = static_cast<unsigned>(total);
    hipStream_t st = static_cast<hipStream_t>(stream);
    // blocks of 32 channels where there are several (fewer rounds of staging and barriers per tile: 101 against 105 us
    // for 128 -> 32 on 256 x 256, batch 8), blocks of 16 for 32 channels (three workgroups per compute unit instead
    // of two: 85 against 106 us for 32 -> 16 on 512 x 512)
    if ((cin & 31) == 0 && cin >= 64)
        hipLaunchKernelGGL((conv3x3_narrow_kernel<32, 2>), dim3(a.total_tiles), dim3(kNwThreads), 0, st, a);
    else
        hipLaunchKernelGGL((conv3x3_narrow_kernel<16, 2>), dim3(a.total_tiles), dim3(kNwThreads), 0, st, a);
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}

}  // namespace spice
