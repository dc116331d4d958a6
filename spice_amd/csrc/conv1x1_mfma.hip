// Convolutions of fp16 NCHW activations on the matrix cores, with the inference batch-norm and the element-wise
// tail of a residual block applied to the product before it leaves the compute unit:
//   Y[n][Cout, HW] = relu?( bn( W[Cout, K] @ X'[n][K, HW] ) [+ R | + bn_r(R)] )
// one launch instead of a convolution and a batch-norm pass over its result (the layers of pruning/models.py:35-39's
// networks: every bottleneck's conv1/bn1, conv2/bn2 and conv3/bn3, projection shortcuts, the stem, the ASPP branches
// and projections).  X' is x itself (1x1), its even pixels (1x1, stride 2), nine shifted views of it (3x3: K = 9 Cin,
// weights tap-major) or 49 shifted, column-halved views (7x7 stride 2: K = 49 Cin); the kernel picks them from x while
// it stages it - no im2col buffer, no layout copies.  The product is rounded to fp16 where a separate convolution
// would have stored it and the tail is the arithmetic of bn_math.h, so the values are those of the separate launches
// up to the order of the fp32 sums.  mean == nullptr leaves the tail out (the plain product).
//
// Tile: 128 (x MT) channels x 64 or 128 pixels of one image, K in steps of 64.
//   LDS   A[128 MT][72]  weights, K contiguous (144-byte rows: a ds_read_b128 of 16 consecutive rows covers all 64 banks once)
//         B[64][64] (x 2 side by side)  activations as they lie in memory, pixels contiguous, 16-byte pieces XOR-swizzled by the row
//   v_mfma_f32_32x32x16_f16 with the PIXELS as the rows of the product (so that a lane ends up holding four consecutive
//   pixels of one channel):
//         A operand: lane l holds X^T[pixel l & 31][k = 8 (l >> 5) + 0..7]      two ds_read_b64_tr_b16, which hand each
//                    lane a COLUMN of a 4 x 16 block of the pixel-contiguous image
//         B operand: lane l holds W^T[k = 8 (l >> 5) + 0..7][channel l & 31]    one ds_read_b128 of a row of A
//         result:    lane l holds channel l & 31, pixels (r & 3) + 8 (r >> 2) + 4 (l >> 5) in register r
//   Global loads are buffer loads with 32-bit offsets; what lies outside an operand reads as zeros, so no load sits
//   behind a branch.  Five forms (STAGES): 1 = four waves (32 MT channels x 64 pixels each), one LDS image, four
//   workgroups per compute unit: many tiles, short K; 3 = eight waves (32 x 32 each), two LDS images, two register
//   stages, one barrier per K step: deep K on one or two tiles per compute unit; 4 = the same on 128 pixels (32 x 64 per
//   wave): deep K on many tiles; 5 = eight waves (32 x 32 each) on 64 channels x 128 pixels: layers of at most 64
//   output channels, which would leave half of a 128-channel tile's waves without rows; 6 = sixteen waves (32 x 32
//   each) on 128 channels x 128 pixels: deep K where such tiles make one round of the compute units.
//   Epilogue: accumulators -> fp16 -> LDS C image (8-byte writes) -> row-major 16-byte pieces -> tail -> global.
//   The residual's pieces are requested before the products start, the batch-norm constants of a lane's rows right
//   behind the last K step, and which tail applies is decided once, outside the rows: behind the last barrier a
//   lane's rows are one straight run of LDS reads, arithmetic and stores, with no wait for memory and no branch per
//   pair of values (constants fetched row by row behind that barrier cost a round trip to memory per row).
#include "bn_math.h"

namespace spice {

constexpr int kMmThreads = 256;
constexpr int kMmBM = 128, kMmBK = 64;   // channels per tile (x MT), K per step; pixels per tile: mm_tile_pixels
constexpr int kMmAPitch = 72;    // halfs
constexpr int kMmBPitch = 64;    // halfs (swizzled, no padding)
constexpr int kMmPreMaxK = 512;  // input channels whose pending batch-norm the kernel can hold in LDS (8 KB)

typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef short short8_t __attribute__((ext_vector_type(8)));
typedef short short4_t __attribute__((__vector_size__(4 * sizeof(short))));
typedef float float16_t __attribute__((ext_vector_type(16)));
typedef __attribute__((address_space(3))) short4_t *lds_short4_ptr;

struct MmArgs {
    const _Float16 *x;      // [n][K][hw]
    const _Float16 *w;      // [M][K]
    _Float16 *y;            // [n][M][hw], images y_image_stride elements apart (a channel slice of a larger tensor)
    int64_t y_image_stride;
    const _Float16 *res;    // [n][M][hw] or null
    const float *mean, *var, *weight, *bias;           // null mean: no batch-norm, no tail
    const float *r_mean, *r_var, *r_weight, *r_bias;   // batch-norm of the residual, or null
    const float *p_mean, *p_var, *p_weight, *p_bias;   // batch-norm (+ ReLU) still to be applied to x, per INPUT channel, or null
    float eps, r_eps, p_eps;
    int p_relu;
    int M, K, hw;           // hw: pixels per image of the PRODUCT
    int x_hw, x_w, y_w;     // stride 2 and 3x3: pixels per image and row width of x, row width of the product
    int x_h, cin, stride, dil;   // 3x3 and 7x7: rows of x, its channels (K = 9 | 49 cin); 3x3: stride (1 | 2) and dilation
    int tiles_m, tiles_n;   // per image
    unsigned total_tiles;
    int relu;
    int chunked;            // 3x3, stride 1: K steps run over the nine taps of 64 channels before they move to the next 64
};

// Workgroups are dealt to the eight XCDs round-robin; consecutive tiles (same pixels, next 128 channels) share
// their X tile, so a run of consecutive tiles goes to ONE XCD's L2 when the tile count allows.
__device__ __forceinline__ unsigned mm_tile_of_block(unsigned id, unsigned total) {
    if (total % 8u != 0) return id;
    return (id & 7u) * (total >> 3) + (id >> 3);
}

// Pixels per tile of the forms: STAGES 1 (four waves) and 3 (eight waves) 64, STAGES 4, 5 (eight waves) and 6 (sixteen) 128.
constexpr int mm_tile_pixels(int stages) { return stages >= 4 ? 128 : 64; }
// Lanes per workgroup
constexpr int mm_threads(int stages) { return stages == 6 ? 4 * kMmThreads : (stages > 1 ? 2 * kMmThreads : kMmThreads); }
// Channels per tile: 128 (x MT), or 64 in form 5 (layers of at most 64 output channels: no half-empty tile).
constexpr int mm_tile_channels(int stages, int mt) { return stages == 5 ? 64 : kMmBM * mt; }
constexpr int mm_lds_bytes(int stages, int mt) {
    return (stages > 1 ? 2 : 1) * (mm_tile_channels(stages, mt) * kMmAPitch + kMmBK * mm_tile_pixels(stages)) * 2;
}

// Offset (halfs) of 16-byte piece `piece` (0..7) of row `k` of the B image.  Rows are 128 bytes, so rows k and
// k + 2 start on the same bank; the transposed read of a half-wave takes 4 rows x 2 pairs of pieces, which land
// on eight different 32-byte bank ranges once the pieces of rows with bit 1 set are exchanged pairwise.
__device__ __forceinline__ int mm_b_offset(int k, int piece) { return k * kMmBPitch + ((piece ^ ((k & 2) << 1)) << 3); }

// STAGES: 1 = one LDS image, two barriers per K step, room for four workgroups per compute unit (the others'
// loads hide this one's latency: the case of many tiles and short K); above 1 = two LDS images and one barrier per
// step, for deep K on few tiles, where a compute unit holds one or two workgroups.
// XMODE kXStride2: the convolution has stride 2 - the product's pixel (y, x) is x's pixel (2y, 2x): a piece of 8 product pixels
// is the even elements of 16 consecutive pixels of x (two 16-byte loads).
// PRE: x is the raw result of the convolution before; its batch-norm and ReLU (per input channel = per row of the X
// tile) are applied to the pieces on their way from the staging registers into LDS, rounded to fp16 as the separate
// batch-norm launch stores them: that launch, its read of x and its write of the activated tensor are gone.
//
// X3: a 3x3 convolution (padding = dilation) as nine shifted 1x1 products in one sum: W comes as [Cout][9][Cin]
// (K = 9 Cin, row kk = tap * Cin + channel), and row kk of the X tile is channel kk % Cin of x read at the tap's offset:
// with stride 1 one 16-byte load from an address that is only 2-byte aligned (started inside the row at the image's
// left and right edge and shifted in registers, zeros coming in), with stride 2 eight single halfs; rows of x above
// and below the image are zeros.
constexpr int kMmOutside = 0x40000000;   // a byte offset outside W and outside an image of x (the host keeps both below 2^30 - 2^20), and still positive after a whole K of steps is added
constexpr int kXPlain = 0, kXStride2 = 1, kX3x3 = 2, kX3x3S2 = 3;   // 3x3 with stride 1 / stride 2
constexpr int kX7x7S2 = 4;   // 7x7, stride 2, padding 3 (a ResNet stem): K row kk is channel kk / 49, tap (kk % 49 / 7, kk % 7)
typedef u32x4_t u32x4_any_t __attribute__((aligned(2)));   // a 16-byte load from a 2-byte aligned address

// v's eight halfs moved up by n elements (n > 0: zeros enter at element 0) or down by -n (zeros enter at element 7)
__device__ __forceinline__ uint4 mm_shift_halfs(uint4 v, int n) {
    unsigned long long lo = (static_cast<unsigned long long>(v.y) << 32) | v.x;
    unsigned long long hi = (static_cast<unsigned long long>(v.w) << 32) | v.z;
    if (n > 0) {
        const int b = 16 * n;
        if (b >= 64) {
            hi = b >= 128 ? 0ull : lo << (b - 64);
            lo = 0ull;
        } else {
            hi = (hi << b) | (lo >> (64 - b));
            lo <<= b;
        }
    } else if (n < 0) {
        const int b = -16 * n;
        if (b >= 64) {
            lo = b >= 128 ? 0ull : hi >> (b - 64);
            hi = 0ull;
        } else {
            lo = (lo >> b) | (hi << (64 - b));
            hi >>= b;
        }
    }
    return make_uint4(static_cast<unsigned>(lo), static_cast<unsigned>(lo >> 32), static_cast<unsigned>(hi),
                      static_cast<unsigned>(hi >> 32));
}

template <int STAGES, int MT, int XMODE = kXPlain, bool PRE = false, bool W_ODD = false>
__global__ __launch_bounds__(mm_threads(STAGES), MT == 1 ? 4 : 2) void conv1x1_bn_act_mfma_kernel(MmArgs a) {
    constexpr int BM = mm_tile_channels(STAGES, MT);   // channels per workgroup: MT 32-channel product tiles per wave
    constexpr int MW = BM / (32 * MT);      // waves along the channels; the others along the pixels
    // deep K: eight waves (sixteen in form 6), a 32-channel x 32-pixel product each (twice the waves to hide each other's
    // waits: these launches leave a compute unit one or two workgroups); otherwise four waves with 32 MT channels x 64
    // pixels each
    constexpr int THREADS = mm_threads(STAGES);
    constexpr int NPX = mm_tile_pixels(STAGES);   // pixels per tile: 64, or 128 as two 64-pixel sub-images side by side
    constexpr int PCS = NPX / 8;                  // 16-byte pieces per row of the X tile and of the product
    constexpr int NT = (STAGES == 3 || STAGES == 5 || STAGES == 6) ? 1 : 2;   // 32-pixel blocks per wave
    constexpr int A_PIECES = BM * 8 / THREADS;            // 16-byte pieces of A per lane and K step
    constexpr int B_PIECES = kMmBK * PCS / THREADS;       // ... of X
    constexpr int ROWS = BM * PCS / THREADS;              // rows per lane in the row-major pass of the epilogue
    constexpr int RSTEP = THREADS / PCS;                  // ... which lie this many rows apart
    constexpr bool DEEP = STAGES > 1;       // two LDS images: a step's products run beside the next step's stores
    constexpr int IMG = BM * kMmAPitch + kMmBK * NPX;   // halfs of one image (A tile, then X tile)
    constexpr int CP = NPX + 8;                   // pitch of the C image, halfs
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];   // (DEEP ? 2 : 1) * IMG halfs
    static_assert(BM * CP <= (DEEP ? 2 : 1) * IMG, "the C image fits in the images' storage");
    __shared__ __attribute__((aligned(16))) float pre[PRE ? kMmPreMaxK * 4 : 4];   // (mean, inv, weight, bias) per input channel
    _Float16 *As = lds;
    _Float16 *Bs = lds + BM * kMmAPitch;

    const unsigned tile = mm_tile_of_block(blockIdx.x, a.total_tiles);
    const int mt = static_cast<int>(tile % static_cast<unsigned>(a.tiles_m));
    const unsigned rest = tile / static_cast<unsigned>(a.tiles_m);
    const int nt = static_cast<int>(rest % static_cast<unsigned>(a.tiles_n));
    const size_t img = rest / static_cast<unsigned>(a.tiles_n);
    const int m0 = mt * BM, n0 = nt * NPX;

    const int tid = threadIdx.x;
    const int lane = tid & 63, wave = tid >> 6;
    if constexpr (PRE) {   // read in store_step, behind the barrier at the top of the K loop
        for (int k = tid; k < a.K; k += THREADS) {
            const BnChannel c = bn_channel_of(a.p_mean, a.p_var, a.p_weight, a.p_bias, a.p_eps, k);
            *reinterpret_cast<float4 *>(pre + 4 * k) = make_float4(c.mean, c.inv, c.weight, c.bias);
        }
    }

    // the row-major pass of the epilogue: lane takes 8 consecutive pixels of one channel, ROWS such pieces
    const int e_pc = (tid % PCS) * 8, e_row = tid / PCS;   // rows e_row + RSTEP i
    const int e_p = n0 + e_pc;
    const bool has_bn = a.mean != nullptr, has_res = a.res != nullptr, res_bn = a.r_mean != nullptr;
    uint4 rres[ROWS];

    constexpr bool IS3 = XMODE == kX3x3 || XMODE == kX3x3S2;
    constexpr bool IS7 = XMODE == kX7x7S2;
    const int x_plane = XMODE != kXPlain ? a.x_hw : a.hw;   // pixels per channel of x
    const int x_chans = (IS3 || IS7) ? a.cin : a.K;
    const _Float16 *xi = a.x + img * static_cast<size_t>(x_chans) * x_plane;
    // W, and this image of x, behind buffer descriptors: 32-bit byte offsets, and an offset outside the buffer reads
    // zeros without touching memory - which is how rows beyond M, pixels beyond the image, channels beyond K and
    // taps above / below the image come out as zeros (kMmOutside is outside every buffer: the host checks the sizes)
    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(a.w), 0, (a.M * a.K * 2 + 3) & ~3, 0x00020000);   // whole dwords: the range check is per dword
    const __amdgpu_buffer_rsrc_t x_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(xi), 0, x_chans * x_plane * 2, 0x00020000);
    // the lane's piece of an X row: pixels n0 + 8 (tid % PCS) .. +7 of the product (the same piece in every K step)
    const int b_pc = tid % PCS;
    const int b_p = n0 + b_pc * 8;
    const bool b_in = b_p < a.hw;   // hw is a multiple of 8: a piece is inside the image or outside it
    int b_src = b_p;
    int b_yo = 0, b_xo = 0;   // the piece's row and first column in the product (a piece lies inside one row: y_w % 8 == 0)
    if constexpr (XMODE != kXPlain) {
        b_yo = b_p / a.y_w;
        b_xo = b_p - b_yo * a.y_w;
        b_src = 2 * b_yo * a.x_w + 2 * b_xo;
    }

    // Byte offsets of the lane's pieces, moved on by load_step, which is called once per K step in ascending order.
    // A: row m, columns 8 (c & 7) ..; rows beyond M stay outside.
    int a_off[A_PIECES];
#pragma unroll
    for (int i = 0; i < A_PIECES; ++i) {
        const int c = tid + THREADS * i;
        const int m = m0 + (c >> 3);
        a_off[i] = m < a.M ? (m * a.K + (c & 7) * 8) * 2 : kMmOutside;
    }
    // X, 1x1: channel c / PCS + 64 step; channels beyond K are beyond the buffer.
    // X, 3x3: K row kk = tap * cin + ci is channel ci shifted by tap (dy, dx) * dilation; x3_chan = byte offset of the
    // channel, x3_dy / x3_dx = the shift, x3_tap > 8 = beyond K.
    int b_off[B_PIECES], x3_tap[B_PIECES], x3_ci[B_PIECES], x3_dy[B_PIECES], x3_dx[B_PIECES];
#pragma unroll
    for (int i = 0; i < B_PIECES; ++i) {
        const int kk = (tid + THREADS * i) / PCS;
        if constexpr (IS3) {
            x3_tap[i] = kk / a.cin;
            x3_ci[i] = kk - x3_tap[i] * a.cin;
            const int ty = x3_tap[i] / 3;
            x3_dy[i] = (ty - 1) * a.dil;
            x3_dx[i] = (x3_tap[i] - 3 * ty - 1) * a.dil;
            b_off[i] = x3_ci[i] * x_plane * 2;
        } else if constexpr (IS7) {
            b_off[i] = kk;   // the K row itself: the tap is decoded per step (three steps for three input channels)
        } else {
            b_off[i] = b_in ? (kk * x_plane + b_src) * 2 : kMmOutside;
        }
    }

    // global -> register staging per K step: A_PIECES 16-byte pieces of A and B_PIECES of X per lane.  No load is
    // behind a branch, so the compiler can count, at each store, how many younger loads may still be in flight, and
    // the stages overlap; a load behind a branch makes it wait for all of them.  rb_meta (3x3): halfs by which the
    // piece is to be shifted when stored.
    constexpr int RS = DEEP ? 2 : 1;   // register stages: K steps whose pieces are in flight
    uint4 ra_all[RS][A_PIECES], rb_all[RS][B_PIECES];
    unsigned ra_hi_all[RS][W_ODD ? A_PIECES : 1];   // odd K: the dword behind each piece of A (see load_step)
    int rb_meta_all[RS][B_PIECES];
    auto as_uint4 = [](u32x4_t v) { return make_uint4(v.x, v.y, v.z, v.w); };
    auto load_step = [&](int k0, auto stage) {
        constexpr int S = decltype(stage)::value;
        auto &ra = ra_all[S];
        auto &rb = rb_all[S];
        auto &rb_meta = rb_meta_all[S];
        [[maybe_unused]] auto &ra_hi = ra_hi_all[S];
        // the columns of W this step takes: K as it lies, or (chunked) tap t % 9 of channels 64 (t / 9) ..
        int kcol = k0;
        if constexpr (XMODE == kX3x3) {
            if (a.chunked) {
                const int t = k0 / kMmBK, blk = t / 9;
                kcol = (t - 9 * blk) * a.cin + blk * kMmBK;
            }
        }
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i) {
            const int off = a_off[i] + 2 * kcol;
            if constexpr (W_ODD) {
                // odd K: every other row of W starts in the middle of a dword.  The range check of a buffer load is
                // per dword as addressed, so a dword that starts on W's last half would be dropped with it: read
                // whole aligned dwords, one more than the piece; store_step moves the bytes down
                ra[i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(w_rsrc, off & ~3, 0, 0));
                ra_hi[i] = __builtin_amdgcn_raw_buffer_load_b32(w_rsrc, (off & ~3) + 16, 0, 0);
            } else {
                ra[i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(w_rsrc, off, 0, 0));
            }
        }
#pragma unroll
        for (int i = 0; i < B_PIECES; ++i) {
            rb_meta[i] = 0;
            if constexpr (IS3) {
                const int iy = b_yo * a.stride + x3_dy[i];
                const bool in = b_in && x3_tap[i] < 9 && static_cast<unsigned>(iy) < static_cast<unsigned>(a.x_h);
                if constexpr (XMODE == kX3x3) {
                    // eight columns from ix0 on; at the row's ends the load is moved inside the row and the halfs
                    // shifted into place when stored
                    const int ix0 = b_xo + x3_dx[i];
                    const int start = min(max(ix0, 0), a.x_w - 8);
                    const int off = b_off[i] + (iy * a.x_w + start) * 2;
                    rb[i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(x_rsrc, in ? off : kMmOutside, 0, 0));
                    rb_meta[i] = start - ix0;
                } else {
                    rb[i] = make_uint4(0, 0, 0, 0);
                    if (in) {
                        const _Float16 *row = xi + (b_off[i] >> 1) + iy * a.x_w;
                        const int dx = x3_dx[i];
                        if (a.dil == 1) {
                            // columns 2 (xo + e) + dx of the 16 from 2 xo on: their even halfs (dx = 0), their odd halfs
                            // (dx = +1), or the column before them and the first seven odd halfs (dx = -1)
                            const _Float16 *src = row + 2 * b_xo;
                            const uint4 lo = *reinterpret_cast<const uint4 *>(src), hi = *reinterpret_cast<const uint4 *>(src + 8);
                            if (dx == 0) {
                                rb[i] = make_uint4((lo.x & 0xffffu) | (lo.y << 16), (lo.z & 0xffffu) | (lo.w << 16),
                                                   (hi.x & 0xffffu) | (hi.y << 16), (hi.z & 0xffffu) | (hi.w << 16));
                            } else {
                                uint4 odd = make_uint4((lo.x >> 16) | (lo.y & 0xffff0000u), (lo.z >> 16) | (lo.w & 0xffff0000u),
                                                       (hi.x >> 16) | (hi.y & 0xffff0000u), (hi.z >> 16) | (hi.w & 0xffff0000u));
                                if (dx < 0) {
                                    odd = mm_shift_halfs(odd, 1);
                                    if (b_xo > 0) odd.x |= static_cast<unsigned>(__builtin_bit_cast(unsigned short, src[-1]));
                                }
                                rb[i] = odd;
                            }
                        } else {
                            half8_t v = {0, 0, 0, 0, 0, 0, 0, 0};
#pragma unroll
                            for (int e = 0; e < 8; ++e) {
                                const int ix = (b_xo + e) * a.stride + dx;
                                if (ix >= 0 && ix < a.x_w) v[e] = row[ix];
                            }
                            rb[i] = __builtin_bit_cast(uint4, v);
                        }
                    }
                }
                // the next call is for the K step after this one: rows 64 further on - or (chunked) the same
                // channels at the next tap, whose lines of x the compute unit's L1 still holds, and 64 channels on
                // after the ninth
                if (XMODE == kX3x3 && a.chunked) {
                    ++x3_tap[i];
                    if (x3_dx[i] == a.dil) {
                        x3_dx[i] = -a.dil;
                        x3_dy[i] += a.dil;
                    } else {
                        x3_dx[i] += a.dil;
                    }
                    if (x3_tap[i] == 9) {
                        x3_tap[i] = 0;
                        x3_dy[i] = -a.dil;
                        b_off[i] += kMmBK * x_plane * 2;   // beyond the last channel: outside the buffer, zeros
                    }
                    continue;
                }
                x3_ci[i] += kMmBK;
                b_off[i] += kMmBK * x_plane * 2;
                while (x3_ci[i] >= a.cin) {
                    x3_ci[i] -= a.cin;
                    b_off[i] -= a.cin * x_plane * 2;
                    ++x3_tap[i];
                    if (x3_dx[i] == a.dil) {
                        x3_dx[i] = -a.dil;
                        x3_dy[i] += a.dil;
                    } else {
                        x3_dx[i] += a.dil;
                    }
                }
            } else if constexpr (IS7) {
                // below, all pieces together
            } else {
                if constexpr (XMODE == kXStride2) {
                    const uint4 lo = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(x_rsrc, b_off[i], 0, 0));
                    const uint4 hi = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(x_rsrc, b_off[i] + 16, 0, 0));
                    rb[i] = make_uint4((lo.x & 0xffffu) | (lo.y << 16), (lo.z & 0xffffu) | (lo.w << 16),
                                       (hi.x & 0xffffu) | (hi.y << 16), (hi.z & 0xffffu) | (hi.w << 16));
                } else {
                    rb[i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(x_rsrc, b_off[i], 0, 0));
                }
                b_off[i] += kMmBK * x_plane * 2;
            }
        }
        if constexpr (IS7) {
            // output columns xo .. xo+7 of tap (ky, kx) read input columns 2 (xo + e) + kx - 3: the even or the odd
            // halfs of the 16 columns from w0 on; at the row's ends the window is moved inside the row and the
            // picked halfs shifted into place, zeros (the padding) coming in.  All loads first, then the picks.
            uint4 lo[B_PIECES], hi[B_PIECES];
            int odd[B_PIECES], shift[B_PIECES];
#pragma unroll
            for (int i = 0; i < B_PIECES; ++i) {
                const int kk = b_off[i];
                const int c = kk / 49, r = kk - 49 * c, ky = r / 7, q = r - 7 * ky - 3;
                const int iy = 2 * b_yo + ky - 3;
                const bool in = b_in && kk < a.K && static_cast<unsigned>(iy) < static_cast<unsigned>(a.x_h);
                const int w0 = 2 * b_xo + (q & ~1);
                const int w0c = min(max(w0, 0), a.x_w - 16);
                const int off = in ? (c * x_plane + iy * a.x_w + w0c) * 2 : kMmOutside;
                lo[i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(x_rsrc, off, 0, 0));
                hi[i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(x_rsrc, off + 16, 0, 0));
                odd[i] = q & 1;
                shift[i] = (w0c - w0) / 2;
                b_off[i] += kMmBK;
            }
#pragma unroll
            for (int i = 0; i < B_PIECES; ++i) {
                const uint4 l = lo[i], h = hi[i];
                const uint4 picked = odd[i] ? make_uint4((l.x >> 16) | (l.y & 0xffff0000u), (l.z >> 16) | (l.w & 0xffff0000u),
                                                         (h.x >> 16) | (h.y & 0xffff0000u), (h.z >> 16) | (h.w & 0xffff0000u))
                                            : make_uint4((l.x & 0xffffu) | (l.y << 16), (l.z & 0xffffu) | (l.w << 16),
                                                         (h.x & 0xffffu) | (h.y << 16), (h.z & 0xffffu) | (h.w << 16));
                rb[i] = shift[i] == 0 ? picked : mm_shift_halfs(picked, shift[i]);
            }
        }
    };
    auto store_step = [&](int k0, auto stage, int image) {
        constexpr int S = decltype(stage)::value;
        const auto &ra = ra_all[S];
        const auto &rb = rb_all[S];
        const auto &rb_meta = rb_meta_all[S];
        [[maybe_unused]] const auto &ra_hi = ra_hi_all[S];
        const bool k_ends_inside = k0 + kMmBK > a.K;   // uniform; only the last step of a K that is no multiple of 64
#pragma unroll
        for (int i = 0; i < A_PIECES; ++i) {
            const int c = tid + THREADS * i;
            uint4 piece = ra[i];
            if constexpr (W_ODD) {
                const unsigned sh = static_cast<unsigned>(a_off[i]) & 2u;   // 2 k0 is a multiple of 4
                piece = make_uint4(__builtin_amdgcn_alignbyte(piece.y, piece.x, sh), __builtin_amdgcn_alignbyte(piece.z, piece.y, sh),
                                   __builtin_amdgcn_alignbyte(piece.w, piece.z, sh), __builtin_amdgcn_alignbyte(ra_hi[i], piece.w, sh));
            }
            if (k_ends_inside) {   // columns from K on are the next row's: keep the first K - k halfs
                const int keep = a.K - (k0 + (c & 7) * 8);
                if (keep < 8) piece = keep <= 0 ? make_uint4(0, 0, 0, 0) : mm_shift_halfs(mm_shift_halfs(piece, 8 - keep), keep - 8);
            }
            *reinterpret_cast<uint4 *>(As + image + (c >> 3) * kMmAPitch + (c & 7) * 8) = piece;
        }
#pragma unroll
        for (int i = 0; i < B_PIECES; ++i) {
            const int c = tid + THREADS * i;
            uint4 piece = rb[i];
            if constexpr (XMODE == kX3x3) {
                if (rb_meta[i] != 0) piece = mm_shift_halfs(piece, rb_meta[i]);
            }
            if constexpr (PRE) {
                const int kk = k0 + c / PCS;
                if (kk < a.K) {   // rows beyond K stay zero
                    const float4 q = *reinterpret_cast<const float4 *>(pre + 4 * kk);
                    const BnChannel ch = {q.x, q.y, q.z, q.w};
                    half8_t v = __builtin_bit_cast(half8_t, piece);
#pragma unroll
                    for (int e = 0; e < 8; e += 2) {
                        const half2_t o = bn_tail_pair<false, false>(half2_t{v[e], v[e + 1]}, half2_t{0, 0}, ch, ch, a.p_relu);
                        v[e] = o[0];
                        v[e + 1] = o[1];
                    }
                    piece = __builtin_bit_cast(uint4, v);
                }
            }
            *reinterpret_cast<uint4 *>(Bs + image + (b_pc >> 3) * (kMmBK * kMmBPitch) + mm_b_offset(c / PCS, b_pc & 7)) = piece;
        }
    };

    float16_t acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][t][r] = 0.0f;

    // operand addresses inside the LDS images (halfs), without the K-step part
    const int h = lane >> 5;
    const int wave_m = wave % MW, wave_n = wave / MW;                     // the wave's place along the channels / the pixels
    const int wave_row = 32 * MT * wave_m;                                // first channel of the wave inside the tile
    const int wave_px = NT == 1 ? wave_n & 1 : 0;                         // first 32-pixel block of the wave ...
    const int wave_sub = NT == 1 ? wave_n >> 1 : wave_n;                  // ... inside this 64-pixel sub-image
    const int w_off = (wave_row + (lane & 31)) * kMmAPitch + 8 * h;       // + 32 mi rows + 16 s
    // transposed read: lane 4q + p of a 16-lane group supplies row q, pixels 4p .. 4p+3 of the group's 16 pixels
    const int g_b = (lane >> 4) & 1, g_q = (lane & 15) >> 2, g_p = lane & 3;
    const int x_row = 8 * h + g_q;                                        // + 16 s (+ 4 for the second read)
    bool has_rows[MT];   // uniform per wave
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) has_rows[mi] = m0 + wave_row + 32 * mi < a.M;

    // one K step of the products: the operands of FD sub-steps of 16 are all requested from LDS before the first
    // product, in straight-line code, so that the reads of a sub-step overlap the products of the one before
    constexpr int FD = (STAGES == 1 && MT == 1) ? 1 : 4;
    auto multiply_step = [&](int image) {
        if (!has_rows[0]) return;   // uniform per wave; rows of a later 32 beyond M are zeros in the A image
#pragma unroll
        for (int s0 = 0; s0 < 4; s0 += FD) {
            half8_t fw[FD][MT], fx[FD][NT];
#pragma unroll
            for (int sd = 0; sd < FD; ++sd) {
                const int s = s0 + sd;
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
                    fw[sd][mi] = *reinterpret_cast<const half8_t *>(As + image + w_off + 32 * mi * kMmAPitch + 16 * s);
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    // pixels 32 t + 16 g_b + 4 g_p ..: piece 4 t + 2 g_b + (g_p >> 1), second half of it for odd g_p
                    const int piece = 4 * (wave_px + t) + 2 * g_b + (g_p >> 1), within = 4 * (g_p & 1);
                    const int k_lo = 16 * s + x_row, k_hi = k_lo + 4;
                    const short4_t lo = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4_ptr)(Bs + image + wave_sub * (kMmBK * kMmBPitch) + mm_b_offset(k_lo, piece) + within));
                    const short4_t hi = __builtin_amdgcn_ds_read_tr16_b64_v4i16((lds_short4_ptr)(Bs + image + wave_sub * (kMmBK * kMmBPitch) + mm_b_offset(k_hi, piece) + within));
                    const short8_t both = {lo[0], lo[1], lo[2], lo[3], hi[0], hi[1], hi[2], hi[3]};
                    fx[sd][t] = __builtin_bit_cast(half8_t, both);
                }
            }
#pragma unroll
            for (int sd = 0; sd < FD; ++sd)
#pragma unroll
                for (int mi = 0; mi < MT; ++mi)
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[mi][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fx[sd][t], fw[sd][mi], acc[mi][t], 0, 0, 0);
        }
    };

    load_step(0, std::integral_constant<int, 0>{});
    if constexpr (DEEP) load_step(kMmBK, std::integral_constant<int, 1>{});
    // the residual's pieces, requested behind the first K steps'
    if (has_res) {
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int m = m0 + e_row + RSTEP * i;
            rres[i] = make_uint4(0, 0, 0, 0);
            if (m < a.M && e_p < a.hw)
                rres[i] = *reinterpret_cast<const uint4 *>(a.res + (img * a.M + m) * static_cast<size_t>(a.hw) + e_p);
        }
    }
    const int n_steps = (a.K + kMmBK - 1) / kMmBK;
    if constexpr (DEEP) {
        // Step t multiplies out of image t & 1.  Behind its products the pieces of step t + 1, requested two steps
        // ago into register stage (t + 1) & 1, are stored into the other image (every wave left that image at the
        // barrier before), the loads of step t + 3 go out into the stage just emptied, and ONE barrier closes the
        // step.  Nothing in the body is conditional - loads beyond K read the operands' first bytes, stores beyond K
        // write zeros into an image nobody reads - so the compiler counts the loads a store may leave in flight.
        if constexpr (PRE) __syncthreads();   // the pending batch-norm's parameters are in LDS
        store_step(0, std::integral_constant<int, 0>{}, 0);
        load_step(2 * kMmBK, std::integral_constant<int, 0>{});
        __syncthreads();
        auto deep_step = [&](int t, auto parity) {
            constexpr int P = decltype(parity)::value;   // t & 1
            multiply_step(P * IMG);
            __builtin_amdgcn_sched_barrier(0);   // the stores wait for loads: not before the products
            store_step((t + 1) * kMmBK, std::integral_constant<int, 1 - P>{}, (1 - P) * IMG);
            load_step((t + 3) * kMmBK, std::integral_constant<int, 1 - P>{});
            __syncthreads();
        };
        int t = 0;
        for (; t + 2 <= n_steps; t += 2) {
            deep_step(t, std::integral_constant<int, 0>{});
            deep_step(t + 1, std::integral_constant<int, 1>{});
        }
        if (t < n_steps) deep_step(t, std::integral_constant<int, 0>{});
    } else {
        for (int t = 0; t < n_steps; ++t) {
            __syncthreads();   // every wave has read the previous step's image
            store_step(t * kMmBK, std::integral_constant<int, 0>{}, 0);
            __syncthreads();
            if (t + 1 < n_steps) load_step((t + 1) * kMmBK, std::integral_constant<int, 0>{});   // in flight from here
            multiply_step(0);
        }
    }

    // The batch-norm constants of the lane's rows (and of the residual's batch-norm), requested here - no row's
    // request behind a branch (rows beyond M ask for the last channel's) - so that they arrive under the two barriers
    // and the writes of the C image instead of costing a round trip to memory per row behind them.
    // (the 3x3 and 7x7 modes, whose sums are long and whose registers are full, fetch them row by row instead)
    constexpr bool RES_MODE = XMODE == kXPlain;   // only the plain 1x1 product is called with a residual
    constexpr bool EARLY = !IS3 && !IS7;
    float tail_p[EARLY ? ROWS : 1][4], tail_q[RES_MODE ? ROWS : 1][4];
    if (EARLY && has_bn) {
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int mi = min(m0 + e_row + RSTEP * i, a.M - 1);
            tail_p[i][0] = a.mean[mi], tail_p[i][1] = a.var[mi], tail_p[i][2] = a.weight[mi], tail_p[i][3] = a.bias[mi];
            if constexpr (RES_MODE) {
                if (res_bn)
                    tail_q[i][0] = a.r_mean[mi], tail_q[i][1] = a.r_var[mi], tail_q[i][2] = a.r_weight[mi], tail_q[i][3] = a.r_bias[mi];
            }
        }
    }
    // accumulators -> fp16 (the rounding of a stored product) -> C image [channel][pixel], four pixels per write
    __syncthreads();
    _Float16 *Cs = lds;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
        if (has_rows[mi]) {
            const int row = wave_row + 32 * mi + (lane & 31);
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int rq = 0; rq < 4; ++rq) {
                    half4_t v;
#pragma unroll
                    for (int e = 0; e < 4; ++e) v[e] = static_cast<_Float16>(acc[mi][t][4 * rq + e]);
                    *reinterpret_cast<half4_t *>(Cs + row * CP + 64 * wave_sub + 32 * (wave_px + t) + 8 * rq + 4 * h) = v;
                }
        }
    __syncthreads();

    if (e_p >= a.hw) return;
    // The rows of a lane as one straight run: C pieces and constants from LDS, the tail, the stores.  Which tail is
    // decided once, outside the rows; inside, only the store depends on the row (rows beyond M hold zeros and
    // whatever constants the array has: computed and dropped).
    _Float16 *yi = a.y + img * a.y_image_stride + e_p;
    auto rows = [&](auto has_res_c, auto res_bn_c, auto silu_c) {
        constexpr bool HR = decltype(has_res_c)::value, RB = decltype(res_bn_c)::value, SI = decltype(silu_c)::value;
        const half2_t lower = bn_lower_bound(a.relu);
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int row = e_row + RSTEP * i;
            const int m = m0 + row;
            const uint4 c = *reinterpret_cast<const uint4 *>(Cs + row * CP + e_pc);
            BnChannel pch;
            if constexpr (EARLY) pch = BnChannel{tail_p[i][0], __builtin_amdgcn_rsqf(fabsf(tail_p[i][1] + a.eps)), tail_p[i][2], tail_p[i][3]};
            else pch = bn_channel_of(a.mean, a.var, a.weight, a.bias, a.eps, static_cast<unsigned>(min(m, a.M - 1)));
            BnChannel qch = pch;
            if constexpr (RB) qch = BnChannel{tail_q[i][0], __builtin_amdgcn_rsqf(fabsf(tail_q[i][1] + a.r_eps)), tail_q[i][2], tail_q[i][3]};
            unsigned d[4] = {c.x, c.y, c.z, c.w};
            const unsigned rr[4] = {rres[i].x, rres[i].y, rres[i].z, rres[i].w};
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const half2_t x2 = __builtin_bit_cast(half2_t, d[e]);
                const half2_t r2 = __builtin_bit_cast(half2_t, rr[e]);
                const half2_t o2 = SI ? bn_tail_pair<HR, RB>(x2, r2, pch, qch, 2) : bn_tail_pair_lb<HR, RB>(x2, r2, pch, qch, lower);
                d[e] = __builtin_bit_cast(unsigned, o2);
            }
            if (m < a.M) *reinterpret_cast<uint4 *>(yi + static_cast<size_t>(m) * a.hw) = make_uint4(d[0], d[1], d[2], d[3]);
        }
    };
    using std::false_type;
    using std::true_type;
    if (!has_bn) {
#pragma unroll
        for (int i = 0; i < ROWS; ++i) {
            const int row = e_row + RSTEP * i;
            const int m = m0 + row;
            const uint4 c = *reinterpret_cast<const uint4 *>(Cs + row * CP + e_pc);
            if (m < a.M) *reinterpret_cast<uint4 *>(yi + static_cast<size_t>(m) * a.hw) = c;
        }
    } else if (a.relu == 2) {
        if (!has_res) rows(false_type{}, false_type{}, true_type{});
        else if (!res_bn) rows(true_type{}, false_type{}, true_type{});
        else rows(true_type{}, true_type{}, true_type{});
    } else {
        if (!has_res) rows(false_type{}, false_type{}, false_type{});
        else if (!res_bn) rows(true_type{}, false_type{}, false_type{});
        else rows(true_type{}, true_type{}, false_type{});
    }
}

// ---- 3x3, stride 1, second form: the patch of x under a tile staged ONCE per 64 input channels --------------------
//
// The form above stages the X tile of every K step from memory, so each pixel of x crosses the L2 -> compute unit
// path nine times (once per tap), and the weights go through LDS although every wave reads rows of its own.  Here
//   * a tile is whole output rows (NPX pixels = rows x width), and the patch of x under it - (rows + 2 dil) x
//     (width + 2 dil) pixels, zeros where it leaves the image - is staged once per block of 64 input channels,
//     PIXEL-major: patch[pixel][64 channels + 8] halfs (144-byte rows).  A tap's view of the patch is the same
//     image moved by (dy * pitch + dx) whole rows, so the A operand of the product (32 pixels x 16 channels) is ONE
//     aligned ds_read_b128 per lane for every tap - nine products out of one staging;
//   * the weights do not touch LDS: a lane's B operand (its channel's row of W, 8 consecutive K) is a 16-byte buffer
//     load straight into the registers the product reads, requested three taps ahead;
//   * four waves per workgroup, (32 MT) channels x (32 NT) pixels each, two workgroups per compute unit.
// W is [Cout][9][Cin] as for the form above; Cin a multiple of 64 (no block of input channels ends inside a K step:
// the pruned layers of a sweep candidate, whose channel counts are arbitrary, stay with the form above).
struct C3Args {
    const _Float16 *x, *w;
    _Float16 *y;
    const float *mean, *var, *weight, *bias;   // null mean: the plain product
    float eps;
    int relu;
    int M, cin, h, w_px, dil;
    int rows;         // output rows per tile
    int prow;         // pixels per patch row: width + 2 dil
    int prows;        // patch rows: rows + 2 dil
    int buf_halfs;    // halfs of one patch buffer (a multiple of 8)
    int buffers;      // 2 when there is more than one block of input channels
    int batches;      // pieces of a patch per lane and tap (1 or 2)
    int tiles_m, tiles_y;
    unsigned total_tiles;
};

constexpr int kC3Pitch = 72;   // halfs per patch pixel

template <int WM, int WN, int MT, int NT, int P>   // P: pieces of the next patch a lane stages per tap (9 P cover a patch)
__global__ __launch_bounds__(256, 2) void conv3x3_halo_mfma_kernel(C3Args a) {
    constexpr int BM = 32 * MT * WM, NPX = 32 * NT * WN, THREADS = 256, PCS = NPX / 8;
    constexpr int ROWS = BM * PCS / THREADS, RSTEP = THREADS / PCS, CP = NPX + 8;
    constexpr int XP = kC3Pitch;
    static_assert(WM * WN == 4, "four waves");
    extern __shared__ __attribute__((aligned(16))) _Float16 lds[];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, h = lane >> 5;
    const unsigned tile = mm_tile_of_block(blockIdx.x, a.total_tiles);
    const int mt = static_cast<int>(tile % static_cast<unsigned>(a.tiles_m));
    const unsigned rest = tile / static_cast<unsigned>(a.tiles_m);
    const int ty = static_cast<int>(rest % static_cast<unsigned>(a.tiles_y));
    const size_t img = rest / static_cast<unsigned>(a.tiles_y);
    const int m0 = mt * BM, y0 = ty * a.rows;
    const int W = a.w_px, plane = a.h * W, G = W >> 3;

    for (int i = tid * 8; i < a.buffers * a.buf_halfs; i += THREADS * 8) *reinterpret_cast<uint4 *>(lds + i) = make_uint4(0, 0, 0, 0);

    const __amdgpu_buffer_rsrc_t w_rsrc =
        __builtin_amdgcn_make_buffer_rsrc(const_cast<_Float16 *>(a.w), 0, (a.M * 9 * a.cin * 2 + 3) & ~3, 0x00020000);
    const __amdgpu_buffer_rsrc_t x_rsrc = __builtin_amdgcn_make_buffer_rsrc(
        const_cast<_Float16 *>(a.x + img * static_cast<size_t>(a.cin) * plane), 0, a.cin * plane * 2, 0x00020000);
    auto as_uint4 = [](u32x4_t v) { return make_uint4(v.x, v.y, v.z, v.w); };

    // Piece q of a patch: 8 consecutive pixels of one channel.  Eight neighbouring lanes take the same 8 pixels of 8
    // consecutive channels (q & 7), then the pieces along the row, the channel octets, the patch rows: a wave's request
    // is 8 channels x 128 contiguous bytes.  The eight lanes transpose their 8 x 8 halfs among themselves, after which
    // a lane holds ONE pixel's 8 channels - a 16-byte row piece of the pixel-major patch, one ds_write_b128.
    // off: byte offset in the image of x (outside: reads zeros); dst: where the lane's transposed piece goes (-1:
    // nowhere; rows outside the image keep the zeros the patch was cleared to).
    // (the width is a power of two of at most 32 pieces, so G divides 32: a lane's j-th piece keeps its place g in
    // the row and moves on by 32 / G octet-rows)
    const int px_of_lane = 4 * ((lane >> 1) & 1) + 2 * ((lane >> 2) & 1) + (lane & 1);   // the pixel the transposition leaves with the lane
    const int g_shift = __builtin_ctz(G);
    const int pc_g = (tid >> 3) & (G - 1), pc_t1 = (tid >> 3) >> g_shift, pc_dt = 32 >> g_shift;
    const int pc_off = ((tid & 7) * plane + (y0 - a.dil) * W + 8 * pc_g) * 2;
    const int pc_dst = (8 * pc_g + px_of_lane + a.dil) * XP;
    auto piece = [&](int j, int cc, bool wanted, int &off, int &dst) {
        const int t1 = pc_t1 + j * pc_dt;
        const int r = t1 >> 3, c8 = t1 & 7;
        const bool in = wanted && r < a.prows && static_cast<unsigned>(y0 - a.dil + r) < static_cast<unsigned>(a.h);
        off = in ? pc_off + ((cc * 8 + c8) * 8 * plane + r * W) * 2 : kMmOutside;   // channels beyond cin: outside the buffer, zeros
        dst = in ? pc_dst + r * (a.prow * XP) + 8 * c8 : -1;
    };
    const unsigned sel16 = (lane & 1) ? 0x03020706u : 0x05040100u;
    auto store_piece = [&](uint4 v, int dst, _Float16 *buf) {
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
        {   // lanes l, l ^ 4: 2 x 2 blocks of dword pairs
            const bool hi = (lane & 4) != 0;
            // lane l ^ 4 of a row of 16: four lanes up for the lanes of banks 0 and 2, four lanes down for banks 1 and 3
            const unsigned s0 = hi ? d0 : d2, s1 = hi ? d1 : d3;
            unsigned r0 = __builtin_amdgcn_update_dpp(0u, s0, 0x104 /* row_shl:4 */, 0xF, 0x5, false);
            r0 = __builtin_amdgcn_update_dpp(r0, s0, 0x114 /* row_shr:4 */, 0xF, 0xA, false);
            unsigned r1 = __builtin_amdgcn_update_dpp(0u, s1, 0x104, 0xF, 0x5, false);
            r1 = __builtin_amdgcn_update_dpp(r1, s1, 0x114, 0xF, 0xA, false);
            const unsigned n0 = hi ? r0 : d0, n1 = hi ? r1 : d1, n2 = hi ? d2 : r0, n3 = hi ? d3 : r1;
            d0 = n0, d1 = n1, d2 = n2, d3 = n3;
        }
        if (dst >= 0) *reinterpret_cast<uint4 *>(buf + dst) = make_uint4(d0, d1, d2, d3);
    };

    // the lane's rows of W: channel m0 + 32 MT wave_m + 32 mi + (lane & 31), K from 8 h on
    const int wave_m = wave % WM, wave_n = wave / WM;
    int w_row[MT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int m = m0 + 32 * (MT * wave_m + mi) + (lane & 31);
        w_row[mi] = m < a.M ? (m * 9 * a.cin + 8 * h) * 2 : kMmOutside;
    }
    constexpr int RD = 3;   // taps of W in flight
    uint4 fw[RD][4][MT];
    auto load_w = [&](int cc, int tap, int S) {   // S: a constant once the tap loop is unrolled
        const int k = (tap * a.cin + cc * 64) * 2;
#pragma unroll
        for (int s = 0; s < 4; ++s)
#pragma unroll
            for (int mi = 0; mi < MT; ++mi)
                fw[S][s][mi] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(w_rsrc, w_row[mi] + k + 32 * s, 0, 0));
    };

    // the lane's pixels: block b = NT wave_n + t, pixel 32 b + (lane & 31) of the tile; its place in the patch at tap (0, 0)
    int px_base[NT];
#pragma unroll
    for (int t = 0; t < NT; ++t) {
        const int p = 32 * (NT * wave_n + t) + (lane & 31);
        const int py = p / W, pxx = p - py * W;
        px_base[t] = ((py + a.dil) * a.prow + pxx + a.dil) * XP + 8 * h;
    }

    float16_t acc[MT][NT];
#pragma unroll
    for (int mi = 0; mi < MT; ++mi)
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int r = 0; r < 16; ++r) acc[mi][t][r] = 0.0f;

    const int n_chunks = (a.cin + 63) >> 6;
    load_w(0, 0, 0);
    load_w(0, 1, 1);
    load_w(0, 2, 2);
    __syncthreads();   // the cleared patches
    {   // the first block's patch: every piece requested before the first is stored
        uint4 v[9 * P];
        int dst[9 * P];
#pragma unroll
        for (int i = 0; i < 9 * P; ++i) {
            int off;
            piece(i, 0, true, off, dst[i]);
            v[i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(x_rsrc, off, 0, 0));
        }
#pragma unroll
        for (int i = 0; i < 9 * P; ++i) store_piece(v[i], dst[i], lds);
    }
    __syncthreads();

    // the next block's patch: 2 P pieces requested per tap in taps 0 .. 4, stored three taps later (taps 3 .. 7), so
    // that nothing of it is still on its way when the block ends
    constexpr int XD = 3, XB = 2 * P;
    uint4 xs[XD + 1][XB];
    int xs_dst[XD + 1][XB];
    for (int cc = 0; cc < n_chunks; ++cc) {
        _Float16 *cur = lds + (a.buffers == 2 ? (cc & 1) * a.buf_halfs : 0);
        _Float16 *nxt = lds + (a.buffers == 2 ? ((cc + 1) & 1) * a.buf_halfs : 0);
        const bool more = cc + 1 < n_chunks;
        half8_t fx[2][NT];
        {
            const int toff0 = (-a.dil * a.prow - a.dil) * XP;
#pragma unroll
            for (int t = 0; t < NT; ++t) fx[0][t] = *reinterpret_cast<const half8_t *>(cur + px_base[t] + toff0);
        }
#pragma unroll
        for (int tap = 0; tap < 9; ++tap) {
            // (no load behind a branch: beyond the last block and beyond the patch the requests lie outside the buffer)
            if (tap < 5) {
#pragma unroll
                for (int i = 0; i < XB; ++i) {
                    int off;
                    piece(XB * tap + i, cc + 1, more && XB * tap + i < 9 * P, off, xs_dst[tap % (XD + 1)][i]);
                    xs[tap % (XD + 1)][i] = as_uint4(__builtin_amdgcn_raw_buffer_load_b128(x_rsrc, off, 0, 0));
                }
            }
            if (tap >= XD && tap < 5 + XD) {
#pragma unroll
                for (int i = 0; i < XB; ++i) store_piece(xs[(tap - XD) % (XD + 1)][i], xs_dst[(tap - XD) % (XD + 1)][i], nxt);
            }
            // the operands of sub-step s + 1 (of the next tap behind the fourth) are requested from LDS before the
            // products of sub-step s are issued: a wave alone on its SIMD hides the reads behind its own products
#pragma unroll
            for (int s = 0; s < 4; ++s) {
                const int nx = 4 * tap + s + 1;   // the sub-step after this one, counted through the block
                if (nx < 36) {
                    const int ntap = nx >> 2, ns = nx & 3;
                    const int ntoff = ((ntap / 3 - 1) * a.dil * a.prow + (ntap % 3 - 1) * a.dil) * XP;
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        fx[nx & 1][t] = *reinterpret_cast<const half8_t *>(cur + px_base[t] + ntoff + 16 * ns);
                }
#pragma unroll
                for (int mi = 0; mi < MT; ++mi) {
                    const uint4 wv = fw[tap % RD][s][mi];
                    const half8_t wh = __builtin_bit_cast(half8_t, wv);
#pragma unroll
                    for (int t = 0; t < NT; ++t)
                        acc[mi][t] = __builtin_amdgcn_mfma_f32_32x32x16_f16(fx[(4 * tap + s) & 1][t], wh, acc[mi][t], 0, 0, 0);
                }
            }
            // W of three taps on, into the registers just read
            if (tap + RD < 9) load_w(cc, tap + RD, tap % RD);
            else load_w(cc + 1, tap + RD - 9, tap % RD);
        }
        __syncthreads();
    }

    // accumulators -> fp16 -> C image [channel][pixel] -> row-major 16-byte pieces -> batch-norm tail -> memory
    _Float16 *Cs = lds;
#pragma unroll
    for (int mi = 0; mi < MT; ++mi) {
        const int row = 32 * (MT * wave_m + mi) + (lane & 31);
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int rq = 0; rq < 4; ++rq) {
                half4_t v;
#pragma unroll
                for (int e = 0; e < 4; ++e) v[e] = static_cast<_Float16>(acc[mi][t][4 * rq + e]);
                *reinterpret_cast<half4_t *>(Cs + row * CP + 32 * (NT * wave_n + t) + 8 * rq + 4 * h) = v;
            }
    }
    __syncthreads();
    const int e_pc = (tid % PCS) * 8, e_row = tid / PCS;
    const int e_p = y0 * W + e_pc;   // the piece's first pixel in the image
    if (e_p >= plane) return;        // rows of the last tile beyond the image (pieces are whole: width % 8 == 0)
    const bool has_bn = a.mean != nullptr;
#pragma unroll
    for (int i = 0; i < ROWS; ++i) {
        const int row = e_row + RSTEP * i;
        const int m = m0 + row;
        if (m >= a.M) break;
        half8_t v = *reinterpret_cast<const half8_t *>(Cs + row * CP + e_pc);
        if (has_bn) {
            const BnChannel pch = bn_channel_of(a.mean, a.var, a.weight, a.bias, a.eps, m);
#pragma unroll
            for (int e = 0; e < 8; e += 2) {
                const half2_t o = bn_tail_pair<false, false>(half2_t{v[e], v[e + 1]}, half2_t{0, 0}, pch, pch, a.relu);
                v[e] = o[0];
                v[e + 1] = o[1];
            }
        }
        *reinterpret_cast<half8_t *>(a.y + (img * a.M + m) * static_cast<size_t>(plane) + e_p) = v;
    }
}

SPICE_ENV_KNOB(mm_stages_knob, "SPICE_CONV1X1_STAGES", 0)
SPICE_ENV_KNOB(mm_mt_knob, "SPICE_CONV1X1_MT", 0)
SPICE_ENV_KNOB(mm_m64_knob, "SPICE_CONV_M64", 1)
SPICE_ENV_KNOB(mm_m64_min_k_knob, "SPICE_CONV_M64_MIN_K", 256)
SPICE_ENV_KNOB(mm_w16_knob, "SPICE_CONV_W16", 1)
SPICE_ENV_KNOB(mm_chunked_knob, "SPICE_CONV3X3_CHUNKED", 1)

}  // namespace spice

// x_h, x_w: the plane of x when the convolution has stride 2 (then hw = (x_h / 2) * (x_w / 2)), else 0
// One launch of a form of the kernel: its LDS images as dynamic shared memory (the 128-pixel form's two images
// take 68 KB, above what a kernel may have without asking).
template <int STAGES, int MT, int XMODE = spice::kXPlain, bool PRE = false, bool W_ODD = false>
static void mm_launch(const spice::MmArgs &a, hipStream_t st) {
    using namespace spice;
    constexpr int bytes = mm_lds_bytes(STAGES, MT);
    const auto kernel = conv1x1_bn_act_mfma_kernel<STAGES, MT, XMODE, PRE, W_ODD>;
    // (beside the images a workgroup has up to 8 KB of static LDS: a pending batch-norm's constants)
    if constexpr (bytes + 12 * 1024 >= 64 * 1024)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL(kernel, dim3(a.total_tiles), dim3(mm_threads(STAGES)), bytes, st, a);
}

constexpr int32_t kStem7x7 = -7;   // conv1x1_launch's c3_stride for the 7x7 stride-2 mode

static int conv1x1_launch(const void *x, const void *weight, void *y, int64_t n, int64_t cin, int64_t hw,
                          int64_t cout, const float *mean, const float *var, const float *bn_weight,
                          const float *bn_bias, float eps, const void *residual, const float *r_mean,
                          const float *r_var, const float *r_weight, const float *r_bias, float r_eps,
                          int32_t relu, int64_t y_image_stride, int32_t x_h, int32_t x_w, const float *p_mean,
                          const float *p_var, const float *p_weight, const float *p_bias, float p_eps, int32_t p_relu,
                          int32_t c3_stride, int32_t c3_dil, void *stream) {
    // c3_stride > 0: a 3x3 convolution of that stride and dilation c3_dil over x [n, cin, x_h, x_w]; weight is [cout][9][cin]
    // c3_stride == kStem7x7: a 7x7 convolution, stride 2, padding 3; weight is [cout][cin][7][7] as it lies
    using namespace spice;
    const bool c3 = c3_stride > 0, c7 = c3_stride == kStem7x7;
    const int64_t kdim = c7 ? 49 * cin : (c3 ? 9 * cin : cin);   // length of the sums
    if (n < 0 || cin <= 0 || hw < 0 || cout <= 0) return SPICE_ERR_ARG;
    if (n == 0 || hw == 0) return SPICE_OK;
    if (x == nullptr || weight == nullptr || y == nullptr) return SPICE_ERR_ARG;
    if (mean != nullptr && (var == nullptr || bn_weight == nullptr || bn_bias == nullptr)) return SPICE_ERR_ARG;
    if (mean == nullptr && (residual != nullptr || r_mean != nullptr)) return SPICE_ERR_ARG;
    if (r_mean != nullptr && (residual == nullptr || r_var == nullptr || r_weight == nullptr || r_bias == nullptr))
        return SPICE_ERR_ARG;
    if (p_mean != nullptr && (p_var == nullptr || p_weight == nullptr || p_bias == nullptr)) return SPICE_ERR_ARG;
    if (p_mean != nullptr && (cin > kMmPreMaxK || x_h > 0 || c3)) return SPICE_ERR_UNSUPPORTED;
    // 16-byte pieces of whole rows of X, R and Y: the plane size must be a multiple of 8 halfs
    if (y_image_stride == 0) y_image_stride = cout * hw;   // dense
    if (y_image_stride < cout * hw) return SPICE_ERR_ARG;
    if ((hw & 7) != 0 || (y_image_stride & 7) != 0 || !aligned16(x) || !aligned16(y) || (residual != nullptr && !aligned16(residual)))
        return SPICE_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(weight) & 3u) != 0) return SPICE_ERR_UNSUPPORTED;   // the base of a buffer descriptor
    if (cin > (1 << 20) || cout > (1 << 20) || hw > (1 << 28)) return SPICE_ERR_UNSUPPORTED;
    {   // the kernel addresses W and one image of x with 32-bit byte offsets
        const int64_t limit = (int64_t{1} << 30) - (int64_t{1} << 20);
        const int64_t x_plane = x_h > 0 ? static_cast<int64_t>(x_h) * x_w : hw;
        if (cout * kdim * 2 > limit || cin * x_plane * 2 > limit) return SPICE_ERR_UNSUPPORTED;
    }
    // 256-channel tiles (two product tiles per wave: half the reads of X and of the LDS images per product) are an
    // option (SPICE_CONV1X1_MT=2), not the default: at over 200 registers a compute unit holds two workgroups instead
    // of four, which on MI355X costs more than the traffic saves on every pair of the DeepLabV3+-ResNet-50 forward
    // but one (tools/conv1x1_bench.py: 958 us per forward against 850 us)
    const bool wide = mm_mt_knob() >= 2 && cout > kMmBM && x_h == 0 && p_mean == nullptr && !c3 && (kdim & 1) == 0;
    const int64_t tiles_m = ceil_div(cout, wide ? 2 * kMmBM : kMmBM);
    // Which form: deep K on a grid that leaves a compute unit one or two workgroups -> eight waves, two LDS images
    // (3); deep K on a grid with at least two 128-pixel tiles per compute unit -> the same with 128-pixel tiles,
    // a third less traffic from L2 per product (4); else four waves, one image, four workgroups per compute unit (1).
    // (256-channel tiles in the eight-wave form - one workgroup per compute unit where 128-channel tiles give two -
    // take the same time: 64.3 against 65.3 us for the 512-channel dilated 3x3 convolution; not kept.)
    const long forced = mm_stages_knob();
    const int64_t cus = device_cu_count();
    const int64_t tiles64 = tiles_m * ceil_div(hw, 64) * n, tiles128 = tiles_m * ceil_div(hw, 128) * n;
    const bool has_wide_pixels = p_mean == nullptr && !(x_h > 0 && !c3) && !(c3 && c3_stride == 2);   // forms with a 128-pixel variant
    int form = 1;
    if (!wide && (kdim & 1) == 0 && !c7) {
        if (forced > 0) form = forced == 6 ? 6 : (forced >= 4 ? 4 : (forced >= 3 ? 3 : 1));
        else if (kdim >= 4 * kMmBK && tiles64 <= 2 * cus)
            // a plain or paired 1x1 product with K >= 512 gains from the wider tile as soon as every compute unit
            // gets one (2048 -> 512 on 8,192 pixels: 25.8 against 28.5 us); the 3x3 mode does not (24.1 against 22.4)
            form = (!c3 && kdim >= 8 * kMmBK && tiles128 >= cus && (hw & 127) == 0) ? 4 : 3;
        else if (kdim >= 4 * kMmBK && tiles128 >= 2 * cus && (hw & 127) == 0) form = 4;
    }
    // One round of 128-pixel tiles (between three quarters of the compute units and all of them): sixteen waves on the
    // tile (form 6), a 32 x 32 product each - the traffic of the 128-pixel form with the waves of two 64-pixel
    // workgroups to hide its waits.  58.6 against 63.1 us for the dilated 512 -> 512 3x3 convolution, 20.5 against
    // 22.4 us for 128 -> 128 on 32,768 pixels; level with form 4 on the 1x1 products it takes from it; slower wherever
    // the tiles make half a round or several (SPICE_CONV_W16=0 switches it off).
    if (forced == 0 && mm_w16_knob() != 0 && form >= 3 && cout > 64 && (hw & 127) == 0 && has_wide_pixels && !c7 &&
        (kdim & 1) == 0 && 4 * tiles128 >= 3 * cus && tiles128 <= cus)
        form = 6;
    if (form == 6 && !has_wide_pixels) form = 4;
    if (form == 4 && !has_wide_pixels) form = forced > 0 ? 3 : 1;
    // A layer of at most 64 output channels fills half a 128-channel tile: half the waves of a workgroup have no
    // rows.  Form 5 gives it 64-channel x 128-pixel tiles with all eight waves at work (a 32 x 32 product each):
    // 31.3 against 36.7 us for layer1's 3x3 convolutions, level for the 1x1 products of K = 256, slower for one K step
    // (10.5 against 10.2 us) and for the stem's 7x7 mode (67.3 against 63.2 us), which therefore stay (SPICE_CONV_M64=2
    // sends the stem there too).
    const long m64 = mm_m64_knob();
    if (m64 != 0 && cout <= 64 && (hw & 127) == 0 && kdim >= mm_m64_min_k_knob() && (has_wide_pixels || c7) &&
        (c7 ? m64 >= 2 : (kdim & 1) == 0) && (forced == 0 || forced == 5))
        form = 5;
    const bool deep = form >= 3;
    const int64_t tiles_m_used = form == 5 ? ceil_div(cout, 64) : tiles_m;
    const int64_t tiles_n = ceil_div(hw, form >= 4 ? 128 : 64);
    if (tiles_m_used * tiles_n > INT32_MAX / n) return SPICE_ERR_UNSUPPORTED;
    MmArgs a;
    a.x = static_cast<const _Float16 *>(x);
    a.w = static_cast<const _Float16 *>(weight);
    a.y = static_cast<_Float16 *>(y);
    a.y_image_stride = y_image_stride;
    a.res = static_cast<const _Float16 *>(residual);
    a.mean = mean;
    a.var = var;
    a.weight = bn_weight;
    a.bias = bn_bias;
    a.r_mean = r_mean;
    a.r_var = r_var;
    a.r_weight = r_weight;
    a.r_bias = r_bias;
    a.p_mean = p_mean;
    a.p_var = p_var;
    a.p_weight = p_weight;
    a.p_bias = p_bias;
    a.eps = eps;
    a.r_eps = r_eps;
    a.p_eps = p_eps;
    a.p_relu = p_relu ? 1 : 0;
    a.M = static_cast<int>(cout);
    a.K = static_cast<int>(kdim);
    a.hw = static_cast<int>(hw);
    const bool s2 = x_h > 0 && !c3 && !c7;
    a.x_hw = x_h > 0 ? x_h * x_w : 0;
    a.x_w = x_w;
    a.y_w = (s2 || c7) ? x_w / 2 : (c3 ? x_w / c3_stride : 0);
    a.x_h = x_h;
    a.cin = static_cast<int>(cin);
    a.stride = c3_stride;
    a.dil = c3_dil;
    a.tiles_m = static_cast<int>(tiles_m_used);
    a.tiles_n = static_cast<int>(tiles_n);
    a.total_tiles = static_cast<unsigned>(tiles_m_used * tiles_n * n);
    a.relu = tail_code(relu);
    a.chunked = (c3 && c3_stride == 1 && cin % kMmBK == 0 && mm_chunked_knob() != 0) ? 1 : 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (c7) {
        if (form == 5 && (kdim & 1)) mm_launch<5, 1, kX7x7S2, false, true>(a, st);
        else if (form == 5) mm_launch<5, 1, kX7x7S2>(a, st);
        else if (kdim & 1) mm_launch<1, 1, kX7x7S2, false, true>(a, st);
        else mm_launch<1, 1, kX7x7S2>(a, st);
    } else if (kdim & 1) {   // pruned to an odd number of input channels: the forms that take rows of W from the middle of a dword
        if (c3 && c3_stride == 2) mm_launch<1, 1, kX3x3S2, false, true>(a, st);
        else if (c3) mm_launch<1, 1, kX3x3, false, true>(a, st);
        else if (p_mean != nullptr) mm_launch<1, 1, kXPlain, true, true>(a, st);
        else if (s2) mm_launch<1, 1, kXStride2, false, true>(a, st);
        else mm_launch<1, 1, kXPlain, false, true>(a, st);
    } else if (c3 && c3_stride == 2) {
        if (deep) mm_launch<3, 1, kX3x3S2>(a, st);
        else mm_launch<1, 1, kX3x3S2>(a, st);
    } else if (c3) {
        if (form == 6) mm_launch<6, 1, kX3x3>(a, st);
        else if (form == 5) mm_launch<5, 1, kX3x3>(a, st);
        else if (form == 4) mm_launch<4, 1, kX3x3>(a, st);
        else if (deep) mm_launch<3, 1, kX3x3>(a, st);
        else mm_launch<1, 1, kX3x3>(a, st);
    } else if (p_mean != nullptr) {
        if (deep) mm_launch<3, 1, kXPlain, true>(a, st);
        else mm_launch<1, 1, kXPlain, true>(a, st);
    } else if (s2) {
        if (deep) mm_launch<3, 1, kXStride2>(a, st);
        else mm_launch<1, 1, kXStride2>(a, st);
    } else if (wide) {
        mm_launch<1, 2>(a, st);
    } else if (form == 6) {
        mm_launch<6, 1>(a, st);
    } else if (form == 5) {
        mm_launch<5, 1>(a, st);
    } else if (form == 4) {
        mm_launch<4, 1>(a, st);
    } else if (deep) {
        mm_launch<3, 1>(a, st);
    } else {
        mm_launch<1, 1>(a, st);
    }
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}

extern "C" int spice_conv1x1_bn_act_f16(const void *x, const void *weight, void *y, int64_t n, int64_t cin, int64_t hw,
                                        int64_t cout, const float *mean, const float *var, const float *bn_weight,
                                        const float *bn_bias, float eps, const void *residual, const float *r_mean,
                                        const float *r_var, const float *r_weight, const float *r_bias, float r_eps,
                                        int32_t relu, int64_t y_image_stride, const float *p_mean, const float *p_var,
                                        const float *p_weight, const float *p_bias, float p_eps, int32_t p_relu,
                                        void *stream) {
    return conv1x1_launch(x, weight, y, n, cin, hw, cout, mean, var, bn_weight, bn_bias, eps, residual, r_mean, r_var,
                          r_weight, r_bias, r_eps, relu, y_image_stride, 0, 0, p_mean, p_var, p_weight, p_bias, p_eps, p_relu,
                          0, 0, stream);
}

extern "C" int spice_conv1x1_stride2_f16(const void *x, const void *weight, void *y, int64_t n, int64_t cin, int32_t h,
                                         int32_t w, int64_t cout, void *stream) {
    if (h <= 0 || w <= 0) return SPICE_ERR_ARG;
    // whole 2x2 cells, and rows of the product that are whole 16-byte pieces
    if ((h & 1) || (w & 15)) return SPICE_ERR_UNSUPPORTED;
    return conv1x1_launch(x, weight, y, n, cin, static_cast<int64_t>(h / 2) * (w / 2), cout, nullptr, nullptr, nullptr,
                          nullptr, 0.0f, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0f, 0, 0, h, w, nullptr, nullptr, nullptr,
                          nullptr, 0.0f, 0, 0, 0, stream);
}

extern "C" int spice_conv7x7_stride2_f16(const void *x, const void *weight, void *y, int64_t n, int64_t cin, int32_t h,
                                         int32_t w, int64_t cout, void *stream) {
    if (h <= 0 || w <= 0) return SPICE_ERR_ARG;
    // whole 2x2 cells; rows of the product that are whole 16-byte pieces; a 16-column window inside a row
    if ((h & 1) || (w & 15)) return SPICE_ERR_UNSUPPORTED;
    if (cin > 1024) return SPICE_ERR_UNSUPPORTED;
    return conv1x1_launch(x, weight, y, n, cin, static_cast<int64_t>(h / 2) * (w / 2), cout, nullptr, nullptr, nullptr,
                          nullptr, 0.0f, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0f, 0, 0, h, w, nullptr, nullptr, nullptr,
                          nullptr, 0.0f, 0, kStem7x7, 0, stream);
}

namespace spice {
int conv3x3_narrow_launch(const void *x, const void *weight9, void *y, int64_t n, int64_t cin, int32_t h, int32_t w,
                          int64_t cout, const float *mean, const float *var, const float *bn_weight,
                          const float *bn_bias, float eps, int32_t relu, void *stream);   // conv3x3_narrow.hip
}

SPICE_ENV_KNOB(c3_halo_knob, "SPICE_CONV3X3_HALO", 1)        // 0 = the 3x3 convolutions stay with the first form
SPICE_ENV_KNOB(c3_halo_form_knob, "SPICE_CONV3X3_HALO_FORM", 0)   // 2 | 3 = that tile shape wherever it applies (A/B measurements)

template <int WM, int WN, int MT, int NT, int P>
static void c3_launch_p(const spice::C3Args &a, size_t lds_bytes, hipStream_t st) {
    using namespace spice;
    const auto kernel = conv3x3_halo_mfma_kernel<WM, WN, MT, NT, P>;
    if (lds_bytes > 64 * 1024)   // (the host bounds it at 150 KB)
        (void)hipFuncSetAttribute(reinterpret_cast<const void *>(kernel), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipLaunchKernelGGL(kernel, dim3(a.total_tiles), dim3(256), lds_bytes, st, a);
}

template <int WM, int WN, int MT, int NT>
static void c3_launch(const spice::C3Args &a, size_t lds_bytes, hipStream_t st) {
    if (a.batches == 1) c3_launch_p<WM, WN, MT, NT, 1>(a, lds_bytes, st);
    else c3_launch_p<WM, WN, MT, NT, 2>(a, lds_bytes, st);
}

// The second 3x3 form where it applies (stride 1, an even number of input channels, a width that divides the tile's
// pixels, a patch the lanes can stage within nine taps and two of which fit the LDS); SPICE_ERR_UNSUPPORTED otherwise,
// for the caller to take the first form.
static int conv3x3_halo_launch(const void *x, const void *weight9, void *y, int64_t n, int64_t cin, int32_t h, int32_t w,
                               int64_t cout, int32_t dilation, const float *mean, const float *var,
                               const float *bn_weight, const float *bn_bias, float eps, int32_t relu, void *stream) {
    using namespace spice;
    if (c3_halo_knob() == 0) return SPICE_ERR_UNSUPPORTED;
    if (n <= 0 || cin <= 0 || cout <= 0 || x == nullptr || weight9 == nullptr || y == nullptr) return SPICE_ERR_UNSUPPORTED;
    if (mean != nullptr && (var == nullptr || bn_weight == nullptr || bn_bias == nullptr)) return SPICE_ERR_UNSUPPORTED;
    if ((cin & 63) || (w & 7) || !aligned16(x) || !aligned16(y) || (reinterpret_cast<uintptr_t>(weight9) & 3u)) return SPICE_ERR_UNSUPPORTED;
    const int64_t limit = (int64_t{1} << 30) - (int64_t{1} << 20);
    if (cout * 9 * cin * 2 > limit || cin * (int64_t)h * w * 2 > limit || cin > (1 << 16) || cout > (1 << 20)) return SPICE_ERR_UNSUPPORTED;
    // Tile shapes: 2 = 128 channels x 128 pixels (a wave 32 x 128), 3 = 64 x 128 (a wave 32 x 64; layers of at most 64
    // channels).  Measured on MI355X against the first form, batch 8 (tools/conv3x3_bench.py, us): 512 -> 512 at
    // 32 x 32, dilation 2: 53.8 against 62.1; 128 -> 128 at 64 x 64: 20.2 against 21.3; 256 -> 256 at 32 x 32: 27.0
    // against 25.0; 64 -> 64 at 128 x 128: 38.7 against 30.8.  It is taken by itself where it wins clearly - K of
    // 4608 and more - and anywhere it applies with SPICE_CONV3X3_HALO_FORM=2|3.
    const int forced = (int)c3_halo_form_knob();
    int form = 0;
    if (forced == 2 || forced == 3) form = forced;
    else if (cin >= 512 && cout > 64) form = 2;
    if (form == 0 || 128 % w != 0) return SPICE_ERR_UNSUPPORTED;
    const int bm = form == 3 ? 64 : 128;
    const int npx = 128;
    C3Args a;
    a.x = static_cast<const _Float16 *>(x);
    a.w = static_cast<const _Float16 *>(weight9);
    a.y = static_cast<_Float16 *>(y);
    a.mean = mean;
    a.var = var;
    a.weight = bn_weight;
    a.bias = bn_bias;
    a.eps = eps;
    a.relu = tail_code(relu);
    a.M = (int)cout;
    a.cin = (int)cin;
    a.h = h;
    a.w_px = w;
    a.dil = dilation;
    a.rows = npx / w;
    a.prow = w + 2 * dilation;
    a.prows = a.rows + 2 * dilation;
    a.buf_halfs = (a.prows * a.prow * kC3Pitch + 7) & ~7;
    const int64_t chunks = ceil_div(cin, 64);
    a.buffers = chunks > 1 ? 2 : 1;
    if ((w & (w - 1)) != 0 || w > 256) return SPICE_ERR_UNSUPPORTED;   // a power of two: the lanes' pieces step through the patch in whole rows
    const int64_t pieces = (int64_t)64 * a.prows * (w / 8);
    a.batches = (int)ceil_div(pieces, 9 * 256);   // pieces per lane and tap
    if (a.batches > 2) return SPICE_ERR_UNSUPPORTED;
    size_t lds_bytes = (size_t)a.buffers * a.buf_halfs * 2;
    const size_t c_bytes = (size_t)bm * (npx + 8) * 2;
    if (lds_bytes < c_bytes) lds_bytes = c_bytes;
    if (lds_bytes > 150 * 1024) return SPICE_ERR_UNSUPPORTED;
    a.tiles_m = (int)ceil_div(cout, bm);
    a.tiles_y = (int)ceil_div(h, a.rows);
    const int64_t total = (int64_t)a.tiles_m * a.tiles_y * n;
    if (total > INT32_MAX) return SPICE_ERR_UNSUPPORTED;
    a.total_tiles = (unsigned)total;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (form == 3) c3_launch<2, 2, 1, 2>(a, lds_bytes, st);
    else c3_launch<4, 1, 1, 4>(a, lds_bytes, st);
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}

extern "C" int spice_conv3x3_bn_act_f16(const void *x, const void *weight9, void *y, int64_t n, int64_t cin, int32_t h,
                                        int32_t w, int64_t cout, int32_t stride, int32_t dilation, const float *mean,
                                        const float *var, const float *bn_weight, const float *bn_bias, float eps,
                                        int32_t relu, void *stream) {
    if (h <= 0 || w <= 0) return SPICE_ERR_ARG;
    if ((stride != 1 && stride != 2) || dilation < 1 || dilation > 7) return SPICE_ERR_UNSUPPORTED;
    // whole stride cells; rows of the result that are whole 16-byte pieces; the in-row start of an edge load needs 8 columns
    if (h % stride != 0 || w % stride != 0 || ((w / stride) & 7) != 0 || w < 8) return SPICE_ERR_UNSUPPORTED;
    if (cin > (1 << 16)) return SPICE_ERR_UNSUPPORTED;
    if (stride == 1 && dilation == 1 && cout <= 32) {   // the form for narrow layers (conv3x3_narrow.hip)
        const int rc = spice::conv3x3_narrow_launch(x, weight9, y, n, cin, h, w, cout, mean, var, bn_weight, bn_bias, eps, relu, stream);
        if (rc != SPICE_ERR_UNSUPPORTED) return rc;
    }
    if (stride == 1) {
        const int rc = conv3x3_halo_launch(x, weight9, y, n, cin, h, w, cout, dilation, mean, var, bn_weight, bn_bias, eps, relu, stream);
        if (rc != SPICE_ERR_UNSUPPORTED) return rc;
    }
    return conv1x1_launch(x, weight9, y, n, cin, static_cast<int64_t>(h / stride) * (w / stride), cout, mean, var, bn_weight,
                          bn_bias, eps, nullptr, nullptr, nullptr, nullptr, nullptr, 0.0f, relu, 0, h, w, nullptr, nullptr,
                          nullptr, nullptr, 0.0f, 0, stride, dilation, stream);
}
