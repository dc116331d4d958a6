// Kernel (3): mIoU confusion-matrix histogram, conf[target, pred] += 1.
//
// HBM-bound streaming read of two int64 arrays (16 B per pixel).  Each workgroup keeps
// several privatised copies of the C*C histogram in LDS (copy chosen by lane, copies offset
// by one bank) so that the same-bin storms of spatially constant label maps spread over
// banks; workgroup totals go to a scratch table with plain stores and a second small kernel
// adds them into the caller's int64 matrix.  Integer arithmetic only: bit-exact and
// independent of scheduling.
#include "common.h"

#include <stdlib.h>

#include <hip/hip_fp16.h>

namespace spice {

constexpr int kConfThreads = 512;
constexpr int kConfMaxBlocks = 1024;
constexpr int kConfLdsBins = 16384;  // 64 KiB of uint32 bins per workgroup at most

struct ConfGeom {
    int copies;
    int stride;  // bins per copy incl. padding; stride % 32 == 1 when copies > 1
};

static inline ConfGeom conf_geom(int C) {
    const int bins = C * C;
    ConfGeom g;
    int copies = 8;
    while (copies > 1) {
        int stride = bins | 31;  // next value with low five bits set ...
        stride += 2;             // ... plus 2 => stride % 32 == 1
        if ((int64_t)copies * stride <= kConfLdsBins) break;
        copies >>= 1;
    }
    g.copies = copies;
    g.stride = (copies > 1) ? ((bins | 31) + 2) : bins;
    return g;
}

static inline int argmax_blocks(int64_t n_threads_needed) {
    int64_t b = ceil_div(n_threads_needed, (int64_t)kConfThreads);
    if (b < 1) b = 1;
    if (b > kConfMaxBlocks) b = kConfMaxBlocks;
    return (int)b;
}

static inline int conf_blocks(int64_t n_pixels) {
    // one workgroup per ~8K pixels (one per CU for an 8x512x512 batch), at most kConfMaxBlocks
    const char *e = getenv("SPICE_CONF_PIXELS_PER_BLOCK");  // tuning knob (host side only)
    const long v = e ? atol(e) : 0;
    int64_t b = ceil_div(n_pixels, v >= 512 ? (int64_t)v : (int64_t)kConfThreads * 16);
    if (b < 1) b = 1;
    if (b > kConfMaxBlocks) b = kConfMaxBlocks;
    return (int)b;
}

struct alignas(16) I64x2 {
    long long x, y;
};

__device__ __forceinline__ void conf_count(unsigned *hist, long long t, long long p, int C,
                                           unsigned &bad) {
    if (t >= 0 && t < C) {
        if (p >= 0 && p < C)
            atomicAdd(&hist[(int)t * C + (int)p], 1u);
        else
            ++bad;
    }
}

// partial: [gridDim.x][C*C] uint32; bad_partial: [gridDim.x] uint32
__global__ __launch_bounds__(kConfThreads) void confusion_hist_kernel(
    const long long *__restrict__ preds, const long long *__restrict__ target, int64_t n, int C,
    int copies, int stride, int vec_ok, unsigned *__restrict__ partial,
    unsigned *__restrict__ bad_partial) {
    extern __shared__ unsigned lds_hist[];
    __shared__ unsigned lds_bad;
    const int tid = threadIdx.x;
    const int bins = C * C;
    for (int i = tid; i < copies * stride; i += kConfThreads) lds_hist[i] = 0;
    if (tid == 0) lds_bad = 0;
    __syncthreads();

    unsigned *hist = lds_hist + (tid & (copies - 1)) * stride;
    unsigned bad = 0;
    const int64_t gthreads = (int64_t)gridDim.x * kConfThreads;
    const int64_t gtid = (int64_t)blockIdx.x * kConfThreads + tid;
    int64_t done = 0;
    if (vec_ok) {
        const int64_t n2 = n >> 1;
        const I64x2 *p2 = reinterpret_cast<const I64x2 *>(preds);
        const I64x2 *t2 = reinterpret_cast<const I64x2 *>(target);
        int64_t i = gtid;
        for (; i + 3 * gthreads < n2; i += 4 * gthreads) {
            const I64x2 ta = ld_stream16(t2 + (i)), tb = ld_stream16(t2 + (i + gthreads)), tc = ld_stream16(t2 + (i + 2 * gthreads)),
                        td = ld_stream16(t2 + (i + 3 * gthreads));
            const I64x2 pa = ld_stream16(p2 + (i)), pb = ld_stream16(p2 + (i + gthreads)), pc = ld_stream16(p2 + (i + 2 * gthreads)),
                        pd = ld_stream16(p2 + (i + 3 * gthreads));
            conf_count(hist, ta.x, pa.x, C, bad);
            conf_count(hist, ta.y, pa.y, C, bad);
            conf_count(hist, tb.x, pb.x, C, bad);
            conf_count(hist, tb.y, pb.y, C, bad);
            conf_count(hist, tc.x, pc.x, C, bad);
            conf_count(hist, tc.y, pc.y, C, bad);
            conf_count(hist, td.x, pd.x, C, bad);
            conf_count(hist, td.y, pd.y, C, bad);
        }
        for (; i < n2; i += gthreads) {
            const I64x2 ta = ld_stream16(t2 + (i));
            const I64x2 pa = ld_stream16(p2 + (i));
            conf_count(hist, ta.x, pa.x, C, bad);
            conf_count(hist, ta.y, pa.y, C, bad);
        }
        done = n2 << 1;
    }
    for (int64_t i = done + gtid; i < n; i += gthreads) conf_count(hist, target[i], preds[i], C, bad);

    if (bad) atomicAdd(&lds_bad, bad);
    __syncthreads();
    unsigned *out = partial + (int64_t)blockIdx.x * bins;
    for (int b = tid; b < bins; b += kConfThreads) {
        unsigned s = 0;
        for (int c = 0; c < copies; ++c) s += lds_hist[c * stride + b];
        out[b] = s;
    }
    if (tid == 0) bad_partial[blockIdx.x] = lds_bad;
}

// Slow path for class counts whose histogram does not fit LDS: 64-bit global atomics.
__global__ __launch_bounds__(256) void confusion_global_kernel(const long long *__restrict__ preds,
                                                               const long long *__restrict__ target,
                                                               int64_t n, int64_t C,
                                                               unsigned long long *conf,
                                                               unsigned long long *bad_total) {
    unsigned long long bad = 0;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (int64_t)gridDim.x * 256) {
        const long long t = target[i], p = preds[i];
        if (t >= 0 && t < C) {
            if (p >= 0 && p < C)
                atomicAdd(&conf[t * C + p], 1ull);
            else
                ++bad;
        }
    }
    if (bad) atomicAdd(bad_total, bad);
}

// conf[b] += sum over workgroups of partial[g][b].  Grid (ceil(bins/64), kConfReduceY): a
// workgroup sums its share of the partial rows for 64 bins (8 row slices per bin, combined in
// LDS) and adds the result with one 64-bit atomic per bin; integer adds commute, so the result
// does not depend on the order in which workgroups arrive.
constexpr int kConfReduceY = 8;

__global__ __launch_bounds__(512) void confusion_reduce_kernel(const unsigned *__restrict__ partial,
                                                               const unsigned *__restrict__ bad_partial,
                                                               int n_blocks, int bins,
                                                               long long *__restrict__ conf,
                                                               long long *__restrict__ bad_total) {
    __shared__ unsigned long long acc[8][64];
    const int bl = threadIdx.x & 63;
    const int slice = threadIdx.x >> 6;
    const int b = blockIdx.x * 64 + bl;
    const int step = 8 * gridDim.y;
    unsigned long long s = 0;
    if (b < bins) {
        int g = blockIdx.y * 8 + slice;
        for (; g + 3 * step < n_blocks; g += 4 * step) {
            const unsigned v0 = partial[(int64_t)g * bins + b];
            const unsigned v1 = partial[(int64_t)(g + step) * bins + b];
            const unsigned v2 = partial[(int64_t)(g + 2 * step) * bins + b];
            const unsigned v3 = partial[(int64_t)(g + 3 * step) * bins + b];
            s += (unsigned long long)v0 + v1 + v2 + v3;
        }
        for (; g < n_blocks; g += step) s += partial[(int64_t)g * bins + b];
    }
    acc[slice][bl] = s;
    __syncthreads();
    if (slice == 0 && b < bins) {
        unsigned long long t = 0;
        for (int k = 0; k < 8; ++k) t += acc[k][bl];
        if (t) atomicAdd(reinterpret_cast<unsigned long long *>(conf) + b, t);
    }
    if (blockIdx.x == 0 && blockIdx.y == 0 && threadIdx.x < 64) {
        unsigned long long t = 0;
        for (int g = threadIdx.x; g < n_blocks; g += 64) t += bad_partial[g];
        t = (unsigned long long)wave_sum((double)t);  // exact: counts stay far below 2^53
        if (threadIdx.x == 0 && t) *bad_total += (long long)t;
    }
}

// ---- fused argmax + histogram -------------------------------------------------------------

template <typename T>
__device__ __forceinline__ float to_f32(T v);
template <>
__device__ __forceinline__ float to_f32<float>(float v) {
    return v;
}
template <>
__device__ __forceinline__ float to_f32<__half>(__half v) {
    return __half2float(v);
}
template <>
__device__ __forceinline__ float to_f32<unsigned short>(unsigned short v) {  // bf16 bits
    return __uint_as_float((unsigned)v << 16);
}

// torch.argmax semantics: a NaN is the maximum; the first maximum wins.
__device__ __forceinline__ void argmax_step(float v, int c, float &best, int &arg) {
    const bool take = (v > best) || (v != v && best == best);
    if (take) {
        best = v;
        arg = c;
    }
}

template <typename T, int VEC>
struct alignas(sizeof(T) * VEC) Pack {
    T v[VEC];
};

// One pack of class-plane values; streamed (read once) when it is a 16-byte pack.
template <typename T, int VEC>
__device__ __forceinline__ Pack<T, VEC> load_pack(const T *p) {
    if constexpr (sizeof(Pack<T, VEC>) == 16) {
        return ld_stream16(reinterpret_cast<const Pack<T, VEC> *>(p));
    } else {
        return *reinterpret_cast<const Pack<T, VEC> *>(p);
    }
}

template <typename T, int VEC>
__global__ __launch_bounds__(kConfThreads) void argmax_confusion_kernel(
    const T *__restrict__ logits, const long long *__restrict__ target, int64_t N, int64_t HW, int C,
    int copies, int stride, unsigned *__restrict__ partial, unsigned *__restrict__ bad_partial) {
    extern __shared__ unsigned lds_hist[];
    const int tid = threadIdx.x;
    const int bins = C * C;
    for (int i = tid; i < copies * stride; i += kConfThreads) lds_hist[i] = 0;
    __syncthreads();
    unsigned *hist = lds_hist + (tid & (copies - 1)) * stride;
    const int64_t per_img = HW / VEC;  // VEC divides HW (checked on the host)
    const int64_t total = N * per_img;
    for (int64_t q = (int64_t)blockIdx.x * kConfThreads + tid; q < total;
         q += (int64_t)gridDim.x * kConfThreads) {
        const int64_t n = q / per_img;
        const int64_t r = (q - n * per_img) * VEC;
        const T *base = logits + n * C * HW + r;
        // labels first, reduced to a row offset at once (-1 = skip): consumed here, so their loads
        // are issued with the first class plane instead of one by one behind the class loop
        const long long *tp = target + n * HW + r;
        int trow[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const long long t = tp[e];
            trow[e] = (t >= 0 && t < C) ? (int)t * C : -1;
        }
        float best[VEC];
        int arg[VEC];
        {
            const Pack<T, VEC> x = load_pack<T, VEC>(base);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                best[e] = to_f32<T>(x.v[e]);
                arg[e] = 0;
            }
        }
        int c = 1;
        for (; c + 3 < C; c += 4) {
            const Pack<T, VEC> x0 = load_pack<T, VEC>(base + (int64_t)c * HW);
            const Pack<T, VEC> x1 = load_pack<T, VEC>(base + (int64_t)(c + 1) * HW);
            const Pack<T, VEC> x2 = load_pack<T, VEC>(base + (int64_t)(c + 2) * HW);
            const Pack<T, VEC> x3 = load_pack<T, VEC>(base + (int64_t)(c + 3) * HW);
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                argmax_step(to_f32<T>(x0.v[e]), c, best[e], arg[e]);
                argmax_step(to_f32<T>(x1.v[e]), c + 1, best[e], arg[e]);
                argmax_step(to_f32<T>(x2.v[e]), c + 2, best[e], arg[e]);
                argmax_step(to_f32<T>(x3.v[e]), c + 3, best[e], arg[e]);
            }
        }
        for (; c < C; ++c) {
            const Pack<T, VEC> x0 = load_pack<T, VEC>(base + (int64_t)c * HW);
#pragma unroll
            for (int e = 0; e < VEC; ++e) argmax_step(to_f32<T>(x0.v[e]), c, best[e], arg[e]);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            if (trow[e] >= 0) atomicAdd(&hist[trow[e] + arg[e]], 1u);
        }
    }
    __syncthreads();
    unsigned *out = partial + (int64_t)blockIdx.x * bins;
    for (int b = tid; b < bins; b += kConfThreads) {
        unsigned s = 0;
        for (int k = 0; k < copies; ++k) s += lds_hist[k * stride + b];
        out[b] = s;
    }
    if (tid == 0) bad_partial[blockIdx.x] = 0;
}

// Workspace layout (bytes): [0,8) int64 count of out-of-range predictions (caller zeroes once),
// [8,16) reserved, [16, 16+4*kConfMaxBlocks) per-workgroup bad counts, then the partial table.
constexpr size_t kConfHeader = 16 + 4 * (size_t)kConfMaxBlocks;

template <typename T, int VEC>
static int launch_argmax(const void *logits, const int64_t *target, int64_t n, int64_t hw, int C,
                         const ConfGeom &g, int blocks, unsigned *partial, unsigned *bad_partial,
                         hipStream_t st) {
    hipLaunchKernelGGL((argmax_confusion_kernel<T, VEC>), dim3(blocks), dim3(kConfThreads),
                       (size_t)g.copies * g.stride * 4, st, static_cast<const T *>(logits),
                       reinterpret_cast<const long long *>(target), n, hw, C, g.copies, g.stride,
                       partial, bad_partial);
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}

}  // namespace spice

using namespace spice;

extern "C" size_t spice_confusion_workspace_bytes(int64_t n_pixels, int32_t num_classes) {
    (void)n_pixels;
    if (num_classes < 1) return kConfHeader;
    const int64_t bins = (int64_t)num_classes * num_classes;
    if (bins > kConfLdsBins) return kConfHeader;  // global-atomic path needs no table
    return kConfHeader + (size_t)kConfMaxBlocks * (size_t)bins * 4;
}

extern "C" int spice_confusion_update(const int64_t *preds, const int64_t *target, int64_t n_pixels,
                                      int32_t num_classes, int64_t *conf, void *workspace,
                                      size_t workspace_bytes, void *stream) {
    if (n_pixels < 0 || num_classes < 1) return SPICE_ERR_ARG;
    if (num_classes > 46340) return SPICE_ERR_UNSUPPORTED;
    if (n_pixels == 0) return SPICE_OK;
    if (!preds || !target || !conf || !workspace) return SPICE_ERR_ARG;
    if (!aligned8(preds) || !aligned8(target) || !aligned8(conf) || !aligned16(workspace))
        return SPICE_ERR_ARG;
    if (workspace_bytes < spice_confusion_workspace_bytes(n_pixels, num_classes))
        return SPICE_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace);
    long long *bad_total = reinterpret_cast<long long *>(ws);
    const int64_t bins = (int64_t)num_classes * num_classes;
    if (bins > kConfLdsBins) {
        int64_t blocks = ceil_div(n_pixels, 256 * 8);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(confusion_global_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                           reinterpret_cast<const long long *>(preds),
                           reinterpret_cast<const long long *>(target), n_pixels,
                           (int64_t)num_classes, reinterpret_cast<unsigned long long *>(conf),
                           reinterpret_cast<unsigned long long *>(bad_total));
        SPICE_LAUNCH_CHECK();
        return SPICE_OK;
    }
    unsigned *bad_partial = reinterpret_cast<unsigned *>(ws + 16);
    unsigned *partial = reinterpret_cast<unsigned *>(ws + kConfHeader);
    const ConfGeom g = conf_geom(num_classes);
    const int blocks = conf_blocks(n_pixels);
    const int vec_ok = aligned16(preds) && aligned16(target);
    hipLaunchKernelGGL(confusion_hist_kernel, dim3(blocks), dim3(kConfThreads),
                       (size_t)g.copies * g.stride * 4, st,
                       reinterpret_cast<const long long *>(preds),
                       reinterpret_cast<const long long *>(target), n_pixels, (int)num_classes,
                       g.copies, g.stride, vec_ok, partial, bad_partial);
    SPICE_LAUNCH_CHECK();
    hipLaunchKernelGGL(confusion_reduce_kernel, dim3((unsigned)ceil_div(bins, 64), kConfReduceY), dim3(512), 0, st,
                       partial, bad_partial, blocks, (int)bins, reinterpret_cast<long long *>(conf),
                       bad_total);
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}

extern "C" int spice_argmax_confusion_update(const void *logits, int32_t logits_dtype,
                                             const int64_t *target, int64_t n, int64_t hw,
                                             int32_t num_classes, int64_t *conf, void *workspace,
                                             size_t workspace_bytes, void *stream) {
    if (n < 0 || hw < 0 || num_classes < 1) return SPICE_ERR_ARG;
    if (logits_dtype < 0 || logits_dtype > 2) return SPICE_ERR_UNSUPPORTED;
    const int64_t bins = (int64_t)num_classes * num_classes;
    if (bins > kConfLdsBins) return SPICE_ERR_UNSUPPORTED;
    if (n == 0 || hw == 0) return SPICE_OK;
    if (!logits || !target || !conf || !workspace) return SPICE_ERR_ARG;
    if (!aligned8(target) || !aligned8(conf) || !aligned16(workspace)) return SPICE_ERR_ARG;
    if (workspace_bytes < spice_confusion_workspace_bytes(n * hw, num_classes))
        return SPICE_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace);
    unsigned *bad_partial = reinterpret_cast<unsigned *>(ws + 16);
    unsigned *partial = reinterpret_cast<unsigned *>(ws + kConfHeader);
    const ConfGeom g = conf_geom(num_classes);
    const int C = num_classes;
    int rc;
    if (logits_dtype == 0) {
        const bool v4 = aligned16(logits) && (hw % 4 == 0);
        const int blocks = argmax_blocks(n * hw / (v4 ? 4 : 1));
        rc = v4 ? launch_argmax<float, 4>(logits, target, n, hw, C, g, blocks, partial, bad_partial, st)
                : launch_argmax<float, 1>(logits, target, n, hw, C, g, blocks, partial, bad_partial, st);
        if (rc) return rc;
        hipLaunchKernelGGL(confusion_reduce_kernel, dim3((unsigned)ceil_div(bins, 64), kConfReduceY), dim3(512), 0,
                           st, partial, bad_partial, blocks, (int)bins,
                           reinterpret_cast<long long *>(conf), reinterpret_cast<long long *>(ws));
    } else {
        const bool v8 = aligned16(logits) && (hw % 8 == 0);
        const int blocks = argmax_blocks(n * hw / (v8 ? 8 : 1));
        if (logits_dtype == 1)
            rc = v8 ? launch_argmax<__half, 8>(logits, target, n, hw, C, g, blocks, partial, bad_partial, st)
                    : launch_argmax<__half, 1>(logits, target, n, hw, C, g, blocks, partial, bad_partial, st);
        else
            rc = v8 ? launch_argmax<unsigned short, 8>(logits, target, n, hw, C, g, blocks, partial,
                                                       bad_partial, st)
                    : launch_argmax<unsigned short, 1>(logits, target, n, hw, C, g, blocks, partial,
                                                       bad_partial, st);
        if (rc) return rc;
        hipLaunchKernelGGL(confusion_reduce_kernel, dim3((unsigned)ceil_div(bins, 64), kConfReduceY), dim3(512), 0,
                           st, partial, bad_partial, blocks, (int)bins,
                           reinterpret_cast<long long *>(conf), reinterpret_cast<long long *>(ws));
    }
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}
