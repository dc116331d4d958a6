// Kernel (3): mIoU confusion-matrix histogram, conf[target, pred] += 1.
//
// HBM-bound streaming read of two int64 arrays (16 B per pixel).  Each workgroup keeps
// several privatised copies of the C*C histogram in LDS (copy chosen by lane, copies offset
// by one bank) so that the same-bin storms of spatially constant label maps spread over
// banks.  ONE launch per update: a workgroup stores its totals write-through into a scratch
// table and takes a ticket of its group of workgroups; the last arriver of a group adds the
// group's rows into the caller's int64 matrix with 64-bit atomics.  Integer arithmetic only:
// bit-exact and independent of scheduling.
#include "common.h"
#include "conf_tail.h"

#include <mutex>

#include <hip/hip_fp16.h>

namespace spice {

SPICE_ENV_KNOB(knob_conf_pixels_per_block, "SPICE_CONF_PIXELS_PER_BLOCK", 0)
SPICE_ENV_KNOB(knob_argmax_max_blocks, "SPICE_ARGMAX_MAX_BLOCKS", 0)

// Workgroups of the fused arg-max kernel: at most two per compute unit, each lane looping over its
// pixel groups.  A workgroup waits for its table row to drain and for its ticket before it leaves,
// so a grid larger than what is resident at once pays that wait twice (measured on MI355X,
// [8,21,512,512] fp32: 33.7 us with 512 workgroups, 34.8 with 768, 38.5 with 1024).
static inline int argmax_blocks(int64_t n_threads_needed) {
    int64_t b = ceil_div(n_threads_needed, (int64_t)kConfThreads);
    const long cap = knob_argmax_max_blocks();
    int64_t lim = (cap >= 1) ? cap : 2 * (int64_t)device_cu_count();
    if (lim > kConfMaxBlocks) lim = kConfMaxBlocks;
    if (b < 1) b = 1;
    if (b > lim) b = lim;
    return (int)b;
}

static inline int conf_blocks(int64_t n_pixels) {
    // one workgroup per ~4K pixels (two per CU for an 8x512x512 batch), at most kConfMaxBlocks
    const long v = knob_conf_pixels_per_block();
    int64_t b = ceil_div(n_pixels, v >= 512 ? (int64_t)v : (int64_t)kConfThreads * 8);
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

__global__ __launch_bounds__(kConfThreads) void confusion_hist_kernel(
    const long long *__restrict__ preds, const long long *__restrict__ target, int64_t n, int C,
    int copies, int stride, int vec_ok, ConfTail tail) {
    extern __shared__ unsigned lds_hist[];
    const int tid = threadIdx.x;
    const int bins = C * C;
    for (int i = tid; i < copies * stride; i += kConfThreads) lds_hist[i] = 0;
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

    conf_finish<kConfThreads>(lds_hist, copies, stride, bins, bad, tail);
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

// Running arg-max of a lane's VEC pixels on the fast path, which is exact for pixels without a NaN:
// `best` is kept with the NaN-propagating maximum (v_maximum3_f32 / v_pk_maximum3_f16), the index
// follows `!(v <= best)` (one compare, one select).  A NaN among a pixel's logits ends up in `best`
// and sends the lane's pixels through the exact one-by-one walk afterwards.
template <typename T, int VEC>
struct ArgFast {
    float best[VEC];
    int arg[VEC];
    __device__ __forceinline__ void init(const Pack<T, VEC> &x) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            best[e] = to_f32<T>(x.v[e]);
            arg[e] = 0;
        }
    }
    __device__ __forceinline__ void step(const Pack<T, VEC> &x, int c) {
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float v = to_f32<T>(x.v[e]);
            arg[e] = !(v <= best[e]) ? c : arg[e];
            best[e] = __builtin_elementwise_maximum(best[e], v);
        }
    }
    __device__ __forceinline__ bool any_nan() const {
        bool r = false;
#pragma unroll
        for (int e = 0; e < VEC; ++e) r |= best[e] != best[e];
        return r;
    }
};

typedef _Float16 f16x2_t __attribute__((ext_vector_type(2)));

// fp16 logits, eight pixels per lane: compared as halves (the order of fp16 values is the order of
// their fp32 conversions), the running maxima packed two to a register.
template <>
struct ArgFast<__half, 8> {
    f16x2_t best[4];
    int arg[8];
    __device__ __forceinline__ void init(const Pack<__half, 8> &x) {
        const f16x2_t *p = reinterpret_cast<const f16x2_t *>(&x);
#pragma unroll
        for (int k = 0; k < 4; ++k) best[k] = p[k];
#pragma unroll
        for (int e = 0; e < 8; ++e) arg[e] = 0;
    }
    __device__ __forceinline__ void step(const Pack<__half, 8> &x, int c) {
        const f16x2_t *p = reinterpret_cast<const f16x2_t *>(&x);
#pragma unroll
        for (int k = 0; k < 4; ++k) {
            const f16x2_t v = p[k];
            arg[2 * k] = !(v.x <= best[k].x) ? c : arg[2 * k];
            arg[2 * k + 1] = !(v.y <= best[k].y) ? c : arg[2 * k + 1];
            best[k] = __builtin_elementwise_maximum(best[k], v);
        }
    }
    __device__ __forceinline__ bool any_nan() const {
        bool r = false;
#pragma unroll
        for (int k = 0; k < 4; ++k) r |= (best[k].x != best[k].x) | (best[k].y != best[k].y);
        return r;
    }
};

template <typename T, int VEC>
__device__ __forceinline__ void load_planes4(Pack<T, VEC> (&x)[4], const T *base, int c, int64_t HW) {
#pragma unroll
    for (int k = 0; k < 4; ++k) x[k] = load_pack<T, VEC>(base + (int64_t)(c + k) * HW);
}

template <typename T, int VEC>
__device__ __forceinline__ void use_planes4(ArgFast<T, VEC> &f, const Pack<T, VEC> (&x)[4], int c) {
#pragma unroll
    for (int k = 0; k < 4; ++k) f.step(x[k], c + k);
}

// The class axis is walked four planes at a time with the next four already requested (two
// register sets, A and B), so every wave overlaps its compares with its own loads instead of
// relying on other waves being in another phase: all workgroups start together and would
// otherwise alternate between loading and comparing in step.
template <typename T, int VEC, bool WIDE>
__global__ __launch_bounds__(kConfThreads, 4) void argmax_confusion_kernel(
    const T *__restrict__ logits, const long long *__restrict__ target, int64_t N, int64_t HW, int C,
    int copies, int stride, ConfTail tail) {
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
        // are issued with the first class planes instead of one by one behind the class loop
        const long long *tp = target + n * HW + r;
        int trow[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const long long t = tp[e];
            trow[e] = (t >= 0 && t < C) ? (int)t * C : -1;
        }
        int arg[VEC];
        bool slow = !WIDE;
        if (WIDE) {
            ArgFast<T, VEC> f;
            Pack<T, VEC> a[4], b[4];
            const Pack<T, VEC> x0 = load_pack<T, VEC>(base);
            int c = 1;
            if (c + 3 < C) load_planes4<T, VEC>(a, base, c, HW);
            f.init(x0);
            if (c + 3 < C) {
                for (; c + 11 < C; c += 8) {
                    load_planes4<T, VEC>(b, base, c + 4, HW);
                    use_planes4<T, VEC>(f, a, c);
                    load_planes4<T, VEC>(a, base, c + 8, HW);
                    use_planes4<T, VEC>(f, b, c + 4);
                }
                if (c + 7 < C) {
                    load_planes4<T, VEC>(b, base, c + 4, HW);
                    use_planes4<T, VEC>(f, a, c);
                    use_planes4<T, VEC>(f, b, c + 4);
                    c += 8;
                } else {
                    use_planes4<T, VEC>(f, a, c);
                    c += 4;
                }
            }
            for (; c < C; ++c) f.step(load_pack<T, VEC>(base + (int64_t)c * HW), c);
#pragma unroll
            for (int e = 0; e < VEC; ++e) arg[e] = f.arg[e];
            slow = f.any_nan();
        }
        if (slow) {  // exact walk, class by class: a NaN is the maximum, the first maximum wins
            float best[VEC];
            {
                const Pack<T, VEC> x = load_pack<T, VEC>(base);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    best[e] = to_f32<T>(x.v[e]);
                    arg[e] = 0;
                }
            }
            for (int c = 1; c < C; ++c) {
                const Pack<T, VEC> x = load_pack<T, VEC>(base + (int64_t)c * HW);
#pragma unroll
                for (int e = 0; e < VEC; ++e) argmax_step(to_f32<T>(x.v[e]), c, best[e], arg[e]);
            }
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            if (trow[e] >= 0) atomicAdd(&hist[trow[e] + arg[e]], 1u);
        }
    }
    conf_finish<kConfThreads>(lds_hist, copies, stride, bins, 0u, tail);
}

template <typename T, int VEC, bool WIDE>
static int launch_argmax(const void *logits, const int64_t *target, int64_t n, int64_t hw, int C,
                         const ConfGeom &g, int blocks, const ConfTail &tail, hipStream_t st) {
    hipLaunchKernelGGL((argmax_confusion_kernel<T, VEC, WIDE>), dim3(blocks), dim3(kConfThreads),
                       (size_t)g.copies * g.stride * 4, st, static_cast<const T *>(logits),
                       reinterpret_cast<const long long *>(target), n, hw, C, g.copies, g.stride, tail);
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}

// The update itself, on a scratch area laid out as above (caller workspace or the static array).
static int confusion_update_on(const int64_t *preds, const int64_t *target, int64_t n_pixels, int num_classes,
                               int64_t *conf, void *scratch, int64_t *invalid_count, hipStream_t st) {
    const ConfGeom g = conf_geom(num_classes);
    const int blocks = conf_blocks(n_pixels);
    const int vec_ok = aligned16(preds) && aligned16(target);
    hipLaunchKernelGGL(confusion_hist_kernel, dim3(blocks), dim3(kConfThreads),
                       (size_t)g.copies * g.stride * 4, st,
                       reinterpret_cast<const long long *>(preds),
                       reinterpret_cast<const long long *>(target), n_pixels, num_classes,
                       g.copies, g.stride, vec_ok, conf_tail(scratch, conf, invalid_count, blocks));
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
    if (num_classes > 46340 || n_pixels > kConfMaxPixels) return SPICE_ERR_UNSUPPORTED;
    if (n_pixels == 0) return SPICE_OK;
    if (!preds || !target || !conf || !workspace) return SPICE_ERR_ARG;
    if (!aligned8(preds) || !aligned8(target) || !aligned8(conf) || !aligned16(workspace))
        return SPICE_ERR_ARG;
    if (workspace_bytes < spice_confusion_workspace_bytes(n_pixels, num_classes))
        return SPICE_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const int64_t bins = (int64_t)num_classes * num_classes;
    if (bins > kConfLdsBins) {
        int64_t blocks = ceil_div(n_pixels, 256 * 8);
        if (blocks > 2048) blocks = 2048;
        hipLaunchKernelGGL(confusion_global_kernel, dim3((unsigned)blocks), dim3(256), 0, st,
                           reinterpret_cast<const long long *>(preds),
                           reinterpret_cast<const long long *>(target), n_pixels,
                           (int64_t)num_classes, reinterpret_cast<unsigned long long *>(conf),
                           reinterpret_cast<unsigned long long *>(workspace));
        SPICE_LAUNCH_CHECK();
        return SPICE_OK;
    }
    return confusion_update_on(preds, target, n_pixels, num_classes, conf, workspace, nullptr, st);
}

// ---- the same update for callers that cannot allocate -------------------------------------
//
// The scratch table lives in a file-scope device array instead of a caller workspace.  Each GPU
// holds its own copy of the array at its own address; in host code the array's name is only a
// stand-in, so the address is looked up once per device (with that device current) and that
// pointer, never the name, is what the kernels receive.
constexpr int kConfStaticMaxClasses = 32;
constexpr size_t kConfStaticBytes =
    kConfHeader + (size_t)kConfMaxBlocks * kConfStaticMaxClasses * kConfStaticMaxClasses * 4;  // 4 KiB + 4 MiB
static __device__ __align__(256) unsigned char scratch[kConfStaticBytes];

constexpr int kMaxDevices = 64;
static std::mutex scratch_mutex;
static void *scratch_ptr[kMaxDevices] = {};

static int scratch_of_current_device(void **out) {
    int dev = -1;
    SPICE_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= kMaxDevices) return SPICE_ERR_UNSUPPORTED;
    std::lock_guard<std::mutex> lock(scratch_mutex);
    if (scratch_ptr[dev] == nullptr) {
        void *p = nullptr;
        SPICE_HIP_TRY(hipGetSymbolAddress(&p, HIP_SYMBOL(scratch)));
        if (p == nullptr) return SPICE_ERR_ARG;
        scratch_ptr[dev] = p;
    }
    *out = scratch_ptr[dev];
    return SPICE_OK;
}

extern "C" int spice_confusion_update_static(const int64_t *preds, const int64_t *target, int64_t n_pixels,
                                             int32_t num_classes, int64_t *conf, int64_t *invalid_count,
                                             void *stream) {
    if (n_pixels < 0 || num_classes < 1) return SPICE_ERR_ARG;
    if (num_classes > kConfStaticMaxClasses || n_pixels > kConfMaxPixels) return SPICE_ERR_STATIC_LIMIT;
    if (n_pixels == 0) return SPICE_OK;
    if (!preds || !target || !conf) return SPICE_ERR_ARG;
    if (!aligned8(preds) || !aligned8(target) || !aligned8(conf) || !aligned8(invalid_count)) return SPICE_ERR_ARG;
    void *dev_scratch = nullptr;
    const int rc = scratch_of_current_device(&dev_scratch);
    if (rc != SPICE_OK) return rc;
    return confusion_update_on(preds, target, n_pixels, num_classes, conf, dev_scratch, invalid_count,
                               static_cast<hipStream_t>(stream));
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
    if (n > kConfMaxPixels / hw) return SPICE_ERR_UNSUPPORTED;
    if (!logits || !target || !conf || !workspace) return SPICE_ERR_ARG;
    if (!aligned8(target) || !aligned8(conf) || !aligned16(workspace)) return SPICE_ERR_ARG;
    if (workspace_bytes < spice_confusion_workspace_bytes(n * hw, num_classes))
        return SPICE_ERR_WORKSPACE;
    hipStream_t st = static_cast<hipStream_t>(stream);
    const ConfGeom g = conf_geom(num_classes);
    const int C = num_classes;
    if (logits_dtype == 0) {
        const bool v4 = aligned16(logits) && (hw % 4 == 0);
        const int blocks = argmax_blocks(n * hw / (v4 ? 4 : 1));
        const ConfTail tail = conf_tail(workspace, conf, nullptr, blocks);
        return v4 ? launch_argmax<float, 4, true>(logits, target, n, hw, C, g, blocks, tail, st)
                  : launch_argmax<float, 1, false>(logits, target, n, hw, C, g, blocks, tail, st);
    }
    const bool v8 = aligned16(logits) && (hw % 8 == 0);
    const int blocks = argmax_blocks(n * hw / (v8 ? 8 : 1));
    const ConfTail tail = conf_tail(workspace, conf, nullptr, blocks);
    if (logits_dtype == 1)
        return v8 ? launch_argmax<__half, 8, true>(logits, target, n, hw, C, g, blocks, tail, st)
                  : launch_argmax<__half, 1, false>(logits, target, n, hw, C, g, blocks, tail, st);
    return v8 ? launch_argmax<unsigned short, 8, true>(logits, target, n, hw, C, g, blocks, tail, st)
              : launch_argmax<unsigned short, 1, false>(logits, target, n, hw, C, g, blocks, tail, st);
}
