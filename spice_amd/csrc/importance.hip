// Kernel (1): per-channel importance reductions over every weight tensor of a model
// in one launch.  HBM-bound: each weight (and gradient) element is read exactly once
// with 16-byte loads; a row is reduced by a fixed tree (per-lane strided partial sums,
// then a butterfly over the lanes that share the row), so results are deterministic.
#include "common.h"

#include <stdlib.h>

namespace spice {

constexpr int kImpThreads = 256;
constexpr int64_t kImpTargetElemsDefault = 4096;  // elements one work item (one workgroup) reduces; measured on MI355X: 4 K -> 20.0 us (L2) / 35.8 us (Taylor), 16 K -> 21.5 / 37.9 us

static int64_t imp_target_elems() {
    const char *e = getenv("SPICE_IMP_TARGET_ELEMS");  // tuning knob, read by the host planner only
    const int64_t v = e ? atoll(e) : 0;
    return v >= 1024 ? v : kImpTargetElemsDefault;
}

template <int MODE>
__device__ __forceinline__ float imp_term(float w, float g) {
    if (MODE == SPICE_IMP_L1) return fabsf(w);
    if (MODE == SPICE_IMP_L2) return w * w;
    return fabsf(g) * fabsf(w);
}

template <int MODE>
__device__ __forceinline__ float imp_term4(const float4 &w, const float4 &g) {
    return (imp_term<MODE>(w.x, g.x) + imp_term<MODE>(w.y, g.y)) +
           (imp_term<MODE>(w.z, g.z) + imp_term<MODE>(w.w, g.w));
}

template <int MODE>
__global__ __launch_bounds__(kImpThreads) void importance_rows_kernel(
    const float *__restrict__ w_base, const float *__restrict__ g_base,
    float *__restrict__ score_base, const spice_rows_item *__restrict__ items, int accumulate,
    int bases_vec_ok) {
    const spice_rows_item it = items[blockIdx.x];
    const int L = it.lanes;
    const int K = it.row_len;
    const int tid = threadIdx.x;
    const int sub = tid / L;
    const int lane = tid & (L - 1);
    const int groups = kImpThreads / L;
    const bool vec = bases_vec_ok && (it.flags & 1) && (MODE != SPICE_IMP_TAYLOR || (it.flags & 2));
    const float *w0 = w_base + it.w_off;
    const float *g0 = (MODE == SPICE_IMP_TAYLOR) ? g_base + it.g_off : w0;
    float *out = score_base + it.out_off;

    for (int r = sub; r < it.n_rows; r += groups) {
        const float *wr = w0 + (int64_t)r * K;
        const float *gr = g0 + (int64_t)r * K;
        float a0 = 0.f, a1 = 0.f, a2 = 0.f, a3 = 0.f;
        if (vec) {
            const float4 *w4 = reinterpret_cast<const float4 *>(wr);
            const float4 *g4 = reinterpret_cast<const float4 *>(gr);
            const int K4 = K >> 2;
            int j = lane;
            for (; j + 7 * L < K4; j += 8 * L) {
                float4 x[8], y[8];
#pragma unroll
                for (int u = 0; u < 8; ++u) x[u] = ld_stream16(w4 + (j + u * L));
#pragma unroll
                for (int u = 0; u < 8; ++u) y[u] = (MODE == SPICE_IMP_TAYLOR) ? ld_stream16(g4 + (j + u * L)) : x[u];
                a0 += imp_term4<MODE>(x[0], y[0]) + imp_term4<MODE>(x[4], y[4]);
                a1 += imp_term4<MODE>(x[1], y[1]) + imp_term4<MODE>(x[5], y[5]);
                a2 += imp_term4<MODE>(x[2], y[2]) + imp_term4<MODE>(x[6], y[6]);
                a3 += imp_term4<MODE>(x[3], y[3]) + imp_term4<MODE>(x[7], y[7]);
            }
            for (; j + 3 * L < K4; j += 4 * L) {
                const float4 x0 = ld_stream16(w4 + (j)), x1 = ld_stream16(w4 + (j + L)), x2 = ld_stream16(w4 + (j + 2 * L)), x3 = ld_stream16(w4 + (j + 3 * L));
                float4 y0 = x0, y1 = x1, y2 = x2, y3 = x3;
                if (MODE == SPICE_IMP_TAYLOR) {
                    y0 = ld_stream16(g4 + (j));
                    y1 = ld_stream16(g4 + (j + L));
                    y2 = ld_stream16(g4 + (j + 2 * L));
                    y3 = ld_stream16(g4 + (j + 3 * L));
                }
                a0 += imp_term4<MODE>(x0, y0);
                a1 += imp_term4<MODE>(x1, y1);
                a2 += imp_term4<MODE>(x2, y2);
                a3 += imp_term4<MODE>(x3, y3);
            }
            for (; j < K4; j += L) {
                const float4 x0 = ld_stream16(w4 + (j));
                float4 y0 = x0;
                if (MODE == SPICE_IMP_TAYLOR) y0 = ld_stream16(g4 + (j));
                a0 += imp_term4<MODE>(x0, y0);
            }
        } else {
            int j = lane;
            for (; j + 3 * L < K; j += 4 * L) {
                const float x0 = wr[j], x1 = wr[j + L], x2 = wr[j + 2 * L], x3 = wr[j + 3 * L];
                float y0 = x0, y1 = x1, y2 = x2, y3 = x3;
                if (MODE == SPICE_IMP_TAYLOR) {
                    y0 = gr[j];
                    y1 = gr[j + L];
                    y2 = gr[j + 2 * L];
                    y3 = gr[j + 3 * L];
                }
                a0 += imp_term<MODE>(x0, y0);
                a1 += imp_term<MODE>(x1, y1);
                a2 += imp_term<MODE>(x2, y2);
                a3 += imp_term<MODE>(x3, y3);
            }
            for (; j < K; j += L) {
                const float x0 = wr[j];
                const float y0 = (MODE == SPICE_IMP_TAYLOR) ? gr[j] : x0;
                a0 += imp_term<MODE>(x0, y0);
            }
        }
        float s = group_sum_sw((a0 + a1) + (a2 + a3), L);
        if (lane == 0) {
            if (accumulate) s += out[r];
            out[r] = s;
        }
    }
}

// Stage 1 of the per-input-channel reduction: a workgroup owns a chunk of flat columns
// (cols*inner granularity) and a slab of rows; thread t sums its 4 (or 1) flat columns over
// the slab and stores the partial sums.  Stage 2 folds slabs and the kh*kw taps.
template <int MODE>
__global__ __launch_bounds__(kImpThreads) void importance_cols_stage1_kernel(
    const float *__restrict__ w_base, float *__restrict__ partial_base,
    const spice_cols_item *__restrict__ items, int bases_vec_ok) {
    const spice_cols_item it = items[blockIdx.x];
    const int tid = threadIdx.x;
    const float *w0 = w_base + it.w_off;
    float *out = partial_base + it.out_off;
    const int64_t stride = it.row_stride;
    if (bases_vec_ok && (it.flags & 1)) {
        const int c = tid * 4;
        if (c >= it.n_flat) return;
        float4 acc = make_float4(0.f, 0.f, 0.f, 0.f);
        const float *p = w0 + c;
        int r = 0;
        for (; r + 3 < it.n_rows; r += 4) {
            const float4 x0 = *reinterpret_cast<const float4 *>(p + (int64_t)r * stride);
            const float4 x1 = *reinterpret_cast<const float4 *>(p + (int64_t)(r + 1) * stride);
            const float4 x2 = *reinterpret_cast<const float4 *>(p + (int64_t)(r + 2) * stride);
            const float4 x3 = *reinterpret_cast<const float4 *>(p + (int64_t)(r + 3) * stride);
            acc.x += (imp_term<MODE>(x0.x, 0.f) + imp_term<MODE>(x1.x, 0.f)) +
                     (imp_term<MODE>(x2.x, 0.f) + imp_term<MODE>(x3.x, 0.f));
            acc.y += (imp_term<MODE>(x0.y, 0.f) + imp_term<MODE>(x1.y, 0.f)) +
                     (imp_term<MODE>(x2.y, 0.f) + imp_term<MODE>(x3.y, 0.f));
            acc.z += (imp_term<MODE>(x0.z, 0.f) + imp_term<MODE>(x1.z, 0.f)) +
                     (imp_term<MODE>(x2.z, 0.f) + imp_term<MODE>(x3.z, 0.f));
            acc.w += (imp_term<MODE>(x0.w, 0.f) + imp_term<MODE>(x1.w, 0.f)) +
                     (imp_term<MODE>(x2.w, 0.f) + imp_term<MODE>(x3.w, 0.f));
        }
        for (; r < it.n_rows; ++r) {
            const float4 x0 = *reinterpret_cast<const float4 *>(p + (int64_t)r * stride);
            acc.x += imp_term<MODE>(x0.x, 0.f);
            acc.y += imp_term<MODE>(x0.y, 0.f);
            acc.z += imp_term<MODE>(x0.z, 0.f);
            acc.w += imp_term<MODE>(x0.w, 0.f);
        }
        *reinterpret_cast<float4 *>(out + c) = acc;
    } else {
        for (int c = tid; c < it.n_flat; c += kImpThreads) {
            float acc = 0.f;
            const float *p = w0 + c;
            for (int r = 0; r < it.n_rows; ++r) acc += imp_term<MODE>(p[(int64_t)r * stride], 0.f);
            out[c] = acc;
        }
    }
}

__global__ __launch_bounds__(kImpThreads) void importance_cols_fold_kernel(
    const float *__restrict__ partial_base, float *__restrict__ score_base,
    const spice_cols_fold *__restrict__ folds) {
    const spice_cols_fold f = folds[blockIdx.x];
    const float *part = partial_base + f.part_off;
    float *out = score_base + f.out_off;
    for (int c = blockIdx.y * kImpThreads + threadIdx.x; c < f.cols; c += gridDim.y * kImpThreads) {
        float acc = 0.f;
        for (int s = 0; s < f.n_slabs; ++s) {
            const float *row = part + (int64_t)s * f.n_flat + (int64_t)c * f.inner;
            float a = 0.f;
            for (int k = 0; k < f.inner; ++k) a += row[k];
            acc += a;
        }
        out[c] = acc;
    }
}

}  // namespace spice

using namespace spice;

extern "C" int64_t spice_importance_plan(const spice_rows_desc *layers, int32_t n_layers,
                                         spice_rows_item *items, int64_t capacity) {
    if (n_layers < 0 || (n_layers > 0 && layers == nullptr)) return -SPICE_ERR_ARG;
    int64_t n = 0;
    for (int32_t l = 0; l < n_layers; ++l) {
        const spice_rows_desc &d = layers[l];
        if (d.rows < 0 || d.row_len < 0 || d.w_off < 0 || d.out_off < 0) return -SPICE_ERR_ARG;
        if (d.rows == 0) continue;
        if (d.row_len > INT32_MAX || d.rows > INT32_MAX) return -SPICE_ERR_UNSUPPORTED;
        const int64_t K = d.row_len;
        const bool w_vec = (d.w_off % 4 == 0) && (K % 4 == 0) && K > 0;
        const bool g_vec = w_vec && (d.g_off % 4 == 0);
        // lanes per row: every lane should own at least four (vector) loads of its row
        const int64_t units = (w_vec ? K / 4 : K) / 4;
        int L = kWave;
        while (L > 1 && L > units) L >>= 1;
        const int64_t groups = kImpThreads / L;
        int64_t rpi = K > 0 ? imp_target_elems() / K : d.rows;
        if (rpi < 1) rpi = 1;
        rpi = ceil_div(rpi, groups) * groups;
        const int64_t n_it = ceil_div(d.rows, rpi);
        rpi = ceil_div(ceil_div(d.rows, n_it), groups) * groups;
        for (int64_t r0 = 0; r0 < d.rows; r0 += rpi, ++n) {
            if (items == nullptr || n >= capacity) continue;
            spice_rows_item &it = items[n];
            it.w_off = d.w_off + r0 * K;
            it.g_off = d.g_off + r0 * K;
            it.out_off = d.out_off + r0;
            it.row_len = (int32_t)K;
            it.n_rows = (int32_t)((d.rows - r0 < rpi) ? d.rows - r0 : rpi);
            it.lanes = L;
            it.flags = (w_vec ? 1 : 0) | (g_vec ? 2 : 0);
        }
    }
    return n;
}

extern "C" int spice_importance_run(const float *w_base, const float *g_base, float *score_base,
                                    const spice_rows_item *d_items, int64_t n_items, int32_t mode,
                                    int32_t accumulate, void *stream) {
    if (n_items < 0 || n_items > INT32_MAX) return SPICE_ERR_ARG;
    if (n_items == 0) return SPICE_OK;
    if (!w_base || !score_base || !d_items) return SPICE_ERR_ARG;
    if (mode == SPICE_IMP_TAYLOR && !g_base) return SPICE_ERR_ARG;
    if (mode != SPICE_IMP_L1 && mode != SPICE_IMP_L2 && mode != SPICE_IMP_TAYLOR)
        return SPICE_ERR_UNSUPPORTED;
    if ((reinterpret_cast<uintptr_t>(w_base) & 3u) || (reinterpret_cast<uintptr_t>(score_base) & 3u))
        return SPICE_ERR_ARG;
    const int vec_ok = aligned16(w_base) && (mode != SPICE_IMP_TAYLOR || aligned16(g_base));
    hipStream_t st = static_cast<hipStream_t>(stream);
    const dim3 grid((unsigned)n_items), block(kImpThreads);
    if (mode == SPICE_IMP_L1)
        hipLaunchKernelGGL(importance_rows_kernel<SPICE_IMP_L1>, grid, block, 0, st, w_base, g_base,
                           score_base, d_items, accumulate, vec_ok);
    else if (mode == SPICE_IMP_L2)
        hipLaunchKernelGGL(importance_rows_kernel<SPICE_IMP_L2>, grid, block, 0, st, w_base, g_base,
                           score_base, d_items, accumulate, vec_ok);
    else
        hipLaunchKernelGGL(importance_rows_kernel<SPICE_IMP_TAYLOR>, grid, block, 0, st, w_base,
                           g_base, score_base, d_items, accumulate, vec_ok);
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}

extern "C" int64_t spice_importance_cols_plan(const spice_cols_desc *layers, int32_t n_layers,
                                              spice_cols_item *items, int64_t capacity,
                                              spice_cols_fold *folds, int64_t *partial_elems) {
    if (n_layers < 0 || (n_layers > 0 && layers == nullptr)) return -SPICE_ERR_ARG;
    int64_t n = 0, part = 0;
    for (int32_t l = 0; l < n_layers; ++l) {
        const spice_cols_desc &d = layers[l];
        if (d.rows < 0 || d.cols < 0 || d.inner < 1 || d.w_off < 0 || d.out_off < 0)
            return -SPICE_ERR_ARG;
        const int64_t F = d.cols * d.inner;
        if (F > INT32_MAX || d.rows > INT32_MAX) return -SPICE_ERR_UNSUPPORTED;
        const bool vec = (d.w_off % 4 == 0) && (F % 4 == 0);
        const int64_t chunk = vec ? 4 * kImpThreads : kImpThreads;
        const int64_t n_chunks = F > 0 ? ceil_div(F, chunk) : 0;
        // rows per slab: about 32K elements per item, at most 16 slabs per layer
        const int64_t width = F < chunk ? (F > 0 ? F : 1) : chunk;
        int64_t rps = 32768 / width;
        if (rps < 8) rps = 8;
        int64_t n_slabs = d.rows > 0 ? ceil_div(d.rows, rps) : 0;
        if (n_slabs > 16) n_slabs = 16;
        if (n_slabs > 0) rps = ceil_div(d.rows, n_slabs);
        if (n_slabs > 0) n_slabs = ceil_div(d.rows, rps);
        part = ceil_div(part, 4) * 4;  // keep every layer's partial block 16-byte aligned
        if (folds != nullptr) {
            spice_cols_fold &f = folds[l];
            f.part_off = part;
            f.out_off = d.out_off;
            f.n_slabs = (int32_t)n_slabs;
            f.n_flat = (int32_t)F;
            f.inner = (int32_t)d.inner;
            f.cols = (int32_t)d.cols;
        }
        for (int64_t s = 0; s < n_slabs; ++s) {
            const int64_t r0 = s * rps;
            const int64_t nr = (d.rows - r0 < rps) ? d.rows - r0 : rps;
            for (int64_t ch = 0; ch < n_chunks; ++ch, ++n) {
                if (items == nullptr || n >= capacity) continue;
                spice_cols_item &it = items[n];
                const int64_t c0 = ch * chunk;
                it.w_off = d.w_off + r0 * F + c0;
                it.out_off = part + s * F + c0;
                it.row_stride = (int32_t)F;
                it.n_rows = (int32_t)nr;
                it.n_flat = (int32_t)((F - c0 < chunk) ? F - c0 : chunk);
                it.flags = vec ? 1 : 0;
            }
        }
        part += n_slabs * F;
    }
    if (partial_elems) *partial_elems = part;
    return n;
}

extern "C" int spice_importance_cols_run(const float *w_base, float *score_base, float *partial_base,
                                         const spice_cols_item *d_items, int64_t n_items,
                                         const spice_cols_fold *d_folds, int64_t n_folds,
                                         int32_t mode, void *stream) {
    if (n_items < 0 || n_items > INT32_MAX || n_folds < 0 || n_folds > 65535) return SPICE_ERR_ARG;
    if (n_folds == 0) return SPICE_OK;
    if (!w_base || !score_base || !d_folds) return SPICE_ERR_ARG;
    if (n_items > 0 && (!partial_base || !d_items)) return SPICE_ERR_ARG;
    if (mode != SPICE_IMP_L1 && mode != SPICE_IMP_L2) return SPICE_ERR_UNSUPPORTED;
    const int vec_ok = aligned16(w_base) && aligned16(partial_base);
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (n_items > 0) {
        const dim3 grid((unsigned)n_items), block(kImpThreads);
        if (mode == SPICE_IMP_L1)
            hipLaunchKernelGGL(importance_cols_stage1_kernel<SPICE_IMP_L1>, grid, block, 0, st,
                               w_base, partial_base, d_items, vec_ok);
        else
            hipLaunchKernelGGL(importance_cols_stage1_kernel<SPICE_IMP_L2>, grid, block, 0, st,
                               w_base, partial_base, d_items, vec_ok);
        SPICE_LAUNCH_CHECK();
    }
    hipLaunchKernelGGL(importance_cols_fold_kernel, dim3((unsigned)n_folds, 16), dim3(kImpThreads), 0,
                       st, partial_base, score_base, d_folds);
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}
