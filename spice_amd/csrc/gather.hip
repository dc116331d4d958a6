// Kernel (4): channel gather.  dst[o, i, k] = src[keep_out[o], keep_in[i], k] for every tensor
// of a model in one launch.  Pure HBM-bound copy of 32-bit words (bit-exact); kept rows are
// copied with 16-byte accesses when no input-channel list breaks their contiguity.
#include "common.h"

#include <stdlib.h>

#include <type_traits>

namespace spice {

constexpr int kGatherThreads = 256;
constexpr int64_t kGatherDefaultTargetElems = 1024;  // measured best on MI355X (1 K: 24.2 us, 4 K: 27.1 us, 16 K: 32 us for 53 MB)

static int64_t gather_target_elems() {
    const char *e = getenv("SPICE_GATHER_TARGET_ELEMS");  // tuning knob (host side only)
    const long v = e ? atol(e) : 0;
    return (v >= 256 && v <= (1 << 22)) ? v : kGatherDefaultTargetElems;
}

// Items with an in-channel list whose kept source rows fit this many 32-bit words are staged in
// LDS: whole source rows arrive with 16-byte loads (every 64-byte sector of a kept row holds kept
// columns anyway when every second channel goes), the kept columns are picked out of LDS, and
// the destination is written with 16-byte stores.
constexpr int kGatherStageWords = 5120;   // 20 KB of source rows per workgroup
constexpr int kGatherStageCols = 2048;    // 8 KB table of kept-column offsets
constexpr size_t kGatherLdsBytes = (size_t)(kGatherStageWords + kGatherStageCols) * 4;
constexpr int64_t kGatherStageTargetElems = 4096;  // destination words per staged item

static int64_t gather_stage_target_elems() {
    const char *e = getenv("SPICE_GATHER_STAGE_TARGET_ELEMS");  // tuning knob (host side only)
    const long v = e ? atol(e) : 0;
    return (v >= 256 && v <= (1 << 22)) ? v : kGatherStageTargetElems;
}

static bool gather_stage_enabled() {
    const char *e = getenv("SPICE_GATHER_STAGE");  // tuning knob (host side only): 0 = never stage rows in LDS
    return !(e && atoi(e) == 0);
}

// One staged work item (flags bit2): rows and kept columns as described above.  Requirements, all
// checked by the planner: 16-byte aligned source rows and destination rows, row_len % 4 == 0,
// src_cols * inner <= kGatherStageWords, dst_cols <= kGatherStageCols.  The item's rows go through
// LDS in batches of as many rows as fit; the next batch's source rows are already on their way
// (held in registers, five 16-byte loads per lane) while the current one is picked apart.
constexpr int kGatherStageRegs = kGatherStageWords / 4 / kGatherThreads;  // 16-byte registers per lane per batch

__device__ __forceinline__ void gather_staged(const spice_gather_item &it, const unsigned *__restrict__ src,
                                              unsigned *__restrict__ dst, const long long *__restrict__ idx_out,
                                              const long long *__restrict__ idx_in, unsigned *lds) {
    const int tid = threadIdx.x;
    const int inner = it.inner;
    const int row_len = it.dst_cols * inner;
    const int src_stride = it.src_cols * inner;
    unsigned *rows = lds;
    int *coltab = reinterpret_cast<int *>(lds + kGatherStageWords);
    for (int i = tid; i < it.dst_cols; i += kGatherThreads) coltab[i] = (int)idx_in[i] * inner;
    const int stride4 = src_stride >> 2;
    const int rows_b = min(it.n_rows, kGatherStageWords / src_stride);  // rows per batch (>= 1)
    const uint4 *src4 = reinterpret_cast<const uint4 *>(src);
    uint4 *rows4 = reinterpret_cast<uint4 *>(rows);
    uint4 *dst4 = reinterpret_cast<uint4 *>(dst);

    uint4 regs[kGatherStageRegs];
    auto load_batch = [&](int r0) {  // source rows r0 .. r0+nr-1 -> registers, flat over (row, 16-byte chunk)
        const int nr = min(rows_b, it.n_rows - r0);
        const int total4 = nr * stride4;
        long long so[kGatherStageRegs];
        int col[kGatherStageRegs];
#pragma unroll
        for (int u = 0; u < kGatherStageRegs; ++u) {
            const int f = min(tid + u * kGatherThreads, total4 - 1);  // surplus lanes repeat the last chunk
            const int row = f / stride4;
            col[u] = f - row * stride4;
            so[u] = idx_out ? idx_out[r0 + row] : (long long)(it.row0 + r0 + row);
        }
#pragma unroll
        for (int u = 0; u < kGatherStageRegs; ++u) regs[u] = ld_stream16(src4 + (so[u] * stride4 + col[u]));
    };

    load_batch(0);
    for (int r0 = 0; r0 < it.n_rows; r0 += rows_b) {
        const int nr = min(rows_b, it.n_rows - r0);
        const int total4 = nr * stride4;
#pragma unroll
        for (int u = 0; u < kGatherStageRegs; ++u) {
            const int f = tid + u * kGatherThreads;
            if (f < total4) rows4[f] = regs[u];
        }
        __syncthreads();
        if (r0 + rows_b < it.n_rows) load_batch(r0 + rows_b);
        // kept columns out of LDS, four consecutive destination words per lane (never across a row)
        const int total = nr * row_len;
        uint4 *out4 = dst4 + (((int64_t)r0 * row_len) >> 2);
        for (int e0 = tid * 4; e0 < total; e0 += 4 * kGatherThreads) {
            const int row = e0 / row_len;
            const int rem = e0 - row * row_len;
            int i = inner == 1 ? rem : rem / inner;
            int k = rem - i * inner;
            const unsigned *r = rows + row * src_stride;
            unsigned v[4];
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                v[u] = r[coltab[i] + k];
                if (++k == inner) {
                    k = 0;
                    ++i;
                }
            }
            st_stream16(out4 + (e0 >> 2), make_uint4(v[0], v[1], v[2], v[3]));
        }
        __syncthreads();
    }
}

// Element gather of one work item: the same flat walk over the destination as the whole-row
// path, four consecutive 32-bit words per lane and two such quads per pass, so stores stay
// contiguous (one 16-byte store per lane where the item's destination is 16-byte aligned) while
// each word comes from src[keep_out[row], keep_in[col], k].  The presence of each list is a
// template parameter so that nothing inside a pass is conditional: all row/column ids of a pass are
// loaded first, then its eight words, then they are stored.
template <bool HAS_OUT, bool HAS_IN>
__device__ __forceinline__ void gather_words(const spice_gather_item &it, const unsigned *__restrict__ src,
                                             unsigned *__restrict__ dst, const long long *__restrict__ idx_out,
                                             const long long *__restrict__ idx_in, bool st4) {
    const int tid = threadIdx.x;
    const int row_len = it.dst_cols * it.inner;
    const int inner = it.inner;
    const int64_t src_row_stride = (int64_t)it.src_cols * inner;
    const int total = it.n_rows * row_len;

    auto locate = [&](int e, int &row, int &i, int &k) {
        row = e / row_len;
        const int rem = e - row * row_len;
        i = inner == 1 ? rem : rem / inner;
        k = rem - i * inner;
    };
    // N words = N/4 quads of four consecutive destination words; one division pair per quad, the
    // other three positions follow by counting (inner index, then column, then row)
    auto fetch_n = [&](int e0, int stride, unsigned *v, auto n_tag) {
        constexpr int N = decltype(n_tag)::value;
        int row[N], i[N], k[N];
        long long so[N], ci[N];
#pragma unroll
        for (int q = 0; q < N / 4; ++q) {
            int r, ii, kk;
            locate(e0 + q * stride, r, ii, kk);
#pragma unroll
            for (int u = 0; u < 4; ++u) {
                row[4 * q + u] = r;
                i[4 * q + u] = ii;
                k[4 * q + u] = kk;
                if (++kk == inner) {
                    kk = 0;
                    if (++ii == it.dst_cols) {
                        ii = 0;
                        ++r;
                    }
                }
            }
        }
#pragma unroll
        for (int u = 0; u < N; ++u) {
            so[u] = HAS_OUT ? idx_out[row[u]] : (long long)(it.row0 + row[u]);
            ci[u] = HAS_IN ? idx_in[i[u]] : (long long)i[u];
        }
#pragma unroll
        for (int u = 0; u < N; ++u) v[u] = src[so[u] * src_row_stride + ci[u] * inner + k[u]];
    };
    using N8 = std::integral_constant<int, 8>;
    using N4 = std::integral_constant<int, 4>;

    int e0 = tid * 4;
    if (st4) {
        for (; e0 + 3 + 4 * kGatherThreads < total; e0 += 8 * kGatherThreads) {
            unsigned v[8];
            fetch_n(e0, 4 * kGatherThreads, v, N8{});
            *reinterpret_cast<uint4 *>(dst + e0) = make_uint4(v[0], v[1], v[2], v[3]);
            *reinterpret_cast<uint4 *>(dst + e0 + 4 * kGatherThreads) = make_uint4(v[4], v[5], v[6], v[7]);
        }
        for (; e0 + 3 < total; e0 += 4 * kGatherThreads) {
            unsigned v[4];
            fetch_n(e0, 0, v, N4{});
            *reinterpret_cast<uint4 *>(dst + e0) = make_uint4(v[0], v[1], v[2], v[3]);
        }
    } else {
        for (; e0 + 3 < total; e0 += 4 * kGatherThreads) {
            unsigned v[4];
            fetch_n(e0, 0, v, N4{});
#pragma unroll
            for (int u = 0; u < 4; ++u) dst[e0 + u] = v[u];
        }
    }
    for (int e = e0; e < total && e < e0 + 4; ++e) {  // at most three words of one lane
        int row, i, k;
        locate(e, row, i, k);
        const long long so = HAS_OUT ? idx_out[row] : (long long)(it.row0 + row);
        const long long ci = HAS_IN ? idx_in[i] : (long long)i;
        dst[e] = src[so * src_row_stride + ci * inner + k];
    }
}

__global__ __launch_bounds__(kGatherThreads) void gather_kernel(
    const unsigned *__restrict__ src_base, unsigned *__restrict__ dst_base,
    const long long *__restrict__ idx_base, const spice_gather_item *__restrict__ items,
    int bases_vec_ok, int have_lds) {
    extern __shared__ unsigned gather_lds[];
    const spice_gather_item it = items[blockIdx.x];
    const int tid = threadIdx.x;
    const int row_len = it.dst_cols * it.inner;
    const int64_t src_row_stride = (int64_t)it.src_cols * it.inner;
    const unsigned *src = src_base + it.src_off;
    unsigned *dst = dst_base + it.dst_off;
    const long long *idx_out = it.idx_out_off >= 0 ? idx_base + it.idx_out_off : nullptr;
    const long long *idx_in = it.idx_in_off >= 0 ? idx_base + it.idx_in_off : nullptr;
    const bool vec = bases_vec_ok && (it.flags & 1);

    if (idx_in == nullptr && vec) {
        // Whole rows, 16 bytes per lane.  The item's destination (n_rows consecutive rows) is one
        // contiguous range, so lanes walk it flat: every store instruction of a wave writes 1 KB of
        // consecutive bytes whatever the row length, and reads are contiguous within each kept row.
        // Four float4 per lane per pass: the four row ids first, then the four loads, then the
        // stores, with no condition inside a pass (a conditional store makes the compiler sink its
        // load next to it, one load in flight at a time); the remainder goes one float4 at a time.
        const int n4 = row_len >> 2;
        const int total4 = it.n_rows * n4;
        const int64_t stride4 = src_row_stride >> 2;
        const uint4 *src4 = reinterpret_cast<const uint4 *>(src);
        uint4 *dst4 = reinterpret_cast<uint4 *>(dst);
        constexpr int kU = 4;
        int f0 = tid;
        for (; f0 + (kU - 1) * kGatherThreads < total4; f0 += kU * kGatherThreads) {
            int col[kU];
            long long so[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) {
                const int f = f0 + u * kGatherThreads;
                const int row = f / n4;
                col[u] = f - row * n4;
                so[u] = idx_out ? idx_out[row] : (long long)(it.row0 + row);
            }
            uint4 v[kU];
#pragma unroll
            for (int u = 0; u < kU; ++u) v[u] = ld_stream16(src4 + (so[u] * stride4 + col[u]));
#pragma unroll
            for (int u = 0; u < kU; ++u) st_stream16(dst4 + (f0 + u * kGatherThreads), v[u]);
        }
        for (; f0 < total4; f0 += kGatherThreads) {
            const int row = f0 / n4;
            const long long so = idx_out ? idx_out[row] : (long long)(it.row0 + row);
            st_stream16(dst4 + f0, ld_stream16(src4 + (so * stride4 + (f0 - row * n4))));
        }
        return;
    }
    if (idx_in != nullptr && bases_vec_ok && have_lds && (it.flags & 4)) {
        gather_staged(it, src, dst, idx_out, idx_in, gather_lds);
        return;
    }
    // Element gather (an in-channel list is present, or the layout rules out 16-byte moves).
    const bool st4 = bases_vec_ok && (it.flags & 2);
    if (idx_out) {
        if (idx_in) gather_words<true, true>(it, src, dst, idx_out, idx_in, st4);
        else gather_words<true, false>(it, src, dst, idx_out, idx_in, st4);
    } else {
        if (idx_in) gather_words<false, true>(it, src, dst, idx_out, idx_in, st4);
        else gather_words<false, false>(it, src, dst, idx_out, idx_in, st4);
    }
}

}  // namespace spice

using namespace spice;

extern "C" int64_t spice_gather_plan(const spice_gather_desc *tensors, int32_t n_tensors,
                                     spice_gather_item *items, int64_t capacity) {
    if (n_tensors < 0 || (n_tensors > 0 && tensors == nullptr)) return -SPICE_ERR_ARG;
    int64_t n = 0;
    const int64_t target = gather_target_elems();
    for (int32_t t = 0; t < n_tensors; ++t) {
        const spice_gather_desc &d = tensors[t];
        if (d.dst_rows < 0 || d.dst_cols < 0 || d.src_cols < 0 || d.inner < 1 || d.src_off < 0 ||
            d.dst_off < 0 || d.idx_out_off < -1 || d.idx_in_off < -1)
            return -SPICE_ERR_ARG;
        if (d.idx_in_off < 0 && d.dst_cols != d.src_cols) return -SPICE_ERR_ARG;
        if (d.dst_cols > d.src_cols) return -SPICE_ERR_ARG;
        const int64_t row_len = d.dst_cols * d.inner;
        const int64_t src_stride = d.src_cols * d.inner;
        // the kernel walks an item with 32-bit offsets and looks a few thousand words ahead
        if (src_stride > INT32_MAX - 8192 || d.dst_rows > INT32_MAX) return -SPICE_ERR_UNSUPPORTED;
        if (d.dst_rows == 0 || row_len == 0) continue;
        const bool vec = d.idx_in_off < 0 && (row_len % 4 == 0) && (d.src_off % 4 == 0) &&
                         (d.dst_off % 4 == 0);
        // rows per item: the kernel walks an item's destination as one contiguous range, so any
        // row count will do; small items keep many workgroups in flight
        // staged path: in-channel list, 16-byte aligned source and destination rows, rows that fit LDS
        const bool stage = gather_stage_enabled() && d.idx_in_off >= 0 && (row_len % 4 == 0) && (src_stride % 4 == 0) && (d.src_off % 4 == 0) &&
                           (d.dst_off % 4 == 0) && src_stride <= kGatherStageWords && d.dst_cols <= kGatherStageCols;
        int64_t rpi = (stage ? gather_stage_target_elems() : target) / row_len;
        if (rpi < 1) rpi = 1;
        const int64_t n_it = ceil_div(d.dst_rows, rpi);
        rpi = ceil_div(d.dst_rows, n_it);
        for (int64_t r0 = 0; r0 < d.dst_rows; r0 += rpi, ++n) {
            if (items == nullptr || n >= capacity) continue;
            spice_gather_item &it = items[n];
            it.src_off = d.src_off;
            it.dst_off = d.dst_off + r0 * row_len;
            it.idx_out_off = d.idx_out_off >= 0 ? d.idx_out_off + r0 : -1;
            it.idx_in_off = d.idx_in_off;
            it.row0 = (int32_t)r0;
            it.n_rows = (int32_t)((d.dst_rows - r0 < rpi) ? d.dst_rows - r0 : rpi);
            it.dst_cols = (int32_t)d.dst_cols;
            it.src_cols = (int32_t)d.src_cols;
            it.inner = (int32_t)d.inner;
            it.lanes = 0;
            it.flags = (vec ? 1 : 0) | (((d.dst_off + r0 * row_len) % 4 == 0) ? 2 : 0) | (stage ? 4 : 0);
            it.pad_ = 0;
        }
    }
    return n;
}

extern "C" int spice_gather_run(const float *src_base, float *dst_base, const int64_t *idx_base,
                                const spice_gather_item *d_items, int64_t n_items, int32_t plan_flags,
                                void *stream) {
    if (n_items < 0 || n_items > INT32_MAX) return SPICE_ERR_ARG;
    if (n_items == 0) return SPICE_OK;
    if (!src_base || !dst_base || !d_items) return SPICE_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(src_base) & 3u) || (reinterpret_cast<uintptr_t>(dst_base) & 3u))
        return SPICE_ERR_ARG;
    const int vec_ok = aligned16(src_base) && aligned16(dst_base);
    // LDS only when the plan holds staged items: the allocation limits the workgroups per CU
    const size_t lds = (plan_flags & 4) ? kGatherLdsBytes : 0;
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)n_items), dim3(kGatherThreads), lds,
                       static_cast<hipStream_t>(stream), reinterpret_cast<const unsigned *>(src_base),
                       reinterpret_cast<unsigned *>(dst_base),
                       reinterpret_cast<const long long *>(idx_base), d_items, vec_ok, lds ? 1 : 0);
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}

// ---- launch plans to the device without a copy command ----
//
// A copy command from page-locked host memory, queued on a stream that already holds launches, keeps the calling
// thread until the stream has run that far (measured on MI355X / ROCm 7: up to 5 ms per call inside a candidate
// loop that is a few milliseconds ahead of the device).  A kernel that reads the host memory through its device
// mapping is a launch like any other: the host moves on at once, and the bytes cross the bus when the stream
// gets there.
namespace spice {
__global__ __launch_bounds__(256) void gather_upload_kernel(uint4 *__restrict__ dst, const uint4 *__restrict__ src, long long n16) {
    const long long step = (long long)gridDim.x * blockDim.x;
    for (long long i = (long long)blockIdx.x * blockDim.x + threadIdx.x; i < n16; i += step) dst[i] = src[i];
}
}  // namespace spice

extern "C" int spice_gather_upload(void *dst, const void *src_host, int64_t nbytes, void *stream) {
    if (nbytes < 0) return SPICE_ERR_ARG;
    if (nbytes == 0) return SPICE_OK;
    if (!dst || !src_host || (nbytes & 15) || !aligned16(dst) || !aligned16(src_host)) return SPICE_ERR_ARG;
    void *mapped = nullptr;
    if (hipHostGetDevicePointer(&mapped, const_cast<void *>(src_host), 0) != hipSuccess || mapped == nullptr) {
        (void)hipGetLastError();
        return SPICE_ERR_ARG;   // not page-locked memory the device can read
    }
    const long long n16 = nbytes >> 4;
    const unsigned blocks = (unsigned)((n16 + 255) / 256 < 1024 ? (n16 + 255) / 256 : 1024);
    hipLaunchKernelGGL(spice::gather_upload_kernel, dim3(blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       static_cast<uint4 *>(dst), static_cast<const uint4 *>(mapped), n16);
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}
