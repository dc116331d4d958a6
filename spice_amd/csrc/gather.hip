// Kernel (4): channel gather.  dst[o, i, k] = src[keep_out[o], keep_in[i], k] for every tensor
// of a model in one launch.  Pure HBM-bound copy of 32-bit words (bit-exact); kept rows are
// copied with 16-byte accesses when no input-channel list breaks their contiguity.
#include "common.h"

#include <stdlib.h>

namespace spice {

constexpr int kGatherThreads = 256;
constexpr int64_t kGatherDefaultTargetElems = 1024;  // measured best on MI355X (1 K: 24.2 us, 4 K: 27.1 us, 16 K: 32 us for 53 MB)

static int64_t gather_target_elems() {
    const char *e = getenv("SPICE_GATHER_TARGET_ELEMS");  // tuning knob (host side only)
    const long v = e ? atol(e) : 0;
    return (v >= 256 && v <= (1 << 22)) ? v : kGatherDefaultTargetElems;
}

__global__ __launch_bounds__(kGatherThreads) void gather_kernel(
    const unsigned *__restrict__ src_base, unsigned *__restrict__ dst_base,
    const long long *__restrict__ idx_base, const spice_gather_item *__restrict__ items,
    int bases_vec_ok) {
    const spice_gather_item it = items[blockIdx.x];
    const int L = it.lanes;
    const int tid = threadIdx.x;
    const int sub = tid / L;
    const int lane = tid & (L - 1);
    const int groups = kGatherThreads / L;
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
            for (int u = 0; u < kU; ++u) v[u] = src4[so[u] * stride4 + col[u]];
#pragma unroll
            for (int u = 0; u < kU; ++u) dst4[f0 + u * kGatherThreads] = v[u];
        }
        for (; f0 < total4; f0 += kGatherThreads) {
            const int row = f0 / n4;
            const long long so = idx_out ? idx_out[row] : (long long)(it.row0 + row);
            dst4[f0] = src4[so * stride4 + (f0 - row * n4)];
        }
        return;
    }
    for (int r = sub; r < it.n_rows; r += groups) {
        const int64_t so = idx_out ? (int64_t)idx_out[r] : (int64_t)(it.row0 + r);
        const unsigned *srow = src + so * src_row_stride;
        unsigned *drow = dst + (int64_t)r * row_len;
        if (idx_in == nullptr) {
            for (int j = lane; j < row_len; j += L) drow[j] = srow[j];
        } else if (it.inner == 1) {
            for (int j = lane; j < row_len; j += L) drow[j] = srow[idx_in[j]];
        } else {
            const int inner = it.inner;
            for (int j = lane; j < row_len; j += L) {
                const int i = j / inner;
                const int k = j - i * inner;
                drow[j] = srow[idx_in[i] * (int64_t)inner + k];
            }
        }
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
        if (src_stride > INT32_MAX || d.dst_rows > INT32_MAX) return -SPICE_ERR_UNSUPPORTED;
        if (d.dst_rows == 0 || row_len == 0) continue;
        const bool vec = d.idx_in_off < 0 && (row_len % 4 == 0) && (d.src_off % 4 == 0) &&
                         (d.dst_off % 4 == 0);
        // lanes per row: every lane should own at least four (vector) copies of its row
        const int64_t units = (vec ? row_len / 4 : row_len) / 4;
        int L = kWave;
        while (L > 1 && L > units) L >>= 1;
        const int64_t groups = kGatherThreads / L;
        // rows per item: whole lane-group passes on the indexed path; any count on the flat
        // whole-row path, which walks the item's destination as one contiguous range
        const int64_t unit = vec ? 1 : groups;
        int64_t rpi = target / row_len;
        if (rpi < 1) rpi = 1;
        rpi = ceil_div(rpi, unit) * unit;
        const int64_t n_it = ceil_div(d.dst_rows, rpi);
        rpi = ceil_div(ceil_div(d.dst_rows, n_it), unit) * unit;
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
            it.lanes = L;
            it.flags = vec ? 1 : 0;
            it.pad_ = 0;
        }
    }
    return n;
}

extern "C" int spice_gather_run(const float *src_base, float *dst_base, const int64_t *idx_base,
                                const spice_gather_item *d_items, int64_t n_items, void *stream) {
    if (n_items < 0 || n_items > INT32_MAX) return SPICE_ERR_ARG;
    if (n_items == 0) return SPICE_OK;
    if (!src_base || !dst_base || !d_items) return SPICE_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(src_base) & 3u) || (reinterpret_cast<uintptr_t>(dst_base) & 3u))
        return SPICE_ERR_ARG;
    const int vec_ok = aligned16(src_base) && aligned16(dst_base);
    hipLaunchKernelGGL(gather_kernel, dim3((unsigned)n_items), dim3(kGatherThreads), 0,
                       static_cast<hipStream_t>(stream), reinterpret_cast<const unsigned *>(src_base),
                       reinterpret_cast<unsigned *>(dst_base),
                       reinterpret_cast<const long long *>(idx_base), d_items, vec_ok);
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}
