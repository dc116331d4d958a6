// The pieces of the confusion-matrix update that more than one kernel file uses: the layout of the
// privatised LDS histogram, the scratch table of per-workgroup totals, and the common end of every
// histogram kernel (totals stored write-through, a ticket per group of workgroups, the last arriver
// of a group adds the group's rows into the caller's int64 matrix).
#pragma once

#include "common.h"

namespace spice {

constexpr int kConfThreads = 512;
constexpr int kConfMaxBlocks = 1024;
constexpr int kConfMaxGroups = 128;
constexpr int kConfLdsBins = 16384;  // 64 KiB of uint32 bins per workgroup at most
constexpr int64_t kConfMaxPixels = (int64_t)1 << 40;  // a workgroup's uint32 totals cannot wrap below this

// Scratch layout (bytes): [0,8) int64 count of out-of-range predictions, [8,16) reserved,
// [16, 16 + 4*kConfMaxGroups) one arrival counter per group of workgroups (zero between calls),
// from kConfHeader on the table of per-workgroup totals, [kConfMaxBlocks][C*C] uint32.
constexpr size_t kConfHeader = 4096;

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

SPICE_ENV_KNOB(knob_conf_group, "SPICE_CONF_GROUP", 0)

// Workgroups per ticket group: the last arriver of a group reads `group` rows of the table and
// issues one 64-bit atomic per non-zero bin, so larger groups mean fewer atomics on the caller's
// matrix and a longer read for the last arriver.
static inline int conf_group_size(int blocks) {
    const long v = knob_conf_group();
    int g = (v >= 1) ? (int)v : 16;
    const int need = (int)ceil_div(blocks, kConfMaxGroups);
    if (g < need) g = need;
    return g;
}

struct ConfTail {
    unsigned *partial;              // [gridDim.x][bins]
    unsigned *tickets;              // [ceil(gridDim.x / group)]
    unsigned long long *conf;       // [bins], the caller's matrix
    unsigned long long *bad_total;  // count of out-of-range predictions
    int group;
};

// Run by the last arriver of a ticket group: the group's rows of the table (agent-scope loads: never a stale
// cached line), `members` of them from row `first` on, summed per bin and added to the caller's matrix.  Eight rows
// are requested at a time: the loads of a bin are independent, and what this costs is round trips, not bytes.
template <int THREADS>
__device__ __forceinline__ void conf_fold_group(const ConfTail &t, int first, int members, int bins) {
    const int tid = threadIdx.x;
    for (int b = tid; b < bins; b += THREADS) {
        const unsigned *col = t.partial + (int64_t)first * bins + b;
        unsigned long long s = 0;
        int m = 0;
        for (; m + 7 < members; m += 8) {
            unsigned v[8];
#pragma unroll
            for (int k = 0; k < 8; ++k)
                v[k] = __hip_atomic_load(col + (int64_t)(m + k) * bins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
#pragma unroll
            for (int k = 0; k < 8; ++k) s += v[k];
        }
        for (; m < members; ++m)
            s += __hip_atomic_load(col + (int64_t)m * bins, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (s) atomicAdd(t.conf + b, s);
    }
}

// Common end of the histogram kernels (workgroups of THREADS threads).  On entry every LDS atomic of the workgroup has been
// issued (the caller has not synchronised yet).  The totals of the privatised copies go to this
// workgroup's row of the table as agent-scope stores (write-through, so no release fence), one
// lane takes the group's ticket after the stores have drained, and the last arriver adds the
// group's rows (agent-scope loads: never a stale cached line) into the caller's matrix and
// returns the ticket to zero for the next call.
template <int THREADS>
__device__ __forceinline__ void conf_finish(const unsigned *lds_hist, int copies, int stride, int bins,
                                            unsigned bad, const ConfTail &t) {
    __shared__ int last_flag;
    __shared__ unsigned lds_bad;
    const int tid = threadIdx.x;
    if (tid == 0) lds_bad = 0;
    __syncthreads();
    if (bad) atomicAdd(&lds_bad, bad);
    unsigned *row = t.partial + (int64_t)blockIdx.x * bins;
    for (int b = tid; b < bins; b += THREADS) {
        unsigned s = 0;
        for (int c = 0; c < copies; ++c) s += lds_hist[c * stride + b];
        __hip_atomic_store(row + b, s, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    const int g = blockIdx.x / t.group;
    const int first = g * t.group;
    const int members = min(t.group, (int)gridDim.x - first);
    if (tid == 0) {
        if (lds_bad) atomicAdd(t.bad_total, (unsigned long long)lds_bad);
        const unsigned k = __hip_atomic_fetch_add(t.tickets + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (k == (unsigned)(members - 1));
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            __hip_atomic_store(t.tickets + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        last_flag = last;
    }
    __syncthreads();
    if (!last_flag) return;
    conf_fold_group<THREADS>(t, first, members, bins);
}

// The scratch area (caller workspace or a static array) seen as the table, the tickets and the counter.
static ConfTail conf_tail(void *scratch, int64_t *conf, int64_t *invalid_count, int blocks) {
    char *ws = static_cast<char *>(scratch);
    ConfTail t;
    t.partial = reinterpret_cast<unsigned *>(ws + kConfHeader);
    t.tickets = reinterpret_cast<unsigned *>(ws + 16);
    t.conf = reinterpret_cast<unsigned long long *>(conf);
    t.bad_total = reinterpret_cast<unsigned long long *>(invalid_count ? (void *)invalid_count : (void *)ws);
    t.group = conf_group_size(blocks);
    return t;
}

}  // namespace spice
