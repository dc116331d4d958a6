// Kernel (2): fused KD-Lite loss.  One coalesced pass over the student and teacher logits
// ([N,C,H,W] fp32, class stride H*W) computes, per pixel, the temperature-softmax KL term,
// the cross-entropy term and (optionally) d(loss)/d(student); all softmax statistics are kept
// in registers by an online (running-maximum) formulation, so every logit is read once for the
// loss (and once more, L2/MALL-warm, when the gradient is requested).
//
// Per pixel, with k = 1/T and centred logits  s'_c = s_c - max s,  t'_c = t_c - max t:
//   Zs = sum exp(k s'_c)   Zt = sum exp(k t'_c)   Z1 = sum exp(s'_c)
//   A  = sum exp(k t'_c) (t'_c - s'_c)
//   KL(softmax(t/T) || softmax(s/T)) = k A / Zt - ln Zt + ln Zs
//   nll = ln Z1 - s'_label
// Sums over pixels are carried in fp64 from the thread level up and combined in a fixed
// order (thread -> wave butterfly -> workgroup -> one finalising workgroup): deterministic.
#include "common.h"
#include "bilinear_math.h"
#include "conf_tail.h"

#include <hip/hip_fp16.h>

#include <stdlib.h>

namespace spice {

constexpr int kKdThreads = 256;
constexpr int kKdMaxBlocks = 2048;  // capacity of the partial-sum tables in the workspace
constexpr size_t kKdMaxStageBytes = 48 * 1024;  // LDS stage of the gradient pass (C <= 24)
constexpr float kNegFill = -1.0e30f;
constexpr float kLog2e = 1.4426950408889634f;
constexpr float kLn2 = 0.6931471805599453f;

// workspace layout (bytes)
constexpr size_t kKdOffBad = 0;       // int64: labels neither ignore_index nor in [0,C)
constexpr size_t kKdOffTicket = 8;    // uint32 arrival counter of the main kernel (returns to 0)
constexpr size_t kKdOffTotals = 16;   // 8 doubles: the eight global sums
constexpr size_t kKdOffPre = 128;     // kKdMaxBlocks x 4 doubles
constexpr size_t kKdOffMain = kKdOffPre + (size_t)kKdMaxBlocks * 4 * 8;  // kKdMaxBlocks x 8 doubles
constexpr size_t kKdWorkspace = kKdOffMain + (size_t)kKdMaxBlocks * 8 * 8;

struct KdArgs {
    const float *s;
    const float *t;
    const long long *labels;
    float *grad;
    const float *class_w;
    const float *upstream;
    double *pre_partial;
    double *main_partial;
    long long *bad;
    unsigned *ticket;
    double *totals;
    float *out3;
    long long N, HW;
    int C, H, W;
    int pre_blocks;
    long long ignore_index;
    float invT, T, alpha, bw, coef_ce, coef_kd;
    int conf;
    int pre_ready;   // host side only: the label pass has run into this workspace already (spice_kdlite_label_pass)
};

__device__ __forceinline__ float fast_exp2(float x) { return __builtin_amdgcn_exp2f(x); }
__device__ __forceinline__ float fast_ln(float x) { return __builtin_amdgcn_logf(x) * kLn2; }

template <int VEC>
struct VecF;
template <>
struct VecF<4> {
    typedef float4 type;
};
template <>
struct VecF<1> {
    typedef float type;
};

template <int VEC>
__device__ __forceinline__ void load_vec(const float *p, float (&v)[VEC]) {
    if (VEC == 4) {
        const float4 x = *reinterpret_cast<const float4 *>(p);
        v[0] = x.x;
        v[1 % VEC] = x.y;
        v[2 % VEC] = x.z;
        v[3 % VEC] = x.w;
    } else {
        v[0] = *p;
    }
}

// Streaming variant: the logits are read once and never again, so they need not displace anything
// in the caches (non-temporal load policy).
template <int VEC>
__device__ __forceinline__ void load_vec_nt(const float *p, float (&v)[VEC]) {
    if (VEC == 4) {
        typedef float f4 __attribute__((ext_vector_type(4)));
        const f4 x = __builtin_nontemporal_load(reinterpret_cast<const f4 *>(p));
        v[0] = x.x;
        v[1 % VEC] = x.y;
        v[2 % VEC] = x.z;
        v[3 % VEC] = x.w;
    } else {
        v[0] = __builtin_nontemporal_load(p);
    }
}

template <int VEC>
__device__ __forceinline__ void store_vec(float *p, const float (&v)[VEC]) {
    if (VEC == 4) {
        *reinterpret_cast<float4 *>(p) = make_float4(v[0], v[1 % VEC], v[2 % VEC], v[3 % VEC]);
    } else {
        *p = v[0];
    }
}

template <int VEC>
__device__ __forceinline__ void load_labels(const long long *p, long long (&v)[VEC]) {
    if (VEC == 4) {
        const longlong2 a = *reinterpret_cast<const longlong2 *>(p);
        const longlong2 b = *reinterpret_cast<const longlong2 *>(p + 2);
        v[0] = a.x;
        v[1 % VEC] = a.y;
        v[2 % VEC] = b.x;
        v[3 % VEC] = b.y;
    } else {
        v[0] = *p;
    }
}

// 3x3 dilate != erode on the label image (borders use in-bounds neighbours only) for the VEC
// consecutive pixels that start at offset r of one image.  Out-of-image neighbours are replaced
// by clamped coordinates: the clamped pixel is itself inside the 3x3 window, so minimum and
// maximum do not change.  When the group lies in one image row its windows share columns: three
// rows of VEC + 2 labels are read once and reduced column-wise first.
template <int VEC>
__device__ __forceinline__ void label_boundary_group(const long long *__restrict__ img, int r, int H, int W,
                                                     bool (&edge)[VEC]) {
    const int y = r / W;
    const int x = r - y * W;
    if (VEC > 1 && x + VEC - 1 < W) {
        long long cmn[VEC + 2], cmx[VEC + 2];
        const long long *rows[3] = {img + (long long)max(y - 1, 0) * W, img + (long long)y * W,
                                    img + (long long)min(y + 1, H - 1) * W};
#pragma unroll
        for (int j = 0; j < VEC + 2; ++j) {
            const int xx = min(max(x - 1 + j, 0), W - 1);
            const long long a = rows[0][xx], b = rows[1][xx], c = rows[2][xx];
            cmn[j] = a < b ? (a < c ? a : c) : (b < c ? b : c);
            cmx[j] = a > b ? (a > c ? a : c) : (b > c ? b : c);
        }
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            long long mn = cmn[e] < cmn[e + 1] ? cmn[e] : cmn[e + 1];
            long long mx = cmx[e] > cmx[e + 1] ? cmx[e] : cmx[e + 1];
            mn = cmn[e + 2] < mn ? cmn[e + 2] : mn;
            mx = cmx[e + 2] > mx ? cmx[e + 2] : mx;
            edge[e] = mn != mx;
        }
        return;
    }
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        const int pe = r + e;
        const int ye = pe / W;
        const int xe = pe - ye * W;
        long long mn = img[pe], mx = mn;
#pragma unroll
        for (int dy = -1; dy <= 1; ++dy) {
            const long long *row = img + (long long)min(max(ye + dy, 0), H - 1) * W;
#pragma unroll
            for (int dx = -1; dx <= 1; ++dx) {
                const long long v = row[min(max(xe + dx, 0), W - 1)];
                mn = v < mn ? v : mn;
                mx = v > mx ? v : mx;
            }
        }
        edge[e] = mn != mx;
    }
}

// Sum `NQ` per-thread doubles over the workgroup; thread 0..NQ-1 of wave 0 end up holding the
// totals in `out[k]` order.  red: [kKdThreads/64][8] doubles of LDS.
template <int NQ>
__device__ __forceinline__ void block_sum_store(const double (&acc)[NQ], double (*red)[8],
                                                double *out) {
    const int wave = threadIdx.x >> 6;
    const int lane = threadIdx.x & 63;
#pragma unroll
    for (int k = 0; k < NQ; ++k) {
        const double v = wave_sum(acc[k]);
        if (lane == 0) red[wave][k] = v;
    }
    __syncthreads();
    if (threadIdx.x < NQ) {
        double v = 0.0;
        for (int wv = 0; wv < kKdThreads / 64; ++wv) v += red[wv][threadIdx.x];
        out[threadIdx.x] = v;
    }
}

// Work decomposition shared by the pre-pass and the main kernel: a tile is 256 consecutive
// pixel groups (VEC pixels each) of ONE image, so the image index and every class-plane base
// address are wave-uniform (scalar registers) and a lane only carries its 32-bit pixel offset.
struct TileGeom {
    int per_img;        // pixel groups per image = HW / VEC
    int tiles_per_img;  // ceil(per_img / 256)
    int n_tiles;        // N * tiles_per_img
};

template <int VEC>
__device__ __forceinline__ TileGeom tile_geom(const KdArgs &a) {
    TileGeom g;
    g.per_img = (int)(a.HW / VEC);
    g.tiles_per_img = (g.per_img + kKdThreads - 1) / kKdThreads;
    g.n_tiles = (int)a.N * g.tiles_per_img;
    return g;
}

// Labels as 32-bit class ids: >= 0 in range, -1 ignore_index, -2 any other out-of-range value.
template <int VEC>
__device__ __forceinline__ void load_label_ids(const long long *p, long long ignore_index, int C,
                                               int (&id)[VEC]) {
    long long lab[VEC];
    load_labels<VEC>(p, lab);
#pragma unroll
    for (int e = 0; e < VEC; ++e)
        id[e] = (lab[e] == ignore_index) ? -1 : ((lab[e] >= 0 && lab[e] < C) ? (int)lab[e] : -2);
}

// Pre-pass (only when a gradient is requested): the loss normalisers, which depend on the
// labels (and, for confidence weighting, on the teacher) but not on the student.
// Per workgroup: {n_valid, sum class weight, n_boundary&valid, sum confidence&valid}.
template <int VEC, bool EXT>
__global__ __launch_bounds__(kKdThreads) void kdlite_prepass_kernel(KdArgs a) {
    __shared__ double red[kKdThreads / 64][8];
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    const TileGeom tg = tile_geom<VEC>(a);
    const bool use_b = EXT && a.bw > 1.0f;
    const bool use_c = EXT && a.conf;
    for (int tile = blockIdx.x; tile < tg.n_tiles; tile += gridDim.x) {
        const int n = tile / tg.tiles_per_img;
        const int qi = (tile - n * tg.tiles_per_img) * kKdThreads + (int)threadIdx.x;
        if (qi >= tg.per_img) continue;
        const int r = qi * VEC;
        const long long *lab_img = a.labels + (long long)n * a.HW;
        int id[VEC];
        load_label_ids<VEC>(lab_img + r, a.ignore_index, a.C, id);
        float zt1[VEC];
        if (use_c) {
            float mt[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                mt[e] = kNegFill;
                zt1[e] = 0.f;
            }
            const float *tb = a.t + (long long)n * a.C * a.HW;
            for (int c = 0; c < a.C; ++c) {
                float tv[VEC];
                load_vec<VEC>(tb + (long long)c * a.HW + r, tv);
#pragma unroll
                for (int e = 0; e < VEC; ++e) {
                    const float mn = fmaxf(mt[e], tv[e]);
                    zt1[e] = zt1[e] * fast_exp2((mt[e] - mn) * kLog2e) + fast_exp2((tv[e] - mn) * kLog2e);
                    mt[e] = mn;
                }
            }
        }
        float nv = 0.f, ws = 0.f, nb = 0.f, cs = 0.f;
        bool edge[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) edge[e] = false;
        if (use_b) label_boundary_group<VEC>(lab_img, r, a.H, a.W, edge);
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const bool valid = id[e] != -1;
            if (valid) nv += 1.f;
            if (id[e] >= 0) ws += a.class_w ? a.class_w[id[e]] : 1.f;
            if (use_b && valid && edge[e]) nb += 1.f;
            if (use_c && valid) cs += 1.f / zt1[e];
        }
        acc[0] += nv;
        acc[1] += ws;
        acc[2] += nb;
        acc[3] += cs;
    }
    block_sum_store<4>(acc, red, a.pre_partial + (long long)blockIdx.x * 4);
}

// The same normalisers when neither the boundary nor the confidence option is on: they depend on the
// labels alone, so the pre-pass is a flat streaming count over all N*H*W labels (16-byte loads, eight
// labels per lane in flight per step; default load policy: the main pass reads the labels again).
constexpr int kKdCountMaxBlocks = 1024;

__device__ __forceinline__ void kd_count_label(long long lab, const KdArgs &a, float &nv, float &ws) {
    if (lab == a.ignore_index) return;
    nv += 1.f;   // every pixel that is not ignored takes part in the distillation term
    if (lab >= 0 && lab < a.C) ws += a.class_w ? a.class_w[lab] : 1.f;
}

__global__ __launch_bounds__(kKdThreads) void kdlite_label_count_kernel(KdArgs a) {
    __shared__ double red[kKdThreads / 64][8];
    const long long P = a.N * a.HW;
    const long long stride = (long long)gridDim.x * kKdThreads;
    const long long gtid = (long long)blockIdx.x * kKdThreads + threadIdx.x;
    float nv = 0.f, ws = 0.f;   // a lane sees at most P / stride labels: exact in fp32 far beyond any image batch
    const bool vec = (reinterpret_cast<uintptr_t>(a.labels) & 15u) == 0;
    const long long P2 = vec ? (P >> 1) : 0;
    const longlong2 *l2 = reinterpret_cast<const longlong2 *>(a.labels);
    long long i = gtid;
    for (; i + 3 * stride < P2; i += 4 * stride) {
        const longlong2 x0 = l2[i], x1 = l2[i + stride], x2 = l2[i + 2 * stride], x3 = l2[i + 3 * stride];
        kd_count_label(x0.x, a, nv, ws);
        kd_count_label(x0.y, a, nv, ws);
        kd_count_label(x1.x, a, nv, ws);
        kd_count_label(x1.y, a, nv, ws);
        kd_count_label(x2.x, a, nv, ws);
        kd_count_label(x2.y, a, nv, ws);
        kd_count_label(x3.x, a, nv, ws);
        kd_count_label(x3.y, a, nv, ws);
    }
    for (; i < P2; i += stride) {
        const longlong2 x0 = l2[i];
        kd_count_label(x0.x, a, nv, ws);
        kd_count_label(x0.y, a, nv, ws);
    }
    for (long long j = 2 * P2 + gtid; j < P; j += stride) kd_count_label(a.labels[j], a, nv, ws);
    const double acc[4] = {(double)nv, (double)ws, 0.0, 0.0};
    block_sum_store<4>(acc, red, a.pre_partial + (long long)blockIdx.x * 4);
}

// Launches the pre-pass that fits the options and records its workgroup count in `a` (the main kernel
// sums that many rows of the table).
template <int VEC>
static void kd_launch_prepass(KdArgs &a, bool ext, hipStream_t st) {
    if (ext) {
        hipLaunchKernelGGL((kdlite_prepass_kernel<VEC, true>), dim3(a.pre_blocks), dim3(kKdThreads), 0, st, a);
        return;
    }
    long long g = (a.N * a.HW + (long long)kKdThreads * 16 - 1) / ((long long)kKdThreads * 16);
    if (g < 1) g = 1;
    if (g > kKdCountMaxBlocks) g = kKdCountMaxBlocks;
    a.pre_blocks = (int)g;   // a function of the label count alone: the same whoever launches the pass
    if (a.pre_ready) return;
    hipLaunchKernelGGL(kdlite_label_count_kernel, dim3((unsigned)g), dim3(kKdThreads), 0, st, a);
}

template <int VEC>
struct PixStats {
    float ms[VEC], mt[VEC], zs[VEC], z1[VEC], zt[VEC], zt1[VEC], at[VEC], sy[VEC];
};

// Fold the U class planes c0..c0+U-1, whose logits are in sv / tv, into the running statistics.
template <int VEC, int U, bool CONF>
__device__ __forceinline__ void kd_fold_stats(PixStats<VEC> &st, const float (&sv)[U][VEC], const float (&tv)[U][VEC],
                                              int c0, const int (&id)[VEC], float k2) {
#pragma unroll
    for (int e = 0; e < VEC; ++e) {
        float msn = fmaxf(fmaxf(st.ms[e], fmaxf(sv[0][e], sv[1 % U][e])), fmaxf(sv[2 % U][e], sv[3 % U][e]));
        float mtn = fmaxf(fmaxf(st.mt[e], fmaxf(tv[0][e], tv[1 % U][e])), fmaxf(tv[2 % U][e], tv[3 % U][e]));
#pragma unroll
        for (int u = 4; u < U; u += 2) {
            msn = fmaxf(msn, fmaxf(sv[u][e], sv[(u + 1) % U][e]));
            mtn = fmaxf(mtn, fmaxf(tv[u][e], tv[(u + 1) % U][e]));
        }
        const float ds = st.ms[e] - msn;  // <= 0
        const float dt = st.mt[e] - mtn;  // <= 0
        const float rt = fast_exp2(dt * k2);
        float Zs = st.zs[e] * fast_exp2(ds * k2);
        float Z1 = st.z1[e] * fast_exp2(ds * kLog2e);
        float At = rt * (st.at[e] + st.zt[e] * (dt - ds));
        float Zt = st.zt[e] * rt;
        float Zt1 = 0.f;
        if (CONF) Zt1 = st.zt1[e] * fast_exp2(dt * kLog2e);
        float sy = st.sy[e];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float sc = sv[u][e] - msn;
            const float tc = tv[u][e] - mtn;
            const float et = fast_exp2(tc * k2);
            Zs += fast_exp2(sc * k2);
            Z1 += fast_exp2(sc * kLog2e);
            Zt += et;
            At += et * (tc - sc);
            if (CONF) Zt1 += fast_exp2(tc * kLog2e);
            sy = (c0 + u == id[e]) ? sv[u][e] : sy;
        }
        st.ms[e] = msn;
        st.mt[e] = mtn;
        st.zs[e] = Zs;
        st.z1[e] = Z1;
        st.zt[e] = Zt;
        st.at[e] = At;
        st.zt1[e] = Zt1;
        st.sy[e] = sy;
    }
}

// Fold U class planes (c0..c0+U-1) into the running softmax statistics of VEC pixels.
// FULL = all U classes exist; otherwise loads are clamped to the last class and the surplus
// values replaced by a large negative fill (their exponentials are exactly zero).
// sv / tv receive the raw logits of the step (the caller of the register-resident gradient pass
// keeps them; everyone else passes scratch arrays).
// STAGE: the raw logits are also parked in LDS (slot (2c+which)*threads + tid, touched by this
// thread only) for the gradient pass.
template <int VEC, int U, bool FULL, bool CONF, bool STAGE>
__device__ __forceinline__ void kd_fold(PixStats<VEC> &st, const float *__restrict__ s_img,
                                        const float *__restrict__ t_img, long long HW, int r,
                                        int c0, int C, const int (&id)[VEC], float k2,
                                        typename VecF<VEC>::type *stage, float (&sv)[U][VEC],
                                        float (&tv)[U][VEC]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int cc = FULL ? c0 + u : min(c0 + u, C - 1);
        load_vec_nt<VEC>(s_img + (long long)cc * HW + r, sv[u]);
        load_vec_nt<VEC>(t_img + (long long)cc * HW + r, tv[u]);
    }
    if (STAGE) {
#pragma unroll
        for (int u = 0; u < U; ++u) {
            if (!FULL && c0 + u >= C) break;
            store_vec<VEC>(reinterpret_cast<float *>(stage + (2 * (c0 + u)) * kKdThreads + threadIdx.x), sv[u]);
            store_vec<VEC>(reinterpret_cast<float *>(stage + (2 * (c0 + u) + 1) * kKdThreads + threadIdx.x), tv[u]);
        }
    }
    if (!FULL) {
#pragma unroll
        for (int u = 1; u < U; ++u) {
            const bool gone = c0 + u >= C;
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                sv[u][e] = gone ? kNegFill : sv[u][e];
                tv[u][e] = gone ? kNegFill : tv[u][e];
            }
        }
    }
    kd_fold_stats<VEC, U, CONF>(st, sv, tv, c0, id, k2);
}

// Gradient of four class planes: softmax probabilities recomputed from the final statistics.
template <int VEC, bool FULL, bool STAGE>
__device__ __forceinline__ void kd_grad4(const PixStats<VEC> &st, const float *__restrict__ s_img,
                                         const float *__restrict__ t_img, float *__restrict__ g_img,
                                         long long HW, int r, int c0, int C, const int (&id)[VEC],
                                         float k2, const float (&gce)[VEC], const float (&gkd)[VEC],
                                         const float (&izs)[VEC], const float (&iz1)[VEC],
                                         const float (&izt)[VEC],
                                         const typename VecF<VEC>::type *stage) {
    float sv[4][VEC], tv[4][VEC];
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        const int cc = FULL ? c0 + u : min(c0 + u, C - 1);
        if (STAGE) {
            load_vec<VEC>(reinterpret_cast<const float *>(stage + (2 * cc) * kKdThreads + threadIdx.x), sv[u]);
            load_vec<VEC>(reinterpret_cast<const float *>(stage + (2 * cc + 1) * kKdThreads + threadIdx.x), tv[u]);
        } else {
            load_vec<VEC>(s_img + (long long)cc * HW + r, sv[u]);
            load_vec<VEC>(t_img + (long long)cc * HW + r, tv[u]);
        }
    }
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        if (!FULL && c0 + u >= C) break;
        float gv[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float sc = sv[u][e] - st.ms[e];
            const float tc = tv[u][e] - st.mt[e];
            const float p1 = fast_exp2(sc * kLog2e) * iz1[e];
            const float ps = fast_exp2(sc * k2) * izs[e];
            const float pt = fast_exp2(tc * k2) * izt[e];
            const float onehot = (c0 + u == id[e]) ? 1.f : 0.f;
            gv[e] = gce[e] * (p1 - onehot) + gkd[e] * (ps - pt);
        }
        store_vec<VEC>(g_img + (long long)(c0 + u) * HW + r, gv);
    }
}

// Gradient of up to eight class planes whose logits are still in registers (C <= 8: the whole
// class axis of a lane's pixels was loaded in ONE kd_fold step): no second read, 16-byte stores.
template <int VEC>
__device__ __forceinline__ void kd_grad_regs(const PixStats<VEC> &st, const float (&sv)[8][VEC],
                                             const float (&tv)[8][VEC], float *__restrict__ g_img,
                                             long long HW, int r, int C, const int (&id)[VEC], float k2,
                                             const float (&gce)[VEC], const float (&gkd)[VEC],
                                             const float (&izs)[VEC], const float (&iz1)[VEC],
                                             const float (&izt)[VEC]) {
#pragma unroll
    for (int u = 0; u < 8; ++u) {
        if (u >= C) break;
        float gv[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const float sc = sv[u][e] - st.ms[e];
            const float tc = tv[u][e] - st.mt[e];
            const float p1 = fast_exp2(sc * kLog2e) * iz1[e];
            const float ps = fast_exp2(sc * k2) * izs[e];
            const float pt = fast_exp2(tc * k2) * izt[e];
            const float onehot = (u == id[e]) ? 1.f : 0.f;
            gv[e] = gce[e] * (p1 - onehot) + gkd[e] * (ps - pt);
        }
        store_vec<VEC>(g_img + (long long)u * HW + r, gv);
    }
}

// Run by ONE workgroup of THREADS threads: fixed-order fp64 sum of the per-workgroup partials,
// then the scalar epilogue of PixelKDLoss / CrossEntropyLoss / KDLiteLoss.
template <int THREADS = kKdThreads>
__device__ __forceinline__ void kd_finalize(const double *partial, int n_blocks, double *totals,
                                            float *out3, float T, float alpha, float bw, int conf) {
    // thread t owns quantity t % 8 of workgroups t/8, t/8 + THREADS/8, ...: coalesced, then an LDS tree
    __shared__ double fin_red[THREADS];
    double v0 = 0.0, v1 = 0.0, v2 = 0.0, v3 = 0.0;
    const int total = n_blocks * 8;
    int i = threadIdx.x;
    const unsigned long long *pu = reinterpret_cast<const unsigned long long *>(partial);
    auto ld = [&](int j) {
        return __longlong_as_double((long long)__hip_atomic_load(pu + j, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT));
    };
    for (; i + 3 * THREADS < total; i += 4 * THREADS) {
        const double x0 = ld(i), x1 = ld(i + THREADS), x2 = ld(i + 2 * THREADS), x3 = ld(i + 3 * THREADS);
        v0 += x0;
        v1 += x1;
        v2 += x2;
        v3 += x3;
    }
    for (; i < total; i += THREADS) v0 += ld(i);
    fin_red[threadIdx.x] = (v0 + v1) + (v2 + v3);
    __syncthreads();
    for (int half = THREADS / 2; half >= 8; half >>= 1) {
        if ((int)threadIdx.x < half) fin_red[threadIdx.x] += fin_red[threadIdx.x + half];
        __syncthreads();
    }
    if (threadIdx.x == 0) {
        double S[8];
        for (int k = 0; k < 8; ++k) S[k] = fin_red[k];
        for (int k = 0; k < 8; ++k) totals[k] = S[k];
        const double T2 = (double)T * (double)T;
        double kd = S[1] > 0.0 ? S[0] / S[1] : 0.0;
        kd *= T2;
        if (bw > 1.0f) {
            const double kb = S[5] > 0.0 ? S[4] / S[5] : 0.0;
            kd += ((double)bw - 1.0) * (kb * T2);
        }
        if (conf) {
            const double kc = S[7] > 0.0 ? S[6] / S[7] : 0.0;
            kd = 0.5 * kd + 0.5 * (kc * T2);
        }
        const double ce = S[2] / S[3];  // 0/0 -> NaN, as CrossEntropyLoss with nothing to average
        out3[0] = (float)((double)alpha * ce + (1.0 - (double)alpha) * kd);
        out3[1] = (float)ce;
        out3[2] = (float)kd;
    }
}

// Hand a workgroup's eight partial sums to whichever workgroup arrives last, which folds all of them
// and writes the loss.  The sums are stored write-through (agent-scope atomic stores), so no release
// fence is needed; the storing wave drains them, one lane takes a ticket, and the last arriver
// acquires before reading.
template <int THREADS = kKdThreads>
__device__ __forceinline__ void kd_publish_and_finalize(const KdArgs &a, const double (&out8)[8],
                                                        double (*red)[8]) {
    const int tid = threadIdx.x;
    {
        const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
        for (int k = 0; k < 8; ++k) {
            const double v = wave_sum(out8[k]);
            if (lane == 0) red[wave][k] = v;
        }
        __syncthreads();
        if (tid < 8) {
            double v = 0.0;
            for (int wv = 0; wv < THREADS / 64; ++wv) v += red[wv][tid];
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.main_partial) + (long long)blockIdx.x * 8 + tid,
                               (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
        }
    }
    __shared__ int last_flag;
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (tid == 0) {
        const unsigned t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        const int last = (t == gridDim.x - 1);
        if (last) {
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        }
        last_flag = last;
    }
    __syncthreads();
    if (!last_flag) return;
    kd_finalize<THREADS>(a.main_partial, (int)gridDim.x, a.totals, a.out3, a.T, a.alpha, a.bw, a.conf);
    if (tid == 0) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
}

// STAGE (gradient only): the tile's logits are parked in dynamic LDS during the statistics pass
// and the gradient pass reads them from there instead of from memory.
// REGS (gradient only, C <= 8): one step covers the class axis, the logits stay in registers.
template <int VEC, bool GRAD, bool EXT, bool STAGE, bool REGS>
__device__ __forceinline__ void kd_main_body(const KdArgs &a) {
    extern __shared__ float4 stage_raw[];
    typename VecF<VEC>::type *stage = reinterpret_cast<typename VecF<VEC>::type *>(stage_raw);
    constexpr int U = (STAGE || REGS) ? 8 : 4;  // class planes per step (loads in flight: 2U per lane)
    __shared__ double red[kKdThreads / 64][8];
    __shared__ double norm[4];
    const int tid = threadIdx.x;
    const bool use_b = EXT && a.bw > 1.0f;
    const bool use_c = EXT && a.conf;
    const float k2 = a.invT * kLog2e;  // exp(x/T) = exp2(x * k2)

    float coef_ce = a.coef_ce, coef_kd = a.coef_kd;
    float inv_nvalid = 0.f, inv_wsum = 0.f, inv_nb = 0.f, inv_cs = 0.f;
    if (GRAD) {
        double v[4] = {0.0, 0.0, 0.0, 0.0};
        for (int b = tid; b < a.pre_blocks; b += kKdThreads) {
#pragma unroll
            for (int k = 0; k < 4; ++k) v[k] += a.pre_partial[(long long)b * 4 + k];
        }
        block_sum_store<4>(v, red, norm);
        __syncthreads();
        inv_nvalid = norm[0] > 0.0 ? (float)(1.0 / norm[0]) : 0.f;
        inv_wsum = norm[1] > 0.0 ? (float)(1.0 / norm[1]) : 0.f;
        inv_nb = norm[2] > 0.0 ? (float)(1.0 / norm[2]) : 0.f;
        inv_cs = norm[3] > 0.0 ? (float)(1.0 / norm[3]) : 0.f;
        if (a.upstream) {
            coef_ce = a.upstream[0] * a.alpha + a.upstream[1];
            coef_kd = a.upstream[0] * (1.0f - a.alpha) + a.upstream[2];
        }
        __syncthreads();
    }

    double acc[EXT ? 8 : 4];
#pragma unroll
    for (int k = 0; k < (EXT ? 8 : 4); ++k) acc[k] = 0.0;
    int bad = 0;
    const TileGeom tg = tile_geom<VEC>(a);
    const int C = a.C;
    const int C4 = C & ~3;
    const int CU = (C / U) * U;
    for (int tile = blockIdx.x; tile < tg.n_tiles; tile += gridDim.x) {
        const int n = tile / tg.tiles_per_img;
        const int qi = (tile - n * tg.tiles_per_img) * kKdThreads + tid;
        if (qi >= tg.per_img) continue;
        const int r = qi * VEC;
        const float *s_img = a.s + (long long)n * C * a.HW;
        const float *t_img = a.t + (long long)n * C * a.HW;
        const long long *lab_img = a.labels + (long long)n * a.HW;
        int id[VEC];
        load_label_ids<VEC>(lab_img + r, a.ignore_index, C, id);

        PixStats<VEC> st;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            st.ms[e] = st.mt[e] = kNegFill;
            st.zs[e] = st.z1[e] = st.zt[e] = st.zt1[e] = st.at[e] = st.sy[e] = 0.f;
        }
        float sv[U][VEC], tv[U][VEC];   // the logits of a step; after the loops, with REGS, the whole class axis
        if (REGS) {
            if (use_c)
                kd_fold<VEC, U, false, true, false>(st, s_img, t_img, a.HW, r, 0, C, id, k2, stage, sv, tv);
            else
                kd_fold<VEC, U, false, false, false>(st, s_img, t_img, a.HW, r, 0, C, id, k2, stage, sv, tv);
        } else if (use_c) {
#pragma unroll 1
            for (int c0 = 0; c0 < CU; c0 += U)
                kd_fold<VEC, U, true, true, STAGE>(st, s_img, t_img, a.HW, r, c0, C, id, k2, stage, sv, tv);
            if (CU < C) kd_fold<VEC, U, false, true, STAGE>(st, s_img, t_img, a.HW, r, CU, C, id, k2, stage, sv, tv);
        } else {
#pragma unroll 1
            for (int c0 = 0; c0 < CU; c0 += U)
                kd_fold<VEC, U, true, false, STAGE>(st, s_img, t_img, a.HW, r, c0, C, id, k2, stage, sv, tv);
            if (CU < C) kd_fold<VEC, U, false, false, STAGE>(st, s_img, t_img, a.HW, r, CU, C, id, k2, stage, sv, tv);
        }

        float wy[VEC], wkd[VEC];
        bool edge[VEC];
#pragma unroll
        for (int e = 0; e < VEC; ++e) edge[e] = false;
        if (use_b) label_boundary_group<VEC>(lab_img, r, a.H, a.W, edge);
        float p_kl = 0.f, p_nv = 0.f, p_ce = 0.f, p_w = 0.f, p_bk = 0.f, p_nb = 0.f, p_ck = 0.f, p_cs = 0.f;
#pragma unroll
        for (int e = 0; e < VEC; ++e) {
            const bool valid = id[e] != -1;
            if (id[e] == -2) ++bad;
            const float kl = a.invT * st.at[e] / st.zt[e] - fast_ln(st.zt[e]) + fast_ln(st.zs[e]);
            const float nll = fast_ln(st.z1[e]) - (st.sy[e] - st.ms[e]);
            wy[e] = (id[e] >= 0) ? (a.class_w ? a.class_w[id[e]] : 1.f) : 0.f;
            float wk = 0.f;
            if (valid) {
                p_kl += kl;
                p_nv += 1.f;
                wk = inv_nvalid;
            }
            if (id[e] >= 0) {
                p_ce += wy[e] * nll;
                p_w += wy[e];
            }
            if (use_b) {
                if (valid && edge[e]) {
                    p_bk += kl;
                    p_nb += 1.f;
                    wk += (a.bw - 1.0f) * inv_nb;
                }
            }
            if (use_c) {
                const float cw = valid ? 1.f / st.zt1[e] : 0.f;
                p_ck += cw * kl;
                p_cs += cw;
                wk = 0.5f * wk + 0.5f * cw * inv_cs;
            }
            wkd[e] = wk;
        }
        acc[0] += p_kl;
        acc[1] += p_nv;
        acc[2] += p_ce;
        acc[3] += p_w;
        if (EXT) {
            acc[4 % (EXT ? 8 : 4)] += p_bk;
            acc[5 % (EXT ? 8 : 4)] += p_nb;
            acc[6 % (EXT ? 8 : 4)] += p_ck;
            acc[7 % (EXT ? 8 : 4)] += p_cs;
        }

        if (GRAD) {
            float gce[VEC], gkd[VEC], izs[VEC], iz1[VEC], izt[VEC];
#pragma unroll
            for (int e = 0; e < VEC; ++e) {
                gce[e] = coef_ce * wy[e] * inv_wsum;
                gkd[e] = coef_kd * a.T * wkd[e];  // T^2 * (1/T)
                izs[e] = 1.f / st.zs[e];
                iz1[e] = 1.f / st.z1[e];
                izt[e] = 1.f / st.zt[e];
            }
            float *g_img = a.grad + (long long)n * C * a.HW;
            if constexpr (REGS) {
                kd_grad_regs<VEC>(st, sv, tv, g_img, a.HW, r, C, id, k2, gce, gkd, izs, iz1, izt);
                continue;
            }
#pragma unroll 1
            for (int c0 = 0; c0 < C4; c0 += 4)
                kd_grad4<VEC, true, STAGE>(st, s_img, t_img, g_img, a.HW, r, c0, C, id, k2, gce, gkd, izs, iz1, izt, stage);
            if (C4 < C)
                kd_grad4<VEC, false, STAGE>(st, s_img, t_img, g_img, a.HW, r, C4, C, id, k2, gce, gkd, izs, iz1, izt, stage);
        }
    }
    if (bad) atomicAdd(reinterpret_cast<unsigned long long *>(a.bad), (unsigned long long)bad);
    double out8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) out8[k] = (k < (EXT ? 8 : 4)) ? acc[k % (EXT ? 8 : 4)] : 0.0;
    kd_publish_and_finalize(a, out8, red);
}

template <int VEC, bool GRAD, bool EXT, bool STAGE>
__global__ __launch_bounds__(kKdThreads) void kdlite_main_kernel(KdArgs a) {
    kd_main_body<VEC, GRAD, EXT, STAGE, false>(a);
}

// Loss + gradient for C <= 8 with the logits of a lane's four pixels held in registers.
template <bool EXT>
__global__ __launch_bounds__(kKdThreads) void kdlite_grad_regs_kernel(KdArgs a) {
    kd_main_body<4, true, EXT, false, true>(a);
}

// Once per teacher (a sweep evaluates hundreds of candidates against the same un-pruned network): the
// temperature-softmax probabilities p_c = softmax(t / T)_c, [N,C,H,W] fp32, and per pixel sum_c p_c ln p_c.  One lane
// per four pixels; the class axis is walked three times (maximum, sums, probabilities), the repeats served by L2.
__global__ __launch_bounds__(256) void kd_teacher_prepare_kernel(const float *__restrict__ t, float *__restrict__ probs,
                                                                 float *__restrict__ hneg, long long N, long long HW,
                                                                 int C, float invT) {
    const float k2 = invT * kLog2e;
    const long long per_img = HW / 4;
    for (long long q = (long long)blockIdx.x * 256 + threadIdx.x; q < N * per_img; q += (long long)gridDim.x * 256) {
        const long long n = q / per_img;
        const long long r = (q - n * per_img) * 4;
        const float *tp = t + n * C * HW + r;
        float m[4] = {kNegFill, kNegFill, kNegFill, kNegFill};
        for (int c = 0; c < C; ++c) {
            float v[4];
            load_vec<4>(tp + (long long)c * HW, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) m[e] = fmaxf(m[e], v[e]);
        }
        float z[4] = {0.f, 0.f, 0.f, 0.f}, a[4] = {0.f, 0.f, 0.f, 0.f};
        for (int c = 0; c < C; ++c) {
            float v[4];
            load_vec<4>(tp + (long long)c * HW, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float tc = v[e] - m[e];
                const float et = fast_exp2(tc * k2);
                z[e] += et;
                a[e] = __builtin_fmaf(et, tc, a[e]);
            }
        }
        float iz[4], hn[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            iz[e] = 1.0f / z[e];
            hn[e] = invT * a[e] * iz[e] - fast_ln(z[e]);   // sum p ln p = (1/T) sum p (t - max) - ln Z
        }
        store_vec<4>(hneg + n * HW + r, hn);
        float *pp = probs + n * C * HW + r;
        for (int c = 0; c < C; ++c) {
            float v[4], p[4];
            load_vec<4>(tp + (long long)c * HW, v);
#pragma unroll
            for (int e = 0; e < 4; ++e) p[e] = fast_exp2((v[e] - m[e]) * k2) * iz[e];
            store_vec<4>(pp + (long long)c * HW, p);
        }
    }
}

// ---- the sweep candidate's head: up-sampling + loss + arg-max histogram in ONE pass ---------------
//
// A sweep candidate's forward ends in the segmentation head's convolution ([N,C,h,w] fp16 under
// autocast) followed by a bilinear up-sampling to [N,C,H,W] fp32, which the loss and the arg-max then
// each read once: 3 x 4*C bytes per pixel written and read back for values that are a function of a
// tensor sixteen times smaller.  This kernel interpolates the student's logits on the fly instead: a
// tile's source rows (all classes) are staged in LDS as fp32, every lane interpolates its four pixels
// class by class with the arithmetic of upsample_bilinear_kernel (same bits), folds them with the
// teacher's logits into the loss statistics (kd_fold_stats, as the loss kernel), tracks the arg-max
// (first maximum wins, a NaN is the maximum) and counts (label, arg-max) into the privatised LDS
// histogram of the confusion-matrix kernels.  HBM traffic: teacher + labels + the small source.
struct HeadArgs {
    KdArgs kd;              // kd.s is unused; kd.t holds the teacher's logits, or (PROBS) its prepared probabilities
    const float *hneg;      // PROBS: [N][H][W] sum_c p_c ln p_c of the teacher at the temperature
    const __half *low;      // [N][C][in_h][in_w]
    int in_h, in_w;
    int align_corners;
    float scale_h, scale_w;
    int lds_rows;           // source rows a tile can touch (upper bound, sized on the host)
    int copies, stride;     // layout of the LDS histogram
    ConfTail tail;
    // the class-innermost form (kdlite_head2_kernel) only:
    int win_cols;           // source columns staged per row (a multiple of 8; the stage row holds one more)
    int windowed;           // every tile lies inside one output row: the staged columns follow the tile
};

struct HeadLane {
    int top[4], bot[4];     // LDS word offsets (within a class plane) of the left neighbours, upper and lower row
    float w0[4], w1[4];
    float h0, h1;
};

// One class plane's four interpolated values of a lane.  A staged row carries one more column, a copy of
// its last one, so the right neighbour is always the next word.
__device__ __forceinline__ void head_values(const float *plane, const HeadLane &g, float (&v)[4]) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const float a = plane[g.top[e]], b = plane[g.top[e] + 1];
        const float c = plane[g.bot[e]], d = plane[g.bot[e] + 1];
        const float t = __builtin_fmaf(g.w0[e], a, g.w1[e] * b);
        const float u = __builtin_fmaf(g.w0[e], c, g.w1[e] * d);
        v[e] = __builtin_fmaf(g.h0, t, g.h1 * u);
    }
}

// The student's half of kd_fold_stats for a teacher whose temperature-softmax probabilities are given (pv):
// running maximum, Zs, Z1, the label's logit, and sum_c p_c s_c (in st.at; the teacher's own statistics are not
// needed: KL = sum p ln p - (1/T) (sum p s - max s) + ln Zs).
template <int U>
__device__ __forceinline__ void kd_fold_student(PixStats<4> &st, const float (&sv)[U][4], const float (&pv)[U][4],
                                                int c0, const int (&id)[4], float k2) {
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        float msn = fmaxf(fmaxf(st.ms[e], fmaxf(sv[0][e], sv[1 % U][e])), fmaxf(sv[2 % U][e], sv[3 % U][e]));
        const float ds = st.ms[e] - msn;  // <= 0
        float Zs = st.zs[e] * fast_exp2(ds * k2);
        float Z1 = st.z1[e] * fast_exp2(ds * kLog2e);
        float ps = st.at[e];
        float sy = st.sy[e];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const float sc = sv[u][e] - msn;
            Zs += fast_exp2(sc * k2);
            Z1 += fast_exp2(sc * kLog2e);
            ps = __builtin_fmaf(pv[u][e], sv[u][e], ps);
            sy = (c0 + u == id[e]) ? sv[u][e] : sy;
        }
        st.ms[e] = msn;
        st.zs[e] = Zs;
        st.z1[e] = Z1;
        st.at[e] = ps;
        st.sy[e] = sy;
    }
}

template <int U, bool FULL, bool PROBS>
__device__ __forceinline__ void head_fold(PixStats<4> &st, const float *stage, int plane_words, const HeadLane &g,
                                          const float *__restrict__ t_img, long long HW, int r, int c0, int C,
                                          const int (&id)[4], float k2, float (&best)[4], int (&arg)[4]) {
    float sv[U][4], tv[U][4];
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int cc = FULL ? c0 + u : min(c0 + u, C - 1);
        load_vec_nt<4>(t_img + (long long)cc * HW + r, tv[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        const int cc = FULL ? c0 + u : min(c0 + u, C - 1);
        head_values(stage + cc * plane_words, g, sv[u]);
    }
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (!FULL && c0 + u >= C) break;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            arg[e] = !(sv[u][e] <= best[e]) ? c0 + u : arg[e];
            best[e] = __builtin_elementwise_maximum(best[e], sv[u][e]);   // NaN-propagating
        }
    }
    if (!FULL) {
#pragma unroll
        for (int u = 1; u < U; ++u) {
            const bool gone = c0 + u >= C;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                sv[u][e] = gone ? kNegFill : sv[u][e];
                tv[u][e] = gone ? (PROBS ? 0.0f : kNegFill) : tv[u][e];
            }
        }
    }
    if constexpr (PROBS)
        kd_fold_student<U>(st, sv, tv, c0, id, k2);
    else
        kd_fold_stats<4, U, false>(st, sv, tv, c0, id, k2);
}

// The arg-max half of head_fold alone: U class planes interpolated and compared, nothing else.
template <int U, bool FULL>
__device__ __forceinline__ void head_fold_argmax(const float *stage, int plane_words, const HeadLane &g, int c0, int C,
                                                 float (&best)[4], int (&arg)[4]) {
#pragma unroll
    for (int u = 0; u < U; ++u) {
        if (!FULL && c0 + u >= C) break;
        float v[4];
        head_values(stage + (c0 + u) * plane_words, g, v);
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            arg[e] = !(v[e] <= best[e]) ? c0 + u : arg[e];
            best[e] = __builtin_elementwise_maximum(best[e], v[e]);   // NaN-propagating
        }
    }
}

// Source rows of one tile: where they lie, and the split of a chunk number (16 bytes = 8 halves of one
// class's rows) into class, row and chunk of the row.
struct HeadRows {
    int y_lo, y_hi;
    int per_class;          // chunks per class
    float inv_per_class;
};

// NCH = 16-byte source chunks a thread stages per tile at most (the host picks 4 or 8 by the stage's size).
// PROBS: the teacher comes as prepared probabilities and per-pixel entropy term (kd_teacher_prepare_kernel).
// LOSS = false: no teacher and no loss, only the arg-max histogram of the interpolated logits (evaluation loops).
template <int U, int NCH, bool PROBS, bool LOSS = true>
__global__ __launch_bounds__(kKdThreads, 2) void kdlite_head_kernel(HeadArgs h) {   // two waves per SIMD at least
    extern __shared__ float4 head_lds_raw[];
    const KdArgs &a = h.kd;
    const int C = a.C;
    const int row_words = h.in_w + 1;
    const int plane_words = h.lds_rows * row_words;
    float *stage = reinterpret_cast<float *>(head_lds_raw);               // [C][lds_rows][in_w + 1] fp32
    unsigned *lds_hist = reinterpret_cast<unsigned *>(stage + C * plane_words);
    __shared__ double red[kKdThreads / 64][8];
    const int tid = threadIdx.x;
    for (int i = tid; i < h.copies * h.stride; i += kKdThreads) lds_hist[i] = 0;
    unsigned *hist = lds_hist + (tid & (h.copies - 1)) * h.stride;
    const float k2 = a.invT * kLog2e;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    int bad = 0;
    const TileGeom tg = tile_geom<4>(a);
    const int W = a.W;
    const int CU = (C / U) * U;
    const int chunks_per_row = h.in_w >> 3;   // 8 halves = 16 bytes; in_w % 8 == 0 (host)
    typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));
    const float inv_per_row = 1.0f / (float)chunks_per_row;
    // the source rows under a tile's output rows
    auto rows_of = [&](int tile) {
        const int n = tile / tg.tiles_per_img;
        const int px_first = (tile - n * tg.tiles_per_img) * kKdThreads * 4;
        const int px_last = min(px_first + kKdThreads * 4, (int)a.HW) - 1;
        HeadRows rw;
        rw.y_lo = (int)src_index(h.scale_h, px_first / W, h.align_corners);
        const int y_last = (int)src_index(h.scale_h, px_last / W, h.align_corners);
        rw.y_hi = min(min(y_last + 1, h.in_h - 1), rw.y_lo + h.lds_rows - 1);
        rw.per_class = (rw.y_hi - rw.y_lo + 1) * chunks_per_row;
        rw.inv_per_class = 1.0f / (float)rw.per_class;
        return rw;
    };
    // A tile's source rows travel through registers: requested while the previous tile is being worked on,
    // converted and written to LDS when the previous tile's readers are done.
    f16x8_t pre[NCH];
    auto request = [&](int tile, const HeadRows &rw) {
        const int n = tile / tg.tiles_per_img;
        const __half *src = h.low + ((long long)n * C * h.in_h + rw.y_lo) * h.in_w;
        const int total = C * rw.per_class;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int i = tid + j * kKdThreads;
            if (i < total) {
                // exact while the numbers stay far below 2^22 (the host bounds the stage)
                const int c = (int)(((float)i + 0.5f) * rw.inv_per_class);
                const int k = i - c * rw.per_class;   // rows of one class are contiguous in memory
                pre[j] = *reinterpret_cast<const f16x8_t *>(src + (long long)c * h.in_h * h.in_w + k * 8);
            }
        }
    };
    auto deposit = [&](const HeadRows &rw) {
        const int total = C * rw.per_class;
#pragma unroll
        for (int j = 0; j < NCH; ++j) {
            const int i = tid + j * kKdThreads;
            if (i < total) {
                const int c = (int)(((float)i + 0.5f) * rw.inv_per_class);
                const int k = i - c * rw.per_class;
                const int rr = (int)(((float)k + 0.5f) * inv_per_row);
                const int kc = k - rr * chunks_per_row;
                float *d = stage + c * plane_words + rr * row_words + kc * 8;
#pragma unroll
                for (int m = 0; m < 8; ++m) d[m] = (float)pre[j][m];
                if (kc == chunks_per_row - 1) d[8] = (float)pre[j][7];
            }
        }
    };
    HeadRows rows = rows_of((int)blockIdx.x);
    if ((int)blockIdx.x < tg.n_tiles) request((int)blockIdx.x, rows);
    for (int tile = blockIdx.x; tile < tg.n_tiles; tile += gridDim.x) {
        const int n = tile / tg.tiles_per_img;
        const int q0 = (tile - n * tg.tiles_per_img) * kKdThreads;
        const int qi = q0 + tid;
        const int y_lo = rows.y_lo, y_hi = rows.y_hi;
        __syncthreads();   // whoever still reads the previous tile's rows (and, first time, the cleared histogram)
        deposit(rows);
        __syncthreads();
        if (tile + (int)gridDim.x < tg.n_tiles) {
            rows = rows_of(tile + (int)gridDim.x);
            request(tile + (int)gridDim.x, rows);
        }
        if (qi >= tg.per_img) continue;
        const int r = qi * 4;
        const int oy = r / W;
        const int ox = r - oy * W;
        HeadLane g;
        {
            const float sy = src_index(h.scale_h, oy, h.align_corners);
            const int y0 = min((int)sy, y_hi);
            const int y1 = min(y0 + 1, y_hi);
            g.h1 = sy - (float)y0;
            g.h0 = 1.0f - g.h1;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float sx = src_index(h.scale_w, ox + e, h.align_corners);
                const int x0 = min((int)sx, h.in_w - 1);
                g.w1[e] = sx - (float)x0;
                g.w0[e] = 1.0f - g.w1[e];
                g.top[e] = (y0 - y_lo) * row_words + x0;
                g.bot[e] = (y1 - y_lo) * row_words + x0;
            }
        }
        const float *t_img = LOSS ? a.t + (long long)n * C * a.HW : nullptr;
        long long lab[4];
        load_labels<4>(a.labels + (long long)n * a.HW + r, lab);
        float hn[4] = {0.f, 0.f, 0.f, 0.f};
        if constexpr (PROBS) load_vec_nt<4>(h.hneg + (long long)n * a.HW + r, hn);
        int id[4];
#pragma unroll
        for (int e = 0; e < 4; ++e)
            id[e] = (lab[e] == a.ignore_index) ? -1 : ((lab[e] >= 0 && lab[e] < C) ? (int)lab[e] : -2);

        PixStats<4> st;
        float best[4];
        int arg[4];
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            st.ms[e] = st.mt[e] = kNegFill;
            st.zs[e] = st.z1[e] = st.zt[e] = st.zt1[e] = st.at[e] = st.sy[e] = 0.f;
            best[e] = -__builtin_inff();
            arg[e] = 0;
        }
        if constexpr (LOSS) {
#pragma unroll 1
            for (int c0 = 0; c0 < CU; c0 += U)
                head_fold<U, true, PROBS>(st, stage, plane_words, g, t_img, a.HW, r, c0, C, id, k2, best, arg);
            // what is left of the class axis: a single plane as a step of its own (21 classes = 5 x 4 + 1), more as
            // one step whose surplus planes are filled
            if (C - CU == 1)
                head_fold<1, true, PROBS>(st, stage, plane_words, g, t_img, a.HW, r, CU, C, id, k2, best, arg);
            else if (CU < C)
                head_fold<U, false, PROBS>(st, stage, plane_words, g, t_img, a.HW, r, CU, C, id, k2, best, arg);
        } else {
#pragma unroll 1
            for (int c0 = 0; c0 < CU; c0 += U) head_fold_argmax<U, true>(stage, plane_words, g, c0, C, best, arg);
            if (CU < C) head_fold_argmax<U, false>(stage, plane_words, g, CU, C, best, arg);
        }

        if ((best[0] != best[0]) | (best[1] != best[1]) | (best[2] != best[2]) | (best[3] != best[3])) {
            // a NaN among a pixel's logits: the exact class-by-class walk
            float bv[4];
            head_values(stage, g, bv);
#pragma unroll
            for (int e = 0; e < 4; ++e) arg[e] = 0;
            for (int c = 1; c < C; ++c) {
                float v[4];
                head_values(stage + c * plane_words, g, v);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const bool take = (v[e] > bv[e]) || (v[e] != v[e] && bv[e] == bv[e]);
                    bv[e] = take ? v[e] : bv[e];
                    arg[e] = take ? c : arg[e];
                }
            }
        }
        if constexpr (!LOSS) {
#pragma unroll
            for (int e = 0; e < 4; ++e)
                if (lab[e] >= 0 && lab[e] < C) atomicAdd(&hist[(int)lab[e] * C + arg[e]], 1u);
            continue;
        }
        float p_kl = 0.f, p_nv = 0.f, p_ce = 0.f, p_w = 0.f;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            if (lab[e] >= 0 && lab[e] < C) atomicAdd(&hist[(int)lab[e] * C + arg[e]], 1u);
            if (id[e] == -2) ++bad;
            const float kl = PROBS ? hn[e] + a.invT * (st.ms[e] - st.at[e]) + fast_ln(st.zs[e])
                                   : a.invT * st.at[e] / st.zt[e] - fast_ln(st.zt[e]) + fast_ln(st.zs[e]);
            const float nll = fast_ln(st.z1[e]) - (st.sy[e] - st.ms[e]);
            const float wy = (id[e] >= 0) ? (a.class_w ? a.class_w[id[e]] : 1.f) : 0.f;
            if (id[e] != -1) {
                p_kl += kl;
                p_nv += 1.f;
            }
            if (id[e] >= 0) {
                p_ce += wy * nll;
                p_w += wy;
            }
        }
        acc[0] += p_kl;
        acc[1] += p_nv;
        acc[2] += p_ce;
        acc[3] += p_w;
    }
    if (LOSS && bad) atomicAdd(reinterpret_cast<unsigned long long *>(a.bad), (unsigned long long)bad);
    conf_finish<kKdThreads>(lds_hist, h.copies, h.stride, C * C, 0u, h.tail);
    if constexpr (!LOSS) return;
    double out8[8];
#pragma unroll
    for (int k = 0; k < 8; ++k) out8[k] = k < 4 ? acc[k % 4] : 0.0;
    __syncthreads();
    kd_publish_and_finalize(a, out8, red);
}


// ---- the same head, second form: class-innermost LDS image, packed arithmetic, one exponential per logit ----
//
// What the form above spends per logit is vector arithmetic, not bytes: four 4-byte LDS reads and an address
// addition per class plane, the interpolation and the softmax statistics one value at a time, a running maximum
// whose every change rescales the sums, two quarter-rate exponentials.  This form removes most of it:
//  * the staged source rows lie [row][column][class] with the class axis padded to CPAD floats (CPAD / 4 odd, so
//    the sixteen lanes one ds_read_b128 serves together hit sixteen different 16-byte slots): a pixel's four
//    corners of FOUR classes are four 16-byte reads at immediate offsets, no address arithmetic in the class loop;
//  * the interpolation and the statistics run on class pairs (v_pk_mul_f32 / v_pk_fma_f32 / v_pk_add_f32), with the
//    fused multiply-adds placed as in upsample_bilinear_kernel: the interpolated logits are the same bits;
//  * the softmax shift of a pixel is not its running maximum but M = the largest logit of its four source pixels
//    (a table filled once per tile): an interpolated logit is a convex combination of the corners', so s - M <= 0
//    up to rounding, and the sums never need rescaling.  Should M lie so far above the pixel's own maximum that
//    sum exp(s - M) leaves the normal range (or a logit is not finite), the pixel is redone with its exact maximum;
//  * for the integer temperatures (T = 1..4) exp(s - M) = exp((s - M) / T)^T: one exponential and two packed
//    multiplications per logit pair instead of two exponentials per logit;
//  * the label's logit is interpolated once per pixel behind the class loop instead of selected per class.
// The teacher comes as prepared probabilities (kd_teacher_prepare_kernel).  LOSS = false: arg-max histogram only.
typedef float f32x2_t __attribute__((ext_vector_type(2)));
typedef float f32x4_t __attribute__((ext_vector_type(4)));

constexpr float kHead2TinySum = 1.0e-20f;   // below it sum exp(s - M) is redone with the exact maximum

__device__ __forceinline__ f32x2_t pk_fma(f32x2_t a, f32x2_t b, f32x2_t c) { return __builtin_elementwise_fma(a, b, c); }

// Bilinear value of two classes: h0 * (w0 * a + w1 * b) + h1 * (w0 * c + w1 * d), the multiply-adds fused as in
// upsample_bilinear_kernel (fma(w0, a, w1 * b), fma(w0, c, w1 * d), fma(h0, t, h1 * u)).
__device__ __forceinline__ f32x2_t head2_pair(f32x2_t a, f32x2_t b, f32x2_t c, f32x2_t d, float w0, float w1, float h0,
                                              float h1) {
    const f32x2_t t = pk_fma((f32x2_t)(w0), a, (f32x2_t)(w1) * b);
    const f32x2_t u = pk_fma((f32x2_t)(w0), c, (f32x2_t)(w1) * d);
    return pk_fma((f32x2_t)(h0), t, (f32x2_t)(h1) * u);
}

__device__ __forceinline__ float head2_one(const float *stage, int top, int bot, int cpad, int c, float w0, float w1,
                                           float h0, float h1) {
    const float a = stage[top + c], b = stage[top + cpad + c], cc = stage[bot + c], d = stage[bot + cpad + c];
    const float t = __builtin_fmaf(w0, a, w1 * b);
    const float u = __builtin_fmaf(w0, cc, w1 * d);
    return __builtin_fmaf(h0, t, h1 * u);
}

// exp(x)^T from e = exp(x / T) for the integer temperatures.
template <int TPOW>
__device__ __forceinline__ f32x2_t pow_t(f32x2_t e) {
    if constexpr (TPOW == 1) return e;
    const f32x2_t e2 = e * e;
    if constexpr (TPOW == 2) return e2;
    if constexpr (TPOW == 3) return e2 * e;
    return e2 * e2;
}

SPICE_ENV_KNOB(knob_head2_threads, "SPICE_HEAD2_THREADS", 0)
// one output pixel per lane: a tile is as many consecutive pixels of one image as the workgroup has lanes
static int head2_threads() { return knob_head2_threads() == 256 ? 256 : 512; }

// A lane's pixel over four class planes (c0 .. c0 + 3).  Classes from C on are padding: a large negative FINITE value
// in the stage (an interpolation weight of exactly zero times -infinity would be a NaN) and probability zero in `tp`,
// so their exponentials and their terms of sum_c p_c s_c are zero, and they win the arg-max only over a pixel whose
// real logits are all -infinity (the caller turns such a winner into class 0, the first maximum).
template <int CPAD, int TPOW, bool LOSS>
__device__ __forceinline__ void head2_step(const float *top, const float *bot, float w0, float w1, float h0, float h1, int c0,
                                           float &best, int &arg, float mk, float k2, f32x2_t &zs, f32x2_t &z1, f32x2_t &ps,
                                           const f32x2_t *tp) {
    const f32x4_t A = *reinterpret_cast<const f32x4_t *>(top + c0);
    const f32x4_t B = *reinterpret_cast<const f32x4_t *>(top + CPAD + c0);
    const f32x4_t Cc = *reinterpret_cast<const f32x4_t *>(bot + c0);
    const f32x4_t D = *reinterpret_cast<const f32x4_t *>(bot + CPAD + c0);
    const f32x2_t v01 = head2_pair(A.lo, B.lo, Cc.lo, D.lo, w0, w1, h0, h1);
    const f32x2_t v23 = head2_pair(A.hi, B.hi, Cc.hi, D.hi, w0, w1, h0, h1);
    const float v[4] = {v01.x, v01.y, v23.x, v23.y};
#pragma unroll
    for (int u = 0; u < 4; ++u) {
        arg = !(v[u] <= best) ? c0 + u : arg;
        best = __builtin_elementwise_maximum(best, v[u]);   // NaN-propagating
    }
    if constexpr (LOSS) {
        // (s - M) / T in base-2 units as one multiply-add: s * k2 - M * k2
        const f32x2_t x01 = pk_fma(v01, (f32x2_t)(k2), (f32x2_t)(mk));
        const f32x2_t x23 = pk_fma(v23, (f32x2_t)(k2), (f32x2_t)(mk));
        f32x2_t e01, e23;
        e01.x = fast_exp2(x01.x);
        e01.y = fast_exp2(x01.y);
        e23.x = fast_exp2(x23.x);
        e23.y = fast_exp2(x23.y);
        zs += e01 + e23;   // the sums as one partial per class parity, added up behind the class loop
        if constexpr (TPOW > 1) z1 += pow_t<TPOW>(e01) + pow_t<TPOW>(e23);
        ps = pk_fma(tp[c0 >> 1], v01, ps);   // sum_c p_c s_c
        ps = pk_fma(tp[(c0 >> 1) + 1], v23, ps);
    }
}

// One output pixel per lane, workgroups of 512.  What hides the memory's latency is the teacher data in flight per
// compute unit; a lane requests the probabilities of ALL its classes (a register each) at the start of its tile, before
// the tile's source rows are staged, so they arrive while the workgroup stages and the class loop never waits for
// memory.  A lane's state is small enough for four waves per SIMD: two workgroups = 1024 pixels x C classes x 4 bytes
// (86 KB at 21 classes) requested per compute unit.
template <int CPAD, int TPOW, bool LOSS, int THREADS>
__global__ __launch_bounds__(THREADS, 4) void kdlite_head2_kernel(HeadArgs h) {
    extern __shared__ float4 head_lds_raw[];
    const KdArgs &a = h.kd;
    const int C = a.C;
    const int WC = h.win_cols + 1;                  // staged columns per row: the window and its right neighbour
    const int n_src = h.lds_rows * WC;
    float *stage = reinterpret_cast<float *>(head_lds_raw);                 // [lds_rows][WC][CPAD] fp32
    float *smax = stage + n_src * CPAD;                                      // [lds_rows][WC]: largest logit of a source pixel
    unsigned *lds_hist = reinterpret_cast<unsigned *>(smax + ((n_src + 3) & ~3));
    __shared__ double red[THREADS / 64][8];
    const int tid = threadIdx.x;
    for (int i = tid; i < h.copies * h.stride; i += THREADS) lds_hist[i] = 0;
    if (CPAD > C) {   // the padding classes, once (the tiles write the real classes only)
        for (int j = tid; j < n_src; j += THREADS)
            for (int c = C; c < CPAD; ++c) stage[j * CPAD + c] = kNegFill;
    }
    unsigned *hist = lds_hist + (tid & (h.copies - 1)) * h.stride;
    const float k2 = a.invT * kLog2e;
    double acc[4] = {0.0, 0.0, 0.0, 0.0};
    int bad = 0;
    const int HW = (int)a.HW;
    const int W = a.W;
    const int tiles_per_img = (HW + THREADS - 1) / THREADS;
    const int n_tiles = (int)a.N * tiles_per_img;
    const int chunks_per_row = h.win_cols >> 3;   // 8 halves = 16 bytes
    const float inv_c = 1.0f / (float)C;
    const float inv_per_row = 1.0f / (float)chunks_per_row;
    const float inv_w = 1.0f / (float)W;
    typedef _Float16 f16x8_t __attribute__((ext_vector_type(8)));

    // A lane's column: the same in every tile when a tile is a whole number of output rows (worked out once).
    const bool same_columns = (THREADS % W) == 0;
    int xoff = 0;            // x0 * CPAD: the column part of the corner's offset, before the window's start is taken off
    float w0 = 0.f, w1 = 0.f;
    auto column = [&](int ox) {
        const float sx = src_index(h.scale_w, ox, h.align_corners);
        const int x0 = min((int)sx, h.in_w - 1);
        w1 = sx - (float)x0;
        w0 = 1.0f - w1;
        xoff = x0 * CPAD;
    };
    const int lane_row = (int)(((float)tid + 0.5f) * inv_w);     // exact: the numbers stay far below 2^22
    const int lane_col = tid - lane_row * W;
    if (same_columns) column(lane_col);

    // a tile's source rows: where they lie
    struct Rows {
        int n, y_lo, y_hi, x_lo, px_first, px_last, oy_first;
    };
    auto rows_of = [&](int tile) {
        Rows q;
        q.n = tile / tiles_per_img;
        q.px_first = (tile - q.n * tiles_per_img) * THREADS;
        q.px_last = min(q.px_first + THREADS, HW) - 1;
        q.oy_first = q.px_first / W;
        q.y_lo = (int)src_index(h.scale_h, q.oy_first, h.align_corners);
        const int y_last = (int)src_index(h.scale_h, q.px_last / W, h.align_corners);
        q.y_hi = min(min(y_last + 1, h.in_h - 1), q.y_lo + h.lds_rows - 1);
        q.x_lo = 0;
        if (h.windowed)
            q.x_lo = min(((int)src_index(h.scale_w, q.px_first - q.oy_first * W, h.align_corners)) & ~7, h.in_w - h.win_cols);
        return q;
    };
    // chunk number -> (class, row, 8-column piece); exact while the numbers stay far below 2^22 (the host bounds the stage)
    auto load_chunk = [&](const __half *src, int i) {
        const int k = (int)(((float)i + 0.5f) * inv_c);
        const int c = i - k * C;
        const int rr = (int)(((float)k + 0.5f) * inv_per_row);
        const int kc = k - rr * chunks_per_row;
        return *reinterpret_cast<const f16x8_t *>(src + ((long long)c * h.in_h + rr) * h.in_w + kc * 8);
    };
    auto deposit_chunk = [&](int i, const f16x8_t &piece) {
        const int k = (int)(((float)i + 0.5f) * inv_c);
        const int c = i - k * C;
        const int rr = (int)(((float)k + 0.5f) * inv_per_row);
        const int kc = k - rr * chunks_per_row;
        float *d = stage + (rr * WC + kc * 8) * CPAD + c;
#pragma unroll
        for (int m = 0; m < 8; ++m) d[m * CPAD] = (float)piece[m];
        if (kc == chunks_per_row - 1) d[8 * CPAD] = (float)piece[7];   // right neighbour of the last column
    };
    constexpr int NPRE = 1024 / THREADS;   // chunks per lane requested one tile ahead (1024 per workgroup)
    f16x8_t pre[NPRE];
    auto request = [&](const Rows &q) {
        const __half *src = h.low + ((long long)q.n * C * h.in_h + q.y_lo) * h.in_w + q.x_lo;
        const int total = C * (q.y_hi - q.y_lo + 1) * chunks_per_row;
#pragma unroll
        for (int j = 0; j < NPRE; ++j) {
            const int i = tid + j * THREADS;
            if (i < total) pre[j] = load_chunk(src, i);
        }
    };
    // A workgroup takes a run of CONSECUTIVE tiles: neighbouring output rows lie under the same source rows (four of
    // them at x4), so most tiles find their rows staged already - no staging, no barrier - and the source rows are
    // fetched by one workgroup instead of by whichever four happen to take those tiles on four XCDs.
    const int per_wg = (n_tiles + (int)gridDim.x - 1) / (int)gridDim.x;
    const int t_first = (int)blockIdx.x * per_wg, t_end = min(t_first + per_wg, n_tiles);
    if (t_first < t_end) request(rows_of(t_first));
    int s_n = -1, s_ylo = 0, s_yhi = 0, s_xlo = 0;   // the rows the stage holds

    for (int tile = t_first; tile < t_end; ++tile) {
        const Rows cur = rows_of(tile);
        const int n = cur.n;
        const int px_first = cur.px_first;
        const int r = px_first + tid;   // the lane's pixel inside the image
        const bool live = r < HW;
        // everything the lane needs from memory is requested first: its teacher probabilities, label, entropy term
        f32x2_t tp[CPAD / 2];   // the teacher's probabilities, class pairs
        long long lab = -1;
        float hn = 0.f;
        if constexpr (LOSS) {
            // The image's teacher planes through a buffer descriptor: a load is ONE instruction (the plane's offset a
            // scalar operand, the lane's 32-bit byte offset the vector one, no address arithmetic in vector registers),
            // and a class past the last lies outside the buffer, which reads as zero without touching memory -
            // probability zero, no branch.
            const __amdgpu_buffer_rsrc_t planes = __builtin_amdgcn_make_buffer_rsrc(
                const_cast<float *>(a.t + (long long)n * C * a.HW), 0, C * HW * 4, 0x00020000);
            const int rb = (live ? r : 0) * 4;
#pragma unroll
            for (int c = 0; c < CPAD; c += 2) {
                tp[c >> 1].x = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(planes, rb, c * HW * 4, 2 /* non-temporal */));
                tp[c >> 1].y = __builtin_bit_cast(float, __builtin_amdgcn_raw_buffer_load_b32(planes, rb, (c + 1) * HW * 4, 2));
            }
            if (live) hn = __builtin_nontemporal_load(h.hneg + (long long)n * a.HW + r);
        }
        if (live) lab = a.labels[(long long)n * a.HW + r];

        const int oy_first = cur.oy_first, y_lo = cur.y_lo, y_hi = cur.y_hi, x_lo = cur.x_lo;
        const int rows = y_hi - y_lo + 1;
        const bool fresh = !(n == s_n && y_lo == s_ylo && y_hi == s_yhi && x_lo == s_xlo);   // uniform
        if (fresh) {
            __syncthreads();   // whoever still reads the staged rows (first time: the cleared histogram, the padding)
            // The tile's source chunks were requested while the tiles before were worked on (the first tile's before
            // the loop): converted and stored now.  Chunk i = (k, c) with the class fastest: a wave's stores fall on
            // consecutive banks.
            const int total = C * rows * chunks_per_row;
#pragma unroll
            for (int j = 0; j < NPRE; ++j) {
                const int i = tid + j * THREADS;
                if (i < total) deposit_chunk(i, pre[j]);
            }
            if (total > NPRE * THREADS) {   // a stage larger than the chunks held ahead: the rest straight from memory
                const __half *src = h.low + ((long long)n * C * h.in_h + y_lo) * h.in_w + x_lo;
                for (int i = tid + NPRE * THREADS; i < total; i += THREADS) deposit_chunk(i, load_chunk(src, i));
            }
            s_n = n, s_ylo = y_lo, s_yhi = y_hi, s_xlo = x_lo;
            __syncthreads();
        }
        if (tile + 1 < t_end) {
            // the next tile's source chunks, if it lies under other rows: on their way while this tile is worked on
            // (the registers that hold them are free: whatever was requested before has been stored)
            const Rows nx = rows_of(tile + 1);
            if (!(nx.n == s_n && nx.y_lo == s_ylo && nx.y_hi == s_yhi && nx.x_lo == s_xlo)) request(nx);
        }
        if constexpr (LOSS) {
            if (fresh) {
                for (int j = tid; j < rows * WC; j += THREADS) {
                    const f32x4_t *p = reinterpret_cast<const f32x4_t *>(stage + j * CPAD);
                    // NaN-propagating maxima (v_maximum3_f32: no canonicalising moves); a NaN here sends the pixels around it
                    // to the exact walk
                    float m = kNegFill;
#pragma unroll
                    for (int s4 = 0; s4 < CPAD / 4; ++s4) {
                        const f32x4_t x = p[s4];
                        m = __builtin_elementwise_maximum(__builtin_elementwise_maximum(m, x.x), x.y);
                        m = __builtin_elementwise_maximum(__builtin_elementwise_maximum(m, x.z), x.w);
                    }
                    smax[j] = m;
                }
                __syncthreads();
            }
        }
        if (!live) continue;
        int oy;
        if (same_columns) {
            oy = oy_first + lane_row;
        } else {
            oy = (HW < (1 << 22)) ? (int)(((float)r + 0.5f) * inv_w) : r / W;
            column(r - oy * W);
        }
        float h0, h1;
        int jt, jb;   // the corner's source pixel in the stage, upper and lower row
        {
            const float sy = src_index(h.scale_h, oy, h.align_corners);
            const int y0 = min((int)sy, y_hi);
            const int y1 = min(y0 + 1, y_hi);
            h1 = sy - (float)y0;
            h0 = 1.0f - h1;
            const int jc = xoff / CPAD - x_lo;   // CPAD is a constant: a multiplication and a shift
            jt = (y0 - y_lo) * WC + jc;
            jb = (y1 - y_lo) * WC + jc;
        }
        const float *top = stage + jt * CPAD, *bot = stage + jb * CPAD;
        float best = -__builtin_inff();
        int arg = 0;
        f32x2_t zs2 = (f32x2_t)(0.f), z12 = (f32x2_t)(0.f), ps2 = (f32x2_t)(0.f);
        float M = 0.f;
        if constexpr (LOSS)
            M = __builtin_elementwise_maximum(__builtin_elementwise_maximum(smax[jt], smax[jt + 1]),
                                              __builtin_elementwise_maximum(smax[jb], smax[jb + 1]));
        const float mk = -M * k2;
#pragma unroll
        for (int c0 = 0; c0 < CPAD; c0 += 4) {
            if (c0 >= C) break;   // wave-uniform
            head2_step<CPAD, TPOW, LOSS>(top, bot, w0, w1, h0, h1, c0, best, arg, mk, k2, zs2, z12, ps2, tp);
        }
        const float zs = zs2.x + zs2.y, z1 = z12.x + z12.y, ps = ps2.x + ps2.y;
        arg = arg >= C ? 0 : arg;   // a padding class wins only where every real logit is -infinity: class 0
        if (best != best) {
            // a NaN among the pixel's logits: the exact class-by-class walk (a NaN is the maximum, the first one wins)
            float bv = head2_one(stage, jt * CPAD, jb * CPAD, CPAD, 0, w0, w1, h0, h1);
            arg = 0;
            for (int c = 1; c < C; ++c) {
                const float v = head2_one(stage, jt * CPAD, jb * CPAD, CPAD, c, w0, w1, h0, h1);
                const bool take = (v > bv) || (v != v && bv == bv);
                bv = take ? v : bv;
                arg = take ? c : arg;
            }
        }
        // one unsigned comparison decides "a class"
        const int id = ((unsigned long long)lab < (unsigned long long)C) ? (int)lab : ((LOSS && lab == a.ignore_index) ? -1 : -2);
        if (id >= 0) atomicAdd(&hist[id * C + arg], 1u);
        if constexpr (!LOSS) continue;

        float Zs = zs, Z1 = TPOW > 1 ? z1 : zs, ms = M;
        if (!(Z1 > kHead2TinySum) || !(Zs < 3.0e38f)) {
            // M far above the pixel's own logits, or a logit that is not finite: the sums about the exact running
            // maximum, as the first form keeps them
            Zs = Z1 = 0.f;
            ms = kNegFill;
            for (int c = 0; c < C; ++c) {
                const float v = head2_one(stage, jt * CPAD, jb * CPAD, CPAD, c, w0, w1, h0, h1);
                const float mn = fmaxf(ms, v);
                const float ds = ms - mn;
                Zs = Zs * fast_exp2(ds * k2) + fast_exp2((v - mn) * k2);
                Z1 = Z1 * fast_exp2(ds * kLog2e) + fast_exp2((v - mn) * kLog2e);
                ms = mn;
            }
        }
        if (id == -2) ++bad;
        const float sy = id >= 0 ? head2_one(stage, jt * CPAD, jb * CPAD, CPAD, id, w0, w1, h0, h1) : 0.f;
        const float kl = hn + a.invT * (ms - ps) + fast_ln(Zs);
        const float nll = fast_ln(Z1) - (sy - ms);
        const float wy = (id >= 0) ? (a.class_w ? a.class_w[id] : 1.f) : 0.f;
        if (id != -1) {
            acc[0] += kl;
            acc[1] += 1.0;
        }
        if (id >= 0) {
            acc[2] += wy * nll;
            acc[3] += wy;
        }
    }
    if (LOSS && bad) atomicAdd(reinterpret_cast<unsigned long long *>(a.bad), (unsigned long long)bad);
    __syncthreads();   // (a workgroup without a tile has not passed a barrier since it cleared the histogram)
    if constexpr (!LOSS) {
        conf_finish<THREADS>(lds_hist, h.copies, h.stride, C * C, 0u, h.tail);
    } else {
        // Both tails at once.  Separately (conf_finish, then kd_publish_and_finalize) they are two chains of dependent
        // trips to memory, one behind the other: stores -> wait -> ticket -> last arriver's loads.  Here the workgroup
        // stores its histogram row AND its four loss sums, waits once, and two lanes take the two tickets (the
        // histogram's group ticket, the loss's grid-wide one) in the same trip.
        const int bins = C * C;
        {
            const int wave = tid >> 6, lane = tid & 63;
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                const double v = wave_sum(acc[k]);
                if (lane == 0) red[wave][k] = v;
            }
        }
        unsigned *row = h.tail.partial + (int64_t)blockIdx.x * bins;
        for (int b = tid; b < bins; b += THREADS) {
            unsigned sum = 0;
            for (int c = 0; c < h.copies; ++c) sum += lds_hist[c * h.stride + b];
            __hip_atomic_store(row + b, sum, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __syncthreads();
        if (tid < 8) {
            double v = 0.0;
            if (tid < 4)
                for (int wv = 0; wv < THREADS / 64; ++wv) v += red[wv][tid];
            __hip_atomic_store(reinterpret_cast<unsigned long long *>(a.main_partial) + (long long)blockIdx.x * 8 + tid,
                               (unsigned long long)__double_as_longlong(v), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        __shared__ int last_of_group, last_of_grid;
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        const int g = blockIdx.x / h.tail.group;
        const int first = g * h.tail.group;
        const int members = min(h.tail.group, (int)gridDim.x - first);
        if (tid == 0) {
            const unsigned k = __hip_atomic_fetch_add(h.tail.tickets + g, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (k == (unsigned)(members - 1));
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                __hip_atomic_store(h.tail.tickets + g, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
            last_of_group = last;
        }
        if (tid == 64) {   // another wave: its ticket travels beside the first
            const unsigned t = __hip_atomic_fetch_add(a.ticket, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            const int last = (t == gridDim.x - 1);
            if (last) {
                __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
                asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            }
            last_of_grid = last;
        }
        __syncthreads();
        if (last_of_group) conf_fold_group<THREADS>(h.tail, first, members, bins);
        if (last_of_grid) {
            kd_finalize<THREADS>(a.main_partial, (int)gridDim.x, a.totals, a.out3, a.T, a.alpha, a.bw, a.conf);
            if (tid == 0) __hip_atomic_store(a.ticket, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
    }
}


__global__ __launch_bounds__(256) void scale_inplace_kernel(float *x, int64_t n, const float *scale) {
    const float s = *scale;
    if (s == 1.0f) return;
    const int64_t n4 = ((reinterpret_cast<uintptr_t>(x) & 15u) == 0) ? (n >> 2) : 0;
    float4 *x4 = reinterpret_cast<float4 *>(x);
    const int64_t stride = (int64_t)gridDim.x * 256;
    for (int64_t i = (int64_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        float4 v = x4[i];
        v.x *= s;
        v.y *= s;
        v.z *= s;
        v.w *= s;
        x4[i] = v;
    }
    for (int64_t i = (n4 << 2) + (int64_t)blockIdx.x * 256 + threadIdx.x; i < n; i += stride) x[i] *= s;
}

constexpr size_t kLdsPerCu = 160 * 1024;

// Workgroups of the loss kernel, each looping over its tiles.  Measured on MI355X (256 CUs): the
// plain loss inside the sweep step 65.8 us with two per CU and 66.8 us with three (standalone on
// cold inputs 65.8 / 64.1 us); with the boundary / confidence options, whose register use allows
// exactly three resident workgroups per CU, 76.9 us with two and 70.1 us with three.  With few classes
// a tile is short (48 bytes per pixel at 5 classes against 176 at 21) and more workgroups are needed
// to keep as many bytes in flight: [32,5,576,576] takes 118 us with two per CU, 102 us with four.
SPICE_ENV_KNOB(knob_kd_max_blocks, "SPICE_KD_MAX_BLOCKS", 0)       // tuning knobs (host side only)
SPICE_ENV_KNOB(knob_kd_stage, "SPICE_KD_STAGE", 1)                 // 0 = never stage the gradient pass in LDS
SPICE_ENV_KNOB(knob_kd_regs, "SPICE_KD_REGS", 1)                   // 0 = never keep the logits in registers for the gradient
SPICE_ENV_KNOB(knob_kd_stage_blocks, "SPICE_KD_STAGE_BLOCKS", 0)

static int kd_max_blocks(bool ext, int64_t classes) {
    const int v = (int)knob_kd_max_blocks();
    if (v >= 1 && v <= kKdMaxBlocks) return v;
    const int per_cu = classes <= 8 ? 4 : (ext ? 3 : 2);
    const int d = per_cu * device_cu_count();
    return d < kKdMaxBlocks ? d : kKdMaxBlocks;
}

static int kd_stage_mode() { return (int)knob_kd_stage(); }

static int kd_regs_mode() { return (int)knob_kd_regs(); }

template <int VEC>
static int kd_launch(KdArgs a, bool grad, bool ext, int blocks, hipStream_t st) {
    const dim3 g(blocks), b(kKdThreads);
    if (grad) {
        kd_launch_prepass<VEC>(a, ext, st);
        SPICE_LAUNCH_CHECK();
        if (ext)
            hipLaunchKernelGGL((kdlite_main_kernel<VEC, true, true, false>), g, b, 0, st, a);
        else
            hipLaunchKernelGGL((kdlite_main_kernel<VEC, true, false, false>), g, b, 0, st, a);
    } else {
        if (ext)
            hipLaunchKernelGGL((kdlite_main_kernel<VEC, false, true, false>), g, b, 0, st, a);
        else
            hipLaunchKernelGGL((kdlite_main_kernel<VEC, false, false, false>), g, b, 0, st, a);
    }
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}

// Gradient with the logits staged in LDS: one pixel per lane, so a workgroup's stage is
// C * 2 * 256 * 4 bytes (43 KB at 21 classes: three workgroups per CU).
static int kd_launch_staged(KdArgs a, bool ext, bool vec4, int blocks, size_t lds_bytes, hipStream_t st) {
    const dim3 g(blocks), b(kKdThreads);
    if (vec4)  // the pre-pass keeps its 16-byte accesses when the layout allows
        kd_launch_prepass<4>(a, ext, st);
    else
        kd_launch_prepass<1>(a, ext, st);
    SPICE_LAUNCH_CHECK();
    if (ext)
        hipLaunchKernelGGL((kdlite_main_kernel<1, true, true, true>), g, b, lds_bytes, st, a);
    else
        hipLaunchKernelGGL((kdlite_main_kernel<1, true, false, true>), g, b, lds_bytes, st, a);
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}

// Gradient with the logits kept in registers (C <= 8, 16-byte layout): label pre-pass + one pass.
static int kd_launch_regs(KdArgs a, bool ext, int blocks, hipStream_t st) {
    const dim3 g(blocks), b(kKdThreads);
    kd_launch_prepass<4>(a, ext, st);
    SPICE_LAUNCH_CHECK();
    if (ext)
        hipLaunchKernelGGL((kdlite_grad_regs_kernel<true>), g, b, 0, st, a);
    else
        hipLaunchKernelGGL((kdlite_grad_regs_kernel<false>), g, b, 0, st, a);
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}

}  // namespace spice

using namespace spice;

extern "C" size_t spice_kdlite_workspace_bytes(int64_t, int64_t, int64_t, int64_t) {
    return kKdWorkspace;
}

static int kdlite_forward_impl(const float *student, const float *teacher, const int64_t *labels,
                               int64_t n, int64_t c, int64_t h, int64_t w,
                               const spice_kdlite_params *params, float *out3, float *grad,
                               float coef_ce, float coef_kd, const float *upstream,
                               void *workspace, size_t workspace_bytes, void *stream, bool pre_ready) {
    if (!params || !out3 || !workspace) return SPICE_ERR_ARG;
    if (n < 0 || c < 1 || h < 0 || w < 0) return SPICE_ERR_ARG;
    if (c > INT32_MAX || h > INT32_MAX || w > INT32_MAX) return SPICE_ERR_UNSUPPORTED;
    if (!(params->temperature > 0.f)) return SPICE_ERR_ARG;
    if (workspace_bytes < kKdWorkspace) return SPICE_ERR_WORKSPACE;
    if (!aligned16(workspace)) return SPICE_ERR_ARG;
    const int64_t hw = h * w;
    const int64_t pixels = n * hw;
    if (pixels > 0 && (!student || !teacher || !labels)) return SPICE_ERR_ARG;
    if ((reinterpret_cast<uintptr_t>(student) & 3u) || (reinterpret_cast<uintptr_t>(teacher) & 3u) ||
        (reinterpret_cast<uintptr_t>(grad) & 3u) || !aligned8(labels))
        return SPICE_ERR_ARG;
    hipStream_t st = static_cast<hipStream_t>(stream);
    char *ws = static_cast<char *>(workspace);

    const bool ext = params->boundary_weight > 1.0f || params->confidence_weight != 0;
    if (pre_ready && (ext || grad == nullptr)) return SPICE_ERR_UNSUPPORTED;   // (those forms have no separate label pass)
    const bool vec4 = (hw % 4 == 0) && aligned16(student) && aligned16(teacher) && aligned16(labels) &&
                      (grad == nullptr || aligned16(grad));
    const int vec = vec4 ? 4 : 1;
    if (hw > INT32_MAX - 4096) return SPICE_ERR_UNSUPPORTED;
    const int64_t tiles = n * ceil_div(hw / vec, kKdThreads);
    if (tiles > INT32_MAX) return SPICE_ERR_UNSUPPORTED;
    int64_t blocks = tiles;
    if (blocks < 1) blocks = 1;
    if (blocks > kd_max_blocks(ext, c)) blocks = kd_max_blocks(ext, c);

    KdArgs a;
    a.s = student;
    a.t = teacher;
    a.labels = reinterpret_cast<const long long *>(labels);
    a.grad = grad;
    a.class_w = params->class_weight;
    a.upstream = upstream;
    a.pre_partial = reinterpret_cast<double *>(ws + kKdOffPre);
    a.main_partial = reinterpret_cast<double *>(ws + kKdOffMain);
    a.bad = reinterpret_cast<long long *>(ws + kKdOffBad);
    a.ticket = reinterpret_cast<unsigned *>(ws + kKdOffTicket);
    a.totals = reinterpret_cast<double *>(ws + kKdOffTotals);
    a.out3 = out3;
    a.N = n;
    a.HW = hw;
    a.C = (int)c;
    a.H = (int)h;
    a.W = (int)w;
    a.pre_blocks = (int)blocks;
    a.ignore_index = params->ignore_index;
    a.T = params->temperature;
    a.invT = 1.0f / params->temperature;
    a.alpha = params->alpha;
    a.bw = params->boundary_weight;
    a.coef_ce = coef_ce;
    a.coef_kd = coef_kd;
    a.conf = params->confidence_weight != 0;
    a.pre_ready = pre_ready ? 1 : 0;

    if (grad != nullptr && vec4 && c <= 8 && kd_regs_mode() != 0) return kd_launch_regs(a, ext, (int)blocks, st);
    const size_t stage_bytes = (size_t)c * 2 * kKdThreads * sizeof(float);
    if (grad != nullptr && kd_stage_mode() != 0 && stage_bytes <= kKdMaxStageBytes) {
        int64_t b1 = n * ceil_div(hw, kKdThreads);  // tiles at one pixel per lane
        if (b1 > INT32_MAX) return SPICE_ERR_UNSUPPORTED;
        if (b1 < 1) b1 = 1;
        {
            // exactly the workgroups that are resident at once (LDS-limited), each looping over
            // its tiles: measured best on MI355X (768: 121 us, 1536: 130 us, 2048: 138 us)
            int64_t per_cu = (int64_t)(kLdsPerCu / (stage_bytes + 1024));
            if (per_cu > 8) per_cu = 8;
            if (per_cu < 1) per_cu = 1;
            int64_t lim = per_cu * device_cu_count();
            const int cap = (int)knob_kd_stage_blocks();
            if (cap >= 1) lim = cap;
            if (lim > kKdMaxBlocks) lim = kKdMaxBlocks;
            if (b1 > lim) b1 = lim;
        }
        a.pre_blocks = (int)b1;
        return kd_launch_staged(a, ext, vec4, (int)b1, stage_bytes, st);
    }
    return vec4 ? kd_launch<4>(a, grad != nullptr, ext, (int)blocks, st)
                : kd_launch<1>(a, grad != nullptr, ext, (int)blocks, st);
}

extern "C" int spice_kdlite_forward(const float *student, const float *teacher, const int64_t *labels,
                                    int64_t n, int64_t c, int64_t h, int64_t w,
                                    const spice_kdlite_params *params, float *out3, float *grad,
                                    float coef_ce, float coef_kd, const float *upstream,
                                    void *workspace, size_t workspace_bytes, void *stream) {
    return kdlite_forward_impl(student, teacher, labels, n, c, h, w, params, out3, grad, coef_ce, coef_kd, upstream,
                               workspace, workspace_bytes, stream, false);
}

extern "C" int spice_kdlite_forward_prepared(const float *student, const float *teacher, const int64_t *labels,
                                             int64_t n, int64_t c, int64_t h, int64_t w,
                                             const spice_kdlite_params *params, float *out3, float *grad,
                                             float coef_ce, float coef_kd, const float *upstream,
                                             void *workspace, size_t workspace_bytes, void *stream) {
    return kdlite_forward_impl(student, teacher, labels, n, c, h, w, params, out3, grad, coef_ce, coef_kd, upstream,
                               workspace, workspace_bytes, stream, true);
}

extern "C" int spice_kdlite_label_pass(const int64_t *labels, int64_t n, int64_t c, int64_t h, int64_t w,
                                       const spice_kdlite_params *params, void *workspace, size_t workspace_bytes,
                                       void *stream) {
    if (!params || !workspace) return SPICE_ERR_ARG;
    if (n < 0 || c < 1 || h < 0 || w < 0) return SPICE_ERR_ARG;
    if (c > INT32_MAX || h > INT32_MAX || w > INT32_MAX) return SPICE_ERR_UNSUPPORTED;
    if (workspace_bytes < kKdWorkspace) return SPICE_ERR_WORKSPACE;
    if (!aligned16(workspace) || !aligned8(labels)) return SPICE_ERR_ARG;
    if (params->boundary_weight > 1.0f || params->confidence_weight != 0) return SPICE_ERR_UNSUPPORTED;   // their pass reads the teacher
    const int64_t hw = h * w;
    if (n * hw > 0 && !labels) return SPICE_ERR_ARG;
    if (hw > INT32_MAX - 4096) return SPICE_ERR_UNSUPPORTED;
    KdArgs a = KdArgs{};
    a.labels = reinterpret_cast<const long long *>(labels);
    a.class_w = params->class_weight;
    a.pre_partial = reinterpret_cast<double *>(static_cast<char *>(workspace) + kKdOffPre);
    a.N = n;
    a.HW = hw;
    a.C = (int)c;
    a.ignore_index = params->ignore_index;
    kd_launch_prepass<1>(a, false, static_cast<hipStream_t>(stream));
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}

SPICE_ENV_KNOB(knob_head_blocks_per_cu, "SPICE_HEAD_BLOCKS_PER_CU", 0)   // tuning knobs (host side only)

constexpr size_t kHeadMaxLdsBytes = 64 * 1024;   // two workgroups per compute unit at least

extern "C" int spice_kd_teacher_prepare(const float *teacher, int64_t n, int64_t c, int64_t h, int64_t w,
                                        float temperature, float *probs, float *hneg, void *stream) {
    if (n < 0 || c < 1 || h < 0 || w < 0 || !(temperature > 0.f)) return SPICE_ERR_ARG;
    const int64_t hw = h * w;
    if (n == 0 || hw == 0) return SPICE_OK;
    if (!teacher || !probs || !hneg) return SPICE_ERR_ARG;
    if (c > INT32_MAX || hw > INT32_MAX - 4096) return SPICE_ERR_UNSUPPORTED;
    if (hw % 4 != 0 || !aligned16(teacher) || !aligned16(probs) || !aligned16(hneg)) return SPICE_ERR_UNSUPPORTED;
    int64_t blocks = ceil_div(n * (hw / 4), 256);
    if (blocks > 16 * (int64_t)device_cu_count()) blocks = 16 * (int64_t)device_cu_count();
    hipLaunchKernelGGL(kd_teacher_prepare_kernel, dim3((unsigned)blocks), dim3(256), 0, static_cast<hipStream_t>(stream),
                       teacher, probs, hneg, (long long)n, (long long)hw, (int)c, 1.0f / temperature);
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}

namespace spice {

// Geometry, LDS stage and grid shared by the head kernel's entry points.  `params` and the loss workspace are
// absent for the arg-max-only form.  `form` 2 sizes the class-innermost stage of kdlite_head2_kernel (returns
// SPICE_ERR_UNSUPPORTED where that form does not apply: more than 28 classes, a stage beyond its LDS budget), `form` 1
// the plane-major stage of kdlite_head_kernel.  Returns SPICE_OK with `ha`, `blocks`, `lds_bytes` and `chunks` set.
SPICE_ENV_KNOB(knob_head_form, "SPICE_HEAD_FORM", 0)   // 1 = always the first form (A/B measurements)

constexpr size_t kHead2MaxLdsBytes = 63 * 1024;   // with the kernel's static LDS within the 64 KiB a launch may ask for unannounced

static int head2_cpad(int64_t c) {   // the class axis padded to a multiple of 4 floats whose quarter is odd
    int cpad = (int)((c + 3) / 4) * 4;
    if ((cpad / 4) % 2 == 0) cpad += 4;
    return cpad;
}

static int head_prepare(HeadArgs &ha, const void *low, const int64_t *labels, int64_t n, int64_t c, int64_t in_h,
                        int64_t in_w, int64_t h, int64_t w, int32_t align_corners, float scale_h, float scale_w,
                        int64_t *conf, void *conf_workspace, size_t conf_workspace_bytes, bool loss, int form, int64_t &blocks,
                        size_t &lds_bytes, int64_t &chunks) {
    if (!conf || !conf_workspace || !low || !labels) return SPICE_ERR_ARG;
    if (n < 1 || c < 1 || h < 1 || w < 1 || in_h < 1 || in_w < 1) return SPICE_ERR_ARG;
    if (!aligned16(conf_workspace) || !aligned8(conf)) return SPICE_ERR_ARG;
    if (c > 1024 || c * c > kConfLdsBins || h > INT32_MAX || w > INT32_MAX || in_h > h || in_w > w) return SPICE_ERR_UNSUPPORTED;
    const int64_t hw = h * w;
    if (hw > INT32_MAX - 4096 || n > kConfMaxPixels / hw) return SPICE_ERR_UNSUPPORTED;
    if (w % 4 != 0 || in_w % 8 != 0) return SPICE_ERR_UNSUPPORTED;   // a lane's four pixels lie in one row; 16-byte source chunks
    if (!aligned16(low) || !aligned16(labels)) return SPICE_ERR_UNSUPPORTED;
    if (conf_workspace_bytes < spice_confusion_workspace_bytes(n * hw, (int32_t)c)) return SPICE_ERR_WORKSPACE;

    ha.low = static_cast<const __half *>(low);
    ha.hneg = nullptr;
    ha.in_h = (int)in_h;
    ha.in_w = (int)in_w;
    ha.align_corners = align_corners ? 1 : 0;
    if (ha.align_corners) {   // torch's area_pixel_compute_scale, as spice_upsample_bilinear
        ha.scale_h = h > 1 ? static_cast<float>(in_h - 1) / static_cast<float>(h - 1) : 0.0f;
        ha.scale_w = w > 1 ? static_cast<float>(in_w - 1) / static_cast<float>(w - 1) : 0.0f;
    } else {
        ha.scale_h = scale_h > 0.0f ? 1.0f / scale_h : static_cast<float>(in_h) / static_cast<float>(h);
        ha.scale_w = scale_w > 0.0f ? 1.0f / scale_w : static_cast<float>(in_w) / static_cast<float>(w);
    }
    if (ha.scale_h > 1.0f || ha.scale_w > 1.0f) return SPICE_ERR_UNSUPPORTED;
    // output rows a tile of consecutive pixels (1024 in the first form, 512 in the second) can touch, and the source
    // rows under them: the span of the upper rows, the lower neighbour of the last, one more where the truncations fall
    const int64_t tile_px = form == 2 ? (int64_t)head2_threads() : (int64_t)kKdThreads * 4;
    int64_t out_rows = (tile_px % w == 0) ? tile_px / w : tile_px / w + 2;
    if (out_rows > h) out_rows = h;
    const bool one_row = (w % tile_px == 0);   // every tile lies inside one output row
    int64_t lds_rows = one_row ? 2 : (int64_t)((float)(out_rows - 1) * ha.scale_h + 1e-3f) + 3;
    if (lds_rows > in_h) lds_rows = in_h;
    ha.lds_rows = (int)lds_rows;
    ConfGeom g = conf_geom((int)c);
    int max_per_cu = loss ? 2 : 3;
    if (form == 2) {
        if (c > 28 || (int64_t)head2_cpad(c) * hw * 4 >= ((int64_t)1 << 31)) return SPICE_ERR_UNSUPPORTED;   // an image's planes as one buffer
        const int cpad = head2_cpad(c);
        // the source columns under a tile: whole rows, or (a tile inside one output row) the columns under its
        // pixels, from a multiple of 8 on
        int64_t win = in_w;
        ha.windowed = 0;
        if (one_row) {
            // first to last left neighbour: the whole part of (tile - 1) * scale, one more where the fractions carry;
            // then the last one's right neighbour
            const int64_t span = (int64_t)((float)(tile_px - 1) * ha.scale_w + 1e-3f) + 3;
            const int64_t need = (span + 7 + 7) / 8 * 8;
            if (need < in_w) {
                win = need;
                ha.windowed = 1;
            }
        }
        ha.win_cols = (int)win;
        const size_t src_px = (size_t)lds_rows * (win + 1);
        const size_t stage_bytes = src_px * cpad * sizeof(float) + ((src_px + 3) / 4 * 4) * sizeof(float);
        // histogram copies: as many as leave room for three workgroups per compute unit
        size_t hist_bytes = (size_t)g.copies * g.stride * 4;
        while (g.copies > 2 && (size_t)(3 * (512 / head2_threads())) * (stage_bytes + hist_bytes + 512) > kLdsPerCu) {
            g.copies >>= 1;
            hist_bytes = (size_t)g.copies * g.stride * 4;
        }
        lds_bytes = stage_bytes + hist_bytes;
        if (lds_bytes > kHead2MaxLdsBytes) return SPICE_ERR_UNSUPPORTED;
        chunks = c * lds_rows * (win / 8);
        if (chunks >= (1 << 20)) return SPICE_ERR_UNSUPPORTED;   // the float reciprocals that split a chunk number stay exact
        max_per_cu = (int)(kLdsPerCu / (lds_bytes + 512));
        // workgroups per compute unit: 512 lanes at four waves per SIMD (the loss form's registers) are two; the
        // arg-max form's registers allow three, but two are faster (29.6 against 30.9 us at [8,21,128,128] -> 512 x 512:
        // fewer rows for the tail to fold), and a second round of workgroups costs either form more than its tail
        // hidden behind the first round saves (loss form: 51.7 / 61.1 / 59.3 us with 2 / 3 / 4 per compute unit)
        // (the arg-max form on long inputs - [32,19,256,512] -> 1024 x 2048 is 131,072 tiles - is all interpolation
        // and no tail to speak of: three per compute unit, 715 against 807 us)
        const bool long_input = !loss && n * ceil_div(hw, tile_px) >= 16 * 3 * device_cu_count();
        const int reg_limit = (long_input ? 3 : 2) * (512 / head2_threads());
        if (max_per_cu > reg_limit) max_per_cu = reg_limit;
        if (max_per_cu < 1) max_per_cu = 1;
    } else {
        ha.win_cols = (int)in_w;
        ha.windowed = 0;
        if (one_row) ha.lds_rows = (int)(lds_rows = (lds_rows + 1 > in_h ? in_h : lds_rows + 1));   // the first form's own bound
        const size_t stage_bytes = (size_t)c * lds_rows * (in_w + 1) * sizeof(float);
        lds_bytes = stage_bytes + (size_t)g.copies * g.stride * 4;
        // the kernel's own static LDS (partial sums, flags) comes on top of the dynamic part
        if (lds_bytes + 1024 > kHeadMaxLdsBytes) return SPICE_ERR_UNSUPPORTED;
        chunks = c * lds_rows * (in_w / 8);   // 16-byte source chunks of a tile
        if (chunks > 8 * kKdThreads) return SPICE_ERR_UNSUPPORTED;
    }
    ha.copies = g.copies;
    ha.stride = g.stride;

    const int64_t tiles = n * ceil_div(hw, tile_px);
    if (tiles > INT32_MAX) return SPICE_ERR_UNSUPPORTED;
    // workgroups per compute unit, each looping over its tiles.  Measured on MI355X at [8,21,128,128] -> 512 x 512
    // (HIP events), first form: the loss forms 109 / 90 / 92 / 95 us with 1 / 2 / 3 / 4 (their registers allow two waves
    // per SIMD), the arg-max-only form 61 / 45 / 40 / 47 us
    const long per_cu = knob_head_blocks_per_cu() >= 1 ? knob_head_blocks_per_cu() : max_per_cu;
    blocks = tiles;
    int64_t resident = per_cu * device_cu_count();
    if (resident > kConfMaxBlocks) resident = kConfMaxBlocks;   // also within kKdMaxBlocks
    if (blocks > resident) {
        // every workgroup the same number of tiles (the last ones one fewer): 2048 tiles on 768 resident workgroups
        // are 683 workgroups of three, not 768 of which two thirds take three
        const int64_t per_wg = ceil_div(tiles, resident);
        blocks = form == 2 ? ceil_div(tiles, per_wg) : resident;
    }

    KdArgs &a = ha.kd;
    a = KdArgs{};
    a.labels = reinterpret_cast<const long long *>(labels);
    a.N = n;
    a.HW = hw;
    a.C = (int)c;
    a.H = (int)h;
    a.W = (int)w;
    a.pre_blocks = (int)blocks;
    a.ignore_index = 255;
    a.T = a.invT = 1.0f;
    ha.tail = conf_tail(conf_workspace, conf, nullptr, (int)blocks);
    return SPICE_OK;
}

template <int CPAD, bool LOSS, int THREADS>
static void head2_launch_tt(int tpow, dim3 grid, size_t lds_bytes, hipStream_t st, const HeadArgs &ha) {
    const dim3 block(THREADS);
    if constexpr (!LOSS) {
        hipLaunchKernelGGL((kdlite_head2_kernel<CPAD, 1, false, THREADS>), grid, block, lds_bytes, st, ha);
    } else {
        switch (tpow) {
            case 1: hipLaunchKernelGGL((kdlite_head2_kernel<CPAD, 1, true, THREADS>), grid, block, lds_bytes, st, ha); break;
            case 2: hipLaunchKernelGGL((kdlite_head2_kernel<CPAD, 2, true, THREADS>), grid, block, lds_bytes, st, ha); break;
            case 3: hipLaunchKernelGGL((kdlite_head2_kernel<CPAD, 3, true, THREADS>), grid, block, lds_bytes, st, ha); break;
            default: hipLaunchKernelGGL((kdlite_head2_kernel<CPAD, 4, true, THREADS>), grid, block, lds_bytes, st, ha); break;
        }
    }
}

template <int CPAD, bool LOSS>
static void head2_launch_t(int tpow, dim3 grid, size_t lds_bytes, hipStream_t st, const HeadArgs &ha) {
    if (head2_threads() == 256)
        head2_launch_tt<CPAD, LOSS, 256>(tpow, grid, lds_bytes, st, ha);
    else
        head2_launch_tt<CPAD, LOSS, 512>(tpow, grid, lds_bytes, st, ha);
}

template <bool LOSS>
static int head2_launch(int cpad, int tpow, int64_t blocks, size_t lds_bytes, hipStream_t st, const HeadArgs &ha) {
    const dim3 grid((unsigned)blocks);
    switch (cpad) {
        case 4: head2_launch_t<4, LOSS>(tpow, grid, lds_bytes, st, ha); break;
        case 12: head2_launch_t<12, LOSS>(tpow, grid, lds_bytes, st, ha); break;
        case 20: head2_launch_t<20, LOSS>(tpow, grid, lds_bytes, st, ha); break;
        case 28: head2_launch_t<28, LOSS>(tpow, grid, lds_bytes, st, ha); break;
        default: return SPICE_ERR_UNSUPPORTED;
    }
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}

}  // namespace spice

extern "C" int spice_sweep_head_f16(const void *low, const float *teacher, const float *teacher_hneg,
                                    const int64_t *labels, int64_t n, int64_t c, int64_t in_h, int64_t in_w, int64_t h,
                                    int64_t w, int32_t align_corners, float scale_h, float scale_w,
                                    const spice_kdlite_params *params, float *out3, int64_t *conf,
                                    void *kd_workspace, size_t kd_workspace_bytes, void *conf_workspace,
                                    size_t conf_workspace_bytes, void *stream) {
    if (!params || !out3 || !kd_workspace || !teacher) return SPICE_ERR_ARG;
    if (!(params->temperature > 0.f)) return SPICE_ERR_ARG;
    if (kd_workspace_bytes < kKdWorkspace) return SPICE_ERR_WORKSPACE;
    if (!aligned16(kd_workspace)) return SPICE_ERR_ARG;
    // what this kernel does not cover is left to the three separate launches
    if (params->boundary_weight > 1.0f || params->confidence_weight != 0) return SPICE_ERR_UNSUPPORTED;
    if (!aligned16(teacher) || !aligned16(teacher_hneg)) return SPICE_ERR_UNSUPPORTED;
    HeadArgs ha;
    int64_t blocks = 0, chunks = 0;
    size_t lds_bytes = 0;
    // the class-innermost form takes the prepared teacher at the integer temperatures
    const float T = params->temperature;
    const int tpow = (T == 1.0f) ? 1 : (T == 2.0f) ? 2 : (T == 3.0f) ? 3 : (T == 4.0f) ? 4 : 0;
    int form = (teacher_hneg != nullptr && tpow != 0 && knob_head_form() != 1) ? 2 : 1;
    int rc = head_prepare(ha, low, labels, n, c, in_h, in_w, h, w, align_corners, scale_h, scale_w, conf,
                          conf_workspace, conf_workspace_bytes, true, form, blocks, lds_bytes, chunks);
    if (rc == SPICE_ERR_UNSUPPORTED && form == 2) {
        form = 1;
        rc = head_prepare(ha, low, labels, n, c, in_h, in_w, h, w, align_corners, scale_h, scale_w, conf,
                          conf_workspace, conf_workspace_bytes, true, form, blocks, lds_bytes, chunks);
    }
    if (rc != SPICE_OK) return rc;
    ha.hneg = teacher_hneg;
    char *ws = static_cast<char *>(kd_workspace);
    KdArgs &a = ha.kd;
    a.t = teacher;
    a.class_w = params->class_weight;
    a.pre_partial = reinterpret_cast<double *>(ws + kKdOffPre);
    a.main_partial = reinterpret_cast<double *>(ws + kKdOffMain);
    a.bad = reinterpret_cast<long long *>(ws + kKdOffBad);
    a.ticket = reinterpret_cast<unsigned *>(ws + kKdOffTicket);
    a.totals = reinterpret_cast<double *>(ws + kKdOffTotals);
    a.out3 = out3;
    a.ignore_index = params->ignore_index;
    a.T = params->temperature;
    a.invT = 1.0f / params->temperature;
    a.alpha = params->alpha;
    a.bw = params->boundary_weight;
    a.coef_ce = params->alpha;
    a.coef_kd = 1.0f - params->alpha;
    a.conf = 0;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (form == 2) return head2_launch<true>(head2_cpad(c), tpow, blocks, lds_bytes, st, ha);
    const dim3 grid((unsigned)blocks), block(kKdThreads);
    const bool probs = teacher_hneg != nullptr;
    if (chunks > 4 * kKdThreads) {
        if (probs)
            hipLaunchKernelGGL((kdlite_head_kernel<4, 8, true>), grid, block, lds_bytes, st, ha);
        else
            hipLaunchKernelGGL((kdlite_head_kernel<4, 8, false>), grid, block, lds_bytes, st, ha);
    } else {
        if (probs)
            hipLaunchKernelGGL((kdlite_head_kernel<4, 4, true>), grid, block, lds_bytes, st, ha);
        else
            hipLaunchKernelGGL((kdlite_head_kernel<4, 4, false>), grid, block, lds_bytes, st, ha);
    }
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}

extern "C" int spice_upsample_argmax_confusion_f16(const void *low, const int64_t *labels, int64_t n, int64_t c,
                                                   int64_t in_h, int64_t in_w, int64_t h, int64_t w,
                                                   int32_t align_corners, float scale_h, float scale_w, int64_t *conf,
                                                   void *conf_workspace, size_t conf_workspace_bytes, void *stream) {
    HeadArgs ha;
    int64_t blocks = 0, chunks = 0;
    size_t lds_bytes = 0;
    int form = knob_head_form() != 1 ? 2 : 1;
    int rc = head_prepare(ha, low, labels, n, c, in_h, in_w, h, w, align_corners, scale_h, scale_w, conf,
                          conf_workspace, conf_workspace_bytes, false, form, blocks, lds_bytes, chunks);
    if (rc == SPICE_ERR_UNSUPPORTED && form == 2) {
        form = 1;
        rc = head_prepare(ha, low, labels, n, c, in_h, in_w, h, w, align_corners, scale_h, scale_w, conf,
                          conf_workspace, conf_workspace_bytes, false, form, blocks, lds_bytes, chunks);
    }
    if (rc != SPICE_OK) return rc;
    hipStream_t st = static_cast<hipStream_t>(stream);
    if (form == 2) return head2_launch<false>(head2_cpad(c), 1, blocks, lds_bytes, st, ha);
    const dim3 grid((unsigned)blocks), block(kKdThreads);
    if (chunks > 4 * kKdThreads)
        hipLaunchKernelGGL((kdlite_head_kernel<4, 8, false, false>), grid, block, lds_bytes, st, ha);
    else
        hipLaunchKernelGGL((kdlite_head_kernel<4, 4, false, false>), grid, block, lds_bytes, st, ha);
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}

extern "C" int spice_scale_inplace(float *x, int64_t n, const float *scale, void *stream) {
    if (n < 0) return SPICE_ERR_ARG;
    if (n == 0) return SPICE_OK;
    if (!x || !scale || (reinterpret_cast<uintptr_t>(x) & 3u)) return SPICE_ERR_ARG;
    int64_t blocks = ceil_div(n, 256 * 16);
    if (blocks > 2048) blocks = 2048;
    hipLaunchKernelGGL(scale_inplace_kernel, dim3((unsigned)blocks), dim3(256), 0,
                       static_cast<hipStream_t>(stream), x, n, scale);
    SPICE_LAUNCH_CHECK();
    return SPICE_OK;
}
