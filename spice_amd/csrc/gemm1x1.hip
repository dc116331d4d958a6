// 1x1 stride-1 convolution of fp16 NCHW activations as the matrix product it is,
//   Y[n][Cout, HW] = W[Cout, Cin] @ X[n][Cin, HW]      (fp32 accumulation, fp16 out),
// through hipBLASLt with everything that does not change from call to call prepared once per
// problem size: the operation and layout descriptors and the algorithm hipBLASLt's heuristic picks.
// A call then costs the library's launch path only (about 5 us of host time on MI355X against
// 20 us through rocBLAS and 34 us through MIOpen's convolution entry point), which matters for a
// forward of ~70 convolutions whose device work is a few milliseconds.
//
// Column-major view handed to hipBLASLt:  Y^T[HW, Cout] = X^T[HW, Cin] @ W^T[Cin, Cout], batch n,
// batch strides Cin*HW (X), 0 (W, shared) and Cout*HW (Y).
#include "common.h"

#include <hipblaslt/hipblaslt.h>

#include <mutex>
#include <unordered_map>

namespace spice {

struct GemmKey {
    int64_t n, cin, hw, cout;
    bool operator==(const GemmKey &o) const { return n == o.n && cin == o.cin && hw == o.hw && cout == o.cout; }
};

struct GemmKeyHash {
    size_t operator()(const GemmKey &k) const {
        uint64_t h = 1469598103934665603ull;
        for (int64_t v : {k.n, k.cin, k.hw, k.cout}) h = (h ^ static_cast<uint64_t>(v)) * 1099511628211ull;
        return static_cast<size_t>(h);
    }
};

struct GemmPlan {
    hipblasLtMatmulDesc_t desc = nullptr;
    hipblasLtMatrixLayout_t lx = nullptr, lw = nullptr, ly = nullptr;
    hipblasLtMatmulAlgo_t algo;
    int status = SPICE_OK;   // SPICE_ERR_UNSUPPORTED: hipBLASLt has no scratch-free algorithm for this size (kept: it will not change)
};

static void destroy_plan(GemmPlan &p) {
    if (p.desc) hipblasLtMatmulDescDestroy(p.desc);
    if (p.lx) hipblasLtMatrixLayoutDestroy(p.lx);
    if (p.lw) hipblasLtMatrixLayoutDestroy(p.lw);
    if (p.ly) hipblasLtMatrixLayoutDestroy(p.ly);
    p.desc = nullptr;
    p.lx = p.lw = p.ly = nullptr;
}

struct GemmDevice {
    hipblasLtHandle_t handle = nullptr;
    std::unordered_map<GemmKey, GemmPlan, GemmKeyHash> plans;
};

static std::mutex g_gemm_mutex;
static GemmDevice g_gemm_devices[64];

static int make_layout(hipblasLtMatrixLayout_t *l, int64_t rows, int64_t cols, int64_t ld, int32_t batch, int64_t stride) {
    if (hipblasLtMatrixLayoutCreate(l, HIP_R_16F, rows, cols, ld) != HIPBLAS_STATUS_SUCCESS) return SPICE_ERR_BLAS;
    if (hipblasLtMatrixLayoutSetAttribute(*l, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &batch, sizeof(batch)) != HIPBLAS_STATUS_SUCCESS ||
        hipblasLtMatrixLayoutSetAttribute(*l, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &stride, sizeof(stride)) !=
            HIPBLAS_STATUS_SUCCESS)
        return SPICE_ERR_BLAS;
    return SPICE_OK;
}

static int build_plan(GemmDevice &d, const GemmKey &k, GemmPlan &p) {
    if (hipblasLtMatmulDescCreate(&p.desc, HIPBLAS_COMPUTE_32F, HIP_R_32F) != HIPBLAS_STATUS_SUCCESS) return SPICE_ERR_BLAS;
    int rc = make_layout(&p.lx, k.hw, k.cin, k.hw, static_cast<int32_t>(k.n), k.cin * k.hw);
    if (rc == SPICE_OK) rc = make_layout(&p.lw, k.cin, k.cout, k.cin, static_cast<int32_t>(k.n), 0);
    if (rc == SPICE_OK) rc = make_layout(&p.ly, k.hw, k.cout, k.hw, static_cast<int32_t>(k.n), k.cout * k.hw);
    if (rc != SPICE_OK) return rc;
    hipblasLtMatmulPreference_t pref = nullptr;
    if (hipblasLtMatmulPreferenceCreate(&pref) != HIPBLAS_STATUS_SUCCESS) return SPICE_ERR_BLAS;
    const uint64_t no_workspace = 0;   // only algorithms that need no scratch: the library brings none
    hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &no_workspace, sizeof(no_workspace));
    hipblasLtMatmulHeuristicResult_t found[1];
    int n_found = 0;
    const hipblasStatus_t st = hipblasLtMatmulAlgoGetHeuristic(d.handle, p.desc, p.lx, p.lw, p.ly, p.ly, pref, 1, found, &n_found);
    hipblasLtMatmulPreferenceDestroy(pref);
    if (st != HIPBLAS_STATUS_SUCCESS || n_found < 1 || found[0].workspaceSize != 0) return SPICE_ERR_UNSUPPORTED;
    p.algo = found[0].algo;
    return SPICE_OK;
}

}  // namespace spice

extern "C" int spice_conv1x1_f16(const void *x, const void *weight, void *y, int64_t n, int64_t cin, int64_t hw,
                                 int64_t cout, void *stream) {
    using namespace spice;
    if (n < 0 || cin <= 0 || hw < 0 || cout <= 0) return SPICE_ERR_ARG;
    if (n == 0 || hw == 0) return SPICE_OK;
    if (x == nullptr || weight == nullptr || y == nullptr) return SPICE_ERR_ARG;
    if (n > INT32_MAX || cin > INT32_MAX || hw > INT32_MAX || cout > INT32_MAX) return SPICE_ERR_UNSUPPORTED;
    int dev = 0;
    SPICE_HIP_TRY(hipGetDevice(&dev));
    if (dev < 0 || dev >= 64) return SPICE_ERR_UNSUPPORTED;
    GemmPlan plan;
    {
        std::lock_guard<std::mutex> lock(g_gemm_mutex);
        GemmDevice &d = g_gemm_devices[dev];
        if (d.handle == nullptr && hipblasLtCreate(&d.handle) != HIPBLAS_STATUS_SUCCESS) {
            d.handle = nullptr;
            return SPICE_ERR_BLAS;
        }
        const GemmKey key{n, cin, hw, cout};
        auto it = d.plans.find(key);
        if (it == d.plans.end()) {
            GemmPlan fresh;
            fresh.status = build_plan(d, key, fresh);
            if (fresh.status != SPICE_OK) destroy_plan(fresh);   // nothing of a failed plan is kept alive
            // a library error may be transient: only a finished plan or a definite "no algorithm" is remembered
            if (fresh.status != SPICE_OK && fresh.status != SPICE_ERR_UNSUPPORTED) return fresh.status;
            it = d.plans.emplace(key, fresh).first;
        }
        plan = it->second;
    }
    if (plan.status != SPICE_OK) return plan.status;
    const float alpha = 1.0f, beta = 0.0f;
    const hipblasStatus_t st = hipblasLtMatmul(g_gemm_devices[dev].handle, plan.desc, &alpha, x, plan.lx, weight, plan.lw, &beta,
                                               y, plan.ly, y, plan.ly, &plan.algo, nullptr, 0, static_cast<hipStream_t>(stream));
    return st == HIPBLAS_STATUS_SUCCESS ? SPICE_OK : SPICE_ERR_BLAS;
}
