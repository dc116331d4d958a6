// Probe: host time of hipblasLtMatmul with descriptors and the algorithm prepared once, for the
// 1x1-convolution product  Y[n][Cout,HW] = W[Cout,Cin] @ X[n][Cin,HW]  (fp16 in/out, fp32 compute).
#include <hip/hip_runtime.h>
#include <hipblaslt/hipblaslt.h>
#include <chrono>
#include <cstdio>
#include <cstdlib>

#define CK(x) do { auto e_ = (x); if (e_ != 0) { printf("error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); return 1; } } while (0)

int main() {
    const int shapes[][5] = {{8, 64, 128, 128, 256}, {8, 256, 128, 128, 64}, {8, 1024, 32, 32, 256}, {8, 2048, 32, 32, 512},
                             {8, 304, 128, 128, 256}, {8, 230, 32, 32, 1000}};
    hipblasLtHandle_t h;
    CK(hipblasLtCreate(&h));
    void *ws;
    const size_t ws_bytes = 64u << 20;
    CK(hipMalloc(&ws, ws_bytes));
    hipStream_t st;
    CK(hipStreamCreate(&st));
    for (auto &s : shapes) {
        const int64_t n = s[0], cin = s[1], hw = (int64_t)s[2] * s[3], cout = s[4];
        void *x, *w, *y;
        CK(hipMalloc(&x, n * cin * hw * 2));
        CK(hipMalloc(&w, cout * cin * 2));
        CK(hipMalloc(&y, n * cout * hw * 2));
        CK(hipMemset(x, 0, n * cin * hw * 2));
        CK(hipMemset(w, 0, cout * cin * 2));
        hipblasLtMatmulDesc_t desc;
        CK(hipblasLtMatmulDescCreate(&desc, HIPBLAS_COMPUTE_32F, HIP_R_32F));
        hipblasLtMatrixLayout_t la, lb, lc;
        CK(hipblasLtMatrixLayoutCreate(&la, HIP_R_16F, hw, cin, hw));
        CK(hipblasLtMatrixLayoutCreate(&lb, HIP_R_16F, cin, cout, cin));
        CK(hipblasLtMatrixLayoutCreate(&lc, HIP_R_16F, hw, cout, hw));
        const int32_t batch = (int32_t)n;
        const int64_t sa = cin * hw, sb = 0, sc = cout * hw;
        for (auto pr : {std::make_pair(la, sa), std::make_pair(lb, sb), std::make_pair(lc, sc)}) {
            CK(hipblasLtMatrixLayoutSetAttribute(pr.first, HIPBLASLT_MATRIX_LAYOUT_BATCH_COUNT, &batch, sizeof(batch)));
            CK(hipblasLtMatrixLayoutSetAttribute(pr.first, HIPBLASLT_MATRIX_LAYOUT_STRIDED_BATCH_OFFSET, &pr.second, sizeof(int64_t)));
        }
        hipblasLtMatmulPreference_t pref;
        CK(hipblasLtMatmulPreferenceCreate(&pref));
        CK(hipblasLtMatmulPreferenceSetAttribute(pref, HIPBLASLT_MATMUL_PREF_MAX_WORKSPACE_BYTES, &ws_bytes, sizeof(ws_bytes)));
        hipblasLtMatmulHeuristicResult_t res[4];
        int found = 0;
        auto t0 = std::chrono::steady_clock::now();
        CK(hipblasLtMatmulAlgoGetHeuristic(h, desc, la, lb, lc, lc, pref, 4, res, &found));
        auto t1 = std::chrono::steady_clock::now();
        if (found == 0) { printf("no algorithm\n"); continue; }
        const float alpha = 1.f, beta = 0.f;
        for (int i = 0; i < 10; ++i)
            CK(hipblasLtMatmul(h, desc, &alpha, x, la, w, lb, &beta, y, lc, y, lc, &res[0].algo, ws, ws_bytes, st));
        CK(hipStreamSynchronize(st));
        const int reps = 300;
        auto a0 = std::chrono::steady_clock::now();
        for (int i = 0; i < reps; ++i)
            CK(hipblasLtMatmul(h, desc, &alpha, x, la, w, lb, &beta, y, lc, y, lc, &res[0].algo, ws, ws_bytes, st));
        auto a1 = std::chrono::steady_clock::now();
        CK(hipStreamSynchronize(st));
        auto a2 = std::chrono::steady_clock::now();
        auto us = [](auto d) { return std::chrono::duration<double, std::micro>(d).count(); };
        printf("[%ld,%ld,%d,%d]->%ld: heuristic %.0f us (%d found, workspace %zu), matmul host %.1f us, wall %.1f us per call\n",
               (long)n, (long)cin, s[2], s[3], (long)cout, us(t1 - t0), found, res[0].workspaceSize, us(a1 - a0) / reps, us(a2 - a0) / reps);
        hipFree(x); hipFree(w); hipFree(y);
    }
    return 0;
}
