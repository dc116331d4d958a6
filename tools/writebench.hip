// Ceiling probe for streaming WRITES and read+write mixes on gfx950: fill, copy (1 read : 1 write)
// and a 2 reads : 1 write mix (the shape of the KD-Lite gradient pass), with plain and
// non-temporal stores.  Build: hipcc --offload-arch=gfx950 -O3 tools/writebench.hip -o build/writebench
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <vector>

typedef float f4 __attribute__((ext_vector_type(4)));

template <bool NT>
__global__ __launch_bounds__(256) void fill_k(f4 *__restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    const f4 v = {1.f, 2.f, 3.f, 4.f};
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n4; i += stride) {
        if (NT) __builtin_nontemporal_store(v, dst + i);
        else dst[i] = v;
    }
}

template <bool NT>
__global__ __launch_bounds__(256) void copy_k(const f4 *__restrict__ src, f4 *__restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        f4 v[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) v[u] = src[i + u * stride];
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            if (NT) __builtin_nontemporal_store(v[u], dst + i + u * stride);
            else dst[i + u * stride] = v[u];
        }
    }
    for (; i < n4; i += stride) dst[i] = src[i];
}

template <bool NT>
__global__ __launch_bounds__(256) void mix21_k(const f4 *__restrict__ a, const f4 *__restrict__ b,
                                                f4 *__restrict__ dst, size_t n4) {
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + 3 * stride < n4; i += 4 * stride) {
        f4 x[4], y[4];
#pragma unroll
        for (int u = 0; u < 4; ++u) { x[u] = a[i + u * stride]; y[u] = b[i + u * stride]; }
#pragma unroll
        for (int u = 0; u < 4; ++u) {
            const f4 r = x[u] - y[u];
            if (NT) __builtin_nontemporal_store(r, dst + i + u * stride);
            else dst[i + u * stride] = r;
        }
    }
    for (; i < n4; i += stride) dst[i] = a[i] - b[i];
}

template <typename F>
static float time_us(F launch, int iters) {
    hipEvent_t a, b;
    hipEventCreate(&a);
    hipEventCreate(&b);
    std::vector<float> t;
    for (int i = 0; i < iters + 3; ++i) {
        hipEventRecord(a);
        launch(i);
        hipEventRecord(b);
        hipEventSynchronize(b);
        float ms;
        hipEventElapsedTime(&ms, a, b);
        if (i >= 3) t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    const size_t MB = 1 << 20;
    const int nc = 4;  // rotating buffer sets, > 256 MiB in total for every case
    for (size_t bytes : {53 * MB, 176 * MB}) {
        std::vector<f4 *> A(nc), B(nc), D(nc);
        for (int i = 0; i < nc; ++i) {
            if (hipMalloc(&A[i], bytes) != hipSuccess || hipMalloc(&B[i], bytes) != hipSuccess ||
                hipMalloc(&D[i], bytes) != hipSuccess) {
                printf("allocation failed\n");
                return 1;
            }
            hipMemset(A[i], 1, bytes);
            hipMemset(B[i], 2, bytes);
            hipMemset(D[i], 0, bytes);
        }
        const size_t n4 = bytes / 16;
        for (int blocks : {2048, 4096, 8192}) {
            const dim3 g(blocks), b(256);
            float f0 = time_us([&](int i) { hipLaunchKernelGGL((fill_k<false>), g, b, 0, 0, D[i % nc], n4); }, 15);
            float f1 = time_us([&](int i) { hipLaunchKernelGGL((fill_k<true>), g, b, 0, 0, D[i % nc], n4); }, 15);
            float c0 = time_us([&](int i) { hipLaunchKernelGGL((copy_k<false>), g, b, 0, 0, A[i % nc], D[i % nc], n4); }, 15);
            float c1 = time_us([&](int i) { hipLaunchKernelGGL((copy_k<true>), g, b, 0, 0, A[i % nc], D[i % nc], n4); }, 15);
            float m0 = time_us([&](int i) { hipLaunchKernelGGL((mix21_k<false>), g, b, 0, 0, A[i % nc], B[i % nc], D[i % nc], n4); }, 15);
            float m1 = time_us([&](int i) { hipLaunchKernelGGL((mix21_k<true>), g, b, 0, 0, A[i % nc], B[i % nc], D[i % nc], n4); }, 15);
            printf("%4zu MB blocks %5d: fill %6.1f us (%4.2f TB/s) nt %6.1f (%4.2f) | copy 1r:1w %6.1f (%4.2f) nt %6.1f (%4.2f) | 2r:1w %6.1f (%4.2f) nt %6.1f (%4.2f)\n",
                   bytes / MB, blocks, f0, bytes / f0 / 1e6, f1, bytes / f1 / 1e6, c0, 2 * bytes / c0 / 1e6, c1,
                   2 * bytes / c1 / 1e6, m0, 3 * bytes / m0 / 1e6, m1, 3 * bytes / m1 / 1e6);
        }
        for (int i = 0; i < nc; ++i) {
            hipFree(A[i]);
            hipFree(B[i]);
            hipFree(D[i]);
        }
    }
    return 0;
}
