// Ceiling probe: how fast can gfx950 stream-read N bytes with a trivial reduction, as a function
// of size, blocks and unroll?  Build: hipcc -w --offload-arch=gfx950 -O3 tools/streambench.hip -o tools/streambench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <algorithm>

template <int UNROLL, bool NT>
__global__ __launch_bounds__(256) void read_sum(const float4* __restrict__ p, size_t n4, float* out) {
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * 256;
    size_t i = (size_t)blockIdx.x * 256 + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) {
            typedef float f4 __attribute__((ext_vector_type(4)));
            if (NT) { f4 t = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p + i + u * stride)); v[u] = make_float4(t.x, t.y, t.z, t.w); }
            else v[u] = p[i + u * stride];
        }
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += (v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w);
    }
    for (; i < n4; i += stride) { float4 v = p[i]; acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + blockIdx.x, acc);
}

// contiguous-chunk variant: block b reads its own contiguous slice (like a per-item kernel)
template <int UNROLL>
__global__ __launch_bounds__(256) void read_sum_chunk(const float4* __restrict__ p, size_t n4, float* out) {
    const size_t per = (n4 + gridDim.x - 1) / gridDim.x;
    const size_t b0 = (size_t)blockIdx.x * per;
    const size_t b1 = b0 + per < n4 ? b0 + per : n4;
    float acc = 0.f;
    size_t i = b0 + threadIdx.x;
    for (; i + (UNROLL - 1) * 256 < b1; i += UNROLL * 256) {
        float4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = p[i + u * 256];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += (v[u].x * v[u].x + v[u].y * v[u].y) + (v[u].z * v[u].z + v[u].w * v[u].w);
    }
    for (; i < b1; i += 256) { float4 v = p[i]; acc += v.x * v.x + v.y * v.y + v.z * v.z + v.w * v.w; }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0) atomicAdd(out + blockIdx.x, acc);
}

template <typename F>
float time_us(F launch, int iters) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    std::vector<float> t;
    for (int i = 0; i < iters + 3; ++i) {
        hipEventRecord(a); launch(i); hipEventRecord(b); hipEventSynchronize(b);
        float ms; hipEventElapsedTime(&ms, a, b); if (i >= 3) t.push_back(ms * 1e3f);
    }
    std::sort(t.begin(), t.end());
    return t[t.size() / 2];
}

int main() {
    const size_t MB = 1 << 20;
    const int copies = 6;
    for (size_t bytes : {107 * MB, 214 * MB, 369 * MB, 1024 * MB}) {
        const int nc = bytes > 512 * MB ? 2 : copies;
        std::vector<float4*> bufs(nc);
        for (auto& b : bufs) { hipMalloc(&b, bytes); hipMemset(b, 1, bytes); }
        float* out; hipMalloc(&out, 65536 * 4); hipMemset(out, 0, 65536 * 4);
        const size_t n4 = bytes / 16;
        for (int blocks : {1024, 2048, 4096, 8192}) {
            float t4 = time_us([&](int i) { hipLaunchKernelGGL((read_sum<4, false>), dim3(blocks), dim3(256), 0, 0, bufs[i % nc], n4, out); }, 15);
            float t8 = time_us([&](int i) { hipLaunchKernelGGL((read_sum<8, false>), dim3(blocks), dim3(256), 0, 0, bufs[i % nc], n4, out); }, 15);
            float t8n = time_us([&](int i) { hipLaunchKernelGGL((read_sum<8, true>), dim3(blocks), dim3(256), 0, 0, bufs[i % nc], n4, out); }, 15);
            float tc = time_us([&](int i) { hipLaunchKernelGGL((read_sum_chunk<4>), dim3(blocks), dim3(256), 0, 0, bufs[i % nc], n4, out); }, 15);
            printf("%5zu MB blocks %5d: strided u4 %7.1f us (%5.2f TB/s) | u8 %7.1f (%5.2f) | u8 nt %7.1f (%5.2f) | chunk u4 %7.1f (%5.2f)\n",
                   bytes / MB, blocks, t4, bytes / t4 / 1e6, t8, bytes / t8 / 1e6, t8n, bytes / t8n / 1e6, tc, bytes / tc / 1e6);
        }
        for (auto b : bufs) hipFree(b);
        hipFree(out);
    }
    // finer sweep at the KD-Lite size (369 MB): workgroups x unroll x load policy
    {
        const size_t bytes = 352 * MB;
        const int nc = 3;
        std::vector<float4*> bufs(nc);
        for (auto& b : bufs) { if (hipMalloc(&b, bytes) != hipSuccess) { printf("alloc failed\n"); return 1; } hipMemset(b, 1, bytes); }
        float* out; hipMalloc(&out, 65536 * 4); hipMemset(out, 0, 65536 * 4);
        const size_t n4 = bytes / 16;
        for (int blocks : {512, 1024, 2048, 4096, 8192, 16384}) {
            float a = time_us([&](int i) { hipLaunchKernelGGL((read_sum<8, false>), dim3(blocks), dim3(256), 0, 0, bufs[i % nc], n4, out); }, 15);
            float b = time_us([&](int i) { hipLaunchKernelGGL((read_sum<8, true>), dim3(blocks), dim3(256), 0, 0, bufs[i % nc], n4, out); }, 15);
            float c = time_us([&](int i) { hipLaunchKernelGGL((read_sum<16, false>), dim3(blocks), dim3(256), 0, 0, bufs[i % nc], n4, out); }, 15);
            float d = time_us([&](int i) { hipLaunchKernelGGL((read_sum<16, true>), dim3(blocks), dim3(256), 0, 0, bufs[i % nc], n4, out); }, 15);
            printf("352 MB blocks %5d: u8 %6.1f us (%4.2f TB/s) | u8 nt %6.1f (%4.2f) | u16 %6.1f (%4.2f) | u16 nt %6.1f (%4.2f)\n",
                   blocks, a, bytes / a / 1e6, b, bytes / b / 1e6, c, bytes / c / 1e6, d, bytes / d / 1e6);
        }
        for (auto b : bufs) hipFree(b);
        hipFree(out);
    }
    return 0;
}
