// Floor probe for the 33.6 MB confusion update: how long do (a) an empty launch, (b) a trivial
// streaming read of 32 MiB from rotating buffers, (c) B workgroups adding 441 64-bit counters each
// into ONE table with agent-scope atomics take on gfx950?  Run under rocprofv3 --kernel-trace.
// Build: hipcc -w --offload-arch=gfx950 -O3 tools/smallbench.hip -o tools/smallbench.bin
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

__global__ void empty_kernel() {}

template <int THREADS, int UNROLL>
__global__ __launch_bounds__(THREADS) void read_sum(const float4* __restrict__ p, size_t n4, float* out) {
    typedef float f4 __attribute__((ext_vector_type(4)));
    float acc = 0.f;
    const size_t stride = (size_t)gridDim.x * THREADS;
    size_t i = (size_t)blockIdx.x * THREADS + threadIdx.x;
    for (; i + (UNROLL - 1) * stride < n4; i += UNROLL * stride) {
        f4 v[UNROLL];
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) v[u] = __builtin_nontemporal_load(reinterpret_cast<const f4*>(p + i + u * stride));
#pragma unroll
        for (int u = 0; u < UNROLL; ++u) acc += (v[u].x + v[u].y) + (v[u].z + v[u].w);
    }
    for (; i < n4; i += stride) { float4 v = p[i]; acc += v.x + v.y + v.z + v.w; }
    for (int o = 32; o > 0; o >>= 1) acc += __shfl_xor(acc, o, 64);
    if ((threadIdx.x & 63) == 0 && acc == 12345.f) out[blockIdx.x] = acc;
}

__global__ __launch_bounds__(512) void atomics_direct(unsigned long long* table, int bins) {
    for (int b = threadIdx.x; b < bins; b += 512) atomicAdd(table + b, (unsigned long long)(blockIdx.x + 1));
}

// replica = blockIdx % R
__global__ __launch_bounds__(512) void atomics_replica(unsigned long long* table, int bins, int R) {
    unsigned long long* t = table + (size_t)(blockIdx.x % R) * 512;
    for (int b = threadIdx.x; b < bins; b += 512) atomicAdd(t + b, (unsigned long long)(blockIdx.x + 1));
}

int main() {
    const size_t bytes = 32u << 20;
    const int nc = 12;
    std::vector<float4*> bufs(nc);
    for (auto& b : bufs) { hipMalloc(&b, bytes); hipMemset(b, 0, bytes); }
    float* out; hipMalloc(&out, 1 << 20); hipMemset(out, 0, 1 << 20);
    unsigned long long* table; hipMalloc(&table, 64 * 512 * 8); hipMemset(table, 0, 64 * 512 * 8);
    const size_t n4 = bytes / 16;
    hipDeviceSynchronize();
    for (int it = 0; it < 24; ++it) {
        hipLaunchKernelGGL(empty_kernel, dim3(256), dim3(512), 0, 0);
        float4* p = bufs[it % nc];
        hipLaunchKernelGGL((read_sum<512, 8>), dim3(256), dim3(512), 0, 0, p, n4, out);
        hipLaunchKernelGGL((read_sum<512, 4>), dim3(512), dim3(512), 0, 0, bufs[(it + 3) % nc], n4, out);
        hipLaunchKernelGGL((read_sum<256, 8>), dim3(1024), dim3(256), 0, 0, bufs[(it + 6) % nc], n4, out);
        hipLaunchKernelGGL((read_sum<256, 4>), dim3(2048), dim3(256), 0, 0, bufs[(it + 9) % nc], n4, out);
        hipLaunchKernelGGL((read_sum<1024, 2>), dim3(1024), dim3(1024), 0, 0, bufs[(it + 1) % nc], n4, out);
        hipLaunchKernelGGL(atomics_direct, dim3(256), dim3(512), 0, 0, table, 441);
        hipLaunchKernelGGL(atomics_direct, dim3(64), dim3(512), 0, 0, table, 441);
        hipLaunchKernelGGL(atomics_replica, dim3(256), dim3(512), 0, 0, table, 441, 8);
        hipLaunchKernelGGL(atomics_replica, dim3(256), dim3(512), 0, 0, table, 441, 32);
        hipDeviceSynchronize();
    }
    printf("done\n");
    return 0;
}
