// Fills the LDS of every compute unit with NaN halves / NaN floats and leaves: kernels that read LDS they have not
// written then compute on NaNs instead of on what the kernel before happened to leave there.
#include <hip/hip_runtime.h>
__global__ __launch_bounds__(256) void poison_kernel(unsigned pattern, int words, unsigned *sink) {
    extern __shared__ unsigned lds[];
    for (int i = threadIdx.x; i < words; i += 256) lds[i] = pattern;
    __syncthreads();
    if (sink != nullptr && lds[(threadIdx.x * 7) % words] == 1u) sink[0] = 1;   // keeps the stores
}
extern "C" int poison_lds(unsigned pattern, void *stream) {
    const int bytes = 80 * 1024;
    (void)hipFuncSetAttribute(reinterpret_cast<const void *>(poison_kernel), hipFuncAttributeMaxDynamicSharedMemorySize, bytes);
    hipLaunchKernelGGL(poison_kernel, dim3(1024), dim3(256), bytes, static_cast<hipStream_t>(stream), pattern, bytes / 4, nullptr);
    return (int)hipGetLastError();
}
