// D2H bandwidth on this box: hipMemcpyAsync (SDMA engine) against a copy kernel that stores straight into mapped pinned host memory.
// Build: hipcc --offload-arch=gfx950 -O3 -o /tmp/ubench_d2h tools/ubench_d2h.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#define CK(x) do { hipError_t e = (x); if (e != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e)); return 1; } } while (0)

__global__ void k_copy(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n)
{
    for (size_t i = blockIdx.x * (size_t)blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x)
        dst[i] = src[i];
}

int main()
{
    const size_t bytes = 64ull << 20;
    void *dev, *host;
    CK(hipMalloc(&dev, bytes));
    CK(hipMemset(dev, 0x5a, bytes));
    CK(hipHostMalloc(&host, bytes, hipHostMallocDefault));
    hipStream_t s;
    CK(hipStreamCreate(&s));
    hipEvent_t a, b;
    CK(hipEventCreate(&a));
    CK(hipEventCreate(&b));
    float ms;
    for (int rep = 0; rep < 2; rep++) {
        CK(hipEventRecord(a, s));
        for (int i = 0; i < 10; i++) CK(hipMemcpyAsync(host, dev, bytes, hipMemcpyDeviceToHost, s));
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
        printf("hipMemcpyAsync D2H   : %.1f GB/s\n", 10.0 * bytes / ms * 1e-6);
    }
    const int grids[] = {64, 256, 1024, 4096};
    for (int g : grids) {
        memset(host, 0, bytes);
        CK(hipEventRecord(a, s));
        for (int i = 0; i < 10; i++) hipLaunchKernelGGL(k_copy, dim3(g), dim3(256), 0, s, (const uint4 *)dev, (uint4 *)host, bytes / 16);
        CK(hipEventRecord(b, s));
        CK(hipEventSynchronize(b));
        CK(hipEventElapsedTime(&ms, a, b));
        printf("copy kernel grid %4d : %.1f GB/s  (check %02x)\n", g, 10.0 * bytes / ms * 1e-6, ((unsigned char *)host)[bytes - 1]);
    }
    // both directions of traffic at once do not matter here; two D2H streams at once:
    hipStream_t s2;
    CK(hipStreamCreate(&s2));
    CK(hipEventRecord(a, s));
    for (int i = 0; i < 10; i++) {
        CK(hipMemcpyAsync(host, dev, bytes / 2, hipMemcpyDeviceToHost, s));
        CK(hipMemcpyAsync((char *)host + bytes / 2, (char *)dev + bytes / 2, bytes / 2, hipMemcpyDeviceToHost, s2));
    }
    CK(hipStreamSynchronize(s2));
    CK(hipEventRecord(b, s));
    CK(hipEventSynchronize(b));
    CK(hipEventElapsedTime(&ms, a, b));
    printf("2 x hipMemcpyAsync   : %.1f GB/s\n", 10.0 * bytes / ms * 1e-6);
    return 0;
}
