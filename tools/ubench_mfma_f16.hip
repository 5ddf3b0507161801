// Issue rate of v_mfma_f32_16x16x32_f16 (and _bf16) on one SIMD: cycles per MFMA by s_memtime, for 1/2/4 accumulators, with
// independent and with in-place dependent chains.  hipcc --offload-arch=gfx950 -O3 tools/ubench_mfma_f16.hip -o /tmp/ub && /tmp/ub
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef __bf16 b8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int NACC, bool BF>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters)
{
    h8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
    f4 acc[NACC];
    for (int i = 0; i < NACC; i++) acc[i] = (f4){0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 16 / NACC; r++)
#pragma unroll
            for (int i = 0; i < NACC; i++) {
                if (BF) asm volatile("v_mfma_f32_16x16x32_bf16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
                else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
            }
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = 0;
    for (int i = 0; i < NACC; i++) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int NACC, bool BF>
void run(const char *name, int waves_per_simd)
{
    float *out; unsigned long long *cyc, h;
    hipMalloc(&out, 4 << 20); hipMalloc(&cyc, 8);
    const int iters = 2000;
    hipLaunchKernelGGL((k<NACC, BF>), dim3(256 * waves_per_simd), dim3(256), 0, 0, out, cyc, iters);
    hipDeviceSynchronize();
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    hipLaunchKernelGGL((k<NACC, BF>), dim3(256 * waves_per_simd), dim3(256), 0, 0, out, cyc, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    const double n = (double)iters * 16;
    printf("%-34s waves/SIMD %d: %6.2f memtime-ticks per MFMA (one wave), kernel %.3f ms -> %.0f TFLOP/s\n", name, waves_per_simd, h / n, ms,
           n * 16384.0 * 1024 * waves_per_simd / (ms * 1e-3) / 1e12);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<1, false>("f16 16x16x32, 1 accumulator", 1);
    run<2, false>("f16 16x16x32, 2 accumulators", 1);
    run<4, false>("f16 16x16x32, 4 accumulators", 1);
    run<16, false>("f16 16x16x32, 16 accumulators", 1);
    run<4, false>("f16 16x16x32, 4 accumulators", 2);
    run<4, true>("bf16 16x16x32, 4 accumulators", 1);
    run<1, true>("bf16 16x16x32, 1 accumulator", 1);
    return 0;
}
