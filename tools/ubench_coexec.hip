// Does a VALU-only wave run beside an MFMA-only wave on the same SIMD at full rate on gfx950 (fp32 MFMA vs v_fma_f32)?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>   // 0: all 8 waves MFMA-only(4)+idle(4) ; 1: 4 MFMA + 4 VALU ; 2: 4 idle + 4 VALU
__global__ __launch_bounds__(512) void k(float *out, int iters)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float b[8];
    for (int i = 0; i < 8; i++) b[i] = lane * 0.01f + i;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    float f[8];
    for (int i = 0; i < 8; i++) f[i] = lane * 0.001f + i;
    if (wave < 4) {
        if (MODE != 2)
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int ks = 0; ks < 64; ks++) {
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[ks & 7], b[(ks + 1) & 7], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(b[(ks + 2) & 7], b[(ks + 3) & 7], acc1, 0, 0, 0);
                }
            }
    } else {
        if (MODE != 0)
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int j = 0; j < 128; j++)
#pragma unroll
                    for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            }
    }
    float s = acc0[0] + acc1[1];
    for (int i = 0; i < 8; i++) s += f[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE>
float run(float *d, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256, 512>>>(d, 2);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<256, 512>>>(d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    float *d; hipMalloc(&d, 256 * 512 * 4);
    const int iters = 400;
    const float m0 = run<0>(d, iters), m2 = run<2>(d, iters), m1 = run<1>(d, iters);
    const double mfma_flop = 1024.0 * iters * 128 * 2048, valu_flop = 1024.0 * iters * 1024 * 64 * 2;
    printf("MFMA alone   : %.3f ms  %.1f TFLOP/s\n", m0, mfma_flop / m0 / 1e9);
    printf("VALU alone   : %.3f ms  %.1f TFLOP/s (1 wave/SIMD)\n", m2, valu_flop / m2 / 1e9);
    printf("both together: %.3f ms  -> MFMA %.1f + VALU %.1f TFLOP/s if both finished at the end\n", m1, mfma_flop / m1 / 1e9, valu_flop / m1 / 1e9);
    return 0;
}
