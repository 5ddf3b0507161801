// Same question as ubench_coexec.hip for the other f32 MFMA shapes: does a VALU-only wave make progress beside a wave issuing
// v_mfma_f32_32x32x2_f32 (64 cycles) or v_mfma_f32_4x4x1_16B_f32 back to back on the same SIMD?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x16 __attribute__((ext_vector_type(16)));

template <int MODE, int SHAPE>   // MODE 0: MFMA only, 1: both, 2: VALU only.  SHAPE 0: 32x32x2, 1: 4x4x1
__global__ __launch_bounds__(512) void k(float *out, int iters)
{
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float b[8];
    for (int i = 0; i < 8; i++) b[i] = lane * 0.01f + i;
    f32x16 a0 = {0}, a1 = {0};
    f32x4 c0 = {0, 0, 0, 0}, c1 = {0, 0, 0, 0};
    float f[8];
    for (int i = 0; i < 8; i++) f[i] = lane * 0.001f + i;
    if (wave < 4) {
        if (MODE != 2)
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int ks = 0; ks < 32; ks++) {
                    if (SHAPE == 0) {
                        a0 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[ks & 7], b[(ks + 1) & 7], a0, 0, 0, 0);
                        a1 = __builtin_amdgcn_mfma_f32_32x32x2f32(b[(ks + 2) & 7], b[(ks + 3) & 7], a1, 0, 0, 0);
                    } else {
                        c0 = __builtin_amdgcn_mfma_f32_4x4x1f32(b[ks & 7], b[(ks + 1) & 7], c0, 0, 0, 0);
                        c1 = __builtin_amdgcn_mfma_f32_4x4x1f32(b[(ks + 2) & 7], b[(ks + 3) & 7], c1, 0, 0, 0);
                    }
                }
            }
    } else {
        if (MODE != 0)
            for (int it = 0; it < iters; it++) {
#pragma unroll
                for (int j = 0; j < 64; j++)
#pragma unroll
                    for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
            }
    }
    float s = a0[0] + a1[1] + c0[0] + c1[1];
    for (int i = 0; i < 8; i++) s += f[i];
    out[blockIdx.x * 512 + threadIdx.x] = s;
}

template <int MODE, int SHAPE>
float run(float *d, int iters)
{
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE, SHAPE><<<256, 512>>>(d, 2);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE, SHAPE><<<256, 512>>>(d, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms;
}

int main()
{
    float *d; hipMalloc(&d, 256 * 512 * 4);
    const int iters = 400;
    printf("v_mfma_f32_32x32x2_f32 : MFMA alone %.3f ms, VALU alone %.3f ms, together %.3f ms\n", run<0, 0>(d, iters), run<2, 0>(d, iters), run<1, 0>(d, iters));
    printf("v_mfma_f32_4x4x1_f32   : MFMA alone %.3f ms, VALU alone %.3f ms, together %.3f ms\n", run<0, 1>(d, iters), run<2, 1>(d, iters), run<1, 1>(d, iters));
    return 0;
}
