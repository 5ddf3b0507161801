// Micro-benchmark: v_mfma_f32_16x16x4_f32 issue rate on gfx950, bare and fed by one ds_read_b32 per two MFMAs
// (the shape of k_conv_features' inner loop).  hipcc -O3 --offload-arch=gfx950 tools/ubench_mfma.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int MODE>
__global__ __launch_bounds__(256, 2) void k(float *out, int iters, long long *clk)
{
    __shared__ float lds[16384];
    for (int i = threadIdx.x; i < 16384; i += blockDim.x) lds[i] = i * 0.001f;
    __syncthreads();
    const int lane = threadIdx.x & 63;
    float b[16];
    for (int i = 0; i < 16; i++) b[i] = lane * 0.01f + i;
    f32x4 acc0 = {0, 0, 0, 0}, acc1 = {0, 0, 0, 0};
    const float *ap = lds + (lane >> 4) * 2056 + (lane & 15) * 3;
    long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int ks = 0; ks < 72; ks++) {
            float a;
            if (MODE == 0) a = b[(ks + 1) & 15];
            else a = ap[(ks & 7) * 257 + (ks >> 3) * 5];
            acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[ks & 15], acc0, 0, 0, 0);
            acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, b[(ks + 3) & 15], acc1, 0, 0, 0);
        }
    }
    long long t1 = __builtin_amdgcn_s_memtime();
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc0[0] + acc0[1] + acc1[2] + acc1[3];
    if (threadIdx.x == 0 && blockIdx.x == 0) clk[0] = t1 - t0;
}

template <int MODE>
void run(const char *name, int blocks_per_cu, float *d, long long *clk)
{
    const int iters = 200;
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<MODE><<<256 * blocks_per_cu, 256>>>(d, 2, clk);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<MODE><<<256 * blocks_per_cu, 256>>>(d, iters, clk);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    long long c; hipMemcpy(&c, clk, 8, hipMemcpyDeviceToHost);
    double mfma_per_simd = (double)blocks_per_cu * iters * 144;
    printf("%-28s waves/SIMD=%d  %.3f ms  %.1f ns/MFMA/SIMD  in-kernel %.1f memtime-ticks per MFMA (one wave)  => %.1f TFLOP/s\n", name, blocks_per_cu, ms,
           ms * 1e6 / mfma_per_simd, (double)c / (iters * 144), 1024.0 * mfma_per_simd * 2048 / (ms * 1e-3) / 1e12);
}

int main()
{
    float *d; long long *clk; hipMalloc(&d, 4 * 256 * 4 * 256); hipMalloc(&clk, 8);
    for (int w : {1, 2}) { run<0>("mfma16x16x4 regs only", w, d, clk); run<1>("mfma16x16x4 + ds_read_b32", w, d, clk); }
    return 0;
}
