// What a filler instruction costs between back-to-back v_mfma_f32_16x16x32_f16 of ONE wave on a SIMD (the consumer waves of
// k_conv_features_h2): cycles per MFMA by s_memtime with N fillers of one kind per MFMA gap.
#include <hip/hip_runtime.h>
#include <cstdio>
typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef float f4 __attribute__((ext_vector_type(4)));

template <int KIND, int NF>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int iters)
{
    __shared__ uint4 lds[1024];
    h8 a, b;
    for (int i = 0; i < 8; i++) { a[i] = (_Float16)(threadIdx.x * 0.001f + i); b[i] = (_Float16)(0.5f - i * 0.01f); }
    lds[threadIdx.x] = make_uint4(1, 2, 3, 4);
    __syncthreads();
    f4 acc[4];
    for (int i = 0; i < 4; i++) acc[i] = (f4){0, 0, 0, 0};
    float f0 = threadIdx.x, f1 = 1.5f, f2 = 2.5f, f3 = 0.25f;
    uint4 ld = make_uint4(0, 0, 0, 0);
    const unsigned lp = (unsigned)(size_t)(lds + (threadIdx.x & 63));
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++)
#pragma unroll
            for (int i = 0; i < 4; i++) {
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[i]) : "v"(a), "v"(b));
#pragma unroll
                for (int n = 0; n < NF; n++) {
                    if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f0) : "v"(f1));
                    if (KIND == 1) asm volatile("s_nop 0");
                    if (KIND == 2) asm volatile("ds_read_b128 %0, %1" : "=v"(ld) : "v"(lp));
                    if (KIND == 3) asm volatile("s_waitcnt lgkmcnt(15)");
                    if (KIND == 4) asm volatile("v_fma_f32 %0, %1, %2, %0" : "+v"(f2) : "v"(f1), "v"(f3));
                    if (KIND == 5) asm volatile("v_max3_f32 %0, %0, %1, %2" : "+v"(f0) : "v"(f1), "v"(f3));
                    if (KIND == 6) asm volatile("s_add_u32 s20, s20, 1" ::: "s20");
                }
            }
        if (KIND == 2) asm volatile("s_waitcnt lgkmcnt(0)");
    }
    asm volatile("s_nop 15\n\ts_nop 15");
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float s = f0 + f2 + (float)ld.x;
    for (int i = 0; i < 4; i++) s += acc[i][0] + acc[i][3];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND, int NF>
void run(const char *name)
{
    float *out; unsigned long long *cyc, h;
    (void)hipMalloc(&out, 4 << 20); (void)hipMalloc(&cyc, 8);
    const int iters = 1000;
    for (int rep = 0; rep < 2; rep++) { hipLaunchKernelGGL((k<KIND, NF>), dim3(256), dim3(256), 0, 0, out, cyc, iters); (void)hipDeviceSynchronize(); }
    (void)hipMemcpy(&h, cyc, 8, hipMemcpyDeviceToHost);
    printf("%-16s x%d per MFMA: %6.2f ticks per MFMA\n", name, NF, h / (iters * 16.0));
    (void)hipFree(out); (void)hipFree(cyc);
}

int main()
{
    run<0, 0>("none");
    run<0, 1>("v_add_f32"); run<0, 2>("v_add_f32"); run<0, 3>("v_add_f32"); run<0, 4>("v_add_f32");
    run<4, 2>("v_fma_f32"); run<5, 2>("v_max3_f32");
    run<1, 1>("s_nop 0"); run<1, 3>("s_nop 0");
    run<3, 1>("s_waitcnt"); run<3, 3>("s_waitcnt");
    run<6, 2>("s_add_u32");
    run<2, 1>("ds_read_b128"); run<2, 2>("ds_read_b128");
    return 0;
}
