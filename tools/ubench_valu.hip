// Micro-benchmark: sustained VALU issue rate on gfx950 for the instruction kinds K1 is made of.
// Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_valu.hip -o tools/ubench_valu ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

#define N_ITERS 512
#define UNROLL 32

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, int n)
{
    float f[8]; unsigned u[8];
    for (int i = 0; i < 8; i++) { f[i] = threadIdx.x * 0.001f + i; u[i] = threadIdx.x * 7 + i; }
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int j = 0; j < UNROLL / 8; j++)
#pragma unroll
            for (int i = 0; i < 8; i++) {
                if (KIND == 0) asm volatile("v_add_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
                if (KIND == 1) asm volatile("v_mul_f32 %0, %0, %1" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
                if (KIND == 2) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(f[i]) : "v"(f[(i + 1) & 7]));
                if (KIND == 3) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (KIND == 4) asm volatile("v_mad_u32_u24 %0, %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (KIND == 5) asm volatile("v_bfe_u32 %0, %0, 8, 8" : "+v"(u[i]));
                if (KIND == 6) asm volatile("v_lshl_or_b32 %0, %0, 8, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (KIND == 7) asm volatile("v_cvt_f32_ubyte1 %0, %1" : "=v"(f[i]) : "v"(u[i]));
                if (KIND == 8) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(*(double *)&f[i & 6]) : "v"(*(double *)&f[(i + 2) & 6]));
                if (KIND == 9) asm volatile("v_pk_mul_f32 %0, %0, %1" : "+v"(*(double *)&f[i & 6]) : "v"(*(double *)&f[(i + 2) & 6]));
                if (KIND == 10) asm volatile("v_perm_b32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
                if (KIND == 11) asm volatile("v_dot4_u32_u8 %0, %0, %1, %0" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (KIND == 12) asm volatile("v_add3_u32 %0, %0, %1, %2" : "+v"(u[i]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 2) & 7]));
                if (KIND == 13) asm volatile("v_rndne_f32 %0, %0" : "+v"(f[i]));
                if (KIND == 14) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(*(double *)&f[i & 6]) : "v"(*(double *)&f[(i + 2) & 6]));
                // round 3: the integer multiplies and the fp64 operations K2 is made of
                if (KIND == 15) asm volatile("v_mul_lo_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (KIND == 16) asm volatile("v_mul_u32_u24 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
                if (KIND == 17) asm volatile("v_mad_u64_u32 %0, vcc, %1, %2, %0" : "+v"(*(unsigned long long *)&u[i & 6]) : "v"(u[(i + 1) & 7]), "v"(u[(i + 3) & 7]) : "vcc");
                if (KIND == 18) asm volatile("v_fma_f64 %0, %0, %1, %0" : "+v"(*(double *)&f[i & 6]) : "v"(*(double *)&f[(i + 2) & 6]));
                if (KIND == 19) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(*(double *)&f[i & 6]) : "v"(*(double *)&f[(i + 2) & 6]));
                if (KIND == 20) asm volatile("v_rcp_f64 %0, %0" : "+v"(*(double *)&f[i & 6]));
                if (KIND == 21) asm volatile("v_lshl_add_u64 %0, %0, 0, %1" : "+v"(*(unsigned long long *)&u[i & 6]) : "v"(*(unsigned long long *)&u[(i + 2) & 6]));
            }
    }
    float s = 0; unsigned t = 0;
    for (int i = 0; i < 8; i++) { s += f[i]; t += u[i]; }
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + t;
}

template <int KIND>
void run(const char *name, float *d, int waves_per_simd)
{
    const int blocks = 256 * waves_per_simd;  // 256 CUs x (waves_per_simd) blocks of 4 waves
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    k<KIND><<<blocks, 256>>>(d, 8);
    hipDeviceSynchronize();
    hipEventRecord(a);
    k<KIND><<<blocks, 256>>>(d, N_ITERS);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    double instr_per_simd = (double)waves_per_simd * N_ITERS * UNROLL;   // wave-instructions each SIMD issues
    printf("%-18s waves/SIMD=%d  %.3f ms  -> %.2f ns per wave-instr per SIMD (= %.2f cycles @2.4GHz)\n", name, waves_per_simd, ms,
           ms * 1e6 / instr_per_simd, ms * 1e6 / instr_per_simd * 2.4);
}

int main()
{
    float *d; hipMalloc(&d, 256 * 8 * 256 * sizeof(float) * 4);
    for (int w : {2, 4}) {
        run<0>("v_add_f32", d, w); run<1>("v_mul_f32", d, w); run<2>("v_fma_f32", d, w); run<3>("v_add_u32", d, w);
        run<4>("v_mad_u32_u24", d, w); run<5>("v_bfe_u32", d, w); run<6>("v_lshl_or_b32", d, w); run<7>("v_cvt_f32_ubyte1", d, w);
        run<8>("v_pk_add_f32", d, w); run<9>("v_pk_mul_f32", d, w); run<10>("v_perm_b32", d, w); run<11>("v_dot4_u32_u8", d, w);
        run<12>("v_add3_u32", d, w); run<13>("v_rndne_f32", d, w); run<14>("v_pk_fma_f32", d, w);
        run<15>("v_mul_lo_u32", d, w); run<16>("v_mul_u32_u24", d, w); run<17>("v_mad_u64_u32", d, w); run<18>("v_fma_f64", d, w);
        run<19>("v_mul_f64", d, w); run<20>("v_rcp_f64", d, w); run<21>("v_lshl_add_u64", d, w);
        printf("\n");
    }
    return 0;
}
