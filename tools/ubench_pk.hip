// Micro-benchmark (round 3): what a v_pk_fma_f32 costs against two v_fma_f32 on gfx950, in SHADER CYCLES (s_memtime), at 4 waves per SIMD,
// with the operand kinds K1 uses (VGPR pairs, an SGPR pair with op_sel broadcast).  Build: hipcc -O3 --offload-arch=gfx950 tools/ubench_pk.hip -o tools/ubench_pk.bin
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x2 __attribute__((ext_vector_type(2)));

template <int KIND>
__global__ __launch_bounds__(256) void k(float *out, unsigned long long *cyc, int n, float ka, float kb)
{
    f32x2 a[8];
    float s[16];
    for (int i = 0; i < 8; i++) { a[i] = (f32x2){threadIdx.x * 0.001f + i, threadIdx.x * 0.002f + i}; s[2 * i] = a[i][0]; s[2 * i + 1] = a[i][1]; }
    const f32x2 kk = {ka, kb};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < n; it++) {
#pragma unroll
        for (int r = 0; r < 4; r++) {
            if (KIND == 0) {          // 16 independent v_fma_f32 (VGPR, SGPR, VGPR)
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(s[i]) : "s"(ka));
            } else if (KIND == 1) {   // 8 independent v_pk_fma_f32 with an SGPR-pair coefficient, op_sel broadcast
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "s"(kk));
            } else if (KIND == 2) {   // 8 v_pk_fma_f32, all-VGPR
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %0" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            } else if (KIND == 3) {   // 16 v_fma_f32 all-VGPR
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(s[i]) : "v"(s[(i + 1) & 15]));
            } else if (KIND == 4) {   // K1-like mix per 16 "elements": 8 f32 fma + 4 pk fma
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_fma_f32 %0, %0, %1, %0" : "+v"(s[i]) : "s"(ka));
#pragma unroll
                for (int i = 4; i < 8; i++) asm volatile("v_pk_fma_f32 %0, %0, %1, %0 op_sel_hi:[1,0,1]" : "+v"(a[i]) : "s"(kk));
            } else if (KIND == 5) {   // 16 v_cvt_f32_ubyte1 ("slow" class)
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_cvt_f32_ubyte1 %0, %0" : "+v"(s[i]));
            } else if (KIND == 6) {   // 16 v_mov_b32 dpp
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_mov_b32_dpp %0, %1 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "=v"(s[i]) : "v"(s[(i + 1) & 15]));
            } else if (KIND == 7) {   // 8 v_pk_add_f32
#pragma unroll
                for (int i = 0; i < 8; i++) asm volatile("v_pk_add_f32 %0, %0, %1" : "+v"(a[i]) : "v"(a[(i + 1) & 7]));
            } else if (KIND == 8) {   // 16 v_floor_f32
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_floor_f32 %0, %0" : "+v"(s[i]));
            } else if (KIND == 9) {   // 16 v_add_f32 dpp (fused neighbour add)
#pragma unroll
                for (int i = 0; i < 16; i++) asm volatile("v_add_f32_dpp %0, %1, %0 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:1" : "+v"(s[i]) : "v"(s[(i + 1) & 15]));
            }
        }
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    float acc = 0;
    for (int i = 0; i < 8; i++) acc += a[i][0] + a[i][1] + s[2 * i] + s[2 * i + 1];
    out[blockIdx.x * blockDim.x + threadIdx.x] = acc;
    if (threadIdx.x == 0 && blockIdx.x == 0) *cyc = t1 - t0;
}

template <int KIND>
void run(const char *name, int per_iter, int elems_per_instr, float *d, unsigned long long *dc, int waves)
{
    const int n = 2048;
    k<KIND><<<256 * waves, 256>>>(d, dc, 64, 1.0001f, 0.9999f);
    hipDeviceSynchronize();
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    hipEventRecord(a);
    k<KIND><<<256 * waves, 256>>>(d, dc, n, 1.0001f, 0.9999f);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    unsigned long long c; hipMemcpy(&c, dc, 8, hipMemcpyDeviceToHost);
    const double instr_per_simd = (double)waves * n * 4 * per_iter;
    // s_memtime ticks at a fixed 100 MHz on this part: report wall-clock ns per wave-instruction per SIMD and the tick count for reference
    printf("%-44s waves/SIMD=%d  %.3f ms  %.3f ns per wave-instr per SIMD = %.3f ns per element-op  (memtime ticks %llu)\n", name, waves, ms, ms * 1e6 / instr_per_simd,
           ms * 1e6 / instr_per_simd / elems_per_instr, c);
}

int main()
{
    float *d; unsigned long long *dc;
    hipMalloc(&d, 256 * 8 * 256 * sizeof(float)); hipMalloc(&dc, 8);
    for (int w : {4, 8}) {
        run<0>("v_fma_f32 (v, s, v)", 16, 1, d, dc, w);
        run<3>("v_fma_f32 (v, v, v)", 16, 1, d, dc, w);
        run<1>("v_pk_fma_f32 (v, s-pair op_sel bcast, v)", 8, 2, d, dc, w);
        run<2>("v_pk_fma_f32 (v, v, v)", 8, 2, d, dc, w);
        run<7>("v_pk_add_f32", 8, 2, d, dc, w);
        run<4>("mix: 8 v_fma_f32 + 4 v_pk_fma_f32", 12, 1, d, dc, w);
        run<5>("v_cvt_f32_ubyte1", 16, 1, d, dc, w);
        run<8>("v_floor_f32", 16, 1, d, dc, w);
        run<6>("v_mov_b32 dpp wave_shr", 16, 1, d, dc, w);
        run<9>("v_add_f32 dpp wave_shr", 16, 1, d, dc, w);
        printf("\n");
    }
    return 0;
}
