// Micro-benchmark: ds_read_b128 throughput of one CU with W waves resident (conflict-free, lane-linear 1-KB reads as an MFMA B-operand fetch does), in
// cycles per wave-instruction per CU -- what the fc kernels' weight fragments cost the LDS when nothing else runs.
#include <hip/hip_runtime.h>
#include <cstdio>
template <int NW>
__global__ __launch_bounds__(64 * NW) void k(unsigned *out, int iters)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[65536];
    for (int i = threadIdx.x; i < 16384; i += 64 * NW) ((unsigned *)lds)[i] = i;
    __syncthreads();
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const unsigned char *p = lds + lane * 16 + (wave & 3) * 16384;
    uint4 acc = {0, 0, 0, 0};
    const unsigned long long t0 = __builtin_amdgcn_s_memtime();
    for (int it = 0; it < iters; it++) {
        uint4 v[16];
#pragma unroll
        for (int j = 0; j < 16; j++) v[j] = *(const uint4 *)(p + 1024 * j);
#pragma unroll
        for (int j = 0; j < 16; j++) { acc.x ^= v[j].x; acc.y ^= v[j].y; acc.z ^= v[j].z; acc.w ^= v[j].w; }
        asm volatile("" ::: "memory");
    }
    const unsigned long long t1 = __builtin_amdgcn_s_memtime();
    if (lane == 0 && blockIdx.x == 0 && (wave == 0 || wave == NW - 1))
        printf("%2d waves, wave %2d: %.1f s_memtime ticks per ds_read_b128 of this wave -> %.1f per wave-instruction for the CU\n", NW, wave, (double)(t1 - t0) / (16.0 * iters),
               (double)(t1 - t0) / (16.0 * iters * NW));
    if (acc.x == 0x12345u) out[threadIdx.x] = acc.y ^ acc.z ^ acc.w;
}
int main()
{
    unsigned *out; hipMalloc(&out, 1 << 16);
    k<1><<<256, 64>>>(out, 2000); hipDeviceSynchronize();
    k<4><<<256, 256>>>(out, 2000); hipDeviceSynchronize();
    k<8><<<256, 512>>>(out, 2000); hipDeviceSynchronize();
    k<12><<<256, 768>>>(out, 2000); hipDeviceSynchronize();
    k<16><<<256, 1024>>>(out, 2000); hipDeviceSynchronize();
    return 0;
}
