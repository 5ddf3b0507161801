// Probe (round 3): do gfx950's typed buffer loads (buffer_load_format_d16_xyzw, DATA_FORMAT 8_8_8_8, NUM_FORMAT USCALED) deliver the bytes of a BGR
// frame as exact f16 values, and what does a K1-like row loop cost with them + v_fma_mix_f32 against global_load_dwordx3 + 12 v_cvt_f32_ubyte?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ h4 llvm_buffer_load_format_v4f16(i32x4 rsrc, int voffset, int soffset, int aux) __asm("llvm.amdgcn.raw.buffer.load.format.v4f16");

__global__ void k_check(const unsigned char *in, int n, float *out)
{
    i32x4 r;
    r[0] = (int)(unsigned)(size_t)in;
    r[1] = (int)((size_t)in >> 32) & 0xFFFF;
    r[2] = n;
    r[3] = (4 << 0) | (5 << 3) | (6 << 6) | (7 << 9) | (2 << 12) | (10 << 15);
    const int t = blockIdx.x * blockDim.x + threadIdx.x;
    if (4 * t + 3 < n) {
        const h4 a = llvm_buffer_load_format_v4f16(r, 4 * t, 0, 0);
        for (int i = 0; i < 4; i++) out[4 * t + i] = (float)a[i];
    }
}

int main()
{
    const int n = 1 << 16;
    std::vector<unsigned char> h(n);
    for (int i = 0; i < n; i++) h[i] = (unsigned char)((i * 37 + (i >> 8)) & 255);
    unsigned char *d; float *o;
    hipMalloc(&d, n); hipMalloc(&o, n * 4);
    hipMemcpy(d, h.data(), n, hipMemcpyHostToDevice);
    k_check<<<n / 4 / 256, 256>>>(d, n, o);
    std::vector<float> r(n);
    hipMemcpy(r.data(), o, n * 4, hipMemcpyDeviceToHost);
    int bad = 0;
    for (int i = 0; i < n; i++) bad += r[i] != (float)h[i];
    printf("typed load (8_8_8_8 USCALED, d16): %d of %d bytes wrong; first values %g %g %g %g (want %d %d %d %d)\n", bad, n, r[0], r[1], r[2], r[3], h[0], h[1], h[2], h[3]);
    return bad != 0;
}
