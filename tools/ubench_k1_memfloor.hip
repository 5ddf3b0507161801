// Micro-benchmark (round 3): the memory floor of K1's access pattern.  Same launch shape, same loads (one 12-byte load per lane per row of a 256-px
// strip, PF rows in flight, rows of a band in order) and same stores (4 bytes per lane per row, or the 16-bit bit-image stores) as
// k_preprocess_march, with trivial arithmetic in between.  What this takes is what K1 cannot go below without changing how it touches HBM.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
typedef unsigned int u32;
struct __attribute__((packed, aligned(4))) u32x3 { u32 a, b, c; };

template <bool BITS, int PF>
__global__ __launch_bounds__(256) void k(const uint8_t *__restrict__ bgr, int H, int W, uint8_t *__restrict__ out, int TH, int nstrips, int nbands, int nitems)
{
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int item = blockIdx.x * 4 + wave;
    if (item >= nitems) return;
    const int strip = item % nstrips, band = (item / nstrips) % nbands, frame = item / (nstrips * nbands);
    const uint8_t *img = bgr + (size_t)frame * H * W * 3;
    uint8_t *dst = out + (size_t)frame * H * (BITS ? W >> 3 : W);
    const int xs0 = strip * 240 - 8, cx0 = xs0 + 4 * lane;
    int cl = cx0 < 0 ? 0 : (cx0 > W - 4 ? W - 4 : cx0);
    const int ldx = 3 * cl;
    const int yb = band * TH, ye = (yb + TH < H) ? yb + TH : H;
    int q0 = yb - 7; if (q0 < 0) q0 = 0;
    u32x3 raw[PF];
#pragma unroll
    for (int i = 0; i < PF; i++) raw[i] = *(const u32x3 *)(img + (size_t)(q0 + i < H ? q0 + i : H - 1) * W * 3 + ldx);
    u32 acc = 0;
    for (int y = q0; y < ye + 7; y += PF) {
#pragma unroll
        for (int p = 0; p < PF; p++) {
            const int yy = y + p;
            const u32x3 d = raw[p];
            const int yn = yy + PF < H ? yy + PF : H - 1;
            raw[p] = *(const u32x3 *)(img + (size_t)yn * W * 3 + ldx);
            acc = (acc >> 1) ^ d.a ^ (d.b << 1) ^ d.c;
            const int yo = yy - 7;
            if (yo >= yb && yo < ye && lane >= 2 && lane < 62 && cx0 < W) {
                if (BITS) { if (((lane - 2) & 3) == 0) *(unsigned short *)(dst + (size_t)yo * (W >> 3) + (cx0 >> 3)) = (unsigned short)acc; }
                else *(u32 *)(dst + (size_t)yo * W + cx0) = acc;
            }
        }
    }
}

int main()
{
    const int n = 256, H = 1080, W = 1920;
    uint8_t *in, *out;
    hipMalloc(&in, (size_t)n * H * W * 3); hipMalloc(&out, (size_t)n * H * W);
    hipMemset(in, 7, (size_t)n * H * W * 3);
    const int nstrips = 8;
    for (int nbands : {3, 4, 6, 9, 12}) {
        const int TH = (H + nbands - 1) / nbands, nitems = n * nstrips * nbands;
        auto run = [&](auto kern, const char *name) {
            for (int i = 0; i < 20; i++) kern<<<(nitems + 3) / 4, 256>>>(in, H, W, out, TH, nstrips, nbands, nitems);
            hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
            hipEventRecord(a);
            for (int i = 0; i < 20; i++) kern<<<(nitems + 3) / 4, 256>>>(in, H, W, out, TH, nstrips, nbands, nitems);
            hipEventRecord(b); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            printf("%-18s bands %2d (%5d waves): %.4f ms per 256 frames\n", name, nbands, nitems, ms / 20);
        };
        run(k<false, 4>, "bytes out, PF 4");
        run(k<true, 4>, "bits out,  PF 4");
        run(k<false, 8>, "bytes out, PF 8");
    }
    return 0;
}
