// K1 -- cv/preprocess.py on MI355X.
//
//   k_preprocess_fused : preprocess_for_grid_detection (preprocess.py:57-65) in one pass:
//                        BGR -> gray -> 5x5 integer Gaussian (REFLECT_101) -> 11x11 f32 Gaussian mean
//                        (REPLICATE) -> round-half-even -> src - mean <= -2 ? 255 : 0.
//                        Reads 3 B/px, writes 1 B/px; every intermediate lives in LDS.
//   k_gray / k_blur / k_adaptive_threshold : the three stages as stand-alone calls (grayscale(),
//                        blur(), threshold() of preprocess.py) for drop-in use; direct form.
//
// The f32 Gaussian rounds after every multiply and every add, in the order OpenCV's scalar
// FilterEngine uses (row: taps left to right; column: centre, then symmetric pairs outward), so the
// output is bit-identical to the CPU oracle.
#include "sv_device.h"
#include "sv_internal.h"

namespace {

constexpr int TW = 64, TH = 32;            // output tile
constexpr int GW = TW + 14, GH = TH + 14;  // gray tile      (halo 2 + 5)
constexpr int BW = TW + 10, BH = TH + 10;  // blurred tile   (halo 5)

struct Taps11 { float k[11]; };

__global__ __launch_bounds__(256) void k_preprocess_fused(const u8 *__restrict__ bgr, int H, int W, ptrdiff_t pitch,
                                                          ptrdiff_t img_stride, u8 *__restrict__ out, Taps11 taps)
{
    __shared__ u8 g[GH][GW + 2];
    __shared__ unsigned short hb[GH][BW + 2];
    __shared__ float bl[BH][BW + 1];
    __shared__ float rw[BH][TW + 1];

    const int tid = threadIdx.x;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const u8 *img = bgr + (ptrdiff_t)blockIdx.z * img_stride;
    u8 *dst = out + (ptrdiff_t)blockIdx.z * H * W;

    // 1. gray tile.  Column p of the extended image is gray[reflect101(clamp(p,-2,W+1))]: the blur at
    //    a replicated border column only ever looks 2 pixels past the image.
    for (int i = tid; i < GH * GW; i += 256) {
        const int ly = i / GW, lx = i - ly * GW;
        const int sx = sv_reflect101(sv_clamp(x0 - 7 + lx, -2, W + 1), W);
        const int sy = sv_reflect101(sv_clamp(y0 - 7 + ly, -2, H + 1), H);
        const u8 *p = img + (ptrdiff_t)sy * pitch + (ptrdiff_t)sx * 3;
        g[ly][lx] = (u8)sv_gray_px(p[0], p[1], p[2]);
    }
    __syncthreads();

    // 2. horizontal 1-4-6-4-1 in 8.8 fixed point (exact).  Blur column q of the REPLICATE-extended
    //    blurred image is the blur at clamp(q, 0, W-1).
    for (int i = tid; i < GH * BW; i += 256) {
        const int ly = i / BW, bq = i - ly * BW;
        const int lc = sv_clamp(x0 - 5 + bq, 0, W - 1) - (x0 - 7);
        hb[ly][bq] = (unsigned short)(16 * (g[ly][lc - 2] + g[ly][lc + 2]) + 64 * (g[ly][lc - 1] + g[ly][lc + 1]) +
                                      96 * g[ly][lc]);
    }
    __syncthreads();

    // 3. vertical pass, (v + 2^15) >> 16, kept as f32 for the float stage.
    for (int i = tid; i < BH * BW; i += 256) {
        const int br = i / BW, bq = i - br * BW;
        const int lr = sv_clamp(y0 - 5 + br, 0, H - 1) - (y0 - 7);
        const unsigned v = 16u * (hb[lr - 2][bq] + hb[lr + 2][bq]) + 64u * (hb[lr - 1][bq] + hb[lr + 1][bq]) + 96u * hb[lr][bq];
        bl[br][bq] = (float)((v + 32768u) >> 16);
    }
    __syncthreads();

    // 4. f32 row pass, taps left to right.
    for (int i = tid; i < BH * TW; i += 256) {
        const int br = i / TW, x = i - br * TW;
        float acc = __fmul_rn(taps.k[0], bl[br][x]);
#pragma unroll
        for (int j = 1; j < 11; j++) acc = __fadd_rn(acc, __fmul_rn(taps.k[j], bl[br][x + j]));
        rw[br][x] = acc;
    }
    __syncthreads();

    // 5. f32 column pass (centre, then pairs), round half to even, compare.
    for (int i = tid; i < TH * TW; i += 256) {
        const int y = i / TW, x = i - y * TW;
        if (x0 + x >= W || y0 + y >= H) continue;
        float acc = __fadd_rn(__fmul_rn(taps.k[5], rw[y + 5][x]), 0.f);
#pragma unroll
        for (int j = 1; j <= 5; j++) acc = __fadd_rn(acc, __fmul_rn(taps.k[5 + j], __fadd_rn(rw[y + 5 + j][x], rw[y + 5 - j][x])));
        const int mean = sv_clamp(__float2int_rn(acc), 0, 255);
        const int src = (int)bl[y + 5][x + 5];
        dst[(ptrdiff_t)(y0 + y) * W + (x0 + x)] = (src - mean <= -2) ? 255 : 0;
    }
}

__global__ void k_gray(const u8 *__restrict__ bgr, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride, u8 *__restrict__ gray)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const u8 *p = bgr + (ptrdiff_t)blockIdx.z * img_stride + (ptrdiff_t)y * pitch + (ptrdiff_t)x * 3;
    gray[((ptrdiff_t)blockIdx.z * H + y) * W + x] = (u8)sv_gray_px(p[0], p[1], p[2]);
}

struct FxTaps { int k[7]; int n; };

__global__ void k_blur(const u8 *__restrict__ src, int H, int W, FxTaps t, u8 *__restrict__ dst)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const u8 *s = src + (ptrdiff_t)blockIdx.z * H * W;
    const int r = t.n / 2;
    unsigned acc = 0;
    for (int i = 0; i < t.n; i++) {
        const u8 *row = s + (ptrdiff_t)sv_reflect101(y + i - r, H) * W;
        unsigned h = 0;
        for (int j = 0; j < t.n; j++) h += (unsigned)t.k[j] * row[sv_reflect101(x + j - r, W)];
        acc += (unsigned)t.k[i] * h;
    }
    dst[((ptrdiff_t)blockIdx.z * H + y) * W + x] = (u8)((acc + 32768u) >> 16);
}

struct FTaps { float k[31]; int n; };

// row value of the f32 Gaussian at (x, row) in FilterEngine order
__device__ __forceinline__ float row_value(const u8 *row, int x, int W, const FTaps &t)
{
    const int r = t.n / 2;
    float acc;
    if (t.n <= 5) {
        acc = __fmul_rn((float)row[x], t.k[r]);
        for (int j = 1; j <= r; j++) {
            const float pr = __fadd_rn((float)row[sv_clamp(x - j, 0, W - 1)], (float)row[sv_clamp(x + j, 0, W - 1)]);
            acc = __fadd_rn(acc, __fmul_rn(pr, t.k[r + j]));
        }
    } else {
        acc = __fmul_rn(t.k[0], (float)row[sv_clamp(x - r, 0, W - 1)]);
        for (int j = 1; j < t.n; j++) acc = __fadd_rn(acc, __fmul_rn(t.k[j], (float)row[sv_clamp(x + j - r, 0, W - 1)]));
    }
    return acc;
}

__global__ void k_adaptive_threshold(const u8 *__restrict__ src, int H, int W, FTaps t, int idelta, int type_inv, u8 *__restrict__ dst)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const u8 *s = src + (ptrdiff_t)blockIdx.z * H * W;
    const int r = t.n / 2;
    float acc = __fadd_rn(__fmul_rn(t.k[r], row_value(s + (ptrdiff_t)y * W, x, W, t)), 0.f);
    for (int j = 1; j <= r; j++) {
        const float lo = row_value(s + (ptrdiff_t)sv_clamp(y + j, 0, H - 1) * W, x, W, t);
        const float hi = row_value(s + (ptrdiff_t)sv_clamp(y - j, 0, H - 1) * W, x, W, t);
        acc = __fadd_rn(acc, __fmul_rn(t.k[r + j], __fadd_rn(lo, hi)));
    }
    const int mean = sv_clamp(__float2int_rn(acc), 0, 255);
    const int d = (int)s[(ptrdiff_t)y * W + x] - mean;
    const bool on = type_inv ? (d <= -idelta) : (d > -idelta);
    dst[((ptrdiff_t)blockIdx.z * H + y) * W + x] = on ? 255 : 0;
}

}  // namespace

int svk_preprocess(sv_ctx *ctx, const u8 *bgr, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride, u8 *binary, hipStream_t s)
{
    Taps11 t;
    sv_gaussian_taps_f32(11, t.k);
    dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, n);
    sv_time_scope ts(ctx, SVK_PREPROCESS, s);
    hipLaunchKernelGGL(k_preprocess_fused, grid, dim3(256), 0, s, bgr, H, W, pitch, img_stride, binary, t);
    SV_LAUNCH_CHECK("k_preprocess_fused");
    return SV_OK;
}

int svk_gray(const u8 *bgr, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride, u8 *gray, hipStream_t s)
{
    dim3 grid((W + 255) / 256, H, n);
    hipLaunchKernelGGL(k_gray, grid, dim3(256), 0, s, bgr, H, W, pitch, img_stride, gray);
    SV_LAUNCH_CHECK("k_gray");
    return SV_OK;
}

int svk_blur(const u8 *src, int n, int H, int W, int ksize, u8 *dst, hipStream_t s)
{
    static const int tab[4][7] = {{256, 0, 0, 0, 0, 0, 0}, {64, 128, 64, 0, 0, 0, 0}, {16, 64, 96, 64, 16, 0, 0}, {8, 28, 56, 72, 56, 28, 8}};
    FxTaps t;
    t.n = ksize;
    for (int i = 0; i < 7; i++) t.k[i] = tab[ksize / 2][i];
    dim3 grid((W + 255) / 256, H, n);
    hipLaunchKernelGGL(k_blur, grid, dim3(256), 0, s, src, H, W, t, dst);
    SV_LAUNCH_CHECK("k_blur");
    return SV_OK;
}

int svk_adaptive_threshold(const u8 *src, int n, int H, int W, int block, const float *taps, int idelta, int type_inv, u8 *dst, hipStream_t s)
{
    FTaps t;
    t.n = block;
    for (int i = 0; i < 31; i++) t.k[i] = i < block ? taps[i] : 0.f;
    dim3 grid((W + 255) / 256, H, n);
    hipLaunchKernelGGL(k_adaptive_threshold, grid, dim3(256), 0, s, src, H, W, t, idelta, type_inv, dst);
    SV_LAUNCH_CHECK("k_adaptive_threshold");
    return SV_OK;
}
