// Device-side building blocks shared by the K1/K2 kernels.  Integer stages are exact by
// construction; float/double stages use the _rn intrinsics so that no multiply-add is ever fused
// (the arithmetic being matched rounds after every operation).
#pragma once
#include <hip/hip_runtime.h>
#include <cstdint>

typedef uint8_t u8;

// cv2.cvtColor(BGR2GRAY), 15-bit coefficients (cv/preprocess.py:19, cv/extract.py:49).
__device__ __forceinline__ int sv_gray_px(int b, int g, int r)
{
    return (b * 3735 + g * 19235 + r * 9798 + 16384) >> 15;
}

__device__ __forceinline__ int sv_clamp(int v, int lo, int hi) { return v < lo ? lo : (v > hi ? hi : v); }

// cv2 BORDER_REFLECT_101
__device__ __forceinline__ int sv_reflect101(int p, int len)
{
    if (len == 1) return 0;
    while (p < 0 || p >= len) p = p < 0 ? -p : 2 * len - 2 - p;
    return p;
}

// Width of the column blocks cv2.warpPerspective evaluates its fp64 coordinates in.
__host__ __device__ __forceinline__ int sv_warp_block_w(int dw, int dh)
{
    int bh0 = 16 < dh ? 16 : dh;
    int bw0 = 1024 / bh0 < dw ? 1024 / bh0 : dw;
    return bw0;
}

// cv2.warpPerspective evaluates the homography at the origin (x0, dy) of each 64-column block of a destination row in double and
// steps from there: these three values depend on the block and the row only.
__device__ __forceinline__ void sv_warp_block_origin(const double *M, int x0, int dy, double &X0, double &Y0, double &W0)
{
    const double fx0 = (double)x0, fy = (double)dy;
    X0 = __dadd_rn(__dadd_rn(__dmul_rn(M[0], fx0), __dmul_rn(M[1], fy)), M[2]);
    Y0 = __dadd_rn(__dadd_rn(__dmul_rn(M[3], fx0), __dmul_rn(M[4], fy)), M[5]);
    W0 = __dadd_rn(__dadd_rn(__dmul_rn(M[6], fx0), __dmul_rn(M[7], fy)), M[8]);
}

// Destination pixel x1 columns right of a block origin -> source cell (sx,sy) and 1/32 fractions (a,b), cv/grid.py:131.
__device__ __forceinline__ void sv_warp_coord_from(const double *M, double X0, double Y0, double W0, int x1, int &sx, int &sy, int &a, int &b)
{
    const double fx1 = (double)x1;
    double Wv = __dadd_rn(W0, __dmul_rn(M[6], fx1));
    Wv = Wv != 0.0 ? __ddiv_rn(32.0, Wv) : 0.0;
    double fX = __dmul_rn(__dadd_rn(X0, __dmul_rn(M[0], fx1)), Wv);
    double fY = __dmul_rn(__dadd_rn(Y0, __dmul_rn(M[3], fx1)), Wv);
    fX = fmax(-2147483648.0, fmin(2147483647.0, fX));
    fY = fmax(-2147483648.0, fmin(2147483647.0, fY));
    const int X = __double2int_rn(fX), Y = __double2int_rn(fY);
    sx = sv_clamp(X >> 5, -32768, 32767);
    sy = sv_clamp(Y >> 5, -32768, 32767);
    a = X & 31;
    b = Y & 31;
}

// Destination pixel (dx,dy) -> source cell (sx,sy) and 1/32 fractions (a,b)
__device__ __forceinline__ void sv_warp_coord(const double *M, int dx, int dy, int bw, int &sx, int &sy, int &a, int &b)
{
    const int x0 = (dx / bw) * bw;
    double X0, Y0, W0;
    sv_warp_block_origin(M, x0, dy, X0, Y0, W0);
    sv_warp_coord_from(M, X0, Y0, W0, dx - x0, sx, sy, a, b);
}

__device__ __forceinline__ int sv_tap(const u8 *img, int H, int W, ptrdiff_t pitch, int C, int x, int y, int c)
{
    if ((unsigned)x >= (unsigned)W || (unsigned)y >= (unsigned)H) return 0;
    return img[(ptrdiff_t)y * pitch + (ptrdiff_t)x * C + c];
}

// One bilinear sample at source cell (sx,sy) with fractions (a,b)/32: 15-bit weights (they sum to 32768), constant-0 border.
template <int C>
__device__ __forceinline__ void sv_warp_sample(const u8 *img, int H, int W, ptrdiff_t pitch, int sx, int sy, int a, int b, int (&out)[C]);

template <int C>
__device__ __forceinline__ void sv_warp_px(const u8 *img, int H, int W, ptrdiff_t pitch, const double *M, int dx, int dy,
                                           int bw, int (&out)[C])
{
    int sx, sy, a, b;
    sv_warp_coord(M, dx, dy, bw, sx, sy, a, b);
    sv_warp_sample<C>(img, H, W, pitch, sx, sy, a, b, out);
}

template <int C>
__device__ __forceinline__ void sv_warp_sample(const u8 *img, int H, int W, ptrdiff_t pitch, int sx, int sy, int a, int b, int (&out)[C])
{
    const int w00 = (32 - a) * (32 - b) * 32, w01 = a * (32 - b) * 32, w10 = (32 - a) * b * 32, w11 = a * b * 32;
    const bool inside = (unsigned)sx < (unsigned)(W - 1) && (unsigned)sy < (unsigned)(H - 1);
    if (inside) {
        const u8 *p0 = img + (ptrdiff_t)sy * pitch + (ptrdiff_t)sx * C;
        const u8 *p1 = p0 + pitch;
#pragma unroll
        for (int c = 0; c < C; c++)
            out[c] = (p0[c] * w00 + p0[C + c] * w01 + p1[c] * w10 + p1[C + c] * w11 + 16384) >> 15;
    } else {
#pragma unroll
        for (int c = 0; c < C; c++)
            out[c] = (sv_tap(img, H, W, pitch, C, sx, sy, c) * w00 + sv_tap(img, H, W, pitch, C, sx + 1, sy, c) * w01 +
                      sv_tap(img, H, W, pitch, C, sx, sy + 1, c) * w10 + sv_tap(img, H, W, pitch, C, sx + 1, sy + 1, c) * w11 +
                      16384) >> 15;
    }
}

// cv2.resize INTER_LINEAR 8-bit axis table entry (cv/extract.py:52): source offset + two 11-bit weights.
__device__ __forceinline__ void sv_resize_axis(int s_len, int d_len, int d, int &ofs, int &w0, int &w1)
{
    const double scale = __ddiv_rn(1.0, __ddiv_rn((double)d_len, (double)s_len));
    float f = (float)__dadd_rn(__dmul_rn(__dadd_rn((double)d, 0.5), scale), -0.5);
    const int s = (int)floorf(f);
    f = __fsub_rn(f, (float)s);
    ofs = s;
    w0 = __float2int_rn(__fmul_rn(__fsub_rn(1.f, f), 2048.f));
    w1 = __float2int_rn(__fmul_rn(f, 2048.f));
}
