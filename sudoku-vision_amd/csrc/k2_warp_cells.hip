// K2 -- cv/grid.py:94-133 (warp_perspective) and cv/extract.py:13-56 (extract_cells) on MI355X.
//
//   k_warp_cells       : the fused path.  One workgroup per (frame, cell): warps only the 40x40 crop
//                        of the 450x450 grid that extract_cells keeps (crop rows/cols 50r+5..50r+44),
//                        converts it to gray in LDS, and resamples 40->28 with cv2.resize's 11-bit
//                        fixed-point bilinear.  Bit-identical to warp_perspective + extract_cells.
//   k_warp_perspective : the full SxS warp (API parity: warp_perspective returns the image).
//   k_extract_cells    : extract_cells on an arbitrary warped grid image.
#include "sv_device.h"
#include "sv_internal.h"
#include "k2_cells_body.h"

namespace {

using namespace sv_k2;

__global__ __launch_bounds__(256) void k_warp_cells(const u8 *__restrict__ frames, int H, int W, ptrdiff_t pitch,
                                                    ptrdiff_t frame_stride, const double *__restrict__ minv,
                                                    u8 *__restrict__ cells)
{
    __shared__ CellsLds L;
    warp_cells_item(frames, H, W, pitch, frame_stride, minv, cells, blockIdx.y, blockIdx.x, L);
}

template <int C>
__global__ void k_warp_perspective(const u8 *__restrict__ img, int H, int W, ptrdiff_t pitch, const double *__restrict__ minv,
                                   int S, u8 *__restrict__ dst)
{
    __shared__ double M[9];
    if (threadIdx.x < 9) M[threadIdx.x] = minv[threadIdx.x];
    __syncthreads();
    const int dx = blockIdx.x * blockDim.x + threadIdx.x, dy = blockIdx.y;
    if (dx >= S) return;
    int px[C];
    sv_warp_px<C>(img, H, W, pitch, M, dx, dy, sv_warp_block_w(S, S), px);
#pragma unroll
    for (int c = 0; c < C; c++) dst[((ptrdiff_t)dy * S + dx) * C + c] = (u8)px[c];
}

// one workgroup per cell; dynamic LDS: gray crop (ch*cw bytes) then the x and y tables
__global__ __launch_bounds__(256) void k_extract_cells(const u8 *__restrict__ grid, int h, int w, ptrdiff_t pitch, int C,
                                                       int cs, int mh, int mw, u8 *__restrict__ cells)
{
    extern __shared__ __attribute__((aligned(16))) u8 smem[];
    const int cell_h = h / 9, cell_w = w / 9;
    const int ch = cell_h - 2 * mh, cw = cell_w - 2 * mw;
    u8 *crop = smem;
    int *xt = (int *)(smem + ((ch * cw + 15) & ~15));
    int *yt = xt + 3 * cs;
    const int tid = threadIdx.x, cell = blockIdx.x, r = cell / 9, c = cell - r * 9;
    const int y1 = r * cell_h + mh, x1 = c * cell_w + mw;
    for (int i = tid; i < 2 * cs; i += 256) {
        int o, w0, w1;
        const bool isx = i < cs;
        const int d = isx ? i : i - cs;
        sv_resize_axis(isx ? cw : ch, cs, d, o, w0, w1);
        int *t = isx ? xt : yt;
        t[3 * d] = o;
        t[3 * d + 1] = w0;
        t[3 * d + 2] = w1;
    }
    for (int i = tid; i < ch * cw; i += 256) {
        const int y = i / cw, x = i - y * cw;
        const u8 *p = grid + (ptrdiff_t)(y1 + y) * pitch + (ptrdiff_t)(x1 + x) * C;
        crop[i] = C == 3 ? (u8)sv_gray_px(p[0], p[1], p[2]) : p[0];
    }
    __syncthreads();
    u8 *dst = cells + (ptrdiff_t)cell * cs * cs;
    const bool same = (ch == cs && cw == cs);  // cv2.resize to the same size is a copy
    for (int i = tid; i < cs * cs; i += 256) {
        const int y = i / cs, x = i - y * cs;
        dst[i] = same ? crop[i]
                      : (u8)resize_px(crop, cw, ch, cw, xt[3 * x], xt[3 * x + 1], xt[3 * x + 2], yt[3 * y], yt[3 * y + 1], yt[3 * y + 2]);
    }
}

// cv2.resize(img, (dw, dh)) INTER_LINEAR on a gray 8-bit image (cv/extract.py:52,93): one thread per output pixel
__global__ void k_resize_linear(const u8 *__restrict__ src, int sh, int sw, ptrdiff_t pitch, u8 *__restrict__ dst, int dh, int dw)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= dw) return;
    if (sh == dh && sw == dw) { dst[(ptrdiff_t)y * dw + x] = src[(ptrdiff_t)y * pitch + x]; return; }   // same size = copy
    int xo, a0, a1, yo, b0, b1;
    sv_resize_axis(sw, dw, x, xo, a0, a1);
    sv_resize_axis(sh, dh, y, yo, b0, b1);
    if (xo < 0) { xo = 0; a0 = 2048; a1 = 0; }
    if (xo >= sw - 1) { xo = sw - 1; a0 = 2048; a1 = 0; }
    const int x1 = xo + 1 < sw ? xo + 1 : xo;
    const u8 *r0 = src + (ptrdiff_t)sv_clamp(yo, 0, sh - 1) * pitch, *r1 = src + (ptrdiff_t)sv_clamp(yo + 1, 0, sh - 1) * pitch;
    const int t0 = r0[xo] * a0 + r0[x1] * a1, t1 = r1[xo] * a0 + r1[x1] * a1;
    dst[(ptrdiff_t)y * dw + x] = (u8)((((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2);
}

// is_cell_empty (cv/extract.py:59-79): Otsu threshold (OpenCV's single-pass fp64 recurrence, executed by one lane
// in the reference's operation order) of each cell, then the share of pixels <= threshold.  One wave per cell.
__global__ __launch_bounds__(64) void k_cell_ink_ratio(const u8 *__restrict__ cells, int npx, float *__restrict__ ratio, int *__restrict__ otsu)
{
    __shared__ int hist[256];
    __shared__ int thr;
    const int lane = threadIdx.x;
    const u8 *cell = cells + (ptrdiff_t)blockIdx.x * npx;
    for (int i = lane; i < 256; i += 64) hist[i] = 0;
    __syncthreads();
    for (int i = lane; i < npx; i += 64) atomicAdd(&hist[cell[i]], 1);
    __syncthreads();
    if (lane == 0) {
        const double scale = __ddiv_rn(1.0, (double)npx);
        double mu = 0;
        for (int i = 0; i < 256; i++) mu = __dadd_rn(mu, __dmul_rn((double)i, (double)hist[i]));
        mu = __dmul_rn(mu, scale);
        double mu1 = 0, q1 = 0, max_sigma = 0;
        int max_val = 0;
        for (int i = 0; i < 256; i++) {
            const double p_i = __dmul_rn((double)hist[i], scale);
            mu1 = __dmul_rn(mu1, q1);
            q1 = __dadd_rn(q1, p_i);
            const double q2 = __dsub_rn(1.0, q1);
            if (fmin(q1, q2) < 1.1920928955078125e-07 || fmax(q1, q2) > 1.0 - 1.1920928955078125e-07) continue;
            mu1 = __ddiv_rn(__dadd_rn(mu1, __dmul_rn((double)i, p_i)), q1);
            const double mu2 = __ddiv_rn(__dsub_rn(mu, __dmul_rn(q1, mu1)), q2);
            const double d = __dsub_rn(mu1, mu2);
            const double sigma = __dmul_rn(__dmul_rn(__dmul_rn(q1, q2), d), d);
            if (sigma > max_sigma) { max_sigma = sigma; max_val = i; }
        }
        thr = max_val;
    }
    __syncthreads();
    int cnt = 0;
    for (int i = lane; i < npx; i += 64) cnt += cell[i] <= thr;
    for (int o = 32; o > 0; o >>= 1) cnt += __shfl_down(cnt, o);
    if (lane == 0) {
        ratio[blockIdx.x] = (float)((double)cnt / (double)npx);
        if (otsu) otsu[blockIdx.x] = thr;
    }
}

}  // namespace

int svk_resize_linear(const u8 *src, int sh, int sw, ptrdiff_t pitch, u8 *dst, int dh, int dw, hipStream_t s)
{
    hipLaunchKernelGGL(k_resize_linear, dim3((dw + 127) / 128, dh), dim3(128), 0, s, src, sh, sw, pitch, dst, dh, dw);
    SV_LAUNCH_CHECK("k_resize_linear");
    return SV_OK;
}

int svk_cell_ink_ratio(const u8 *cells, long B, int npx, float *ratio, int *otsu, hipStream_t s)
{
    hipLaunchKernelGGL(k_cell_ink_ratio, dim3((unsigned)B), dim3(64), 0, s, cells, npx, ratio, otsu);
    SV_LAUNCH_CHECK("k_cell_ink_ratio");
    return SV_OK;
}

int svk_warp_cells(sv_ctx *ctx, const u8 *frames, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t frame_stride, const double *minv, u8 *cells, hipStream_t s)
{
    sv_time_scope ts(ctx, SVK_WARP_CELLS, s);
    hipLaunchKernelGGL(k_warp_cells, dim3(81, n), dim3(256), 0, s, frames, H, W, pitch, frame_stride, minv, cells);
    SV_LAUNCH_CHECK("k_warp_cells");
    return SV_OK;
}

int svk_warp_perspective(const u8 *img, int H, int W, ptrdiff_t pitch, int channels, const double *minv, int out_size, u8 *dst, hipStream_t s)
{
    dim3 grid((out_size + 127) / 128, out_size);
    if (channels == 3)
        hipLaunchKernelGGL(k_warp_perspective<3>, grid, dim3(128), 0, s, img, H, W, pitch, minv, out_size, dst);
    else
        hipLaunchKernelGGL(k_warp_perspective<1>, grid, dim3(128), 0, s, img, H, W, pitch, minv, out_size, dst);
    SV_LAUNCH_CHECK("k_warp_perspective");
    return SV_OK;
}

int svk_extract_cells(const u8 *grid, int h, int w, ptrdiff_t pitch, int channels, int cell_size, int margin_h, int margin_w, u8 *cells, hipStream_t s)
{
    const int ch = h / 9 - 2 * margin_h, cw = w / 9 - 2 * margin_w;
    const size_t lds = (size_t)((ch * cw + 15) & ~15) + sizeof(int) * 6 * (size_t)cell_size;
    if (lds > 64 * 1024) return sv_fail(SV_ERR_UNSUPPORTED, "extract_cells: crop %dx%d -> %d needs %zu B of LDS (max 65536)", ch, cw, cell_size, lds);
    hipLaunchKernelGGL(k_extract_cells, dim3(81), dim3(256), lds, s, grid, h, w, pitch, channels, cell_size, margin_h, margin_w, cells);
    SV_LAUNCH_CHECK("k_extract_cells");
    return SV_OK;
}
