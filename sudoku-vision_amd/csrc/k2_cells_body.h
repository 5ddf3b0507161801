// The body of k_warp_cells (k2_warp_cells.hip) as a device function, so that the fused K1 + K2 launch (k1_threshold.hip,
// k_preprocess_warp_fused) runs the very same code.  One workgroup of 256 threads per (frame, cell).
#pragma once
#include "sv_device.h"

namespace sv_k2 {

constexpr int OUT = 450, CELL = 50, MARGIN = 5, CROP = 40, CS = 28;

struct CellsLds {
    u8 crop[CROP * CROP];
    int tab[CS][3];              // offset, w0, w1 (same table for x and y: the crop is square)
    double M[9];
    double org[CROP][2][3];      // per crop row and 64-column block of the destination: the block origin's (X0, Y0, W0)
};

// vertical/horizontal bilinear of cv2.resize on a gray crop held in LDS
__device__ __forceinline__ int resize_px(const u8 *crop, int cw, int sh, int sw, int xo, int xa0, int xa1, int yo, int b0, int b1)
{
    int sx = xo, a0 = xa0, a1 = xa1;
    if (sx < 0) { sx = 0; a0 = 2048; a1 = 0; }
    if (sx >= sw - 1) { sx = sw - 1; a0 = 2048; a1 = 0; }
    const int sx1 = sx + 1 < sw ? sx + 1 : sx;
    const int sy0 = sv_clamp(yo, 0, sh - 1), sy1 = sv_clamp(yo + 1, 0, sh - 1);
    const int t0 = crop[sy0 * cw + sx] * a0 + crop[sy0 * cw + sx1] * a1;
    const int t1 = crop[sy1 * cw + sx] * a0 + crop[sy1 * cw + sx1] * a1;
    return (((b0 * (t0 >> 4)) >> 16) + ((b1 * (t1 >> 4)) >> 16) + 2) >> 2;
}

__device__ __forceinline__ void warp_cells_item(const u8 *__restrict__ frames, int H, int W, ptrdiff_t pitch, ptrdiff_t frame_stride,
                                                const double *__restrict__ minv, u8 *__restrict__ cells, int frame, int cell, CellsLds &L)
{
    const int tid = threadIdx.x;
    const int r = cell / 9, c = cell - r * 9;
    const u8 *img = frames + (ptrdiff_t)frame * frame_stride;
    if (tid < 9) L.M[tid] = minv[(ptrdiff_t)frame * 9 + tid];
    if (tid >= 64 && tid < 64 + CS) {
        int o, w0, w1;
        sv_resize_axis(CROP, CS, tid - 64, o, w0, w1);
        L.tab[tid - 64][0] = o;
        L.tab[tid - 64][1] = w0;
        L.tab[tid - 64][2] = w1;
    }
    __syncthreads();

    // the homography at a block origin is the same for every pixel of a (row, block): 80 threads evaluate it once (a 40-px crop row
    // touches at most two 64-column blocks) instead of 1600 pixels evaluating it each -- same operations, same order, same bits
    const int bw = sv_warp_block_w(OUT, OUT);
    const int xlo = c * CELL + MARGIN, ylo = r * CELL + MARGIN, blk0 = xlo / bw;
    if (tid < 2 * CROP) {
        const int y = tid >> 1, k = tid & 1;
        sv_warp_block_origin(L.M, (blk0 + k) * bw, ylo + y, L.org[y][k][0], L.org[y][k][1], L.org[y][k][2]);
    }
    __syncthreads();
    for (int i = tid; i < CROP * CROP; i += 256) {
        const int y = i / CROP, x = i - y * CROP;
        const int dx = xlo + x, k = dx / bw - blk0;
        int sx, sy, a, b, px[3];
        sv_warp_coord_from(L.M, L.org[y][k][0], L.org[y][k][1], L.org[y][k][2], dx - (blk0 + k) * bw, sx, sy, a, b);
        sv_warp_sample<3>(img, H, W, pitch, sx, sy, a, b, px);
        L.crop[i] = (u8)sv_gray_px(px[0], px[1], px[2]);
    }
    __syncthreads();

    u8 *dst = cells + ((ptrdiff_t)frame * 81 + cell) * (CS * CS);
    for (int i = tid; i < CS * CS; i += 256) {
        const int y = i / CS, x = i - y * CS;
        dst[i] = (u8)resize_px(L.crop, CROP, CROP, CROP, L.tab[x][0], L.tab[x][1], L.tab[x][2], L.tab[y][0], L.tab[y][1], L.tab[y][2]);
    }
}

}  // namespace sv_k2
