// Device half of the JPEG front end (scope row N4; cv2.imread at pipeline/run.py:250): Huffman-decoded coefficient
// blocks -> BGR frame in HBM.  Integer arithmetic throughout, bit-exact with libjpeg's defaults (the decoder behind
// cv2.imread and Pillow): JDCT_ISLOW inverse DCT, "fancy" (triangle-filter) chroma up-sampling, 16-bit fixed-point
// YCbCr -> RGB, EXIF orientation.
//
//   k_jpeg_idct    one thread per block row/column, 8 threads per 8x8 block, 32 blocks per workgroup.  Coefficients come in as
//                  one 16-byte load per thread (a block row), are dequantised into LDS, transformed in place column-wise then
//                  row-wise (LDS rows padded to 9 words: both passes conflict-free), and leave as one 8-byte store per thread
//                  into the component plane.  HBM-bound: 2 B in + 1 B out per sample.
//   k_jpeg_colour  one thread per OUTPUT pixel (so stores stay coalesced under every orientation): luma sample, the two
//                  interpolated chroma samples in closed form (no intermediate full-resolution chroma planes), colour
//                  conversion, 3-byte store.
#include "sv_internal.h"

namespace {

struct JpegGeom {
    int ncomp, W, H, OW, OH, hmax, vmax, orientation;
    int bw[3], bh[3];
    long coef_off[3], plane_off[3];
    long blk_start[4];                 // prefix sums of blocks per component
};

__device__ __forceinline__ int descale(int x, int n) { return (x + (1 << (n - 1))) >> n; }

// libjpeg's post-IDCT range-limit table: index (x & 1023), centred on 128
__device__ __forceinline__ unsigned range_limit(int x)
{
    const int v = x & 1023;
    return (unsigned)(v < 128 ? v + 128 : v < 512 ? 255 : v < 896 ? 0 : v - 896);
}

// one 8-point pass of jidctint.c (CONST_BITS 13); the caller applies the pass-specific descale
__device__ __forceinline__ void idct8(const int (&in)[8], int (&out)[8])
{
    int z2 = in[2], z3 = in[6];
    int z1 = (z2 + z3) * 4433;
    int tmp2 = z1 + z3 * (-15137), tmp3 = z1 + z2 * 6270;
    int tmp0 = (in[0] + in[4]) * 8192, tmp1 = (in[0] - in[4]) * 8192;
    const int tmp10 = tmp0 + tmp3, tmp13 = tmp0 - tmp3, tmp11 = tmp1 + tmp2, tmp12 = tmp1 - tmp2;
    tmp0 = in[7]; tmp1 = in[5]; tmp2 = in[3]; tmp3 = in[1];
    z1 = tmp0 + tmp3; z2 = tmp1 + tmp2; z3 = tmp0 + tmp2;
    int z4 = tmp1 + tmp3;
    const int z5 = (z3 + z4) * 9633;
    tmp0 *= 2446; tmp1 *= 16819; tmp2 *= 25172; tmp3 *= 12299;
    z1 *= -7373; z2 *= -20995; z3 *= -16069; z4 *= -3196;
    z3 += z5; z4 += z5;
    tmp0 += z1 + z3; tmp1 += z2 + z4; tmp2 += z2 + z3; tmp3 += z1 + z4;
    out[0] = tmp10 + tmp3; out[7] = tmp10 - tmp3;
    out[1] = tmp11 + tmp2; out[6] = tmp11 - tmp2;
    out[2] = tmp12 + tmp1; out[5] = tmp12 - tmp1;
    out[3] = tmp13 + tmp0; out[4] = tmp13 - tmp0;
}

// natural (row-major) position -> zigzag index
__constant__ unsigned char kZigzagOf[64] = {0, 1, 5, 6, 14, 15, 27, 28, 2, 4, 7, 13, 16, 26, 29, 42, 3, 8, 12, 17, 25, 30, 41, 43, 9, 11, 18, 24, 31, 40, 44, 53,
                                            10, 19, 23, 32, 39, 45, 52, 54, 20, 22, 33, 38, 46, 51, 55, 60, 21, 34, 37, 47, 50, 56, 59, 61, 35, 36, 48, 49, 57, 58, 62, 63};

// SPARSE: the block arrives as (mask over zigzag positions, offset of its first value) + the packed value stream; a thread
// picks its row's eight coefficients by rank: value index = offset + popcount(mask below the position's zigzag bit).
template <bool SPARSE>
__global__ __launch_bounds__(256) void k_jpeg_idct(const short *__restrict__ coef, const unsigned long long *__restrict__ masks,
                                                    const unsigned *__restrict__ offsets, const short *__restrict__ values,
                                                    const unsigned short *__restrict__ quant, JpegGeom g, u8 *__restrict__ planes)
{
    __shared__ int ws[32][8][9];
    const int tid = threadIdx.x, b = tid >> 3, r = tid & 7;
    const long blk = (long)blockIdx.x * 32 + b;
    const bool live = blk < g.blk_start[g.ncomp];
    int c = 0;
    if (live) { if (blk >= g.blk_start[1]) c = 1; if (blk >= g.blk_start[2]) c = 2; }
    const long lb = blk - g.blk_start[c];                       // block index within the component
    if (live) {
        const int4 qr = *(const int4 *)(quant + c * 64 + r * 8);
        const int qw[4] = {qr.x, qr.y, qr.z, qr.w};
        if (SPARSE) {
            const long gb = g.coef_off[c] / 64 + lb;            // block index over all components
            const unsigned long long mask = masks[gb];
            const short *v = values + offsets[gb];
#pragma unroll
            for (int i = 0; i < 8; i++) {
                const int z = kZigzagOf[r * 8 + i];
                const int q = (i & 1) ? (int)((unsigned)qw[i >> 1] >> 16) : (int)(qw[i >> 1] & 0xffff);
                int val = 0;
                if ((mask >> z) & 1) val = v[__popcll(mask & ((1ull << z) - 1))];
                ws[b][r][i] = val * q;
            }
        } else {
            const int4 raw = *(const int4 *)(coef + g.coef_off[c] + lb * 64 + r * 8);
            const int cw[4] = {raw.x, raw.y, raw.z, raw.w};
#pragma unroll
            for (int i = 0; i < 4; i++) {
                ws[b][r][2 * i] = (int)(short)(cw[i] & 0xffff) * (int)(qw[i] & 0xffff);
                ws[b][r][2 * i + 1] = (cw[i] >> 16) * (int)((unsigned)qw[i] >> 16);
            }
        }
    }
    __syncthreads();
    int v[8], o[8];
    if (live) {                                                 // pass 1: column r, results scaled up by 2^PASS1_BITS
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = ws[b][i][r];
        idct8(v, o);
#pragma unroll
        for (int i = 0; i < 8; i++) ws[b][i][r] = descale(o[i], 11);
    }
    __syncthreads();
    if (live) {                                                 // pass 2: row r
#pragma unroll
        for (int i = 0; i < 8; i++) v[i] = ws[b][r][i];
        idct8(v, o);
        unsigned lo = 0, hi = 0;
#pragma unroll
        for (int i = 0; i < 4; i++) {
            lo |= range_limit(descale(o[i], 18)) << (8 * i);
            hi |= range_limit(descale(o[4 + i], 18)) << (8 * i);
        }
        const int bx = (int)(lb % g.bw[c]), by = (int)(lb / g.bw[c]);
        const long pw = (long)g.bw[c] * 8;
        *(uint2 *)(planes + g.plane_off[c] + ((long)by * 8 + r) * pw + bx * 8) = make_uint2(lo, hi);
    }
}

__device__ __forceinline__ int chroma_sample(const u8 *__restrict__ p, long pw, int dw, int dh, int hmax, int vmax, int x, int y)
{
    if (hmax == 1) return p[(long)y * pw + x];
    const int cx = x >> 1;
    if (vmax == 1) {                                            // h2v1
        const u8 *row = p + (long)y * pw;
        if (dw <= 2) return row[cx];
        const int cur = row[cx];
        if (x & 1) return cx == dw - 1 ? cur : (3 * cur + row[cx + 1] + 2) >> 2;
        return cx == 0 ? cur : (3 * cur + row[cx - 1] + 1) >> 2;
    }
    const int cy = y >> 1;                                      // h2v2
    const u8 *near = p + (long)cy * pw;
    if (dw <= 2) return near[cx];
    int fy = (y & 1) ? cy + 1 : cy - 1;
    fy = fy < 0 ? 0 : fy > dh - 1 ? dh - 1 : fy;
    const u8 *far = p + (long)fy * pw;
    const int cur = 3 * near[cx] + far[cx];
    if (x & 1) return cx == dw - 1 ? (4 * cur + 7) >> 4 : (3 * cur + 3 * near[cx + 1] + far[cx + 1] + 7) >> 4;
    return cx == 0 ? (4 * cur + 8) >> 4 : (3 * cur + 3 * near[cx - 1] + far[cx - 1] + 8) >> 4;
}

__device__ __forceinline__ int clamp255(int v) { return v < 0 ? 0 : v > 255 ? 255 : v; }

__global__ __launch_bounds__(256) void k_jpeg_colour(const u8 *__restrict__ planes, JpegGeom g, u8 *__restrict__ bgr, long pitch)
{
    const int ox = blockIdx.x * 256 + threadIdx.x, oy = blockIdx.y;
    if (ox >= g.OW) return;
    int x, y;                                                   // source position of this output pixel
    switch (g.orientation) {
    case 2: x = g.W - 1 - ox; y = oy; break;
    case 3: x = g.W - 1 - ox; y = g.H - 1 - oy; break;
    case 4: x = ox; y = g.H - 1 - oy; break;
    case 5: x = oy; y = ox; break;
    case 6: x = oy; y = g.H - 1 - ox; break;
    case 7: x = g.W - 1 - oy; y = g.H - 1 - ox; break;
    case 8: x = g.W - 1 - oy; y = ox; break;
    default: x = ox; y = oy;
    }
    const int Y = planes[g.plane_off[0] + (long)y * g.bw[0] * 8 + x];
    int r = Y, gr = Y, b = Y;
    if (g.ncomp == 3) {
        const int dw = (g.W + g.hmax - 1) / g.hmax, dh = (g.H + g.vmax - 1) / g.vmax;
        const int cb = chroma_sample(planes + g.plane_off[1], (long)g.bw[1] * 8, dw, dh, g.hmax, g.vmax, x, y) - 128;
        const int cr = chroma_sample(planes + g.plane_off[2], (long)g.bw[2] * 8, dw, dh, g.hmax, g.vmax, x, y) - 128;
        r = clamp255(Y + ((91881 * cr + 32768) >> 16));         // jdcolor.c: FIX(1.40200), FIX(1.77200), FIX(0.34414), FIX(0.71414)
        b = clamp255(Y + ((116130 * cb + 32768) >> 16));
        gr = clamp255(Y + ((-22554 * cb + 32768 - 46802 * cr) >> 16));
    }
    u8 *px = bgr + (long)oy * pitch + 3L * ox;
    px[0] = (u8)b; px[1] = (u8)gr; px[2] = (u8)r;
}

}  // namespace

int svk_jpeg_reconstruct(sv_ctx *ctx, const sv_jpeg_info *info, const int16_t *coef, const uint64_t *masks, const uint32_t *offsets, const int16_t *values,
                         const uint16_t *quant, u8 *bgr, ptrdiff_t pitch, hipStream_t s)
{
    JpegGeom g;
    g.ncomp = info->components; g.W = info->width; g.H = info->height; g.OW = info->out_width; g.OH = info->out_height;
    g.hmax = info->h_samp; g.vmax = info->v_samp; g.orientation = info->orientation;
    const int mcu_cols = (g.W + 8 * g.hmax - 1) / (8 * g.hmax), mcu_rows = (g.H + 8 * g.vmax - 1) / (8 * g.vmax);
    long coff = 0, poff = 0, blocks = 0;
    for (int c = 0; c < 3; c++) {
        const int h = c == 0 ? g.hmax : 1, v = c == 0 ? g.vmax : 1;
        g.bw[c] = c < g.ncomp ? mcu_cols * h : 0;
        g.bh[c] = c < g.ncomp ? mcu_rows * v : 0;
        g.coef_off[c] = coff; g.plane_off[c] = poff; g.blk_start[c] = blocks;
        const long nb = (long)g.bw[c] * g.bh[c];
        coff += nb * 64; poff += nb * 64; blocks += nb;
    }
    g.blk_start[3] = blocks;
    if (g.ncomp == 1) g.blk_start[1] = g.blk_start[2] = blocks;   // the component pick in the kernel compares against these
    if (coff != info->coef_count) return sv_fail(SV_ERR_BAD_ARG, "sv_jpeg_reconstruct_bgr_u8: coef_count %ld does not match the geometry (%ld)", info->coef_count, coff);
    if ((size_t)poff > ctx->cap_jpeg) {
        SV_HIP(hipSetDevice(ctx->device));
        if (ctx->jpeg_planes) SV_HIP(hipFree(ctx->jpeg_planes));
        ctx->jpeg_planes = nullptr; ctx->cap_jpeg = 0;
        SV_HIP(hipMalloc((void **)&ctx->jpeg_planes, (size_t)poff));
        ctx->cap_jpeg = (size_t)poff;
    }
    if (coef)
        hipLaunchKernelGGL(k_jpeg_idct<false>, dim3((unsigned)((blocks + 31) / 32)), dim3(256), 0, s, (const short *)coef, nullptr, nullptr, nullptr, quant, g, ctx->jpeg_planes);
    else
        hipLaunchKernelGGL(k_jpeg_idct<true>, dim3((unsigned)((blocks + 31) / 32)), dim3(256), 0, s, nullptr, (const unsigned long long *)masks, offsets, (const short *)values, quant, g,
                           ctx->jpeg_planes);
    SV_LAUNCH_CHECK("k_jpeg_idct");
    hipLaunchKernelGGL(k_jpeg_colour, dim3((unsigned)((g.OW + 255) / 256), (unsigned)g.OH), dim3(256), 0, s, ctx->jpeg_planes, g, bgr, (long)pitch);
    SV_LAUNCH_CHECK("k_jpeg_colour");
    return SV_OK;
}
