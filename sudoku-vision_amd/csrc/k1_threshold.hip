// K1 -- cv/preprocess.py on MI355X.
//
//   k_preprocess_any : preprocess_for_grid_detection (preprocess.py:57-65) in one pass:
//                        BGR -> gray -> 5x5 integer Gaussian (REFLECT_101) -> 11x11 f32 Gaussian mean
//                        (REPLICATE) -> round-half-even -> src - mean <= -2 ? 255 : 0.
//                        Reads 3 B/px, writes 1 B/px; every intermediate lives in LDS.
//   k_gray / k_blur / k_adaptive_threshold : the three stages as stand-alone calls (grayscale(),
//                        blur(), threshold() of preprocess.py) for drop-in use; direct form.
//
// The f32 Gaussian follows OpenCV's FilterEngine order with the fused multiply-adds its AVX2/FMA3 and
// NEON vector code executes (row: s = x0*k0, then fma(x_j, k_j, s) left to right; column: centre*k0,
// then fma(below + above, k_j, s) outward), so the output is bit-identical to the CPU oracle.
#include "sv_device.h"
#include "sv_internal.h"
#include "k2_cells_body.h"

namespace {

constexpr int TW = 64, TH = 32;            // output tile
constexpr int GW = TW + 14, GH = TH + 14;  // gray tile      (halo 2 + 5)
constexpr int BW = TW + 10, BH = TH + 10;  // blurred tile   (halo 5)

struct Taps11 { float k[11]; };

__global__ __launch_bounds__(256) void k_preprocess_any(const u8 *__restrict__ bgr, int H, int W, ptrdiff_t pitch,
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
        for (int j = 1; j < 11; j++) acc = __builtin_fmaf(bl[br][x + j], taps.k[j], acc);
        rw[br][x] = acc;
    }
    __syncthreads();

    // 5. f32 column pass (centre, then pairs), round half to even, compare.
    for (int i = tid; i < TH * TW; i += 256) {
        const int y = i / TW, x = i - y * TW;
        if (x0 + x >= W || y0 + y >= H) continue;
        float acc = __fmul_rn(taps.k[5], rw[y + 5][x]);
#pragma unroll
        for (int j = 1; j <= 5; j++) acc = __builtin_fmaf(__fadd_rn(rw[y + 5 + j][x], rw[y + 5 - j][x]), taps.k[5 + j], acc);
        const int mean = sv_clamp(__float2int_rn(acc), 0, 255);
        const int src = (int)bl[y + 5][x + 5];
        dst[(ptrdiff_t)(y0 + y) * W + (x0 + x)] = (src - mean <= -2) ? 255 : 0;
    }
}


typedef unsigned int u32;
struct __attribute__((packed, aligned(4))) u32x3 { u32 a, b, c; };
typedef float f32x4_t __attribute__((ext_vector_type(4)));

// ---------------------------------------------------------------------------------------------------
// The LDS-tiled kernel: any width/alignment (the marching kernel below needs W % 4 == 0).  One workgroup
// per 64 x 64 output tile, every stage on 4-pixel groups; column index space of the LDS arrays starts at
// x0-8, rows at y0-7.  gfx950 issues f32
// add/mul/fma at full rate but VOP3 integer and conversion instructions at about half of it
// (profiles/r01_ubench_valu_rates.txt), and every integer stage here is exact in f32 (all values
// < 2^24), so after the byte -> float conversion everything runs on the f32 pipe:
//   AB  gray = floor(fma chain with coefficients/2^15 + 0.5)            (== (.. + 2^14) >> 15)
//       horizontal 1-4-6-4-1 in the same pass: a wave owns whole rows (3 rows x 20 groups = 60 lanes), the
//       two neighbouring pixels on either side come from the adjacent lanes by DPP wave shifts
//   C   vertical 1-4-6-4-1, floor(fma(v, 1/256, 0.5))                   (== (v + 128) >> 8)
//   D/E as in the integer variant; the compare is  rint(mean) - src >= 2.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float lane_prev(float v)   // value held by lane-1
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x138 /*wave_shr:1*/, 0xf, 0xf, true));
}
__device__ __forceinline__ float lane_next(float v)   // value held by lane+1
{
    return __int_as_float(__builtin_amdgcn_update_dpp(0, __float_as_int(v), 0x130 /*wave_shl:1*/, 0xf, 0xf, true));
}
__device__ __forceinline__ u32 lane_next_u32(u32 v) { return __float_as_uint(lane_next(__uint_as_float(v))); }
// (value of `a` held by lane-1 / lane+1) + b, in one instruction (lanes past the wave's ends read 0, as lane_prev / lane_next)
__device__ __forceinline__ float add_prev(float a, float b)
{
    float r;
    asm("v_add_f32_dpp %0, %1, %2 wave_shr:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}
__device__ __forceinline__ float add_next(float a, float b)
{
    float r;
    asm("v_add_f32_dpp %0, %1, %2 wave_shl:1 row_mask:0xf bank_mask:0xf bound_ctrl:0" : "=v"(r) : "v"(a), "v"(b));
    return r;
}

__device__ __forceinline__ float gray_f32(float b, float g, float r)
{
    return floorf(__builtin_fmaf(b, 3735.f / 32768.f, __builtin_fmaf(g, 19235.f / 32768.f, __builtin_fmaf(r, 9798.f / 32768.f, 0.5f))));
}
__device__ __forceinline__ float blur5(float a, float b, float c, float d, float e)   // a + 4b + 6c + 4d + e, exact
{
    return __builtin_fmaf(6.f, c, __builtin_fmaf(4.f, b + d, a + e));
}

template <int TW, int TH, int NT, int RC, int RE>
__global__ __launch_bounds__(NT) void k_preprocess_f32(const u8 *__restrict__ bgr, int H, int W, ptrdiff_t pitch,
                                                       ptrdiff_t img_stride, u8 *__restrict__ out, Taps11 taps, int aligned4)
{
    constexpr int GW = TW + 16, GH = TH + 14, BH = TH + 10, NG = GW / 4, NK = TW / 4, NWAVE = NT / 64;
    constexpr int ROWS_PER_WAVE = 64 / NG;                    // 3 rows of 20 groups, 4 lanes idle
    static_assert(BH * TW <= GH * GW, "rw must fit in hf");
    __shared__ __attribute__((aligned(16))) float smem[GH * GW + BH * GW];
    f32x4_t *hf4 = (f32x4_t *)smem;                           // [GH][NG]  horizontal pass (x 16 scale kept)
    f32x4_t *rw4 = (f32x4_t *)smem;                           // [BH][NK]  f32 row pass (after hf is dead)
    f32x4_t *bf4 = (f32x4_t *)(smem + GH * GW);               // [BH][NG]  blurred image as f32
    float *bf = smem + GH * GW;

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int x0 = blockIdx.x * TW, y0 = blockIdx.y * TH;
    const u8 *img = bgr + (ptrdiff_t)blockIdx.z * img_stride;
    u8 *dst = out + (ptrdiff_t)blockIdx.z * H * W;

    // ---- AB: gray + horizontal pass, whole rows per wave
    {
        const int rl = lane / NG, g = lane - rl * NG;
        const int gx = x0 - 8 + 4 * g;
        const bool fast = aligned4 && gx >= 0 && gx + 3 < W;
        for (int base = 0; base < GH; base += NWAVE * ROWS_PER_WAVE) {
            const int ly = base + wave * ROWS_PER_WAVE + rl;
            const int lyc = ly < GH ? ly : GH - 1;
            const int sy = sv_reflect101(sv_clamp(y0 - 7 + lyc, -2, H + 1), H);
            const u8 *row = img + (ptrdiff_t)sy * pitch;
            float p0, p1, p2, p3;
            if (fast) {
                const u32x3 d = *(const u32x3 *)(row + 3 * gx);
                p0 = gray_f32((float)(d.a & 255), (float)((d.a >> 8) & 255), (float)((d.a >> 16) & 255));
                p1 = gray_f32((float)(d.a >> 24), (float)(d.b & 255), (float)((d.b >> 8) & 255));
                p2 = gray_f32((float)((d.b >> 16) & 255), (float)(d.b >> 24), (float)(d.c & 255));
                p3 = gray_f32((float)((d.c >> 8) & 255), (float)((d.c >> 16) & 255), (float)(d.c >> 24));
            } else {
                float q[4];
                for (int i = 0; i < 4; i++) {
                    const int sx = sv_reflect101(sv_clamp(gx + i, -2, W + 1), W);
                    const u8 *p = row + (ptrdiff_t)sx * 3;
                    q[i] = gray_f32((float)p[0], (float)p[1], (float)p[2]);
                }
                p0 = q[0]; p1 = q[1]; p2 = q[2]; p3 = q[3];
            }
            // neighbours across the group boundary (garbage only in columns 0,1 and GW-2,GW-1, which nobody reads)
            const float l2 = lane_prev(p2), l3 = lane_prev(p3), r0 = lane_next(p0), r1 = lane_next(p1);
            f32x4_t h;
            h[0] = blur5(l2, l3, p0, p1, p2);
            h[1] = blur5(l3, p0, p1, p2, p3);
            h[2] = blur5(p0, p1, p2, p3, r0);
            h[3] = blur5(p1, p2, p3, r0, r1);
            if (lane < ROWS_PER_WAVE * NG && ly < GH) hf4[ly * NG + g] = h;
        }
    }
    __syncthreads();

    // ---- C: vertical pass, sliding window down the column
    {
        constexpr int NCH = (BH + RC - 1) / RC;
        for (int it = tid; it < NCH * NG; it += NT) {
            const int ch = it / NG, g = it - ch * NG;
            const int br0 = ch * RC;
            f32x4_t win[RC + 4];
#pragma unroll
            for (int r = 0; r < RC + 4; r++) win[r] = hf4[(br0 + r < GH ? br0 + r : GH - 1) * NG + g];
#pragma unroll
            for (int r = 0; r < RC; r++) {
                if (br0 + r < BH) {
                    f32x4_t o;
#pragma unroll
                    for (int c = 0; c < 4; c++)
                        o[c] = floorf(__builtin_fmaf(blur5(win[r][c], win[r + 1][c], win[r + 2][c], win[r + 3][c], win[r + 4][c]), 1.f / 256.f, 0.5f));
                    bf4[(br0 + r) * NG + g] = o;
                }
            }
        }
    }
    __syncthreads();

    // ---- edge tiles: blurred halo outside the image = REPLICATE of the blurred image
    if (x0 < 5 || x0 + TW + 5 > W || y0 < 5 || y0 + TH + 5 > H) {
        for (int it = tid; it < BH * GW; it += NT) {
            const int br = it / GW, c = it - br * GW;
            const int q = x0 - 8 + c;
            if (q < 0 || q >= W) {
                const int sc = sv_clamp(q, 0, W - 1) - (x0 - 8);
                if (sc >= 0 && sc < GW) bf[br * GW + c] = bf[br * GW + sc];
            }
        }
        __syncthreads();
        for (int it = tid; it < BH * NG; it += NT) {
            const int br = it / NG, g = it - br * NG;
            const int r = y0 - 5 + br;
            if (r < 0 || r >= H) {
                const int sr = sv_clamp(r, 0, H - 1) - (y0 - 5);
                if (sr >= 0 && sr < BH) bf4[it] = bf4[sr * NG + g];
            }
        }
        __syncthreads();
    }

    // ---- D: f32 row pass.  Outputs x = 4k..4k+3 read blurred columns 4k+3 .. 4k+16 (array index space)
    for (int it = tid; it < BH * NK; it += NT) {
        const int br = it / NK, k = it - br * NK;
        const f32x4_t *src = bf4 + br * NG + k;
        const f32x4_t a0 = src[0], a1 = src[1], a2 = src[2], a3 = src[3], a4 = src[4];
        const float v[14] = {a0[3], a1[0], a1[1], a1[2], a1[3], a2[0], a2[1], a2[2], a2[3], a3[0], a3[1], a3[2], a3[3], a4[0]};
        f32x4_t o;
#pragma unroll
        for (int c = 0; c < 4; c++) {
            float acc = __fmul_rn(taps.k[0], v[c]);
#pragma unroll
            for (int j = 1; j < 11; j++) acc = __builtin_fmaf(v[c + j], taps.k[j], acc);
            o[c] = acc;
        }
        rw4[it] = o;       // rw aliases hf: hf's last reader (stage C) is two barriers behind
    }
    __syncthreads();

    // ---- E: f32 column pass, round, compare, store
    {
        constexpr int NCH = (TH + RE - 1) / RE;
        for (int it = tid; it < NCH * NK; it += NT) {
            const int ch = it / NK, k = it - ch * NK;
            const int yb = ch * RE;
            f32x4_t win[RE + 10];
#pragma unroll
            for (int r = 0; r < RE + 10; r++) win[r] = rw4[(yb + r < BH ? yb + r : BH - 1) * NK + k];
#pragma unroll
            for (int r = 0; r < RE; r++) {
                const int y = yb + r;
                if (y >= TH || y0 + y >= H) continue;
                const f32x4_t srcv = bf4[(y + 5) * NG + k + 2];
                u32 o = 0;
#pragma unroll
                for (int c = 0; c < 4; c++) {
                    // (FilterEngine adds delta = +0.f first; every operand is >= +0, so the sum is unchanged)
                    float acc = __fmul_rn(taps.k[5], win[r + 5][c]);
#pragma unroll
                    for (int j = 1; j <= 5; j++) acc = __builtin_fmaf(__fadd_rn(win[r + 5 + j][c], win[r + 5 - j][c]), taps.k[5 + j], acc);
                    // mean = rint(acc) lies in [0,255] without clamping (acc <= 255*(1+1e-8)); src - mean <= -2
                    o |= (__fsub_rn(rintf(acc), srcv[c]) >= 2.f ? 255u : 0u) << (8 * c);
                }
                const int gx = x0 + 4 * k;
                u8 *d = dst + (ptrdiff_t)(y0 + y) * W + gx;
                if (aligned4 && gx + 3 < W) *(u32 *)d = o;
                else
                    for (int c = 0; c < 4; c++)
                        if (gx + c < W) d[c] = (u8)(o >> (8 * c));
            }
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// The marching variant: no LDS tiles, no barriers.  One WAVE owns a 256-pixel-wide strip (64 lanes x 4
// pixels; the outer 2 lanes on each side are halo, so 240 output columns) and marches down a band of
// rows.  Per input row it loads 12 B per lane, converts to gray, does the horizontal 1-4-6-4-1 with DPP
// neighbour exchange, keeps the last five h-rows in registers for the vertical 1-4-6-4-1, runs the 11-tap
// f32 row pass on the fresh blurred row (again DPP), and pushes the result into an 11-row register ring
// from which the f32 column pass + threshold of the row five pushes back is emitted.  The only memory
// besides the frame and the output is a per-wave LDS delay line that returns the blurred row (the
// threshold's `src`) five rows later.  Vertical REPLICATE = pushing the first/last row several times.
// ---------------------------------------------------------------------------------------------------
__device__ __forceinline__ float pick4(const f32x4_t &v, int e) { return e == 0 ? v[0] : e == 1 ? v[1] : e == 2 ? v[2] : v[3]; }

constexpr int MARCH_STRIP = 240;   // output columns per wave

// Requires W % 4 == 0 and 4-byte aligned rows (aligned4); other shapes use the tiled kernel.
// Straight-line per-row body: no memory operation sits under a divergent or data-dependent branch, so the
// compiler can keep PF row loads in flight (counted s_waitcnt vmcnt) instead of draining the queue.
// BITS: the binary as 1 bit per pixel (LSB = leftmost, W/32 words per row; needs W % 32 == 0) -- what the host corner search reads in the
// end-to-end pipeline, where nothing else consumes the byte image: 8x fewer store bytes here and a despeckle pass that reads 66 MB per 256
// frames instead of 531.  A lane's 4 pixels are a nibble; 4 lanes' nibbles are gathered by lane shifts into 16-bit stores (a strip of 240
// pixels starts on a 16-bit boundary).
// One (frame, band, strip) item = the work of one wave; sdl = that wave's delay line of blurred rows in LDS.
// ---------------------------------------------------------------------------------------------------
// The marching item, on packed f32 math since round 3 (same data flow and arithmetic as the scalar-f32 body of rounds 1-2, bit-identical
// output, git history).  K1 is VALU-issue bound and on gfx950 a v_pk_{fma,add,mul}_f32 does two f32 operations in 2.0 ns of a SIMD's
// issue time where two all-VGPR v_fma_f32 take 2.7 (tools/ubench_pk.hip, 4 waves per SIMD), so every stage whose work is elementwise
// over a lane's 4 pixels runs on pixel PAIRS (the gray conversion, the vertical 1-4-6-4-1, its normalisation, the f32 column pass):
// half the instructions.  The 11-tap f32 row pass is not elementwise -- pixel i needs v[i..i+10] of the 14 values around it, and a
// packed operand must be an even-aligned register pair -- but its two lanes need not be at the same tap: with pixel 1 one tap BEHIND
// pixel 0, both lanes of step j read the SAME value v[j] (a broadcast, free through op_sel) against the coefficient pair
// (k_j, k_{j-1}): 12 packed steps give two pixels' 11-tap sums, each in cv2's order (lane 1's first step is fma(v1, k0, 0) = the
// rounded product, its idle first and lane 0's idle last step multiply by 0: exact, all values are >= +0).  The horizontal
// 1-4-6-4-1 stays scalar (its packed form costs more than it saves).  The threshold compare shifts its result straight into the
// output nibble (v_cmp + v_addc: 2 instructions per pixel instead of 3-4), against src + 2 kept in the delay line.
// ---------------------------------------------------------------------------------------------------
typedef float f32x2_t __attribute__((ext_vector_type(2)));
struct RowPairs { f32x2_t p[12]; };      // p[j] = (k_j, k_{j-1}), k_{-1} = k_11 = 0

__device__ __forceinline__ f32x2_t pkfma(f32x2_t a, f32x2_t b, f32x2_t c) { return __builtin_elementwise_fma(a, b, c); }
__device__ __forceinline__ f32x2_t bc2(float v) { return (f32x2_t){v, v}; }

template <bool BITS>
__device__ __forceinline__ void march_item(const u8 *__restrict__ bgr, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride, u8 *__restrict__ out,
                                              const Taps11 &taps, const RowPairs &rp, int TH, int nstrips, int nbands, int item, f32x4_t (*sdl)[64])
{
    const int lane = threadIdx.x & 63;
    const int strip = item % nstrips, band = (item / nstrips) % nbands, frame = item / (nstrips * nbands);
    const u8 *img = bgr + (ptrdiff_t)frame * img_stride;
    u8 *dst = out + (ptrdiff_t)frame * H * (BITS ? W >> 3 : W);

    const int xs0 = strip * MARCH_STRIP - 8;
    const int cx0 = xs0 + 4 * lane;
    const int ldx = 3 * sv_clamp(cx0, 0, W - 4);             // every lane loads an in-bounds, aligned pixel group
    const bool edge_l = xs0 < 0, edge_r = xs0 + 256 > W;    // wave-uniform
    const int yb = band * TH, ye = (yb + TH < H) ? yb + TH : H;
    const int qs = sv_clamp(yb - 5, 0, H - 1);              // first blurred row this wave produces
    const int N = (ye + 4) - qs + 1;                        // pushes: rows qs .. ye+4 (rows > H-1 are REPLICATE repeats of H-1)
    const int emit_from = yb + 5 - qs;                      // push index whose window centre is output row yb
    const int srcl_hi = (W - 1 - xs0) >> 2;                 // lane holding column W-1 (element 3 since W % 4 == 0)
    const bool store_lane = lane >= 2 && lane < 62 && cx0 < W && (!BITS || ((lane - 2) & 3) == 0);

    auto load_raw = [&](int gy) -> u32x3 { return *(const u32x3 *)(img + (ptrdiff_t)sv_reflect101(gy, H) * pitch + ldx); };
    // gray (pixel pairs) + horizontal 1-4-6-4-1 (scalar) -> 4 un-normalised h values
    auto hrow = [&](const u32x3 &d) -> f32x4_t {
        const f32x2_t cb = bc2(3735.f / 32768.f), cg = bc2(19235.f / 32768.f), cr = bc2(9798.f / 32768.f), half = bc2(0.5f);
        const f32x2_t b01 = {(float)(d.a & 255), (float)(d.a >> 24)}, g01 = {(float)((d.a >> 8) & 255), (float)(d.b & 255)},
                      r01 = {(float)((d.a >> 16) & 255), (float)((d.b >> 8) & 255)};
        const f32x2_t b23 = {(float)((d.b >> 16) & 255), (float)((d.c >> 8) & 255)}, g23 = {(float)(d.b >> 24), (float)((d.c >> 16) & 255)},
                      r23 = {(float)(d.c & 255), (float)(d.c >> 24)};
        const f32x2_t y01 = __builtin_elementwise_floor(pkfma(b01, cb, pkfma(g01, cg, pkfma(r01, cr, half))));
        const f32x2_t y23 = __builtin_elementwise_floor(pkfma(b23, cb, pkfma(g23, cg, pkfma(r23, cr, half))));
        float p0 = y01[0], p1 = y01[1], p2 = y23[0], p3 = y23[1];
        if (edge_l) {            // REFLECT_101: columns -2,-1 are columns 2,1 (held by the next lane)
            const float n1 = lane_next(p1), n2 = lane_next(p2);
            if (cx0 == -4) { p2 = n2; p3 = n1; }
        }
        if (edge_r) {            // columns W, W+1 are columns W-2, W-3 (held by the previous lane)
            const float q1 = lane_prev(p1), q2 = lane_prev(p2);
            if (cx0 == W) { p0 = q2; p1 = q1; }
        }
        // the neighbour lanes' pixels enter as DPP operands of the sums they belong to (six v_add_f32_dpp; as values of their own, the two that
        // are used twice, l3 and r0, each cost a v_mov_b32_dpp on top)
        f32x4_t h;
        h[0] = __builtin_fmaf(6.f, p0, __builtin_fmaf(4.f, add_prev(p3, p1), add_prev(p2, p2)));      // blur5(l2, l3, p0, p1, p2)
        h[1] = __builtin_fmaf(6.f, p1, __builtin_fmaf(4.f, p0 + p2, add_prev(p3, p3)));               // blur5(l3, p0, p1, p2, p3)
        h[2] = __builtin_fmaf(6.f, p2, __builtin_fmaf(4.f, p1 + p3, add_next(p0, p0)));               // blur5(p0, p1, p2, p3, r0)
        h[3] = __builtin_fmaf(6.f, p3, __builtin_fmaf(4.f, add_next(p0, p2), add_next(p1, p1)));      // blur5(p1, p2, p3, r0, r1)
        return h;
    };
    // blurred row (as f32) from the 5-row window, with the horizontal REPLICATE of the blurred image, and its f32 row pass
    auto finish_row = [&](const f32x4_t &a, const f32x4_t &b, const f32x4_t &c, const f32x4_t &d, const f32x4_t &e, f32x4_t &bfv, f32x4_t &rwv) {
        const f32x2_t four = bc2(4.f), six = bc2(6.f), inv = bc2(1.f / 256.f), half = bc2(0.5f);
        const f32x2_t s01 = pkfma(six, c.xy, pkfma(four, b.xy + d.xy, a.xy + e.xy)), s23 = pkfma(six, c.zw, pkfma(four, b.zw + d.zw, a.zw + e.zw));
        const f32x2_t n01 = __builtin_elementwise_floor(pkfma(s01, inv, half)), n23 = __builtin_elementwise_floor(pkfma(s23, inv, half));
        bfv = (f32x4_t){n01[0], n01[1], n23[0], n23[1]};
        if (edge_l) {
            const float v0 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bfv[0]), 2));   // column 0 = lane 2, element 0
            if (cx0 < 0) { bfv[0] = v0; bfv[1] = v0; bfv[2] = v0; bfv[3] = v0; }
        }
        if (edge_r) {
            const float v1 = __int_as_float(__builtin_amdgcn_readlane(__float_as_int(bfv[3]), srcl_hi));
            if (cx0 >= W) { bfv[0] = v1; bfv[1] = v1; bfv[2] = v1; bfv[3] = v1; }
        }
        f32x4_t pv, nx;
#pragma unroll
        for (int i = 0; i < 4; i++) { pv[i] = lane_prev(bfv[i]); nx[i] = lane_next(bfv[i]); }
        const float pp3 = lane_prev(pv[3]), nn0 = lane_next(nx[0]);
        const float v[14] = {pp3, pv[0], pv[1], pv[2], pv[3], bfv[0], bfv[1], bfv[2], bfv[3], nx[0], nx[1], nx[2], nx[3], nn0};
        // pixels (0,1) over v[0..11], pixels (2,3) over v[2..13]; lane 1 of a pair runs one tap behind lane 0
        f32x2_t d01 = bc2(v[0]) * rp.p[0], d23 = bc2(v[2]) * rp.p[0];
#pragma unroll
        for (int j = 1; j < 12; j++) {
            d01 = pkfma(bc2(v[j]), rp.p[j], d01);
            d23 = pkfma(bc2(v[2 + j]), rp.p[j], d23);
        }
        rwv = (f32x4_t){d01[0], d01[1], d23[0], d23[1]};
    };

    constexpr int PF = 4, NPH = 12;
    f32x4_t hw[6];
    hw[2] = hrow(load_raw(qs - 2)); hw[3] = hrow(load_raw(qs - 1)); hw[4] = hrow(load_raw(qs)); hw[5] = hrow(load_raw(qs + 1));
    hw[0] = hw[2]; hw[1] = hw[2];
    u32x3 raw[PF];
#pragma unroll
    for (int i = 0; i < PF; i++) raw[i] = load_raw(qs + 2 + i);

    f32x4_t cur_bf = {0.f, 0.f, 0.f, 0.f};
    f32x4_t win[NPH];
#pragma unroll
    for (int k = 0; k < NPH; k++) win[k] = (f32x4_t){0.f, 0.f, 0.f, 0.f};

    for (int base = 0; base < N; base += NPH) {
#pragma unroll
        for (int ph = 0; ph < NPH; ph++) {
            const int i = base + ph;
            if (i < N) {
                const int q = qs + i;                            // blurred row of this push (rows > H-1 repeat row H-1)
                const u32x3 raw_cur = raw[ph % PF];
                raw[ph % PF] = load_raw(q + 2 + PF < H + 2 ? q + 2 + PF : H + 1);   // row that push i + PF will consume
                if (q <= H - 1) {                                // wave-uniform; bottom REPLICATE pushes the last blurred row again
                    hw[ph % 6] = hrow(raw_cur);                  // h row q+2; rows q-2 .. q+1 sit in the 4 slots before it
                    finish_row(hw[(ph + 2) % 6], hw[(ph + 3) % 6], hw[(ph + 4) % 6], hw[(ph + 5) % 6], hw[ph % 6], cur_bf, win[ph]);
                } else {
                    win[ph] = win[(ph + NPH - 1) % NPH];
                }
                if (i == 0) {
#pragma unroll
                    for (int k = 1; k < NPH; k++) win[k] = win[0];   // top REPLICATE: rows above the first one read the first one
                }
                // the threshold compares rint(mean) with src + 2; both sides carry 2^23, which makes the rint an f32 add (round to
                // nearest even at integer granularity) on the packed pipe
                sdl[i & 7][lane] = cur_bf + 8388610.f;
                if (i >= emit_from) {                            // emit output row yo; its window centre was pushed 5 pushes ago
                    const int yo = q - 5;
                    const f32x4_t thr = sdl[(i - 5) & 7][lane];
                    const f32x4_t &wc = win[(ph + NPH - 5) % NPH];
                    f32x2_t a01 = bc2(taps.k[5]) * wc.xy, a23 = bc2(taps.k[5]) * wc.zw;
#pragma unroll
                    for (int j = 1; j <= 5; j++) {
                        const f32x4_t &wp = win[(ph + NPH - 5 + j) % NPH], &wm = win[(ph + 2 * NPH - 5 - j) % NPH];
                        a01 = pkfma(wp.xy + wm.xy, bc2(taps.k[5 + j]), a01);
                        a23 = pkfma(wp.zw + wm.zw, bc2(taps.k[5 + j]), a23);
                    }
                    const f32x2_t r01 = a01 + bc2(8388608.f), r23 = a23 + bc2(8388608.f);      // 2^23 + rint(mean), exactly
                    const float m0 = r01[0], m1 = r01[1], m2 = r23[0], m3 = r23[1];
                    // src - mean <= -2  <=>  rint(mean) >= src + 2: each compare shifts its bit into the nibble (pixel 3 first)
                    u32 o = 0;
                    asm("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(o) : "v"(m3), "v"(thr[3]) : "vcc");
                    asm("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(o) : "v"(m2), "v"(thr[2]) : "vcc");
                    asm("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(o) : "v"(m1), "v"(thr[1]) : "vcc");
                    asm("v_cmp_ge_f32 vcc, %1, %2\n\tv_addc_co_u32 %0, vcc, %0, %0, vcc" : "+v"(o) : "v"(m0), "v"(thr[0]) : "vcc");
                    if (BITS) {
                        const u32 a = o | (lane_next_u32(o) << 4);                     // lanes l, l+1
                        const u32 w16 = a | (lane_next_u32(lane_next_u32(a)) << 8);    // lanes l .. l+3
                        if (store_lane) *(unsigned short *)(dst + (ptrdiff_t)yo * (W >> 3) + (cx0 >> 3)) = (unsigned short)w16;
                    } else if (store_lane) {
                        *(u32 *)(dst + (ptrdiff_t)yo * W + cx0) = ((o * 0x00204081u) & 0x01010101u) * 255u;   // nibble -> four 0/255 bytes
                    }
                }
            }
        }
    }
}

template <bool BITS>
__global__ __launch_bounds__(256) void k_preprocess_march(const u8 *__restrict__ bgr, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride,
                                                             u8 *__restrict__ out, Taps11 taps, RowPairs rp, int TH, int nstrips, int nbands, int nitems)
{
    __shared__ __attribute__((aligned(16))) f32x4_t sdl[4][8][64];   // per-wave delay line of blurred rows (+ 2: the compare's right-hand side)
    const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int item = blockIdx.x * 4 + wave;
    if (item >= nitems) return;
    march_item<BITS>(bgr, H, W, pitch, img_stride, out, taps, rp, TH, nstrips, nbands, item, sdl[wave]);
}

// BASELINE configs[4]'s fused threshold + warp launch for the device-only mode (corners known before K1 runs, so K1 and K2 are independent
// and both read the frame): one grid holds the marching items and the (frame, cell) workgroups of K2 (sv_k2::warp_cells_item, the very code
// k_warp_cells runs), laid out so that everything that touches frame f is dispatched together and on the same XCD (workgroup b runs on XCD
// b % 8; frame f belongs to XCD f % 8) -- whichever of the two reads a frame region first leaves it in that XCD's L2 for the other.
__global__ __launch_bounds__(256) void k_preprocess_warp_fused(const u8 *__restrict__ bgr, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride,
                                                               u8 *__restrict__ out, Taps11 taps, RowPairs rp, int TH, int nstrips, int nbands,
                                                               const double *__restrict__ minv, u8 *__restrict__ cells)
{
    __shared__ __attribute__((aligned(16))) f32x4_t sdl[4][8][64];
    __shared__ sv_k2::CellsLds cl;
    const int items = nstrips * nbands, k1_wgs = (items + 3) / 4, per_frame = k1_wgs + 81;
    const int j = blockIdx.x >> 3, frame = (j / per_frame) * 8 + (blockIdx.x & 7), within = j % per_frame;
    if (frame >= n) return;
    if (within < k1_wgs) {
        const int wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6), local = within * 4 + wave;
        if (local < items) march_item<false>(bgr, H, W, pitch, img_stride, out, taps, rp, TH, nstrips, nbands, frame * items + local, sdl[wave]);
    } else {
        sv_k2::warp_cells_item(bgr, H, W, pitch, img_stride, minv, cells, frame, within - k1_wgs, cl);
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
            acc = __builtin_fmaf(pr, t.k[r + j], acc);
        }
    } else {
        acc = __fmul_rn(t.k[0], (float)row[sv_clamp(x - r, 0, W - 1)]);
        for (int j = 1; j < t.n; j++) acc = __builtin_fmaf((float)row[sv_clamp(x + j - r, 0, W - 1)], t.k[j], acc);
    }
    return acc;
}

__global__ void k_adaptive_threshold(const u8 *__restrict__ src, int H, int W, FTaps t, int idelta, int type_inv, u8 *__restrict__ dst)
{
    const int x = blockIdx.x * blockDim.x + threadIdx.x, y = blockIdx.y;
    if (x >= W) return;
    const u8 *s = src + (ptrdiff_t)blockIdx.z * H * W;
    const int r = t.n / 2;
    float acc = __fmul_rn(t.k[r], row_value(s + (ptrdiff_t)y * W, x, W, t));
    for (int j = 1; j <= r; j++) {
        const float lo = row_value(s + (ptrdiff_t)sv_clamp(y + j, 0, H - 1) * W, x, W, t);
        const float hi = row_value(s + (ptrdiff_t)sv_clamp(y - j, 0, H - 1) * W, x, W, t);
        acc = __builtin_fmaf(__fadd_rn(lo, hi), t.k[r + j], acc);
    }
    const int mean = sv_clamp(__float2int_rn(acc), 0, 255);
    const int d = (int)s[(ptrdiff_t)y * W + x] - mean;
    const bool on = type_inv ? (d <= -idelta) : (d > -idelta);
    dst[((ptrdiff_t)blockIdx.z * H + y) * W + x] = on ? 255 : 0;
}

}  // namespace

static RowPairs row_pairs(const Taps11 &t)
{
    RowPairs rp;
    for (int j = 0; j < 12; j++) rp.p[j] = (f32x2_t){j < 11 ? t.k[j] : 0.f, j > 0 ? t.k[j - 1] : 0.f};
    return rp;
}

// the march kernel's launch shape for n frames
static void march_shape(int n, int H, int W, int &nstrips, int &nbands, int &TH)
{
    nstrips = (W + MARCH_STRIP - 1) / MARCH_STRIP;
    nbands = (12288 + n * nstrips - 1) / (n * nstrips);      // ~3 rounds of 4 waves per SIMD on 256 CUs (flat from 4k to 20k waves, tools/dev/k1_shape_sweep.py)
    if (nbands > H / 32) nbands = H / 32;
    if (nbands < 1) nbands = 1;
    TH = (H + nbands - 1) / nbands;
    nbands = (H + TH - 1) / TH;
}

int svk_preprocess_bits(sv_ctx *ctx, const u8 *bgr, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride, uint32_t *bits, hipStream_t s)
{
    if (H < 16 || W < 16 || (W & 31) || (pitch % 4) || (img_stride % 4) || ((uintptr_t)bgr % 4))
        return sv_fail(SV_ERR_UNSUPPORTED, "sv_preprocess_bits_u8: needs H, W >= 16, W %% 32 == 0 and a 4-byte aligned frame layout");
    Taps11 t;
    sv_gaussian_taps_f32(11, t.k);
    sv_time_scope ts(ctx, SVK_PREPROCESS, s);
    int nstrips, nbands, TH;
    march_shape(n, H, W, nstrips, nbands, TH);
    const int nitems = n * nstrips * nbands;
    hipLaunchKernelGGL(k_preprocess_march<true>, dim3((nitems + 3) / 4), dim3(256), 0, s, bgr, H, W, pitch, img_stride, (u8 *)bits, t, row_pairs(t), TH, nstrips, nbands, nitems);
    SV_LAUNCH_CHECK("k_preprocess_march<bits>");
    return SV_OK;
}

int svk_preprocess_warp_fused(sv_ctx *ctx, const u8 *bgr, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride, u8 *binary, const double *minv, u8 *cells,
                              hipStream_t s)
{
    if (H < 16 || W < 16 || (W & 3) || (pitch % 4) || (img_stride % 4) || ((uintptr_t)bgr % 4) || ((uintptr_t)binary % 4))
        return sv_fail(SV_ERR_UNSUPPORTED, "sv_preprocess_warp_cells_u8: needs H, W >= 16, W %% 4 == 0 and a 4-byte aligned layout");
    Taps11 t;
    sv_gaussian_taps_f32(11, t.k);
    sv_time_scope ts(ctx, SVK_FUSED12, s);
    int nstrips, nbands, TH;
    march_shape(n, H, W, nstrips, nbands, TH);
    const long per_frame = (nstrips * nbands + 3) / 4 + 81;
    const long blocks = (long)((n + 7) / 8) * per_frame * 8;
    hipLaunchKernelGGL(k_preprocess_warp_fused, dim3((unsigned)blocks), dim3(256), 0, s, bgr, n, H, W, pitch, img_stride, binary, t, row_pairs(t), TH, nstrips, nbands, minv, cells);
    SV_LAUNCH_CHECK("k_preprocess_warp_fused");
    return SV_OK;
}

int svk_preprocess(sv_ctx *ctx, const u8 *bgr, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride, u8 *binary, hipStream_t s)
{
    Taps11 t;
    sv_gaussian_taps_f32(11, t.k);
    sv_time_scope ts(ctx, SVK_PREPROCESS, s);
    if (H < 16 || W < 16) {   // tiny images: the general per-pixel variant
        dim3 grid((W + TW - 1) / TW, (H + TH - 1) / TH, n);
        hipLaunchKernelGGL(k_preprocess_any, grid, dim3(256), 0, s, bgr, H, W, pitch, img_stride, binary, t);
    } else {
        constexpr int FW = 64, FH = 64;
        const int aligned4 = (pitch % 4 == 0) && (img_stride % 4 == 0) && (W % 4 == 0) && ((uintptr_t)bgr % 4 == 0) && ((uintptr_t)binary % 4 == 0);
        dim3 grid((W + FW - 1) / FW, (H + FH - 1) / FH, n);
        if (aligned4) {
            int nstrips, nbands, TH;
            march_shape(n, H, W, nstrips, nbands, TH);
            const int nitems = n * nstrips * nbands;
            hipLaunchKernelGGL(k_preprocess_march<false>, dim3((nitems + 3) / 4), dim3(256), 0, s, bgr, H, W, pitch, img_stride, binary, t, row_pairs(t), TH, nstrips, nbands, nitems);
        } else {
            hipLaunchKernelGGL((k_preprocess_f32<64, 64, 512, 3, 2>), grid, dim3(512), 0, s, bgr, H, W, pitch, img_stride, binary, t, aligned4);
        }
    }
    SV_LAUNCH_CHECK("k_preprocess");
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
