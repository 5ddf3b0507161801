// K1 on the matrix pipe: preprocess_for_grid_detection (cv/preprocess.py:57-65) with its four separable filter passes as
// banded-Toeplitz GEMMs on v_mfma_f32_16x16x32_f16, bit-identical to k_preprocess_march (k1_threshold.hip) and the CPU oracle.
//
// STATUS: an alternative formulation, NOT the default.  It is bit-exact (the whole GPU suite passes with it) but slower than the marching
// kernel: 1.30 ms against 0.64 ms per 256 1080p frames.  Why it loses, measured (stage timestamps of one workgroup): a 64 x 128 tile needs
// an 80 x 144 gray region (1.41x the pixels: 39 % of the kernel's VALU work is gray conversion, against 15 % in the marching kernel, whose
// halo is 6 %), and five short passes separated by barriers with five or six MFMA jobs per wave each are latency, not throughput: 13 % S0,
// 8 % S1, 10 % S2, 25 % S3, 26 % S4, 12 % exact re-decision, 6 % output.  Kept (entry sv_preprocess_mm_u8 of the test-only library libsudokuvision_xcheck.so) as an independent
// second implementation of K1 for the tests, and as the record of the experiment.
//
// The idea: the marching kernel is VALU-issue bound (53.8 instructions per pixel, 0.64 ms per 256 1080p frames at 0.42 of the HBM roof),
// and 142 of its 211 instructions per 4 pixels are the 5-tap and 11-tap passes.  The f16 MFMA runs beside the VALU.
//
// How it stays exact:
//   * gray and the 5x5 integer Gaussian are integer arithmetic on values <= 255 with weights 1,4,6,4,1: every operand is exact in
//     f16 (the un-normalised horizontal sums, <= 4080, are carried as h - 2040 in [-2040, 2040]), every product and partial sum
//     exact in the MFMA's f32 accumulator -- any summation order gives the same integers.  The blurred image is exact.
//   * the 11x11 f32 Gaussian mean is NOT reproduced operation by operation (cv2's result depends on its rounding sequence).  It is
//     approximated: taps as f16 pairs (hi + lo, 22 bits), the blurred image exact in f16, the row-pass result as an f16 pair, f32
//     accumulation -- within ~1e-4 of cv2's float chain (bound below).  A pixel's output depends on the mean only through
//     "rint(mean) - src >= 2"  <=>  mean >= src + 1.5 (src = blurred pixel, an integer), so wherever the approximate mean is further
//     than EPS from src + 1.5 the decision is the exact one.  The other pixels (a few in ten thousand) go on a per-tile list in LDS and
//     are decided, before the tile is written, with the exact chain (row: s = x0*k0, fma left to right; column: centre, then
//     fma(below + above)) on the blurred values the tile already holds.
//     Error budget for EPS = 2^-9 = 1.95e-3: cv2's chain vs the real-number sum <= 22 roundings of values <= 510 -> 1.5e-4; the
//     approximation vs the real-number sum: taps 2 x 255 x 2^-22 = 1.2e-4, row-pass pair 255 x 2^-22 = 6e-5, f32 accumulation inside the
//     MFMAs (5 MFMAs of 32 terms; at worst one rounding per term) 160 x 1.5e-5 = 2.4e-4 (scaled values) -- 5.7e-4 in total, a third of
//     EPS; tests/test_gpu_parity.py::test_k1_matrix_pipe_form measures the actual maximum against the exact chain.
//   * a tile whose list overflows (more than 1/8 of its pixels ambiguous: not a photograph) decides every pixel that way.
//
// Data flow of a workgroup (512 threads, one 64 x 128 output tile, 79 KB LDS, 2 workgroups per CU).  A filter pass is
// D[owner][out] = sum_k A[owner][k] * T[k][out] with the data as the A operand (lane = owner, 8 consecutive positions along the
// filtered dimension) and the Toeplitz matrix T[k][n] = tap(k - n - 8) as B: 32 input positions -> 16 outputs.  The D layout gives a
// lane 4 consecutive OWNERS of one output position, i.e. the transposed orientation -- exactly what the next pass (which filters the
// other dimension) wants as its A operand, so every pass writes its result transposed and no pass ever shuffles:
//   S0 load + gray (VALU)          -> G   f16 [y][x]
//   S1 horizontal 1-4-6-4-1        -> H1T f16 [x][y]     (h - 2040)
//   S2 vertical   1-4-6-4-1, >> 8  -> BL  f16 [y][x]     blurred image, REPLICATE border applied in place afterwards
//   S3 horizontal 11 taps (approx) -> RT  f16 pair [x][y]
//   S4 vertical   11 taps (approx) -> decision vs BL, ambiguous pixels listed, 64 x 128 result staged in LDS, written as full rows
#include "sv_device.h"
#include "sv_internal.h"
#include <cstdlib>

typedef _Float16 h8 __attribute__((ext_vector_type(8)));
typedef _Float16 h4 __attribute__((ext_vector_type(4)));
typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef unsigned int u32;
typedef unsigned long long u64;

namespace {

struct Taps11 { float k[11]; };
struct __attribute__((packed, aligned(4))) u32x3 { u32 a, b, c; };

constexpr int TOW = 128, TOH = 64;
// strides in f16 elements; all are odd multiples of 16 bytes so that the 16 owners of a ds_read_b128 lane group hit 16 different
// 16-byte bank slots
constexpr int GS = 168;    // G  [gy = y_rel + 8 : 80][gx = x_rel + 16 : 160]
constexpr int HS = 104;    // H1T[hx = x_rel + 8 : 144][hy = y_rel + 16 : 96]   (hy 0..7 and 88..95 are zero padding)
constexpr int BS = 152;    // BL [by = y_rel + 8 : 80][bx = x_rel + 8 : 144]
constexpr int RS = 88;     // RT [rx = x_rel : 128][ry = y_rel + 8 : 80], hi plane then lo plane
constexpr int G_OFF = 0, H_OFF = G_OFF + 80 * GS * 2;                 // 26,880
constexpr int RTH_OFF = 0, RTL_OFF = RTH_OFF + 128 * RS * 2;          // RT aliases G + H1T (dead after S2): 2 x 22,528
constexpr int OUT_OFF = RTL_OFF + 128 * RS * 2;                       // 45,056: the 64 x 128 result bytes (8 KB)
constexpr int B_OFF = H_OFF + 144 * HS * 2;                           // 56,832
constexpr int LDS_BYTES = B_OFF + 80 * BS * 2;                        // 81,152
constexpr int AMB_OFF = OUT_OFF + TOH * TOW, AMB_MAX = 1024;         // 53,248: per-tile list of ambiguous pixels (16 + 2 KB)
static_assert(AMB_OFF + 16 + 2 * AMB_MAX <= B_OFF, "result tile and ambiguity list must fit below BL");

constexpr float EPS = 1.f / 512.f;
constexpr float TAP_SCALE = 65536.f, TAP_SCALE_INV = 1.f / 65536.f;   // largest tap 0.199: hi <= 13,041, lo stays a normal f16

__device__ __forceinline__ float gray_f32(float b, float g, float r)
{
    return floorf(__builtin_fmaf(b, 3735.f / 32768.f, __builtin_fmaf(g, 19235.f / 32768.f, __builtin_fmaf(r, 9798.f / 32768.f, 0.5f))));
}

__device__ __forceinline__ f32x4 mfma(h8 a, h8 b, f32x4 c) { return __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, c, 0, 0, 0); }

// Persistent: a workgroup walks over tiles; the frame bytes of its next tile are loaded (into registers) while the current tile's last two
// passes run.
template <bool BITS>
__global__ __launch_bounds__(512, 4) void k_preprocess_mm(const u8 *__restrict__ bgr, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride, u8 *__restrict__ out,
                                                          Taps11 taps, int tiles_x, int tiles_y, long ntiles, long tiles_per_xcd, int wgs_per_xcd,
                                                          float *__restrict__ mean_dbg, unsigned *__restrict__ amb_total)
{
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_BYTES];
    _Float16 *const G = (_Float16 *)(lds + G_OFF), *const H1T = (_Float16 *)(lds + H_OFF), *const BL = (_Float16 *)(lds + B_OFF);
    _Float16 *const RTH = (_Float16 *)(lds + RTH_OFF), *const RTL = (_Float16 *)(lds + RTL_OFF);
    unsigned char *const OUT = lds + OUT_OFF;
    u32 *const amb_n = (u32 *)(lds + AMB_OFF);                                 // this tile's ambiguous pixels: count, then yr << 8 | xr
    unsigned short *const amb_e = (unsigned short *)(amb_n + 4);

    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int n16 = lane & 15, q = lane >> 4;

    // spatially adjacent tiles share halo pixels: each XCD (own L2; workgroup b runs on XCD b % 8) gets a contiguous run of tiles, which
    // its workgroups walk through side by side
    const long xcd_first = (long)(blockIdx.x & 7) * tiles_per_xcd;
    const long xcd_end = xcd_first + tiles_per_xcd < ntiles ? xcd_first + tiles_per_xcd : ntiles;
    long tile = xcd_first + (blockIdx.x >> 3);
    if (tile >= xcd_end) return;

    // Toeplitz B operands: lane (n, q) holds T[8q + i][n] = tap(8q + i - n - 8)
    h8 t5, t11h, t11l;
#pragma unroll
    for (int i = 0; i < 8; i++) {
        const int d = 8 * q + i - n16 - 8;
        const float w5 = (d == 0) ? 6.f : ((d == 1 || d == -1) ? 4.f : ((d == 2 || d == -2) ? 1.f : 0.f));
        t5[i] = (_Float16)w5;
        float t = 0.f;
#pragma unroll
        for (int j = 0; j < 11; j++) t = (d == j - 5) ? taps.k[j] : t;
        t *= TAP_SCALE;
        const _Float16 hi = (_Float16)t;
        t11h[i] = hi;
        t11l[i] = (_Float16)(t - (float)hi);
    }

    // S0 item k of this thread: row and 4-pixel group inside the 80 x 144 gray region (the same for every tile)
    constexpr int NI = (80 * 36 + 511) / 512;
    int s0_row[NI], s0_grp[NI];
#pragma unroll
    for (int k = 0; k < NI; k++) { const int i = tid + 512 * k; s0_row[k] = i / 36; s0_grp[k] = i - 36 * s0_row[k]; }
    u32x3 raw[NI];
    auto load_tile = [&](long t) {
        const int tx = (int)(t % tiles_x), ty = (int)((t / tiles_x) % tiles_y), frame = (int)(t / ((long)tiles_x * tiles_y));
        const u8 *img = bgr + (ptrdiff_t)frame * img_stride;
#pragma unroll
        for (int k = 0; k < NI; k++) {
            const int sy = sv_reflect101(sv_clamp(ty * TOH + s0_row[k] - 8, -2, H + 1), H);
            const int lx = sv_clamp(tx * TOW - 8 + 4 * s0_grp[k], 0, W - 4);
            if (s0_row[k] < 80) raw[k] = *(const u32x3 *)(img + (ptrdiff_t)sy * pitch + 3 * lx);
        }
    };
    load_tile(tile);

    for (; tile < xcd_end; tile += wgs_per_xcd) {
        const int tx = (int)(tile % tiles_x), ty = (int)((tile / tiles_x) % tiles_y), frame = (int)(tile / ((long)tiles_x * tiles_y));
        const int ox = tx * TOW, oy = ty * TOH;
        // the LDS addresses of all passes depend only on the lane and the wave: hoisted out of this loop they would occupy ~70 registers (and
        // spill); hidden behind an opaque copy of the wave index they are recomputed per tile with a scalar add each
        int wv = wave;
        asm volatile("" : "+s"(wv));

        // ---- S0: zero padding, gray of the loaded bytes -> G ---------------------------------------------------------------------
        if (tid < 160) {
            const uint4 z = {0, 0, 0, 0};
            *(uint4 *)(G + (tid >> 1) * GS + ((tid & 1) ? 152 : 0)) = z;        // gx 0..7 and 152..159 are never loaded but lie in windows
        }
        if (tid < 288) {
            const uint4 z = {0, 0, 0, 0};
            *(uint4 *)(H1T + (tid >> 1) * HS + ((tid & 1) ? 88 : 0)) = z;       // hy 0..7 and 88..95
        }
#pragma unroll
        for (int k = 0; k < NI; k++) {
            if (s0_row[k] >= 80) break;
            const u32x3 d = raw[k];
            const int x = ox - 8 + 4 * s0_grp[k];
            float p0 = gray_f32((float)(d.a & 255), (float)((d.a >> 8) & 255), (float)((d.a >> 16) & 255));
            float p1 = gray_f32((float)(d.a >> 24), (float)(d.b & 255), (float)((d.b >> 8) & 255));
            float p2 = gray_f32((float)((d.b >> 16) & 255), (float)(d.b >> 24), (float)(d.c & 255));
            float p3 = gray_f32((float)((d.c >> 8) & 255), (float)((d.c >> 16) & 255), (float)(d.c >> 24));
            if (x == -4) { const float a1 = p1; p1 = p3; p3 = a1; }   // REFLECT_101: columns -3,-2,-1 = 3,2,1 (the group loaded is 0..3; column -4 is never used)
            if (x == W) p0 = p2;                                       // columns W, W+1 = W-2, W-3 (the group loaded is W-4..W-1)
            const h4 o = {(_Float16)p0, (_Float16)p1, (_Float16)p2, (_Float16)p3};
            *(h4 *)(G + s0_row[k] * GS + 8 + 4 * s0_grp[k]) = o;
        }
        __syncthreads();

        // Every pass: all of a wave's A-operand reads, then its MFMAs, then the post-processing and the (transposed) writes -- the compiler
        // cannot reorder LDS reads over LDS writes by itself.
        // ---- S1: horizontal 1-4-6-4-1 of G -> H1T (transposed), carried as h - 2040 ---------------------------------------------
        // (45 jobs on 8 waves: the three surplus slots repeat jobs 0..2 -- same values to the same places -- so that no slot is conditional)
#pragma unroll
        for (int half = 0; half < 2; half++) {
            h8 a[3];
            f32x4 d[3];
#pragma unroll
            for (int u = 0; u < 3; u++) {
                int job = wv + 8 * (3 * half + u);
                job = job < 45 ? job : job - 45;
                const int rb = job / 9, jt = job - 9 * rb;
                a[u] = *(const h8 *)(G + (16 * rb + n16) * GS + 16 * jt + 8 * q);
            }
#pragma unroll
            for (int u = 0; u < 3; u++) d[u] = mfma(a[u], t5, (f32x4){-2040.f, -2040.f, -2040.f, -2040.f});
#pragma unroll
            for (int u = 0; u < 3; u++) {
                int job = wv + 8 * (3 * half + u);
                job = job < 45 ? job : job - 45;
                const int rb = job / 9, jt = job - 9 * rb;
                const h4 o = {(_Float16)d[u][0], (_Float16)d[u][1], (_Float16)d[u][2], (_Float16)d[u][3]};   // rows gy = 16rb + 4q + r of column hx = 16jt + n
                *(h4 *)(H1T + (16 * jt + n16) * HS + 16 * rb + 4 * q + 8) = o;
            }
        }
        __syncthreads();

        // ---- S2: vertical 1-4-6-4-1 of H1T, (v + 128) >> 8 -> BL ---------------------------------------------------------------
#pragma unroll
        for (int half = 0; half < 2; half++) {
            h8 a[3];
            f32x4 d[3];
#pragma unroll
            for (int u = 0; u < 3; u++) {
                int job = wv + 8 * (3 * half + u);
                job = job < 45 ? job : job - 45;
                const int xb = job / 5, jt = job - 5 * xb;
                a[u] = *(const h8 *)(H1T + (16 * xb + n16) * HS + 16 * jt + 8 * q);
            }
#pragma unroll
            for (int u = 0; u < 3; u++) d[u] = mfma(a[u], t5, (f32x4){32640.f, 32640.f, 32640.f, 32640.f});       // + 16 x 2040
#pragma unroll
            for (int u = 0; u < 3; u++) {
                int job = wv + 8 * (3 * half + u);
                job = job < 45 ? job : job - 45;
                const int xb = job / 5, jt = job - 5 * xb;
                h4 o;
#pragma unroll
                for (int r = 0; r < 4; r++) o[r] = (_Float16)floorf(__builtin_fmaf(d[u][r], 1.f / 256.f, 0.5f));
                *(h4 *)(BL + (16 * jt + n16) * BS + 16 * xb + 4 * q) = o;                                        // row by = 16jt + n, columns bx = 16xb + 4q + r
            }
        }
        __syncthreads();

        // BORDER_REPLICATE of the blurred image: positions outside the image take the value at the clamped position (inside, never rewritten)
        if (ox < 5 || oy < 5 || ox + TOW + 5 > W || oy + TOH + 5 > H) {
            for (int i = tid; i < 80 * 144; i += 512) {
                const int by = i / 144, bx = i - 144 * by;
                const int y = oy + by - 8, x = ox + bx - 8;
                const int cy = sv_clamp(y, 0, H - 1), cx = sv_clamp(x, 0, W - 1);
                const int sby = cy - oy + 8, sbx = cx - ox + 8;
                if ((cy != y || cx != x) && sby >= 0 && sby < 80 && sbx >= 0 && sbx < 144) BL[by * BS + bx] = BL[sby * BS + sbx];
            }
            __syncthreads();
        }

        // the next tile's frame bytes start their way into registers now (G is not free yet: RT aliases it)
        if (tile + wgs_per_xcd < xcd_end) load_tile(tile + wgs_per_xcd);
        if (tid == 0) *amb_n = 0;          // (H1T's region is dead since the barrier after S2; the barrier after S3 orders this before S4's atomics)

        // ---- S3: horizontal 11-tap pass (approximate mean) -> RT pair (transposed) ----------------------------------------------
        {
            h8 a[5];
            f32x4 d[5];
#pragma unroll
            for (int u = 0; u < 5; u++) {
                const int job = wv + 8 * u, rb = job >> 3, jt = job & 7;
                a[u] = *(const h8 *)(BL + (16 * rb + n16) * BS + 16 * jt + 8 * q);
            }
#pragma unroll
            for (int u = 0; u < 5; u++) d[u] = mfma(a[u], t11h, (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int u = 0; u < 5; u++) d[u] = mfma(a[u], t11l, d[u]);
#pragma unroll
            for (int u = 0; u < 5; u++) {
                const int job = wv + 8 * u, rb = job >> 3, jt = job & 7;
                h4 hi, lo;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float v = d[u][r] * TAP_SCALE_INV;
                    hi[r] = (_Float16)v;
                    lo[r] = (_Float16)(v - (float)hi[r]);
                }
                const int off = (16 * jt + n16) * RS + 16 * rb + 4 * q;                     // column rx = 16jt + n, rows ry = by
                *(h4 *)(RTH + off) = hi;
                *(h4 *)(RTL + off) = lo;
            }
        }
        __syncthreads();

        // ---- S4: vertical 11-tap pass, decision against the blurred pixel ------------------------------------------------------
        {
            h8 ah[4], al[4];
            h4 src[4];
            f32x4 d[4];
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int job = wv + 8 * u, xb = job >> 2, jt = job & 3;
                const int aoff = (16 * xb + n16) * RS + 16 * jt + 8 * q;
                ah[u] = *(const h8 *)(RTH + aoff);
                al[u] = *(const h8 *)(RTL + aoff);
                src[u] = *(const h4 *)(BL + (16 * jt + n16 + 8) * BS + 16 * xb + 4 * q + 8);
            }
#pragma unroll
            for (int u = 0; u < 4; u++) d[u] = mfma(ah[u], t11h, (f32x4){0.f, 0.f, 0.f, 0.f});
#pragma unroll
            for (int u = 0; u < 4; u++) d[u] = mfma(ah[u], t11l, d[u]);
#pragma unroll
            for (int u = 0; u < 4; u++) d[u] = mfma(al[u], t11h, d[u]);
#pragma unroll
            for (int u = 0; u < 4; u++) {
                const int job = wv + 8 * u, xb = job >> 2, jt = job & 3;
                const int yr = 16 * jt + n16, xr = 16 * xb + 4 * q;                         // this lane: row yr, columns xr .. xr + 3 of the tile
                u32 o = 0;
#pragma unroll
                for (int r = 0; r < 4; r++) {
                    const float mean = d[u][r] * TAP_SCALE_INV, diff = mean - (float)src[u][r] - 1.5f;
                    if (diff > 0.f) o |= 255u << (8 * r);
                    if (fabsf(diff) < EPS) {                                                // rare: goes to the exact evaluation below
                        const u32 slot = atomicAdd(amb_n, 1u);
                        if (slot < AMB_MAX) amb_e[slot] = (unsigned short)((yr << 8) | (xr + r));
                    }
                    if (mean_dbg && oy + yr < H && ox + xr + r < W) mean_dbg[((size_t)frame * H + oy + yr) * W + ox + xr + r] = mean;
                }
                *(u32 *)(OUT + yr * TOW + xr) = o;
            }
        }
        __syncthreads();

        // ---- the ambiguous pixels, exactly: cv2's float sequence on the blurred values in BL (which already has its REPLICATE border).  16
        // lanes per pixel: lane i < 11 runs the row pass of window row i (s = x0*k0, fma left to right), lane 0 gathers the 11 results for the
        // column pass (centre, then fma(below + above) outward).  A tile with more than AMB_MAX ambiguous pixels (not a photograph) re-decides
        // all of its pixels.
        {
            const u32 na_raw = *amb_n;
            if (na_raw) {
                const bool all = na_raw > AMB_MAX;
                const u32 na = all ? TOH * TOW : na_raw;
                const int li = tid & 15, grp = tid >> 4;
                for (u32 base = 0; base < na; base += 32) {
                    const u32 idx = base + grp;
                    const u32 e = idx < na ? (all ? ((idx >> 7) << 8) | (idx & 127) : amb_e[idx]) : 0;
                    const int yr = e >> 8, xr = e & 255;
                    const _Float16 *row = BL + (yr + 3 + (li < 11 ? li : 0)) * BS + xr + 3;    // window row li: by = yr + 8 - 5 + li, bx from xr + 8 - 5
                    float acc = __fmul_rn(taps.k[0], (float)row[0]);
#pragma unroll
                    for (int j = 1; j < 11; j++) acc = __builtin_fmaf((float)row[j], taps.k[j], acc);
                    float rw[11];
#pragma unroll
                    for (int j = 0; j < 11; j++) rw[j] = __shfl(acc, j, 16);
                    float m = __fmul_rn(taps.k[5], rw[5]);
#pragma unroll
                    for (int j = 1; j <= 5; j++) m = __builtin_fmaf(__fadd_rn(rw[5 + j], rw[5 - j]), taps.k[5 + j], m);
                    const int mean = sv_clamp(__float2int_rn(m), 0, 255);
                    const int srcv = (int)(float)BL[(yr + 8) * BS + xr + 8];
                    if (li == 0 && idx < na) OUT[yr * TOW + xr] = (srcv - mean <= -2) ? 255 : 0;
                }
                if (amb_total && tid == 0) atomicAdd(amb_total, na_raw > AMB_MAX ? (u32)(TOH * TOW) : na_raw);
                __syncthreads();
            }
        }

        // ---- the 64 x 128 result as full rows -----------------------------------------------------------------------------------
        if (BITS) {
            if (tid < 256) {
                const int row = tid >> 2, wd = tid & 3, y = oy + row, x = ox + 32 * wd;
                if (y < H && x < W) {
                    const uint4 b0 = *(const uint4 *)(OUT + row * TOW + 32 * wd), b1 = *(const uint4 *)(OUT + row * TOW + 32 * wd + 16);
                    const u32 v[8] = {b0.x, b0.y, b0.z, b0.w, b1.x, b1.y, b1.z, b1.w};
                    u32 w = 0;
#pragma unroll
                    for (int k = 0; k < 8; k++) w |= ((((v[k] & 0x01010101u) * 0x10204080u) >> 28) & 0xFu) << (4 * k);   // bytes 0..3 -> bits 0..3
                    ((u32 *)out)[((size_t)frame * H + y) * (W >> 5) + (x >> 5)] = w;
                }
            }
        } else {
            const int row = tid >> 3, c = tid & 7, y = oy + row, x = ox + 16 * c;
            if (y < H && x < W) *(uint4 *)(out + ((size_t)frame * H + y) * W + x) = *(const uint4 *)(OUT + row * TOW + 16 * c);
        }
        __syncthreads();                   // OUT and the ambiguity list lie where the next tile's padding and H1T go
    }
}

}  // namespace

// diagnostics: how many pixels of the launches since the counter was last read went to the exact evaluation (synchronises the device)
int svk_preprocess_mm_stats(sv_ctx *ctx, unsigned *ambiguous, unsigned long *capacity)
{
    *ambiguous = 0; *capacity = 0;
    if (!ctx->k1_list) return SV_OK;
    SV_HIP(hipDeviceSynchronize());
    SV_HIP(hipMemcpy(ambiguous, ctx->k1_list, 4, hipMemcpyDeviceToHost));
    SV_HIP(hipMemset(ctx->k1_list, 0, 4));
    return SV_OK;
}

bool svk_preprocess_mm_supported(const u8 *bgr, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride, const void *out, bool bits)
{
    return H >= 16 && W >= 16 && W % (bits ? 32 : 16) == 0 && pitch % 4 == 0 && img_stride % 4 == 0 && (uintptr_t)bgr % 4 == 0 && (uintptr_t)out % 16 == 0;
}

int svk_preprocess_mm(sv_ctx *ctx, const u8 *bgr, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride, u8 *out, bool bits, float *mean_dbg, hipStream_t s)
{
    Taps11 t;
    sv_gaussian_taps_f32(11, t.k);
    sv_time_scope ts(ctx, SVK_PREPROCESS, s);
    const int tiles_x = (W + TOW - 1) / TOW, tiles_y = (H + TOH - 1) / TOH;
    const long ntiles = (long)n * tiles_x * tiles_y, per_xcd = (ntiles + 7) / 8;
    const int cus = ctx->num_cus > 0 ? ctx->num_cus : 256;
    long wgs_per_xcd = 2L * cus / 8;                                   // two workgroups per CU
    if (wgs_per_xcd > per_xcd) wgs_per_xcd = per_xcd;
    if (wgs_per_xcd < 1) wgs_per_xcd = 1;
    const dim3 grid((unsigned)(wgs_per_xcd * 8));
    unsigned *amb_total = (unsigned *)ctx->k1_list;                    // optional counter (sv_preprocess_stats); null until someone asks
    if (bits)
        hipLaunchKernelGGL(k_preprocess_mm<true>, grid, dim3(512), 0, s, bgr, H, W, pitch, img_stride, out, t, tiles_x, tiles_y, ntiles, per_xcd, (int)wgs_per_xcd, mean_dbg, amb_total);
    else
        hipLaunchKernelGGL(k_preprocess_mm<false>, grid, dim3(512), 0, s, bgr, H, W, pitch, img_stride, out, t, tiles_x, tiles_y, ntiles, per_xcd, (int)wgs_per_xcd, mean_dbg, amb_total);
    SV_LAUNCH_CHECK("k_preprocess_mm");
    return SV_OK;
}

// turn the diagnostic counter on (allocates 16 bytes in the context)
int svk_preprocess_mm_enable_stats(sv_ctx *ctx)
{
    if (ctx->k1_list) return SV_OK;
    SV_HIP(hipMalloc(&ctx->k1_list, 16));
    SV_HIP(hipMemset(ctx->k1_list, 0, 16));
    return SV_OK;
}
