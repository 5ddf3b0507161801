// Despeckle -- an exact accelerator for the HOST corner search (cv/grid.py:37-71), not a reference stage.
//
// The thresholded frame holds ~24,000 connected components, almost all of them noise specks a few pixels
// across; following their borders is what the host search spends its time on.  A component that lies strictly
// inside a 64x64 tile (touches none of the tile's outermost pixels) has a bounding box under 62x62 px, so it can
// never reach find_grid_contour's area floor (10 % of the frame), and erasing it cannot change how any other
// border is followed (Suzuki-Abe only ever looks at 8-neighbours of the component being traced) nor whether a
// later component counts as external (a traced speck always leaves a "right-exit" mark behind, i.e. the same
// outside-state the scan had before it).  So find_grid_contour(despeckle(b)) == find_grid_contour(b); the tests
// check exactly that.  preprocess_for_grid_detection's own output is never altered -- this runs on a copy.
//
// One wave per tile, one lane per tile row, the row held as a 64-bit mask.  Seeds = foreground pixels on the
// tile's outer ring; flood fill by Jacobi iteration: within a row a seed spreads along its run with one
// carry-propagating add per direction, between rows through DPP lane shifts -- one row per iteration.  A tile crossed by
// the grid's vertical lines needed up to 64 such iterations (38 % of a synthetic frame's tiles, most of the kernel's time), so
// since round 3 the fill alternates orientation: after every iteration the reached set is transposed (64x64 bits across the
// wave: one v_permlane32_swap, then five exchange steps of shuffle + v_alignbit + v_bfi per word) and the next iteration runs
// on the transposed tile, where the columns are the carry-filled direction.  A grid tile converges in 4-6 iterations instead
// of 64.  Stops when an iteration changes nothing (a fixed point of either orientation is one of both: each iteration also
// steps once across its slow direction) or after MAX_IT iterations, in which case the tile is left untouched.  Two passes
// with the tile grid offset by (32,32) catch specks that straddle a tile edge of the first pass.
#include "sv_device.h"
#include "sv_internal.h"

namespace {

typedef unsigned long long u64;
typedef unsigned int u32;

__device__ __forceinline__ u64 lane_shift_up(u64 v)     // lane i receives lane i-1's value (0 into lane 0)
{
    const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)v, 0x138, 0xf, 0xf, true);
    const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(v >> 32), 0x138, 0xf, 0xf, true);
    return ((u64)hi << 32) | lo;
}
__device__ __forceinline__ u64 lane_shift_down(u64 v)   // lane i receives lane i+1's value (0 into lane 63)
{
    const u32 lo = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)v, 0x130, 0xf, 0xf, true);
    const u32 hi = (u32)__builtin_amdgcn_update_dpp(0, (int)(u32)(v >> 32), 0x130, 0xf, 0xf, true);
    return ((u64)hi << 32) | lo;
}
// all bits of the runs of `f` that contain a bit of `g` (g subset of f); fr = __brevll(f), hoisted by the caller
__device__ __forceinline__ u64 fill_runs(u64 f, u64 fr, u64 g)
{
    u64 up = ((f + g) ^ f) & f;                       // from each seed towards the MSB end of its run
    const u64 gr = __brevll(g | up);
    const u64 dn = ((fr + gr) ^ fr) & fr;             // and, bit-reversed, towards the LSB end
    return g | up | __brevll(dn);
}

constexpr int T = 64, MAX_IT = 96;
// The first PLAIN_FIRST iterations of a fill stay in the row orientation: most tiles hold nothing but specks a few rows deep, whose fill
// needs that many steps across rows and no transposition (two to three 64x64 transposes cost more than the iterations they save there);
// only a tile still changing after them -- something tall, a grid line -- starts alternating.  Any iteration that changes nothing is a
// fixed point of both orientations (each also steps across its slow direction), so the result is the same for every PLAIN_FIRST.
#ifndef PLAIN_FIRST
#define PLAIN_FIRST 2
#endif

// 64x64 bit transpose across the wave: lane r holds row r (bit c = column c) -> lane c holds column c (bit r).
// Scale 32: lanes r < 32 exchange their high word with the low word of lane r + 32 (v_permlane32_swap).  Scale k < 32, per 32-bit
// word: with p = the word of lane r ^ k, lane r keeps its blocks on the diagonal and takes the partner's off-diagonal blocks moved
// by k bits -- rotr(p, 32 - k) for (r & k) == 0, rotr(p, k) otherwise (the bits a rotation wraps around fall under the kept mask).
template <int K>
__device__ __forceinline__ u32 xchg(u32 v)          // value of lane ^ K
{
    if (K == 1) return (u32)__builtin_amdgcn_mov_dpp((int)v, 0xB1 /*quad_perm [1,0,3,2]*/, 0xf, 0xf, true);
    if (K == 2) return (u32)__builtin_amdgcn_mov_dpp((int)v, 0x4E /*quad_perm [2,3,0,1]*/, 0xf, 0xf, true);
    return (u32)__builtin_amdgcn_ds_swizzle((int)v, (K << 10) | 0x1F);    // xor mask K within each half of the wave
}
template <int K>
__device__ __forceinline__ void transpose_step(u32 &lo, u32 &hi, int lane)
{
    constexpr u32 M = K == 16 ? 0x0000FFFFu : K == 8 ? 0x00FF00FFu : K == 4 ? 0x0F0F0F0Fu : K == 2 ? 0x33333333u : 0x55555555u;
    const bool upper = (lane & K) != 0;
    const u32 keep = upper ? ~M : M, rot = upper ? K : 32 - K;
    const u32 pl = xchg<K>(lo), ph = xchg<K>(hi);
    lo = (lo & keep) | (__builtin_amdgcn_alignbit(pl, pl, rot) & ~keep);
    hi = (hi & keep) | (__builtin_amdgcn_alignbit(ph, ph, rot) & ~keep);
}
__device__ __forceinline__ u64 transpose64(u64 v, int lane)
{
    u32 lo = (u32)v, hi = (u32)(v >> 32);
    const auto sw = __builtin_amdgcn_permlane32_swap(lo, hi, false, false);   // lo of lanes 32..63 <-> hi of lanes 0..31
    lo = sw[0]; hi = sw[1];
    transpose_step<16>(lo, hi, lane);
    transpose_step<8>(lo, hi, lane);
    transpose_step<4>(lo, hi, lane);
    transpose_step<2>(lo, hi, lane);
    transpose_step<1>(lo, hi, lane);
    return ((u64)hi << 32) | lo;
}

// The pixels of tile f (lane = row) that are 8-connected to the tile's outer ring; f itself if the fill did not converge.
__device__ __forceinline__ u64 ring_connected(u64 f, int lane)
{
    const u64 ring = (lane == 0 || lane == T - 1) ? ~0ull : 0x8000000000000001ull;      // the same mask in either orientation
    const u64 fr = __brevll(f);
    u64 g = fill_runs(f, fr, f & ring);
    if (!__any(f != g)) return f;                                     // everything hangs on the ring already (or the tile is empty)
    if (!__any(g != 0)) return 0;                                     // nothing touches the ring: every component lies strictly inside
    u64 ft = 0, ftr = 0;                                              // the transposed tile, made when the fill first has to turn
    bool have_ft = false, transposed = false, converged = false;
    for (int it = 0; it < MAX_IT; it++) {
        const u64 cur = transposed ? ft : f, curr = transposed ? ftr : fr;
        const u64 nb = g | lane_shift_up(g) | lane_shift_down(g);
        const u64 seeds = cur & (nb | (nb << 1) | (nb >> 1));          // 8-connectivity
        const u64 g2 = fill_runs(cur, curr, g | seeds);
        const bool changed = g2 != g;
        g = g2;
        if (!__any(changed)) { converged = true; break; }
        if (it < PLAIN_FIRST - 1) continue;                               // (see PLAIN_FIRST)
        if (!have_ft) { ft = transpose64(f, lane); ftr = __brevll(ft); have_ft = true; }
        g = transpose64(g, lane);
        transposed = !transposed;
    }
    if (!converged) return f;
    return transposed ? transpose64(g, lane) : g;
}

__global__ __launch_bounds__(256) void k_despeckle(const u8 *__restrict__ src, u8 *__restrict__ dst, u32 *__restrict__ packed, int H, int W,
                                                   int ox, int oy, int tiles_x, int tiles_y, long ntiles)
{
    const int lane = threadIdx.x & 63;
    const long tile = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (tile >= ntiles) return;
    const int tx = (int)(tile % tiles_x), ty = (int)((tile / tiles_x) % tiles_y);
    const long frame = tile / ((long)tiles_x * tiles_y);
    const int x0 = tx * T - ox, y0 = ty * T - oy, y = y0 + lane;
    const u8 *row = src + (frame * H + (y < 0 ? 0 : (y >= H ? H - 1 : y))) * (long)W;
    const bool row_ok = y >= 0 && y < H;
    const bool fast = row_ok && x0 >= 0 && x0 + T <= W && ((W | x0) & 3) == 0;

    u64 f = 0;
    if (fast) {
        const u32 *p = (const u32 *)(row + x0);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const u32 d = p[k] & 0x01010101u;          // pixels are 0 or 255
            f |= (u64)(((d * 0x10204080u) >> 28) & 0xF) << (4 * k);   // bytes 0..3 -> bits 0..3
        }
    } else if (row_ok) {
        for (int k = 0; k < T; k++) {
            const int x = x0 + k;
            if (x >= 0 && x < W && row[x]) f |= 1ull << k;
        }
    }
    const u64 keep = ring_connected(f, lane);
    if (packed) {                     // bit-packed result (1 bit per pixel, LSB first, W/32 words per row) for the D2H copy
        if (row_ok) {
            u32 *prow = packed + (frame * H + y) * (long)(W >> 5);
            if (x0 >= 0 && x0 < W) prow[x0 >> 5] = (u32)keep;
            if (x0 + 32 >= 0 && x0 + 32 < W) prow[(x0 + 32) >> 5] = (u32)(keep >> 32);
        }
        return;
    }
    if (src == dst && keep == f) return;                              // nothing to erase in this row
    u8 *orow = dst + (frame * H + y) * (long)W;
    if (fast) {
        u32 *q = (u32 *)(orow + x0);
#pragma unroll
        for (int k = 0; k < 16; k++) {
            const u32 b = (u32)(keep >> (4 * k)) & 0xF;
            q[k] = (((b * 0x00204081u) & 0x01010101u) * 255u);       // bits 0..3 -> bytes 0..3 of 0/255
        }
    } else if (row_ok) {
        for (int k = 0; k < T; k++) {
            const int x = x0 + k;
            if (x >= 0 && x < W) orow[x] = (keep >> k) & 1 ? 255 : 0;
        }
    }
}

// The same filter on a bit image, in place (1 bit per pixel, LSB = leftmost, wpr = W/32 words per row): a tile row is two words.  Tiles of
// one pass are disjoint, so reading and writing the same array is safe within a pass.
// TPW consecutive tiles of a tile row per wave, their words loaded up front: with one tile per wave the kernel was a chain of one memory
// latency, one short computation and one store per wave, and 65,000 waves per pass of a 128-frame chunk.
constexpr int TPW = 4;
__global__ __launch_bounds__(256) void k_despeckle_bits(u32 *__restrict__ bits, int H, int wpr, int ox, int oy, int tiles_x, int tiles_y, long ngroups)
{
    const int lane = threadIdx.x & 63;
    const long group = (long)blockIdx.x * 4 + (threadIdx.x >> 6);
    if (group >= ngroups) return;
    const int gx = (tiles_x + TPW - 1) / TPW;                              // groups per tile row
    const int tg = (int)(group % gx), ty = (int)((group / gx) % tiles_y);
    const long frame = group / ((long)gx * tiles_y);
    const int y = ty * T - oy + lane;
    const bool row_ok = y >= 0 && y < H;
    u32 *row = bits + (frame * H + (row_ok ? y : 0)) * (long)wpr;
    u64 f[TPW];
    bool ok0[TPW], ok1[TPW];
    int k0[TPW];
#pragma unroll
    for (int t = 0; t < TPW; t++) {
        const int tx = tg * TPW + t;
        k0[t] = (tx * T - ox) >> 5;                                        // first word of the tile row (-1 for the first tile of the offset grid)
        ok0[t] = row_ok && tx < tiles_x && k0[t] >= 0 && k0[t] < wpr;
        ok1[t] = row_ok && tx < tiles_x && k0[t] + 1 < wpr;
        f[t] = (ok0[t] ? (u64)row[k0[t]] : 0ull) | (ok1[t] ? (u64)row[k0[t] + 1] << 32 : 0ull);
    }
#pragma unroll
    for (int t = 0; t < TPW; t++) {
        const u64 keep = ring_connected(f[t], lane);
        if (keep == f[t]) continue;
        if (ok0[t] && (u32)keep != (u32)f[t]) row[k0[t]] = (u32)keep;
        if (ok1[t] && (u32)(keep >> 32) != (u32)(f[t] >> 32)) row[k0[t] + 1] = (u32)(keep >> 32);
    }
}

// The same, one 512-thread workgroup per band of 64 rows (all tiles of a tile row): the band is read row by row -- a wave-load is a
// row's 4 W/32 contiguous bytes -- into the LDS (row pitch odd: lane = row reads are conflict-free), and the waves take their tiles'
// two words per row from there.  With lane = row loads straight from memory (k_despeckle_bits) every lane of a wave-load is its own
// 4-byte access in its own line and the texture addresser, not the fill, paces the kernel (tools/dev/despeckle_time.py with the fill
// compiled out: 0.07 of the 0.15 ms per 256 frames).
__global__ __launch_bounds__(512) void k_despeckle_bits_band(u32 *__restrict__ bits, int H, int wpr, int pitch, int ox, int oy, int tiles_x, int tiles_y)
{
    extern __shared__ u32 band[];                                          // [T][pitch]
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int ty = blockIdx.x % tiles_y;
    const long frame = blockIdx.x / tiles_y;
    const int y0 = ty * T - oy;
    u32 *img = bits + frame * H * (long)wpr;
    for (int r = wave; r < T; r += 8) {
        const int y = y0 + r;
        const bool ok = y >= 0 && y < H;
        for (int k = lane; k < wpr; k += 64) band[r * pitch + k] = ok ? img[(long)y * wpr + k] : 0u;
    }
    __syncthreads();
    const int y = y0 + lane;
    const bool row_ok = y >= 0 && y < H;
    u32 *row = img + (long)(row_ok ? y : 0) * wpr;
    const u32 *brow = band + lane * pitch;
    for (int tx = wave; tx < tiles_x; tx += 8) {
        const int k0 = (tx * T - ox) >> 5;                                 // first word of the tile row (-1 for the first tile of the offset grid)
        const bool ok0 = k0 >= 0 && k0 < wpr, ok1 = k0 + 1 < wpr;
        const u64 f = (ok0 ? (u64)brow[k0] : 0ull) | (ok1 ? (u64)brow[k0 + 1] << 32 : 0ull);
        const u64 keep = ring_connected(f, lane);
        if (keep == f || !row_ok) continue;
        if (ok0 && (u32)keep != (u32)f) row[k0] = (u32)keep;
        if (ok1 && (u32)(keep >> 32) != (u32)(f >> 32)) row[k0 + 1] = (u32)(keep >> 32);
    }
}

// Device -> mapped pinned host memory with plain 16-byte stores.  On this platform the shader's PCIe writes run at ~55 GB/s where
// hipMemcpyAsync's DMA engine delivers 22-30 (tools/ubench_d2h.hip), and 64 small workgroups are enough to saturate the link.
__global__ __launch_bounds__(256) void k_copy_to_host(const uint4 *__restrict__ src, uint4 *__restrict__ dst, size_t n16, const u8 *__restrict__ tail_src,
                                                      u8 *__restrict__ tail_dst, int tail)
{
    for (size_t i = blockIdx.x * (size_t)256 + threadIdx.x; i < n16; i += (size_t)gridDim.x * 256) dst[i] = src[i];
    if (blockIdx.x == 0 && (int)threadIdx.x < tail) tail_dst[threadIdx.x] = tail_src[threadIdx.x];
}

// The despeckled bit image of a frame is mostly zero words (1080p synthetic feed: 13 k of 64.8 k words are not), and the D2H copy of it is
// what bounds the end-to-end rate once the host search is fast.  Sparse record of a frame (sv_pack_sparse_bits):
//   u32 n_values, u32 cap_values, u64 mask[H * gpr], u32 value[cap_values]
// one mask per row and group of 64 words (gpr = ceil(W/32/64) groups per row), bit k = word 64*group + k of the row is non-zero; the
// non-zero words follow in raster order.  n_values > cap_values = the record overflowed (values truncated; the caller falls back to the
// dense image).  One workgroup per frame: masks + counts, a scan of the counts in LDS, then the scatter.
// KEEP: the frame's words stay in registers between the counting and the scatter phase (16 waves x TRIPS x 8 row groups >= G; a 1080p frame
// is 1080 groups), so the image is read once; otherwise (taller frames) the scatter phase reads it again.
template <bool KEEP>
__global__ __launch_bounds__(1024) void k_pack_sparse(const u32 *__restrict__ bits, int H, int wpr, int gpr, u8 *__restrict__ records, long stride,
                                                      unsigned cap_values)
{
    extern __shared__ u32 cnt[];                     // [G] counts, then exclusive offsets; [16] wave sums behind them
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, G = H * gpr;
    u32 *wsum = cnt + G;
    const u32 *fb = bits + (size_t)blockIdx.x * H * wpr;
    u8 *rec = records + (size_t)blockIdx.x * stride;
    u64 *masks = (u64 *)(rec + 8);
    u32 *values = (u32 *)(rec + 8 + 8 * (size_t)G);
    // (U row groups per wave per trip, their loads issued together: one load per trip left the kernel waiting out a full memory latency 68 times)
    constexpr int U = 8, TRIPS = 9;
    u32 kept[KEEP ? TRIPS : 1][U];
    auto load_trip = [&](int g0, u32 (&w)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const int g = g0 + u, y = g / gpr, k = (g - y * gpr) * 64 + lane;
            w[u] = (g < G && k < wpr) ? fb[(size_t)y * wpr + k] : 0u;
        }
    };
    auto count_trip = [&](int g0, const u32 (&w)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u64 m = __ballot(w[u] != 0);
            if (lane == 0 && g0 + u < G) { masks[g0 + u] = m; cnt[g0 + u] = (u32)__popcll(m); }
        }
    };
    auto scatter_trip = [&](int g0, const u32 (&w)[U]) {
#pragma unroll
        for (int u = 0; u < U; u++) {
            const u64 m = __ballot(w[u] != 0);
            if (w[u]) {                                          // (a non-zero word implies g0 + u < G)
                const u32 pos = cnt[g0 + u] + (u32)__popcll(m & ((1ull << lane) - 1));
                if (pos < cap_values) values[pos] = w[u];
            }
        }
    };
    if (KEEP) {
#pragma unroll
        for (int t = 0; t < TRIPS; t++) load_trip((t * 16 + wave) * U, kept[t]);
#pragma unroll
        for (int t = 0; t < TRIPS; t++) count_trip((t * 16 + wave) * U, kept[t]);
    } else {
        for (int g0 = wave * U; g0 < G; g0 += 16 * U) {
            u32 w[U];
            load_trip(g0, w);
            count_trip(g0, w);
        }
    }
    __syncthreads();
    const int per = (G + 1023) / 1024, lo = tid * per, hi = lo + per < G ? lo + per : G;
    u32 mine = 0;
    for (int i = lo; i < hi; i++) mine += cnt[i];
    u32 incl = mine;
#pragma unroll
    for (int d = 1; d < 64; d <<= 1) {
        const u32 v = __shfl_up(incl, d);
        if (lane >= d) incl += v;
    }
    if (lane == 63) wsum[wave] = incl;
    __syncthreads();
    u32 base = 0, total = 0;
    for (int w2 = 0; w2 < 16; w2++) { const u32 v = wsum[w2]; if (w2 < wave) base += v; total += v; }
    u32 run = base + incl - mine;
    for (int i = lo; i < hi; i++) { const u32 c = cnt[i]; cnt[i] = run; run += c; }
    __syncthreads();
    if (KEEP) {
#pragma unroll
        for (int t = 0; t < TRIPS; t++) scatter_trip((t * 16 + wave) * U, kept[t]);
    } else {
        for (int g0 = wave * U; g0 < G; g0 += 16 * U) {
            u32 w[U];
            load_trip(g0, w);
            scatter_trip(g0, w);
        }
    }
    if (tid == 0) { ((u32 *)rec)[0] = total; ((u32 *)rec)[1] = cap_values; }
}

}  // namespace

int svk_despeckle_bits(uint32_t *bits, int n, int H, int W, hipStream_t s)
{
    for (int pass = 0; pass < 2; pass++) {
        const int o = pass ? T / 2 : 0;
        const int tiles_x = (W + o + T - 1) / T, tiles_y = (H + o + T - 1) / T;
        const int wpr = W >> 5, pitch = wpr | 1;
        if ((size_t)T * pitch * 4 <= 65536) {                             // a band fits the LDS: rows up to 8,160 px wide
            hipLaunchKernelGGL(k_despeckle_bits_band, dim3((unsigned)(n * tiles_y)), dim3(512), (size_t)T * pitch * 4, s, bits, H, wpr, pitch, o, o, tiles_x, tiles_y);
            SV_LAUNCH_CHECK("k_despeckle_bits_band");
            continue;
        }
        const long ngroups = (long)n * ((tiles_x + TPW - 1) / TPW) * tiles_y;
        hipLaunchKernelGGL(k_despeckle_bits, dim3((unsigned)((ngroups + 3) / 4)), dim3(256), 0, s, bits, H, wpr, o, o, tiles_x, tiles_y, ngroups);
        SV_LAUNCH_CHECK("k_despeckle_bits");
    }
    return SV_OK;
}

int svk_pack_sparse_bits(const u32 *bits, int n, int H, int W, u8 *records, long stride, hipStream_t s)
{
    const int wpr = W >> 5, gpr = (wpr + 63) / 64, G = H * gpr;
    const long cap = (stride - 8 - 8L * G) / 4;
    const unsigned capu = (unsigned)(cap > 0xffffffffL ? 0xffffffffL : cap);
    if (G <= 16 * 9 * 8)
        hipLaunchKernelGGL(k_pack_sparse<true>, dim3((unsigned)n), dim3(1024), (size_t)(G + 16) * 4, s, bits, H, wpr, gpr, records, stride, capu);
    else
        hipLaunchKernelGGL(k_pack_sparse<false>, dim3((unsigned)n), dim3(1024), (size_t)(G + 16) * 4, s, bits, H, wpr, gpr, records, stride, capu);
    SV_LAUNCH_CHECK("k_pack_sparse");
    return SV_OK;
}

int svk_copy_to_host(const void *src, void *dst_host, size_t bytes, hipStream_t s)
{
    const size_t n16 = bytes / 16;
    const int tail = (int)(bytes - n16 * 16);
    size_t wgs = (n16 + 255) / 256;
    wgs = wgs < 1 ? 1 : (wgs > 64 ? 64 : wgs);
    hipLaunchKernelGGL(k_copy_to_host, dim3((unsigned)wgs), dim3(256), 0, s, (const uint4 *)src, (uint4 *)dst_host, n16, (const u8 *)src + n16 * 16,
                       (u8 *)dst_host + n16 * 16, tail);
    SV_LAUNCH_CHECK("k_copy_to_host");
    return SV_OK;
}

// two passes (tile grids offset by half a tile); dst may equal src.  With `packed` (needs W % 32 == 0) the second pass
// writes 1 bit per pixel there instead of bytes into dst (dst then holds the first pass only and serves as scratch).
int svk_despeckle(const u8 *src, int n, int H, int W, u8 *dst, u32 *packed, hipStream_t s)
{
    for (int pass = 0; pass < 2; pass++) {
        const int o = pass ? T / 2 : 0;
        const int tiles_x = (W + o + T - 1) / T, tiles_y = (H + o + T - 1) / T;
        const long ntiles = (long)n * tiles_x * tiles_y;
        hipLaunchKernelGGL(k_despeckle, dim3((unsigned)((ntiles + 3) / 4)), dim3(256), 0, s, pass ? dst : src, dst, pass ? packed : (u32 *)nullptr, H, W, o, o,
                           tiles_x, tiles_y, ntiles);
        SV_LAUNCH_CHECK("k_despeckle");
    }
    return SV_OK;
}
