// K3 -- DigitCNN.forward (ml/model.py:34-42, eval mode) on MI355X: the f32-MFMA kernels and the small stages around the CNN.
// (The default conv/fc pair -- f32-grade arithmetic on the f16 matrix pipe -- is k3_cnn_h2.hip; svk_cnn_forward at the end of this file
// picks between the two, see sv_ctx_set_cnn_kernels.)
//
//   k_conv_features_pc : true f32 throughout, the reference's own numeric range: what runs when the loaded weights or an f32 input
//        leave the range the f16-pair kernels carry exactly (and for misaligned 8-bit buffers).  Persistent, one 512-thread workgroup per
//        CU working through PAIRS of cells.  conv1 (1->32, 3x3, pad 1) + ReLU + 2x2 max-pool on the VALU into zero-bordered 16x16
//        planes in LDS; conv2 (32->64) as an implicit GEMM on v_mfma_f32_16x16x4_f32:
//        M = 4 pooling windows x 4 positions, N = 16 output channels, K = 4 input channels of one
//        3x3 tap per instruction.  A comes straight from the LDS planes (one ds_read_b32 with an
//        immediate offset per step, no im2col buffer); B (all 288x32 weights a wave needs) stays in
//        144 VGPRs for the life of the kernel.  The 16x16 accumulator holds the 4 positions of a
//        pooling window in the 4 registers of one lane, so bias + ReLU + max-pool are 3 v_max and
//        never leave the lane.  Output: features [cell][window 49][oc 64] f32.
//   k_fc_head : fc1 (3136->128) on the same MFMA with cells as M (16 cells per wave, weights streamed per wave) + ReLU, fc2 (128->10),
//        argmax (pipeline/run.py:142) and softmax[argmax] (run.py:141-143).
//   k_softmax_topk, k_preprocess_cells: the run_v2 top-k epilogue and run.py's preprocess_cell (scope rows N3, N1).
//   Under SV_XCHECK (libsudokuvision_xcheck.so, test-only): k_conv_features_wstream / _wsplit (round 1's Winograd stream on f32 and on
//        split-bf16 MFMA) and k_fc_head_frame -- independent implementations the tests compare the product with.
//
// Weight images are packed on the host by sv_load_weights_f32 (sv_api.cpp) into exactly the
// per-lane register order the kernels load.
#include "sv_device.h"
#include "sv_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));

namespace {

constexpr int PLANE = 257;               // 16x16 plane + 1 float of bank skew
constexpr int C1_CELL = 32 * PLANE;      // conv1 output of one cell
constexpr int IN_W = 30, IN_CELL = 900;  // zero-padded 30x30 input
constexpr int FEAT = 3136;

__device__ __forceinline__ float glue_norm(u8 c)
{
    // x = ((255 - cell)/255 - 0.5)/0.5, one rounding per operation (pipeline/run.py:129-135)
    const float t = __fdiv_rn((float)(255 - (int)c), 255.0f);
    return __fdiv_rn(__fsub_rn(t, 0.5f), 0.5f);
}

// ---------------------------------------------------------------------------------------------------
// N1 -- the per-cell glue of pipeline/run.py:73-95 (preprocess_cell), one wave per cell (k_preprocess_cells):
//   cv2.createCLAHE(2.0, (4,4)) on 28x28: 16 tiles of 7x7; the clip limit is max(1, int(2*49/256)) = 1, so a
//   tile's clipped histogram is its set of present values (a 256-bit bitmap) plus the redistributed residual
//   (one extra count every 256/residual bins); LUT[v] = rint(cumsum[v] * 255/49) is evaluated on demand from
//   the bitmap's prefix popcounts -- no histogram or LUT arrays.  Bilinear blend of the 4 neighbouring tiles
//   in f32 with one rounding per operation, then adaptiveThreshold(GAUSSIAN_C, BINARY, 11, 2) on the 28x28
//   result (same f32 FMA-chain Gaussian as K1, REPLICATE border).  (Fusing this into the conv kernel's producer
//   waves was tried: the producers share the consumers' 216-VGPR allocation and the extra code spilled.)
// ---------------------------------------------------------------------------------------------------
struct N1Scratch {
    float cf[784];          // CLAHE output as f32
    float rw[784];          // Gaussian row pass
    unsigned bits[16][8];   // per tile: which of the 256 values occur
    unsigned pre[16][8];    // per tile: number of present values below word w
    int residual[16], step[16];
};

__device__ __forceinline__ void wave_lds_fence() { __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront"); __builtin_amdgcn_s_waitcnt(0xC07F); __builtin_amdgcn_wave_barrier(); }

__device__ __forceinline__ float n1_lut(const N1Scratch &sc, int tile, int v)
{
    const int w = v >> 5;
    const unsigned mask = 0xFFFFFFFFu >> (31 - (v & 31));
    int cum = (int)sc.pre[tile][w] + __popc(sc.bits[tile][w] & mask);
    const int res = sc.residual[tile];
    if (res != 0) { const int extra = v / sc.step[tile] + 1; cum += extra < res ? extra : res; }
    const float f = rintf(__fmul_rn((float)cum, 255.0f / 49.0f));
    return fminf(fmaxf(f, 0.f), 255.f);
}

// one wave: cell (u8[784], global) -> normalised CNN input written into the zero-bordered 30x30 tile `in_dst`
__device__ void n1_preprocess_cell(const u8 *__restrict__ cell, float *in_dst, N1Scratch &sc, int lane, const float (&taps)[11])
{
    int v[13];
#pragma unroll
    for (int k = 0; k < 13; k++) { const int i = lane + 64 * k; v[k] = i < 784 ? cell[i] : 0; }
    for (int i = lane; i < 128; i += 64) (&sc.bits[0][0])[i] = 0;
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 13; k++) {
        const int i = lane + 64 * k;
        if (i < 784) { const int y = i / 28, x = i - 28 * y; atomicOr(&sc.bits[(y / 7) * 4 + x / 7][v[k] >> 5], 1u << (v[k] & 31)); }
    }
    wave_lds_fence();
    if (lane < 16) {
        int n = 0;
        for (int w = 0; w < 8; w++) { sc.pre[lane][w] = n; n += __popc(sc.bits[lane][w]); }
        const int residual = 49 - n;                 // clipped mass: every present value keeps one count
        sc.residual[lane] = residual;
        sc.step[lane] = residual > 0 ? (256 / residual > 1 ? 256 / residual : 1) : 1;
    }
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 13; k++) {
        const int i = lane + 64 * k;
        if (i < 784) {
            const int y = i / 28, x = i - 28 * y;
            const float txf = __fsub_rn(__fmul_rn((float)x, 1.0f / 7.0f), 0.5f), tyf = __fsub_rn(__fmul_rn((float)y, 1.0f / 7.0f), 0.5f);
            int tx1 = (int)floorf(txf), ty1 = (int)floorf(tyf);
            const float xa = __fsub_rn(txf, (float)tx1), ya = __fsub_rn(tyf, (float)ty1), xa1 = __fsub_rn(1.0f, xa), ya1 = __fsub_rn(1.0f, ya);
            int tx2 = tx1 + 1, ty2 = ty1 + 1;
            tx1 = tx1 < 0 ? 0 : tx1; ty1 = ty1 < 0 ? 0 : ty1; tx2 = tx2 > 3 ? 3 : tx2; ty2 = ty2 > 3 ? 3 : ty2;
            const float l11 = n1_lut(sc, ty1 * 4 + tx1, v[k]), l12 = n1_lut(sc, ty1 * 4 + tx2, v[k]);
            const float l21 = n1_lut(sc, ty2 * 4 + tx1, v[k]), l22 = n1_lut(sc, ty2 * 4 + tx2, v[k]);
            const float top = __fadd_rn(__fmul_rn(l11, xa1), __fmul_rn(l12, xa)), bot = __fadd_rn(__fmul_rn(l21, xa1), __fmul_rn(l22, xa));
            const float res = __fadd_rn(__fmul_rn(top, ya1), __fmul_rn(bot, ya));
            sc.cf[i] = fminf(fmaxf(rintf(res), 0.f), 255.f);
        }
    }
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 13; k++) {      // Gaussian row pass, taps left to right, REPLICATE
        const int i = lane + 64 * k;
        if (i < 784) {
            const int y = i / 28, x = i - 28 * y;
            const float *row = sc.cf + 28 * y;
            float acc = __fmul_rn(taps[0], row[x - 5 < 0 ? 0 : x - 5]);
#pragma unroll
            for (int j = 1; j < 11; j++) { int xx = x + j - 5; xx = xx < 0 ? 0 : (xx > 27 ? 27 : xx); acc = __builtin_fmaf(row[xx], taps[j], acc); }
            sc.rw[i] = acc;
        }
    }
    wave_lds_fence();
#pragma unroll
    for (int k = 0; k < 13; k++) {      // column pass (centre, then pairs), threshold BINARY, invert + normalise
        const int i = lane + 64 * k;
        if (i < 784) {
            const int y = i / 28, x = i - 28 * y;
            float acc = __fmul_rn(taps[5], sc.rw[i]);
#pragma unroll
            for (int j = 1; j <= 5; j++) {
                const int yl = y + j > 27 ? 27 : y + j, yh = y - j < 0 ? 0 : y - j;
                acc = __builtin_fmaf(__fadd_rn(sc.rw[28 * yl + x], sc.rw[28 * yh + x]), taps[5 + j], acc);
            }
            const float mean = fminf(fmaxf(rintf(acc), 0.f), 255.f);
            const bool white = __fsub_rn(sc.cf[i], mean) > -2.f;     // THRESH_BINARY: src - mean > -C
            in_dst[(y + 1) * IN_W + x + 1] = white ? -1.0f : 1.0f;   // ((255 - 255)/255 - .5)/.5 = -1 ; ((255 - 0)/255 - .5)/.5 = +1
        }
    }
}

// ---------------------------------------------------------------------------------------------------
// k_conv_features_pc: one 512-thread workgroup per CU, software-pipelined over cell pairs with ONE
// barrier per pair (producer/consumer wave specialisation).
//   waves 0-3 (consumers): conv2 of pair i on the MFMA pipe, from c1[i & 1]           (weights in VGPRs)
//   waves 4-7 (producers): conv1 of pair i+1 on the VALU into c1[(i+1) & 1], and the 28x28 inputs of
//                          pair i+2 into in_s[i & 1]
// The matrix pipe and the vector pipe of a SIMD run side by side, so the consumers never leave the
// MFMA stream for conv1 or for input staging.  LDS: 2 x 65,792 (conv1 planes) + 2 x 7,200 (inputs) B.
// ---------------------------------------------------------------------------------------------------
struct GaussTaps { float k[11]; };

template <bool U8IN>
__global__ __launch_bounds__(512, 2) void k_conv_features_pc(const void *__restrict__ xin, long B,
                                                             const float *__restrict__ w1, const float *__restrict__ b1,
                                                             const float *__restrict__ w2reg, const float *__restrict__ b2,
                                                             float *__restrict__ feat, const int *__restrict__ run_if_set)
{
    if (run_if_set && *run_if_set == 0) return;      // (svk_cnn_forward: the f16-pair kernels took this batch)
    __shared__ __attribute__((aligned(16))) float lds[4 * IN_CELL + 4 * C1_CELL];
    float *in_base = lds;                 // [2 buffers][2 cells][900]
    float *c1_base = lds + 4 * IN_CELL;   // [2 buffers][2 cells][32][257]

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    const int np = (wave >> 1) & 1, par = wave & 1;   // consumer roles
    const int pw = wave & 3, ptid = tid & 255;        // producer roles

    float breg[2][72];
    float bias2_0 = 0.f, bias2_1 = 0.f;
    if (consumer) {
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int ks = 0; ks < 72; ks++) breg[t][ks] = w2reg[((np * 2 + t) * 72 + ks) * 64 + lane];
        bias2_0 = b2[32 * np + (lane & 15)];
        bias2_1 = b2[32 * np + 16 + (lane & 15)];
    }
    for (int i = tid; i < 4 * IN_CELL + 4 * C1_CELL; i += 512) lds[i] = 0.f;   // zero borders, for good
    __syncthreads();

    const long npairs = (B + 1) / 2;
    const long first = blockIdx.x, stride = gridDim.x;
    const long iters = first < npairs ? (npairs - first + stride - 1) / stride : 0;

    auto stage = [&](long it) {          // producers: inputs of local iteration `it` -> in_s[it & 1]
        const long pair = first + it * stride;
        float *in_s = in_base + (it & 1) * 2 * IN_CELL;
        for (int i = ptid; i < 2 * 784; i += 256) {
            const int cl = i / 784, p = i - cl * 784, y = p / 28, x = p - y * 28;
            long cg = pair * 2 + cl;
            if (cg >= B) cg = B - 1;
            float v;
            if (U8IN) v = glue_norm(((const u8 *)xin)[cg * 784 + p]);
            else v = ((const float *)xin)[cg * 784 + p];
            in_s[cl * IN_CELL + (y + 1) * IN_W + x + 1] = v;
        }
    };
    auto conv1 = [&](long it) {          // producers: in_s[it & 1] -> c1[it & 1]
        const float *in_s = in_base + (it & 1) * 2 * IN_CELL;
        float *c1 = c1_base + (it & 1) * 2 * C1_CELL;
        for (int rnd = 0; rnd < 7; rnd++) {
            const int idx = rnd * 64 + lane;
            if (idx < 392) {
                const int cl = idx / 196, pp = idx - cl * 196, py = pp / 14, px = pp - py * 14;
                float patch[4][4];
                const float *src = in_s + cl * IN_CELL + (2 * py) * IN_W + 2 * px;
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 4; j++) patch[i][j] = src[i * IN_W + j];
                float *dstp = c1 + cl * C1_CELL + (py + 1) * 16 + px + 1;
#pragma unroll
                for (int o = 0; o < 8; o++) {
                    const int oc = pw * 8 + o;
                    const float *w = w1 + oc * 9;
                    const float bias = b1[oc];
                    float m = -3.0e38f;
#pragma unroll
                    for (int dy = 0; dy < 2; dy++)
#pragma unroll
                        for (int dx = 0; dx < 2; dx++) {
                            float acc = bias;
#pragma unroll
                            for (int ky = 0; ky < 3; ky++)
#pragma unroll
                                for (int kx = 0; kx < 3; kx++) acc = __builtin_fmaf(w[ky * 3 + kx], patch[dy + ky][dx + kx], acc);
                            m = fmaxf(m, acc);
                        }
                    dstp[oc * PLANE] = fmaxf(m, 0.f);
                }
            }
        }
    };

    // prologue: fill the pipeline
    if (!consumer && iters > 0) stage(0);
    __syncthreads();
    if (!consumer) {
        if (iters > 0) conv1(0);
        if (iters > 1) stage(1);
    }
    __syncthreads();

    for (long it = 0; it < iters; it++) {
        if (consumer) {
            const long pair = first + it * stride;
            const float *c1 = c1_base + (it & 1) * 2 * C1_CELL;
            for (int j = par; j < 25; j += 2) {
                const int i16 = lane & 15, q = lane >> 4;
                int g = 4 * j + (i16 >> 2);
                if (g > 97) g = 97;
                const int cl = g >= 49 ? 1 : 0, wl = g - 49 * cl, wy = wl / 7, wx = wl - 7 * wy, s = i16 & 3;
                const float *ap = c1 + cl * C1_CELL + q * 8 * PLANE + (2 * wy + (s >> 1)) * 16 + 2 * wx + (s & 1);
                f32x4 acc0 = {0.f, 0.f, 0.f, 0.f}, acc1 = {0.f, 0.f, 0.f, 0.f};
#pragma unroll
                for (int ks = 0; ks < 72; ks++) {
                    const int tap = ks >> 3, icb = ks & 7;
                    const float a = ap[icb * PLANE + (tap / 3) * 16 + (tap % 3)];
                    acc0 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, breg[0][ks], acc0, 0, 0, 0);
                    acc1 = __builtin_amdgcn_mfma_f32_16x16x4f32(a, breg[1][ks], acc1, 0, 0, 0);
                }
                const int gw = 4 * j + q;
                if (gw < 98) {
                    const int ocl = gw >= 49 ? 1 : 0, owl = gw - 49 * ocl;
                    const long cg = pair * 2 + ocl;
                    if (cg < B) {
                        float *o = feat + cg * FEAT + owl * 64 + 32 * np + i16;
                        o[0] = fmaxf(fmaxf(fmaxf(acc0[0], acc0[1]), fmaxf(acc0[2], acc0[3])) + bias2_0, 0.f);
                        o[16] = fmaxf(fmaxf(fmaxf(acc1[0], acc1[1]), fmaxf(acc1[2], acc1[3])) + bias2_1, 0.f);
                    }
                }
            }
        } else {
            if (it + 1 < iters) conv1(it + 1);
            if (it + 2 < iters) stage(it + 2);
        }
        __syncthreads();
    }
}

#ifdef SV_XCHECK   // cross-check kernels: built into libsudokuvision_xcheck.so only (csrc/Makefile), never into the product
// ---------------------------------------------------------------------------------------------------
// k_conv_features_wstream (round 1's default; now a cross-check): conv2 by Winograd F(2x2, 3x3), streamed.  The 7x7 grid of 2x2 output tiles of
// the 14x14 map IS the grid of pooling windows, so per tile:  V = B^T d B over 32 input channels (VALU, 32 add/sub),
// 16 independent GEMMs  M[xi] = V[xi] (tiles x 32) * U[xi] (32 x 64)  on v_mfma_f32_16x16x4_f32, and
// Y = A^T M A + bias, ReLU, 2x2 max -- all four outputs of a tile live in the lane that owns (tile, channel), because
// the 16 GEMMs of one (M tile, N tile) accumulate into 16 register quads of the same lanes.  1568 MFMAs per cell
// instead of 3600; U (G g G^T, 16 x 32 x 16 per wave) stays in 128 VGPRs.  A workgroup owns a contiguous run of cells
// and treats their 49-tile grids as ONE sequence of tiles cut into M tiles of 16 (no 49 -> 64 padding; a per-cell
// variant wasted 23 % of the MFMA rows and needed all of a cell's V, 106 KB, in LDS; here V is two 32-KB slots).
// Winograd reorders the f32 sums: logits differ from the direct kernel by ~1e-6, inside the 1e-4 tolerance
// (tests/test_gpu_parity.py::test_conv_algorithms_agree).
//   step m:  waves 0-3 (one N tile each): 16 GEMMs x 8 k-steps on V[m & 1] for M tile m, output transform, store
//            waves 4-7: B^T d B of M tile m+1 into V[(m+1) & 1] (512 (channel, tile) items = 2 per thread),
//                       conv1 of the next cell whose tiles come up (c1 planes double-buffered by cell parity),
//                       28x28 input of the cell after that (double-buffered too);  one barrier per step.
// The schedule's hazards (a plane is never overwritten while tiles of its previous cell are still to be transformed,
// and so on) are checked exhaustively in tests/test_abi.py::test_winograd_stream_schedule.
// ---------------------------------------------------------------------------------------------------
__host__ __device__ inline long wstream_need(long m, long ncell)   // last cell that M tile m+2 touches
{
    const long c = (16 * (m + 2) + 15) / 49;
    return c < ncell - 1 ? c : ncell - 1;
}

template <bool U8IN>
__global__ __launch_bounds__(512, 2) void k_conv_features_wstream(const void *__restrict__ xin, long B, long cells_per_wg,
                                                                  const float *__restrict__ w1, const float *__restrict__ b1,
                                                                  const float *__restrict__ ureg_img, const float *__restrict__ b2,
                                                                  float *__restrict__ feat)
{
    constexpr int VSLOT = 16 * 32 * 16;       // [xi][ic][tile]
    __shared__ __attribute__((aligned(16))) float lds[2 * IN_CELL + 2 * C1_CELL + 2 * VSLOT];
    float *in_base = lds, *c1_base = lds + 2 * IN_CELL, *v_base = lds + 2 * IN_CELL + 2 * C1_CELL;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    const int nt = wave & 3, ptid = tid & 255;
    const int r16 = lane & 15, q = lane >> 4;

    const long c0 = (long)blockIdx.x * cells_per_wg;
    long ncell = B - c0;
    if (ncell > cells_per_wg) ncell = cells_per_wg;
    if (ncell <= 0) return;
    const int ntiles = (int)ncell * 49, NM = (ntiles + 15) / 16;   // tile arithmetic in 32 bits (a workgroup never owns 2^31 / 49 cells)
    float *featw = feat + c0 * FEAT;                               // this workgroup's first cell

    for (int i = tid; i < 2 * IN_CELL + 2 * C1_CELL + 2 * VSLOT; i += 512) lds[i] = 0.f;
    __syncthreads();

    // Input of a cell -> in_s[c & 1], in two phases so that the global-load latency hides behind the step's other work:
    // stage_load issues one load per value into registers (u8: the cell is 196 dwords, one per thread, 4 pixels of one row
    // each; f32: 784 values, up to 4 per thread), stage_store converts and writes them into the zero-bordered LDS image.
    unsigned sraw[4];
    auto stage_load = [&](long c) {
        if (U8IN) {
            if (ptid < 196) sraw[0] = ((const unsigned *)((const u8 *)xin + (c0 + c) * 784))[ptid];
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++)
                if (ptid + 256 * j < 784) sraw[j] = __float_as_uint(((const float *)xin)[(c0 + c) * 784 + ptid + 256 * j]);
        }
    };
    auto stage_store = [&](long c) {
        float *in_s = in_base + (c & 1) * IN_CELL;
        if (U8IN) {
            if (ptid < 196) {
                const int y = ptid / 7, x = 4 * (ptid - 7 * y);
                float *d = in_s + (y + 1) * IN_W + x + 1;
#pragma unroll
                for (int j = 0; j < 4; j++) d[j] = glue_norm((u8)(sraw[0] >> (8 * j)));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 4; j++) {
                const int i = ptid + 256 * j;
                if (i < 784) { const int y = i / 28, x = i - y * 28; in_s[(y + 1) * IN_W + x + 1] = __uint_as_float(sraw[j]); }
            }
        }
    };
    auto stage = [&](long c) { stage_load(c); stage_store(c); };
    // conv1 + ReLU + 2x2 max: a producer thread owns one group of 4 output channels (og = ptid / 32, the same for every item
    // and every cell, so its 36 weights + 4 biases stay in registers as {w, w} pairs) and walks the 196 pooled pixels in
    // steps of 32.  The two halves of a wave read the same input patches (LDS broadcast).
    const int og = ptid >> 5, pl = ptid & 31;
    f32x2 wreg[4][9], breg[4];
    auto conv1_load_weights = [&]() {
#pragma unroll
        for (int o = 0; o < 4; o++) {
            const float bias = b1[og * 4 + o];
            breg[o] = (f32x2){bias, bias};
#pragma unroll
            for (int t = 0; t < 9; t++) { const float w = w1[(og * 4 + o) * 9 + t]; wreg[o][t] = (f32x2){w, w}; }
        }
    };
    auto conv1 = [&](long c) {                // producers: in_s[c & 1] -> c1[c & 1]
        const float *in_s = in_base + (c & 1) * IN_CELL;
        float *c1 = c1_base + (c & 1) * C1_CELL + og * 4 * PLANE;
        for (int pp = pl; pp < 196; pp += 32) {
            const int py = pp / 14, px = pp - py * 14;
            // the 4x4 input patch as overlapping horizontal pairs: one v_pk_fma_f32 does the two outputs of a pooling-window row
            f32x2 pr[4][3];
            const float *src = in_s + (2 * py) * IN_W + 2 * px;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 3; j++) pr[i][j] = (f32x2){src[i * IN_W + j], src[i * IN_W + j + 1]};
            float *dstp = c1 + (py + 1) * 16 + px + 1;
#pragma unroll
            for (int o = 0; o < 4; o++) {
                f32x2 a0 = breg[o], a1 = breg[o];                    // output rows dy = 0, 1; lanes = dx 0, 1
#pragma unroll
                for (int ky = 0; ky < 3; ky++)
#pragma unroll
                    for (int kx = 0; kx < 3; kx++) {
                        a0 = __builtin_elementwise_fma(wreg[o][ky * 3 + kx], pr[ky][kx], a0);
                        a1 = __builtin_elementwise_fma(wreg[o][ky * 3 + kx], pr[ky + 1][kx], a1);
                    }
                dstp[o * PLANE] = fmaxf(fmaxf(fmaxf(a0[0], a0[1]), fmaxf(a1[0], a1[1])), 0.f);
            }
        }
    };
    auto transform = [&](int m, int first, int last) {   // V[m & 1] = B^T d B for the 16 tiles of M tile m, items [first, last)
        float *Vs = v_base + (m & 1) * VSLOT;
        for (int it = first + ptid; it < last; it += 256) {
            const int ic = it >> 4, tl = it & 15;
            int T = 16 * m + tl;
            if (T > ntiles - 1) T = ntiles - 1;
            const int c = T / 49;
            const int t = T - 49 * c, wy = t / 7, wx = t - 7 * wy;
            const float *d = c1_base + (c & 1) * C1_CELL + ic * PLANE + (2 * wy) * 16 + 2 * wx;
            float tt[4][4];
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const float d0 = d[x], d1 = d[16 + x], d2 = d[32 + x], d3 = d[48 + x];
                tt[0][x] = d0 - d2; tt[1][x] = d1 + d2; tt[2][x] = d2 - d1; tt[3][x] = d1 - d3;
            }
            float *vp = Vs + it;              // (xi*32 + ic)*16 + tl = xi*512 + it
#pragma unroll
            for (int y = 0; y < 4; y++) {
                vp[(y * 4 + 0) * 512] = tt[y][0] - tt[y][2];
                vp[(y * 4 + 1) * 512] = tt[y][1] + tt[y][2];
                vp[(y * 4 + 2) * 512] = tt[y][2] - tt[y][1];
                vp[(y * 4 + 3) * 512] = tt[y][1] - tt[y][3];
            }
        }
    };

    // The two roles run separate loops (one barrier per step in each, so the counts match): register liveness then stays
    // within a role -- U (128 VGPRs) is never live in producer code, nor the conv1 weights in consumer code.
    if (!consumer) {
        conv1_load_weights();
        long conv_done = ncell > 1 ? 1 : 0, staged = conv_done;
        stage(0);
        if (ncell > 1) stage(1);
        __syncthreads();
        conv1(0);
        if (ncell > 1) conv1(1);
        __syncthreads();
        transform(0, 0, 512);
        __syncthreads();
        for (int m = 0; m < NM; m++) {
            const long sc = wstream_need(m + 1, ncell);
            if (sc > staged) stage_load(sc);                      // lands while the transform and conv1 below run
            if (m + 1 < NM) transform(m + 1, 0, 256);
            const long cc = wstream_need(m, ncell);
            if (cc > conv_done) { conv1(cc); conv_done = cc; }
            if (sc > staged) { stage_store(sc); staged = sc; }
            __syncthreads();
        }
        return;
    }

    float ureg[16][8];
#pragma unroll
    for (int xi = 0; xi < 16; xi++)
#pragma unroll
        for (int ks = 0; ks < 8; ks++) ureg[xi][ks] = ureg_img[((nt * 16 + xi) * 8 + ks) * 64 + lane];
    const float bias2 = b2[16 * nt + r16];
    __syncthreads();
    __syncthreads();
    __syncthreads();
    for (int m = 0; m < NM; m++) {
        const float *ap = v_base + (m & 1) * VSLOT + q * 16 + r16;
        f32x4 acc[16];
#pragma unroll
        for (int xi = 0; xi < 16; xi++) acc[xi] = (f32x4){0.f, 0.f, 0.f, 0.f};
        // k-step outermost: consecutive MFMAs hit 16 different accumulators (a dependent 16x16x4 f32 MFMA needs 40
        // cycles, an independent one issues every 32)
#pragma unroll
        for (int ks = 0; ks < 8; ks++)
#pragma unroll
            for (int xi = 0; xi < 16; xi++)
                acc[xi] = __builtin_amdgcn_mfma_f32_16x16x4f32(ap[(xi * 32 + 4 * ks) * 16], ureg[xi][ks], acc[xi], 0, 0, 0);
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int T = 16 * m + 4 * q + reg;
            float s0[4], s1[4];
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const float m0 = acc[x][reg], m1 = acc[4 + x][reg], m2 = acc[8 + x][reg], m3 = acc[12 + x][reg];
                s0[x] = m0 + m1 + m2;
                s1[x] = m1 - m2 - m3;
            }
            const float y00 = s0[0] + s0[1] + s0[2], y01 = s0[1] - s0[2] - s0[3];
            const float y10 = s1[0] + s1[1] + s1[2], y11 = s1[1] - s1[2] - s1[3];
            const float pooled = fmaxf(fmaxf(fmaxf(y00, y01), fmaxf(y10, y11)) + bias2, 0.f);
            if (T < ntiles) featw[(unsigned)(T * 64 + 16 * nt + r16)] = pooled;   // (cell c, tile t) sits at c*3136 + t*64 = T*64
        }
        // The MFMA stream above starves the producer waves (f32 MFMA and VALU share the SIMD's issue); what is left of the
        // step is VALU-only, and two waves per SIMD issue VALU faster than one: take half of the next input transform.
        if (m + 1 < NM) transform(m + 1, 256, 512);
        __syncthreads();
    }
}

// ---------------------------------------------------------------------------------------------------
// k_conv_features_wsplit (experimental, SV_CNN_X_WSPLIT; measured 0.83 ms against the default's 0.81 ms): the Winograd stream
// with its 16 GEMMs on the bf16 matrix pipe at f32 accuracy.
// On gfx950 an f32 MFMA runs at the VALU's rate and blocks the SIMD's VALU issue while it runs (profiles/
// r01_ubench_mfma_valu_coexec.txt); v_mfma_f32_16x16x32_bf16 does 8x the work per cycle and co-issues with VALU work.  Each
// f32 operand is therefore split, without error, into three bf16 parts (x = h + m + l, each the next 8 mantissa bits, by
// truncation: one v_and + one exact v_sub per part), and a product a*b becomes the six partial products
// ah*bh + ah*bm + am*bh + ah*bl + al*bh + am*bm accumulated in f32 -- what is dropped (am*bl, al*bm, al*bl) is below 2^-23 of
// the product, i.e. below f32's own rounding.  U is split once on the host (sv_load_weights_f32), V when the input transform
// writes it.  6 MFMAs of K = 32 replace 8 of K = 4: 96 matrix-pipe cycles per (M tile, N tile, xi) instead of 256.
//
// All 8 waves do everything (no producer/consumer split; the 96-register B image of a wave is (N tile nt, half of the xi)):
//   phase A  every thread: one (channel, tile) item of V = B^T d B for M tile m, split and stored as bf16 [part][xi][tile][ic];
//            waves 0-3 first finish M tile m-1: add the other half's partial output transform, bias, ReLU, pool, store
//   barrier
//   phase B  waves 0-3: 48 MFMAs (their 8 xi), partial output transform, then their conv1 share;
//            waves 4-7: conv1 share first, then 48 MFMAs, partial output transform -> LDS
//            (waves w and w+4 share a SIMD, so one's MFMAs run beside the other's VALU work)
//   barrier
// conv1: wave = 4-channel group (weights wave-uniform: scalar registers), lanes = pooled pixels, spread over the 3-4 steps in
// which its plane buffer is free (wsplit_conv1_rounds); input staging in two phases as in k_conv_features_wstream.
// ---------------------------------------------------------------------------------------------------
__host__ __device__ inline int wsplit_conv1_rounds(long m, long tc, int done, int total)
{
    if (16 * m + 15 < 49 * (tc - 2) + 48) return 0;          // M tile m (transformed in this step's phase A) still short of cell tc-2's last tile
    const long left = (49 * tc) / 16 - m;                      // steps m .. F-1, F = first M tile that touches cell tc
    const int remaining = total - done;
    return left <= 1 ? remaining : (int)((remaining + left - 1) / left);
}

typedef __bf16 bf16x8_t __attribute__((ext_vector_type(8)));

template <bool U8IN>
__global__ __launch_bounds__(512, 2) void k_conv_features_wsplit(const void *__restrict__ xin, long B, long cells_per_wg,
                                                                 const float *__restrict__ w1, const float *__restrict__ b1,
                                                                 const uint4 *__restrict__ usplit, const float *__restrict__ b2,
                                                                 float *__restrict__ feat)
{
    constexpr int VROW = 80;                  // bytes per (xi, tile): 32 bf16 channels + 16 B of bank skew
    constexpr int VPLANE = 16 * VROW;         // one xi: 16 tiles
    constexpr int VPART = 16 * VPLANE;        // one bf16 part: 16 xi
    __shared__ __attribute__((aligned(16))) float lds[2 * IN_CELL + 2 * C1_CELL];
    __shared__ __attribute__((aligned(16))) unsigned char v3[3 * VPART];
    __shared__ float ypart[4][16][64];        // partial output transforms of waves 4-7: [N tile][4 tiles x 4 values][lane]
    __shared__ __attribute__((aligned(16))) float w1s[32][12];   // conv1 weights [0..8] and bias [9] (broadcast ds_reads; wave-uniform rows)
    float *in_base = lds, *c1_base = lds + 2 * IN_CELL;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int nt = wave & 3, xh = wave >> 2;
    const int r16 = lane & 15, q = lane >> 4;

    const long c0 = (long)blockIdx.x * cells_per_wg;
    long ncell = B - c0;
    if (ncell > cells_per_wg) ncell = cells_per_wg;
    if (ncell <= 0) return;
    const int ntiles = (int)ncell * 49, NM = (ntiles + 15) / 16;
    float *featw = feat + c0 * FEAT;

    uint4 breg[8][3];                         // [xi within this wave's half][part]: B[k = 8q + j][col r16] of U[xi], oc = 16nt + r16
#pragma unroll
    for (int j = 0; j < 8; j++)
#pragma unroll
        for (int p = 0; p < 3; p++) breg[j][p] = usplit[((nt * 16 + 8 * xh + j) * 3 + p) * 64 + lane];
    const float bias2 = b2[16 * nt + r16];
    for (int i = tid; i < 2 * IN_CELL + 2 * C1_CELL; i += 512) lds[i] = 0.f;
    for (int i = tid; i < 3 * VPART / 4; i += 512) ((unsigned *)v3)[i] = 0;
    for (int i = tid; i < 320; i += 512) {
        const int oc = i / 10, t = i - 10 * oc;
        const float w = t < 9 ? w1[oc * 9 + t] : b1[oc];
        w1s[oc][t] = w;
    }
    __syncthreads();

    unsigned sraw[2];
    auto stage_load = [&](long c) {
        if (U8IN) {
            if (tid < 196) sraw[0] = ((const unsigned *)((const u8 *)xin + (c0 + c) * 784))[tid];
        } else {
#pragma unroll
            for (int j = 0; j < 2; j++)
                if (tid + 512 * j < 784) sraw[j] = __float_as_uint(((const float *)xin)[(c0 + c) * 784 + tid + 512 * j]);
        }
    };
    auto stage_store = [&](long c) {
        float *in_s = in_base + (c & 1) * IN_CELL;
        if (U8IN) {
            if (tid < 196) {
                const int y = tid / 7, x = 4 * (tid - 7 * y);
                float *d = in_s + (y + 1) * IN_W + x + 1;
#pragma unroll
                for (int j = 0; j < 4; j++) d[j] = glue_norm((u8)(sraw[0] >> (8 * j)));
            }
        } else {
#pragma unroll
            for (int j = 0; j < 2; j++) {
                const int i = tid + 512 * j;
                if (i < 784) { const int y = i / 28, x = i - y * 28; in_s[(y + 1) * IN_W + x + 1] = __uint_as_float(sraw[j]); }
            }
        }
    };
    constexpr int C1_ROUNDS = 4;              // 196 pooled pixels in rounds of 64 lanes; the wave is the 4-channel group
    auto conv1 = [&](long c, int r0, int r1) {
        const float *in_s = in_base + (c & 1) * IN_CELL;
        float *c1 = c1_base + (c & 1) * C1_CELL + wave * 4 * PLANE;
        for (int pp = lane + 64 * r0; pp < 196 && pp < 64 * r1; pp += 64) {
            const int py = pp / 14, px = pp - py * 14;
            // plain v_fma_f32 here, not v_pk_fma_f32: a packed f32 op issued beside the other wave's MFMAs costs ~20 cycles more
            float pt[4][4];
            const float *src = in_s + (2 * py) * IN_W + 2 * px;
#pragma unroll
            for (int i = 0; i < 4; i++)
#pragma unroll
                for (int j = 0; j < 4; j++) pt[i][j] = src[i * IN_W + j];
            float *dstp = c1 + (py + 1) * 16 + px + 1;
#pragma unroll
            for (int o = 0; o < 4; o++) {
                const float *w = w1s[wave * 4 + o];
                float a00 = w[9], a01 = w[9], a10 = w[9], a11 = w[9];
#pragma unroll
                for (int ky = 0; ky < 3; ky++)
#pragma unroll
                    for (int kx = 0; kx < 3; kx++) {
                        const float wv = w[ky * 3 + kx];
                        a00 = __builtin_fmaf(wv, pt[ky][kx], a00);
                        a01 = __builtin_fmaf(wv, pt[ky][kx + 1], a01);
                        a10 = __builtin_fmaf(wv, pt[ky + 1][kx], a10);
                        a11 = __builtin_fmaf(wv, pt[ky + 1][kx + 1], a11);
                    }
                dstp[o * PLANE] = fmaxf(fmaxf(fmaxf(a00, a01), fmaxf(a10, a11)), 0.f);
            }
        }
    };
    // one (channel, tile) item of V = B^T d B for M tile m, each value split into three bf16 parts
    auto transform = [&](int m) {
        const int ic = tid >> 4, tl = tid & 15;
        int T = 16 * m + tl;
        if (T > ntiles - 1) T = ntiles - 1;
        const int c = T / 49;
        const int t = T - 49 * c, wy = t / 7, wx = t - 7 * wy;
        const float *d = c1_base + (c & 1) * C1_CELL + ic * PLANE + (2 * wy) * 16 + 2 * wx;
        float tt[4][4];
#pragma unroll
        for (int x = 0; x < 4; x++) {
            const float d0 = d[x], d1 = d[16 + x], d2 = d[32 + x], d3 = d[48 + x];
            tt[0][x] = d0 - d2; tt[1][x] = d1 + d2; tt[2][x] = d2 - d1; tt[3][x] = d1 - d3;
        }
        unsigned char *vp = v3 + tl * VROW + ic * 2;
#pragma unroll
        for (int y = 0; y < 4; y++) {
            const float v4[4] = {tt[y][0] - tt[y][2], tt[y][1] + tt[y][2], tt[y][2] - tt[y][1], tt[y][1] - tt[y][3]};
#pragma unroll
            for (int x = 0; x < 4; x++) {
                const float v = v4[x];
                const unsigned uh = __float_as_uint(v) & 0xffff0000u;
                const float r = v - __uint_as_float(uh);                       // exact
                const unsigned um = __float_as_uint(r) & 0xffff0000u;
                const float r2 = r - __uint_as_float(um);                      // exact; its top 16 bits are the third part
                unsigned char *o = vp + (y * 4 + x) * VPLANE;
                *(unsigned short *)o = (unsigned short)(uh >> 16);
                *(unsigned short *)(o + VPART) = (unsigned short)(um >> 16);
                *(unsigned short *)(o + 2 * VPART) = (unsigned short)(__float_as_uint(r2) >> 16);
            }
        }
    };

    // prologue: cells 0 and 1 convolved, cell 2 staged
    long conv_done = ncell > 1 ? 1 : 0;
    int conv_round = 0;
    stage_load(0); stage_store(0);
    __syncthreads();
    conv1(0, 0, C1_ROUNDS);
    if (ncell > 1) { stage_load(1); stage_store(1); }
    __syncthreads();
    if (ncell > 1) conv1(1, 0, C1_ROUNDS);
    if (ncell > 2) { stage_load(2); stage_store(2); }
    __syncthreads();

    float p0[4][4];                           // waves 0-3: this wave's partial output transform, [tile reg][y00, y01, y10, y11]
    auto finish = [&](int m) {                // waves 0-3: M tile m's outputs = own partial + the other half's (in LDS)
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const int T = 16 * m + 4 * q + reg;
            const float y00 = p0[reg][0] + ypart[nt][reg * 4 + 0][lane], y01 = p0[reg][1] + ypart[nt][reg * 4 + 1][lane];
            const float y10 = p0[reg][2] + ypart[nt][reg * 4 + 2][lane], y11 = p0[reg][3] + ypart[nt][reg * 4 + 3][lane];
            const float pooled = fmaxf(fmaxf(fmaxf(y00, y01), fmaxf(y10, y11)) + bias2, 0.f);
            if (T < ntiles) featw[(unsigned)(T * 64 + 16 * nt + r16)] = pooled;
        }
    };

    for (int m = 0; m < NM; m++) {
        // ---- phase A
        if (xh == 0 && m > 0) finish(m - 1);
        transform(m);
        __syncthreads();
        // ---- phase B
        const long tc = conv_done + 1;
        const int rounds = tc < ncell ? wsplit_conv1_rounds(m, tc, conv_round, C1_ROUNDS) : 0;
        const bool completes = rounds > 0 && conv_round + rounds == C1_ROUNDS;
        if (completes && tc + 1 < ncell) stage_load(tc + 1);
        if (xh == 1 && rounds > 0) conv1(tc, conv_round, conv_round + rounds);

        f32x4 acc[8];
        {
            const unsigned char *ap = v3 + r16 * VROW + q * 16;
#pragma unroll
            for (int j = 0; j < 8; j++) {
                const unsigned char *a = ap + (8 * xh + j) * VPLANE;
                const bf16x8_t ah = __builtin_bit_cast(bf16x8_t, *(const uint4 *)a);
                const bf16x8_t am = __builtin_bit_cast(bf16x8_t, *(const uint4 *)(a + VPART));
                const bf16x8_t al = __builtin_bit_cast(bf16x8_t, *(const uint4 *)(a + 2 * VPART));
                const bf16x8_t bh = __builtin_bit_cast(bf16x8_t, breg[j][0]), bm = __builtin_bit_cast(bf16x8_t, breg[j][1]),
                               bl = __builtin_bit_cast(bf16x8_t, breg[j][2]);
                f32x4 s = {0.f, 0.f, 0.f, 0.f};                               // smallest terms first
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(al, bh, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bl, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bm, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(am, bh, s, 0, 0, 0);
                s = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bm, s, 0, 0, 0);
                acc[j] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(ah, bh, s, 0, 0, 0);
            }
        }
        // partial output transform over this wave's two rows of the 4x4 xi grid (xi = 4x + y, x in {2xh, 2xh+1}):
        // r[x][0] = M[x][0] + M[x][1] + M[x][2], r[x][1] = M[x][1] - M[x][2] - M[x][3];
        // Y[0][b] = r[0][b] + r[1][b] + r[2][b], Y[1][b] = r[1][b] - r[2][b] - r[3][b]
#pragma unroll
        for (int reg = 0; reg < 4; reg++) {
            const float ra0 = acc[0][reg] + acc[1][reg] + acc[2][reg], ra1 = acc[1][reg] - acc[2][reg] - acc[3][reg];   // first row of the half
            const float rb0 = acc[4][reg] + acc[5][reg] + acc[6][reg], rb1 = acc[5][reg] - acc[6][reg] - acc[7][reg];   // second row
            if (xh == 0) {                    // rows x = 0, 1: Y[0][b] += r0 + r1, Y[1][b] += r1
                p0[reg][0] = ra0 + rb0; p0[reg][1] = ra1 + rb1; p0[reg][2] = rb0; p0[reg][3] = rb1;
            } else {                          // rows x = 2, 3: Y[0][b] += r2, Y[1][b] += -r2 - r3
                ypart[nt][reg * 4 + 0][lane] = ra0; ypart[nt][reg * 4 + 1][lane] = ra1;
                ypart[nt][reg * 4 + 2][lane] = -ra0 - rb0; ypart[nt][reg * 4 + 3][lane] = -ra1 - rb1;
            }
        }
        if (xh == 0 && rounds > 0) conv1(tc, conv_round, conv_round + rounds);
        if (rounds > 0) {
            conv_round += rounds;
            if (completes) {
                if (tc + 1 < ncell) stage_store(tc + 1);
                conv_done = tc;
                conv_round = 0;
            }
        }
        __syncthreads();
    }
    if (xh == 0) finish(NM - 1);
}

#endif  // SV_XCHECK

// 64 cells per workgroup, 16 per wave; K = 3136 in 196 chunks of 16.
__global__ __launch_bounds__(256) void k_fc_head(const float *__restrict__ feat, long B, const float *__restrict__ w1reg,
                                                 const float *__restrict__ b1, const float *__restrict__ w2,
                                                 const float *__restrict__ b2, float *__restrict__ logits,
                                                 u8 *__restrict__ digits, float *__restrict__ conf, const int *__restrict__ run_if_set)
{
    if (run_if_set && *run_if_set == 0) return;
    __shared__ float hs[4][16][129];
    __shared__ float w2s[10][128];
    __shared__ float lg[4][16][12];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const long cell0 = (long)blockIdx.x * 64 + wave * 16;
    long crow = cell0 + r;
    if (crow >= B) crow = B - 1;
    const f32x4 *ap = (const f32x4 *)(feat + crow * FEAT + 4 * q);
    const f32x4 *bp = (const f32x4 *)w1reg + lane;

    for (int i = tid; i < 1280; i += 256) w2s[i >> 7][i & 127] = w2[i];

    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    // explicit two-deep register pipeline: the 9 loads of chunk c+2 are issued before the MFMAs of chunk c
    f32x4 a0 = ap[0], a1 = ap[4], wb0[8], wb1[8];
#pragma unroll
    for (int t = 0; t < 8; t++) { wb0[t] = bp[t * 64]; wb1[t] = bp[(8 + t) * 64]; }
    for (int c = 0; c < 196; c += 2) {
        f32x4 na0 = a0, na1 = a1, nwb0[8], nwb1[8];
        const int c2 = c + 2 < 196 ? c + 2 : c, c3 = c + 3 < 196 ? c + 3 : c + 1;   // (tail: harmless re-loads)
        na0 = ap[c2 * 4];
#pragma unroll
        for (int t = 0; t < 8; t++) nwb0[t] = bp[(c2 * 8 + t) * 64];
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int t = 0; t < 8; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a0[e], wb0[t][e], acc[t], 0, 0, 0);
        na1 = ap[c3 * 4];
#pragma unroll
        for (int t = 0; t < 8; t++) nwb1[t] = bp[(c3 * 8 + t) * 64];
#pragma unroll
        for (int e = 0; e < 4; e++)
#pragma unroll
            for (int t = 0; t < 8; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a1[e], wb1[t][e], acc[t], 0, 0, 0);
        a0 = na0; a1 = na1;
#pragma unroll
        for (int t = 0; t < 8; t++) { wb0[t] = nwb0[t]; wb1[t] = nwb1[t]; }
    }

    // acc[t][reg]: cell row 4q+reg, hidden unit 16t + r
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const float bias = b1[16 * t + r];
#pragma unroll
        for (int reg = 0; reg < 4; reg++) hs[wave][4 * q + reg][16 * t + r] = fmaxf(acc[t][reg] + bias, 0.f);
    }
    __syncthreads();

    // fc2: lane (cell r, class group q) -> classes q, q+4, q+8
    for (int jj = 0; jj < 3; jj++) {
        const int j = q + 4 * jj;
        if (j < 10) {
            float s = b2[j];
            for (int n = 0; n < 128; n++) s = __builtin_fmaf(hs[wave][r][n], w2s[j][n], s);
            lg[wave][r][j] = s;
            if (cell0 + r < B) logits[(cell0 + r) * 10 + j] = s;
        }
    }
    __syncthreads();
    if (q == 0 && cell0 + r < B && (digits || conf)) {
        float best = lg[wave][r][0];
        int arg = 0;
        for (int j = 1; j < 10; j++)
            if (lg[wave][r][j] > best) { best = lg[wave][r][j]; arg = j; }
        if (digits) digits[cell0 + r] = (u8)arg;
        if (conf) {
            float den = 0.f;
            for (int j = 0; j < 10; j++) den += expf(lg[wave][r][j] - best);
            conf[cell0 + r] = 1.0f / den;
        }
    }
}

#ifdef SV_XCHECK
// ---------------------------------------------------------------------------------------------------
// k_fc_head_frame: one 512-thread workgroup per 81 cells (one frame), i.e. one per CU at 256 frames, so the MFMA work is
// spread evenly, and the fc1 weight image is fetched ONCE per workgroup: each 32-wide K stage (16 KB) goes global ->
// registers -> LDS (double-buffered, one barrier per stage) and all waves read their B operands from LDS.  The plain
// k_fc_head streams the whole 1.6 MB image through every wave and is bound by L1 bandwidth (and by 324 workgroups on 256 CUs).
// 81 cells = 6 M tiles: waves 0-3 take M tiles 0-3 (all 8 N tiles), waves 4-7 take M tiles 4,5 split in two halves of N,
// which gives every SIMD (waves s and s+4) three half-tile jobs.
// ---------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(512) void k_fc_head_frame(const float *__restrict__ feat, long B, const float *__restrict__ w1reg,
                                                       const float *__restrict__ b1, const float *__restrict__ w2,
                                                       const float *__restrict__ b2, float *__restrict__ logits,
                                                       u8 *__restrict__ digits, float *__restrict__ conf)
{
    constexpr int CELLS = 81, ROWS = 96, CH = 4, WPT = CH * 8 * 64 / 512;   // WPT float4 of weights per thread per stage                      // cells per workgroup, padded rows, 16-wide chunks per stage
    __shared__ __attribute__((aligned(16))) f32x4 wt[2][CH * 8 * 64]; // 2 x 32 KB weight stages
    __shared__ float hs[ROWS][129];
    __shared__ float w2s[10][128];
    __shared__ float lg[ROWS][12];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int mt = wave < 4 ? wave : 4 + ((wave - 4) >> 1);           // M tile of this wave
    const int t0 = wave < 4 ? 0 : 4 * ((wave - 4) & 1);               // first N tile
    const int nt_cnt = wave < 4 ? 8 : 4;                              // N tiles of this wave
    const long cellbase = (long)blockIdx.x * CELLS;
    long crow = cellbase + mt * 16 + r;
    const long last = (cellbase + CELLS < B ? cellbase + CELLS : B) - 1;
    if (crow > last) crow = last;
    const f32x4 *ap = (const f32x4 *)(feat + crow * FEAT + 4 * q);
    const f32x4 *wp = (const f32x4 *)w1reg;                           // [196][8][64] float4

    for (int i = tid; i < 1280; i += 512) w2s[i >> 7][i & 127] = w2[i];

    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};

    constexpr int NSTAGE = 196 / CH;                                  // 49 stages of 2048 float4 = 4 per thread
    f32x4 wreg[WPT], areg[CH];
#pragma unroll
    for (int j = 0; j < WPT; j++) wreg[j] = wp[512 * j + tid];
#pragma unroll
    for (int c = 0; c < CH; c++) areg[c] = ap[c * 4];
#pragma unroll
    for (int j = 0; j < WPT; j++) wt[0][512 * j + tid] = wreg[j];
    __syncthreads();

    for (int st = 0; st < NSTAGE; st++) {
        const int cur = st & 1;
        f32x4 a[CH];
#pragma unroll
        for (int c = 0; c < CH; c++) a[c] = areg[c];
        if (st + 1 < NSTAGE) {                                        // next stage: global -> registers while this one computes
#pragma unroll
            for (int j = 0; j < WPT; j++) wreg[j] = wp[(long)(st + 1) * (CH * 8 * 64) + 512 * j + tid];
#pragma unroll
            for (int c = 0; c < CH; c++) areg[c] = ap[((st + 1) * CH + c) * 4];
        }
#pragma unroll
        for (int c = 0; c < CH; c++) {
            if (nt_cnt == 8) {
                f32x4 b[8];
#pragma unroll
                for (int t = 0; t < 8; t++) b[t] = wt[cur][(c * 8 + t) * 64 + lane];
#pragma unroll
                for (int e = 0; e < 4; e++)
#pragma unroll
                    for (int t = 0; t < 8; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][e], b[t][e], acc[t], 0, 0, 0);
            } else {
                f32x4 b[4];
#pragma unroll
                for (int t = 0; t < 4; t++) b[t] = wt[cur][(c * 8 + t0 + t) * 64 + lane];
#pragma unroll
                for (int e = 0; e < 4; e++)
#pragma unroll
                    for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x4f32(a[c][e], b[t][e], acc[t], 0, 0, 0);
            }
        }
        if (st + 1 < NSTAGE) {
#pragma unroll
            for (int j = 0; j < WPT; j++) wt[cur ^ 1][512 * j + tid] = wreg[j];
        }
        __syncthreads();
    }

    // acc[t][reg]: cell row mt*16 + 4q + reg, hidden unit 16*(t0 + t) + r
#pragma unroll
    for (int t = 0; t < 8; t++)
        if (t < nt_cnt) {
            const int n = 16 * (t0 + t) + r;
            const float bias = b1[n];
#pragma unroll
            for (int reg = 0; reg < 4; reg++) hs[mt * 16 + 4 * q + reg][n] = fmaxf(acc[t][reg] + bias, 0.f);
        }
    __syncthreads();
    for (int it = tid; it < CELLS * 10; it += 512) {                  // fc2
        const int cl = it / 10, j = it - 10 * cl;
        float sacc = b2[j];
        for (int n = 0; n < 128; n++) sacc = __builtin_fmaf(hs[cl][n], w2s[j][n], sacc);
        lg[cl][j] = sacc;
        if (cellbase + cl < B) logits[(cellbase + cl) * 10 + j] = sacc;
    }
    __syncthreads();
    if (tid < CELLS && cellbase + tid < B && (digits || conf)) {
        float best = lg[tid][0];
        int arg = 0;
        for (int j = 1; j < 10; j++)
            if (lg[tid][j] > best) { best = lg[tid][j]; arg = j; }
        if (digits) digits[cellbase + tid] = (u8)arg;
        if (conf) {
            float den = 0.f;
            for (int j = 0; j < 10; j++) den += expf(lg[tid][j] - best);
            conf[cellbase + tid] = 1.0f / den;
        }
    }
}

#endif  // SV_XCHECK

// preprocess_cell (pipeline/run.py:73-95) as a stand-alone call: one wave per cell, u8 in -> u8 {0,255} out
__global__ __launch_bounds__(256) void k_preprocess_cells(const u8 *__restrict__ cells, long B, GaussTaps taps, u8 *__restrict__ out)
{
    __shared__ N1Scratch sc[4];
    __shared__ float tile[4][IN_CELL];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const long c = (long)blockIdx.x * 4 + wave;
    if (c >= B) return;
    n1_preprocess_cell(cells + c * 784, tile[wave], sc[wave], lane, taps.k);
    wave_lds_fence();
    for (int i = lane; i < 784; i += 64) {
        const int y = i / 28, x = i - 28 * y;
        out[c * 784 + i] = tile[wave][(y + 1) * IN_W + x + 1] < 0.f ? 255 : 0;
    }
}

}  // namespace

int svk_preprocess_cells(const u8 *cells, long B, u8 *out, hipStream_t s)
{
    GaussTaps taps;
    sv_gaussian_taps_f32(11, taps.k);
    hipLaunchKernelGGL(k_preprocess_cells, dim3((unsigned)((B + 3) / 4)), dim3(256), 0, s, cells, B, taps, out);
    SV_LAUNCH_CHECK("k_preprocess_cells");
    return SV_OK;
}

// F.softmax + topk (pipeline/run_v2.py:165-178): one thread per cell, 10 logits in registers, k selection passes
__global__ __launch_bounds__(256) void k_softmax_topk(const float *__restrict__ logits, long B, int k, u8 *__restrict__ index, float *__restrict__ prob)
{
    const long cell = (long)blockIdx.x * 256 + threadIdx.x;
    if (cell >= B) return;
    float p[10];
    float best = -INFINITY;
#pragma unroll
    for (int j = 0; j < 10; j++) { p[j] = logits[cell * 10 + j]; best = fmaxf(best, p[j]); }
    float den = 0.f;
#pragma unroll
    for (int j = 0; j < 10; j++) { p[j] = expf(p[j] - best); den += p[j]; }
    unsigned taken = 0;
    for (int r = 0; r < k; r++) {
        float top = -1.f;
        int arg = 0;
#pragma unroll
        for (int j = 0; j < 10; j++)
            if (!(taken >> j & 1) && p[j] > top) { top = p[j]; arg = j; }
        taken |= 1u << arg;
        index[cell * k + r] = (u8)arg;
        prob[cell * k + r] = top / den;
    }
}

int svk_softmax_topk(const float *logits, long B, int k, u8 *index, float *prob, hipStream_t s)
{
    hipLaunchKernelGGL(k_softmax_topk, dim3((unsigned)((B + 255) / 256)), dim3(256), 0, s, logits, B, k, index, prob);
    SV_LAUNCH_CHECK("k_softmax_topk");
    return SV_OK;
}

// Which conv/fc kernels run (sv_ctx_set_cnn_kernels; include/sudoku_vision_hip.h):
//   SV_CNN_AUTO (default)  the f16 hi/lo operand-pair kernels (k3_cnn_h2.hip) whenever the loaded weights keep every activation inside their
//                          range, else the f32-MFMA kernels below, which have the reference's own range (sv_load_weights_f32 decides; f32
//                          inputs are range-checked on the device per call)
//   SV_CNN_F16PAIR / SV_CNN_F32MFMA   one of the two, unconditionally
//   (libsudokuvision_xcheck.so only) SV_CNN_X_WINOGRAD, SV_CNN_X_WSPLIT: the round-1 Winograd stream on f32 / on split-bf16 MFMA, and
//   SV_FC_X_FRAME for the one-workgroup-per-frame fc kernel -- independent implementations the tests compare the product with
static int effective_algo(const sv_ctx *ctx)
{
    switch (ctx->cnn_kernels) {
    case SV_CNN_F16PAIR: return 4;
    case SV_CNN_F32MFMA: return 0;
#ifdef SV_XCHECK
    case SV_CNN_X_WINOGRAD: return 2;
    case SV_CNN_X_WSPLIT: return 3;
#endif
    default: return ctx->w.h2_in_range ? 4 : 0;
    }
}

extern "C" int sv_ctx_set_cnn_kernels(sv_ctx *ctx, int which)
{
    if (!ctx) return sv_fail(SV_ERR_BAD_ARG, "sv_ctx_set_cnn_kernels: NULL context");
    bool ok = which == SV_CNN_AUTO || which == SV_CNN_F16PAIR || which == SV_CNN_F32MFMA;
#ifdef SV_XCHECK
    ok = ok || which == SV_CNN_X_WINOGRAD || which == SV_CNN_X_WSPLIT;
#endif
    if (!ok) return sv_fail(SV_ERR_BAD_ARG, "sv_ctx_set_cnn_kernels: unknown selection %d", which);
    ctx->cnn_kernels = which;
    return SV_OK;
}

#ifdef SV_XCHECK
extern "C" int svx_ctx_set_fc_frame_kernel(sv_ctx *ctx, int on)
{
    if (!ctx) return sv_fail(SV_ERR_BAD_ARG, "svx_ctx_set_fc_frame_kernel: NULL context");
    ctx->x_fc_frame = on != 0;
    return SV_OK;
}
#endif

extern "C" int sv_conv_kernel_info(sv_ctx *ctx, int *algo, int *mfma_f32_conv2_per_cell, int *mfma_f32_conv1_per_cell, int *mfma_f16_conv_per_cell,
                                   int *mfma_f16_fc_per_cell)
{
    if (!ctx || !algo || !mfma_f32_conv2_per_cell || !mfma_f32_conv1_per_cell || !mfma_f16_conv_per_cell || !mfma_f16_fc_per_cell)
        return sv_fail(SV_ERR_BAD_ARG, "sv_conv_kernel_info: NULL argument");
    const int a = effective_algo(ctx);
    *algo = a;
    // f32 Winograd: 49 tiles x 16 xi x 8 k-steps x 4 N tiles / 16 tiles per M tile; f32 direct: 196 positions / 16 x 72 k-steps x 4 N tiles
    *mfma_f32_conv2_per_cell = a == 2 ? 1568 : ((a == 3 || a == 4) ? 0 : 3600);
    *mfma_f32_conv1_per_cell = 0;
    // f16 pairs: conv2 13 M tiles x 9 taps x 4 N tiles x 3 products + conv1 13 M tiles x 8 N tiles x 2; fc1 98 k-steps x 8 N tiles x 3 per 16 cells
    *mfma_f16_conv_per_cell = a == 4 ? 13 * 9 * 4 * 3 + 13 * 8 * 2 : 0;
    *mfma_f16_fc_per_cell = a == 4 ? 98 * 8 * 3 / 16 : 0;
    return SV_OK;
}

// |x| of an f32 input batch against the range the f16-pair kernels carry exactly: flag = 0 (use them) when lo <= max|x| <= hi and no
// value is NaN/Inf, else 1 (the f32-MFMA kernels).  One pass over the input; both kernel pairs are launched and the one the flag does
// not name returns at once, so the decision never costs a host synchronisation.
__global__ __launch_bounds__(256) void k_input_range(const float *__restrict__ x, long n, float lo, float hi, int *__restrict__ flag)
{
    __shared__ float part[4];
    float m = 0.f;
    bool bad = false;
    for (long i = (long)blockIdx.x * 256 + threadIdx.x; i < n; i += (long)gridDim.x * 256) {
        const float v = fabsf(x[i]);
        bad |= !(v <= hi);                                          // also NaN
        m = fmaxf(m, v);
    }
    for (int d = 32; d > 0; d >>= 1) m = fmaxf(m, __shfl_xor(m, d));
    if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6] = m;
    if (__any(bad) && (threadIdx.x & 63) == 0) atomicOr(flag, 1);
    __syncthreads();
    if (threadIdx.x == 0) {
        m = fmaxf(fmaxf(part[0], part[1]), fmaxf(part[2], part[3]));
        if (m >= lo) atomicOr(flag + 1, 1);                         // some block saw a value of at least lo
    }
}
__global__ void k_input_range_finish(int *flag)                     // flag[0] |= nothing reached lo (all-tiny input); flag[1] is scratch
{
    if (flag[1] == 0) flag[0] = 1;
}

int svk_cnn_forward(sv_ctx *ctx, const void *x, bool x_is_u8, int glue, long B, float *logits, u8 *digits, float *conf, hipStream_t s)
{
    const sv_weights &w = ctx->w;
    const long npairs = (B + 1) / 2;
    if (x_is_u8 && glue == SV_GLUE_RUNPY) {      // run.py's preprocess_cell as its own pass; its {0,255} output then takes the plain glue
        int rc = svk_preprocess_cells((const u8 *)x, B, ctx->cells2, s);
        if (rc) return rc;
        x = ctx->cells2;
    }
    int conv_algo = effective_algo(ctx);
    // (the kernels that read 8-bit cells as dwords need a 4-byte-aligned buffer; a misaligned one takes the direct f32 kernel)
    if (x_is_u8 && ((uintptr_t)x & 3)) conv_algo = 0;
    const int *flag = nullptr;                   // device flag: 0 = the f16-pair kernels run, 1 = the f32-MFMA kernels (both are launched)
    if (conv_algo == 4 && !x_is_u8 && ctx->cnn_kernels == SV_CNN_AUTO) {
        // 8-bit cells are in [-1, 1] after the glue (what sv_load_weights_f32 sized the activations for); f32 input can be anything
        if (!ctx->range_flag) SV_HIP(hipMalloc((void **)&ctx->range_flag, 2 * sizeof(int)));
        SV_HIP(hipMemsetAsync(ctx->range_flag, 0, 2 * sizeof(int), s));
        const long n = B * 784;
        hipLaunchKernelGGL(k_input_range, dim3((unsigned)(n / 4096 + 1 < 1024 ? n / 4096 + 1 : 1024)), dim3(256), 0, s, (const float *)x, n, w.h2_x_lo, w.h2_x_hi, ctx->range_flag);
        hipLaunchKernelGGL(k_input_range_finish, dim3(1), dim3(1), 0, s, ctx->range_flag);
        SV_LAUNCH_CHECK("k_input_range");
        flag = ctx->range_flag;
    }
    if (conv_algo == 4) {
        const int rc = svk_cnn_forward_h2(ctx, x, x_is_u8, B, logits, digits, conf, flag, s);
        if (rc || !flag) return rc;
    }
#ifdef SV_XCHECK
    if (conv_algo == 3 || conv_algo == 2) {
        long cpw = (B + ctx->num_cus - 1) / ctx->num_cus;
        if (cpw < 1) cpw = 1;
        const int grid_s = (int)((B + cpw - 1) / cpw);
        sv_time_scope ts(ctx, SVK_CONV_FEATURES, s);
        if (conv_algo == 3) {
            if (x_is_u8)
                hipLaunchKernelGGL(k_conv_features_wsplit<true>, dim3(grid_s), dim3(512), 0, s, x, B, cpw, w.conv1_w, w.conv1_b, (const uint4 *)w.conv2_wsplit, w.conv2_b, ctx->features);
            else
                hipLaunchKernelGGL(k_conv_features_wsplit<false>, dim3(grid_s), dim3(512), 0, s, x, B, cpw, w.conv1_w, w.conv1_b, (const uint4 *)w.conv2_wsplit, w.conv2_b, ctx->features);
        } else {
            if (x_is_u8)
                hipLaunchKernelGGL(k_conv_features_wstream<true>, dim3(grid_s), dim3(512), 0, s, x, B, cpw, w.conv1_w, w.conv1_b, w.conv2_wino, w.conv2_b, ctx->features);
            else
                hipLaunchKernelGGL(k_conv_features_wstream<false>, dim3(grid_s), dim3(512), 0, s, x, B, cpw, w.conv1_w, w.conv1_b, w.conv2_wino, w.conv2_b, ctx->features);
        }
    } else
#endif
    {
        const int grid_pc = (int)(npairs < (long)ctx->num_cus ? npairs : (long)ctx->num_cus);
        sv_time_scope ts(ctx, SVK_CONV_FEATURES, s);
        if (x_is_u8)
            hipLaunchKernelGGL(k_conv_features_pc<true>, dim3(grid_pc), dim3(512), 0, s, x, B, w.conv1_w, w.conv1_b, w.conv2_wreg, w.conv2_b, ctx->features, flag);
        else
            hipLaunchKernelGGL(k_conv_features_pc<false>, dim3(grid_pc), dim3(512), 0, s, x, B, w.conv1_w, w.conv1_b, w.conv2_wreg, w.conv2_b, ctx->features, flag);
    }
    SV_LAUNCH_CHECK("k_conv_features");
    sv_time_scope ts(ctx, SVK_FC_HEAD, s);
#ifdef SV_XCHECK
    if (ctx->x_fc_frame && B >= 81 * 64)       // enough frames to give most CUs a workgroup
        hipLaunchKernelGGL(k_fc_head_frame, dim3((unsigned)((B + 80) / 81)), dim3(512), 0, s, ctx->features, B, w.fc1_wreg, w.fc1_b, w.fc2_w, w.fc2_b, logits, digits, conf);
    else
#endif
    hipLaunchKernelGGL(k_fc_head, dim3((unsigned)((B + 63) / 64)), dim3(256), 0, s, ctx->features, B, w.fc1_wreg, w.fc1_b, w.fc2_w, w.fc2_b, logits, digits, conf, flag);
    SV_LAUNCH_CHECK("k_fc_head");
    return SV_OK;
}
