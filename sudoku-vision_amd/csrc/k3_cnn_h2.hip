// K3, the default f32-accuracy form: DigitCNN.forward (ml/model.py:34-42) with conv2 and fc1 on the f16 matrix pipe.
//
// Why not the f32 MFMA: on gfx950 v_mfma_f32_16x16x4_f32 runs at the f32 VALU's rate (157 TF) and blocks the SIMD's VALU
// issue while it runs (profiles/r01_ubench_mfma_valu_coexec.txt), so an f32 forward is bounded by the sum of its matrix and
// vector work.  v_mfma_f32_16x16x32_f16 does 16x the work per cycle and co-issues with VALU work.  Each f32 operand x is
// therefore carried as an unevaluated sum of two halves,  x ~= hi + lo,  hi = f16(x), lo = f16(x - hi)  (22 significant bits;
// weights are pre-scaled by a power of two so that lo stays a normal f16), and a product a*w becomes the three partial
// products  ah*wh + ah*wl + al*wh  accumulated in f32 by the MFMA (f16 x f16 products are exact in f32; the dropped al*wl
// is < 2^-22 of the product).  What this does to the logits, measured against an exact (f64) evaluation of the same model on
// the golden inputs: max error 1e-6, the same as PyTorch-CPU's own f32 forward (1.4e-6) -- tests/test_oracle_cnn.py holds the
// simulation, tests/test_gpu_parity.py the measured kernel (<= 1e-4 is the contract, ~1e-6 is what comes out).
// bias, ReLU, pooling, fc2 and the softmax stay in f32 on the VALU.
//
//   k_conv_features_h2 : persistent, one 512-thread workgroup per CU, producer waves (input staging, conv1) and consumer waves
//        (conv2), both on the matrix pipe, software-pipelined over cells with one barrier per cell; a producer and a consumer
//        share each SIMD.  conv1 is a GEMM over the 4x4 input patch of a pooling window (see the producer code); it writes its
//        ReLU/pool output split into two f16 planes, channel-last (position-major, 32 channels = 64 B per position, rows padded
//        against bank conflicts): one ds_read_b128 = one MFMA A operand (8 input channels of one 3x3 tap at one output
//        position).  conv2 as an implicit GEMM: M = 4 pooling windows x 4 positions (so that the 4 accumulator registers of a
//        lane are one pooling window: bias + ReLU + max never leave the lane), N = 2 x 16 channels per wave, K = 9 taps x 32
//        channels; 3 MFMAs per (tap, N tile).  A wave keeps its 9 x 2 x 2 B operands in 144 VGPRs.
//        The pooled, ReLU'd features leave the kernel already split into f16 pairs, 16 B of hi parts and 16 B of lo parts per
//        group of 8: what the fc kernels' MFMAs take as they stand.
//   k_fc_head_h2p : fc1 (3136 -> 128) with cells as M, one workgroup per CU (12 waves = 6 M tiles x 2 N halves), the 1.6 MB weight
//        image (hi and lo halves) streamed once per CU, weights and features global -> LDS by DMA; fc2 + argmax + softmax[argmax]
//        epilogue in f32 (pipeline/run.py:139-143).  Serves every batch whose share per CU fits a 96-cell pass.
//   k_fc_head_h2  : round 2's form (one 16-cell tile per wave, 64 cells per workgroup, the weight image staged once per workgroup
//        through double-buffered LDS, A fragments straight from global memory) for larger batches.
//
// Range: inputs and activations are carried as f16 pairs, so their magnitudes must stay below 65,504 (and a pair holds 22 significant bits
// only while its low half is a normal f16; below that the absolute error is f16's subnormal step, 2^-24).  sv_load_weights_f32 bounds the
// activations from the weights
// and svk_cnn_forward (k3_cnn.hip) routes anything outside to the f32-MFMA kernels; `run_if_clear` is that decision for f32 inputs, made
// on the device.  SV_DEV builds (tools/dev) add an ablation switch and s_memtime stamps; the product is compiled without them.
#include "sv_device.h"
#include "sv_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef _Float16 h8 __attribute__((ext_vector_type(8)));

namespace {

// One f16 plane of a cell: 16x16 zero-bordered positions, 32 channels = 64 B per position, 32 B of padding per row of 16
// positions.  With these strides the 16 lanes of every ds_read_b128 lane group of an MFMA A-operand fetch (4 pooling windows x
// 2x2 positions x 2 of the 4 k-groups) fall on 16 different 16-byte bank slots unless the tile's windows wrap to the next row
// of the 7x7 grid (1.46 LDS cycles per group on average; the 80-B-per-position layout of the bf16 kernel is 4-way conflicted:
// its two pooling-window rows are 1280 B apart, a multiple of the 256-B bank row).
constexpr int POS_STRIDE = 64, ROW_STRIDE = 16 * POS_STRIDE + 32;
constexpr int PLANE_B = 16 * ROW_STRIDE;
constexpr int IN_PLANE_B = 32 * 64;             // one f16 plane of a cell's input: 30x30 zero-bordered pixels in 32 rows of 64 B
constexpr int FEAT = 3136;

__device__ __forceinline__ float glue_norm(u8 c)
{
    // x = ((255 - cell)/255 - 0.5)/0.5, one rounding per operation (pipeline/run.py:129-135)
    const float t = __fdiv_rn((float)(255 - (int)c), 255.0f);
    return __fdiv_rn(__fsub_rn(t, 0.5f), 0.5f);
}

// The compiler puts the s_waitcnt for a value loaded before a loop at its first use INSIDE the loop, where it then also waits, every
// iteration, for whatever the loop itself has in flight (the next cell's input load, the previous tile's feature stores).  Touching the
// registers in an empty asm statement ahead of the loop makes it wait there, once.
__device__ __forceinline__ void settle(uint4 &v) { asm volatile("" : "+v"(v.x), "+v"(v.y), "+v"(v.z), "+v"(v.w)); }
__device__ __forceinline__ void settle(float &v) { asm volatile("" : "+v"(v)); }

__device__ __forceinline__ unsigned pack2h(_Float16 a, _Float16 b)
{
    return (unsigned)__builtin_bit_cast(unsigned short, a) | ((unsigned)__builtin_bit_cast(unsigned short, b) << 16);
}

// v -> (hi, lo) with hi + lo = v to 22 bits
__device__ __forceinline__ void split_h2(float v, _Float16 &hi, _Float16 &lo)
{
    hi = (_Float16)v;
    lo = (_Float16)(v - (float)hi);
}

// Persistent, one 512-thread workgroup per CU, two roles with one barrier per cell (register liveness stays within a role: the
// 144 B-operand registers are never live in conv1 code, nor the conv1 weights in conv2 code):
//   waves 4-7 (producers): conv1 of cell k on the VALU into c1[k & 1]; input of cell k+1 (loaded into registers at the top
//                          of the step, written to LDS at its end)
//   waves 0-3 (consumers): conv2 of cell k-1 from c1[(k-1) & 1] on the f16 matrix pipe (wave = N half x M-tile parity)
// A consumer and a producer share each SIMD: f16 MFMAs co-issue with the other wave's VALU work.
template <bool U8IN>
__global__ __launch_bounds__(512, 2) void k_conv_features_h2(const void *__restrict__ xin, long B, const uint4 *__restrict__ w1img,
                                                             const float *__restrict__ b1, float scale1_inv, const uint4 *__restrict__ w2img,
                                                             const float *__restrict__ b2, float scale_inv, float *__restrict__ feat, const int *__restrict__ run_if_clear
#ifdef SV_DEV
                                                             , int ablate
#endif
                                                             )
{
#ifndef SV_DEV
    constexpr int ablate = 0;
#endif
    if (run_if_clear && *run_if_clear != 0) return;          // (svk_cnn_forward: this batch is outside the f16-pair range; the f32-MFMA kernels take it)
    __shared__ __attribute__((aligned(16))) unsigned char c1[2][2 * PLANE_B];  // [cell parity][part hi/lo][position][32 ch f16 + pad]
    __shared__ __attribute__((aligned(16))) unsigned char inh[2][2][IN_PLANE_B]; // input [slot][part hi/lo][32 rows][32 f16], zero border
    __shared__ __attribute__((aligned(8))) unsigned short lut_in[208], lut_out[208];

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const bool consumer = wave < 4;
    const int c16 = lane & 15, q = lane >> 4;

    for (int i = tid; i < 4 * PLANE_B / 4; i += 512) ((unsigned *)c1)[i] = 0;   // zero borders, for good
    for (int i = tid; i < 4 * IN_PLANE_B / 4; i += 512) ((unsigned *)inh)[i] = 0;
    for (int g = tid; g < 208; g += 512) {                                      // pooled pixel g of conv1 (13 tiles of 16; 196 real)
        const int gc = g < 196 ? g : 195, wy = gc / 14, wx = gc - 14 * wy;
        lut_in[g] = (unsigned short)((2 * wy) * 64 + 4 * wx);                   // its 4x4 input patch in an input plane
        lut_out[g] = (unsigned short)((wy + 1) * ROW_STRIDE + (wx + 1) * POS_STRIDE);   // its position in a c1 plane (g > 195: pixel 195 again)
    }
    __syncthreads();

    const long first = blockIdx.x, stride = gridDim.x;
    const long ncell = first < B ? (B - first + stride - 1) / stride : 0;       // cells first, first + stride, ...
    if (ncell == 0) return;

    if (!consumer) {
        // the producers are the younger half of the workgroup and would lose every issue arbitration against the consumers' MFMA
        // stream on their SIMD; their instructions are few and the consumers wait for them at the barrier: static priority
        __builtin_amdgcn_s_setprio(1);                                          // (0.42 -> 0.375 ms)
        const int ptid = tid & 255;
        // Input of a cell -> inh[slot]: one dword (4 pixels of a row) per thread for 8-bit cells, up to 4 floats for f32 input;
        // every value is split into its f16 pair on the way (pixel (y, x) sits at row y + 1, column x + 1 of a 64-byte row)
        unsigned sraw[4] = {0, 0, 0, 0};
        auto stage_load = [&](long c) {
            if (U8IN) {
                if (ptid < 196) sraw[0] = ((const unsigned *)((const u8 *)xin + c * 784))[ptid];
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++)
                    if (ptid + 256 * j < 784) sraw[j] = __float_as_uint(((const float *)xin)[c * 784 + ptid + 256 * j]);
            }
        };
        auto put = [&](int slot, int y, int x, float v) {
            _Float16 h, l;
            split_h2(v, h, l);
            unsigned char *d = inh[slot][0] + (y + 1) * 64 + (x + 1) * 2;
            *(_Float16 *)d = h;
            *(_Float16 *)(d + IN_PLANE_B) = l;
        };
        auto stage_store = [&](int slot) {
            if (U8IN) {
                if (ptid < 196) {
                    const int y = ptid / 7, x = 4 * (ptid - 7 * y);
#pragma unroll
                    for (int j = 0; j < 4; j++) put(slot, y, x + j, glue_norm((u8)(sraw[0] >> (8 * j))));
                }
            } else {
#pragma unroll
                for (int j = 0; j < 4; j++) {
                    const int i = ptid + 256 * j;
                    if (i < 784) { const int y = i / 28; put(slot, y, i - y * 28, __uint_as_float(sraw[j])); }
                }
            }
        };
        // conv1 on the f16 matrix pipe too (as v_fma_f32 work it took 1330 VALU issue slots per wave per cell, and beside the
        // consumers' MFMA stream a SIMD has only ~2 VALU slots per MFMA to give): rows = 16 pooling windows, K = the window's 4x4
        // input patch twice -- k 0..15 from the hi plane, k 16..31 from the lo plane --, columns = 16 channels of one of the
        // window's 4 conv positions (weights shifted accordingly, zero where a tap falls outside the 3x3).  Two MFMAs per
        // (tile, position): [xh | xl] x [wh | wh]  and  [xh | xl] x [wl | 0].  The 4 positions are 4 accumulators of the same lane:
        // max, scale, bias, ReLU in the lane; the result is split into its f16 pair and stored channel-last for conv2.
        // this wave: channels 16*chalf.., M tiles ppar, ppar+2, ...  The parity is the opposite of the consumer's on the same SIMD (waves w
        // and w+4 share one): a cell has 13 M tiles, so one parity does 7 and the other 6, in conv1 and in conv2 alike, and a SIMD should not
        // get the 7 of both (434 vs 372 MFMAs per cell before, 426 vs 380 now; the barrier waits for the slowest SIMD)
        const int chalf = (wave >> 1) & 1, ppar = (wave & 1) ^ 1;
        uint4 b1reg[4][2];
#pragma unroll
        for (int pos = 0; pos < 4; pos++)
#pragma unroll
            for (int m = 0; m < 2; m++) b1reg[pos][m] = w1img[((chalf * 4 + pos) * 2 + m) * 64 + lane];
        float bias1 = b1[16 * chalf + c16];
#pragma unroll
        for (int pos = 0; pos < 4; pos++) { settle(b1reg[pos][0]); settle(b1reg[pos][1]); }
        settle(bias1);
        stage_load(first);
        stage_store(0);
        __syncthreads();                                                        // (A) input of cell 0 visible
        // One producer wave per SIMD: nothing else hides its latencies, so the tile loop is software-pipelined by hand (tile jj's
        // patch fetch and MFMAs are issued before tile jj-1's epilogue, accumulators double-buffered) and the window -> LDS offset
        // arithmetic (divisions by 14) comes from two small tables.
        const int ntile1 = (ablate & 1) ? 0 : (ppar ? 6 : 7);
        for (long k = 0; k < ncell; k++) {
            if (k + 1 < ncell) stage_load(first + (k + 1) * stride);            // lands while conv1 runs
            const unsigned char *src_part = inh[k & 1][q >> 1] + 2 * (q & 1) * 64;   // this lane's plane (hi: k-groups 0,1; lo: 2,3) and patch rows
            unsigned char *dst_cell = c1[k & 1] + (16 * chalf + c16) * 2;
            f32x4 acc[2][4];
            auto fetch_mfma = [&](int j, int set) {
                const unsigned *src = (const unsigned *)(src_part + lut_in[16 * j + c16]);
                const uint4 araw = {src[0], src[1], src[16], src[17]};          // patch rows 2(q&1), 2(q&1)+1: 4 halfs each
                const h8 a = __builtin_bit_cast(h8, araw);
#pragma unroll
                for (int pos = 0; pos < 4; pos++) {
                    acc[set][pos] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, b1reg[pos][0]), (f32x4){0.f, 0.f, 0.f, 0.f}, 0, 0, 0);
                    acc[set][pos] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, __builtin_bit_cast(h8, b1reg[pos][1]), acc[set][pos], 0, 0, 0);
                }
            };
            auto finish = [&](int j, int set) {
                const uint2 oo = *(const uint2 *)&lut_out[16 * j + 4 * q];      // 4 x u16 store offsets of this lane's 4 pooled pixels
#pragma unroll
                for (int reg = 0; reg < 4; reg++) {                             // rows 4q + reg = pooled pixels 16j + 4q + reg
                    const float m = fmaxf(fmaxf(acc[set][0][reg], acc[set][1][reg]), fmaxf(acc[set][2][reg], acc[set][3][reg]));
                    const float v = fmaxf(__builtin_fmaf(m, scale1_inv, bias1), 0.f);
                    _Float16 h, l;
                    split_h2(v, h, l);
                    const unsigned off = ((reg & 2) ? oo.y : oo.x) >> (16 * (reg & 1)) & 0xFFFFu;
                    *(_Float16 *)(dst_cell + off) = h;                          // (rows past pixel 195 recompute and rewrite pixel 195)
                    *(_Float16 *)(dst_cell + off + PLANE_B) = l;
                }
            };
            if (ntile1 > 0) fetch_mfma(ppar, 0);
#pragma unroll
            for (int jj = 1; jj < 7; jj++) {
                if (jj >= ntile1) break;
                fetch_mfma(ppar + 2 * jj, jj & 1);
                finish(ppar + 2 * (jj - 1), (jj - 1) & 1);
            }
            if (ntile1 > 0) finish(ppar + 2 * (ntile1 - 1), (ntile1 - 1) & 1);
            if (k + 1 < ncell) stage_store((k + 1) & 1);
            __syncthreads();                                                    // (B_k) c1[k & 1] complete; consumers done with c1[(k-1) & 1]
        }
        __syncthreads();                                                        // (C) matches the consumers' last step
        return;
    }

    const int np = wave >> 1, par = wave & 1;  // N half (channels 32np..32np+31), M-tile parity
    uint4 breg[9][2][2];                       // [tap][t][part]: B[k = 8q+j][col c16] = W2s[oc = 32np + 2*c16 + t][ic = 8q + j][tap]
#pragma unroll
    for (int tap = 0; tap < 9; tap++)
#pragma unroll
        for (int t = 0; t < 2; t++)
#pragma unroll
            for (int p = 0; p < 2; p++) breg[tap][t][p] = w2img[(((tap * 2 + np) * 2 + t) * 2 + p) * 64 + lane];
    float bias2_0 = b2[32 * np + 2 * c16], bias2_1 = b2[32 * np + 2 * c16 + 1];
#pragma unroll
    for (int tap = 0; tap < 9; tap++)
#pragma unroll
        for (int t = 0; t < 2; t++) { settle(breg[tap][t][0]); settle(breg[tap][t][1]); }
    settle(bias2_0);
    settle(bias2_1);
    __syncthreads();                                                            // (A)
    __syncthreads();                                                            // (B_0)
    // A consumer is alone on its SIMD's matrix pipe, so every cycle it spends outside MFMAs is idle pipe time.  Measured on this
    // hardware (tools/ubench_mfma_fill.hip, one wave per SIMD, back-to-back v_mfma_f32_16x16x32_f16 = 16.5 cycles each): ONE
    // VALU instruction or s_waitcnt in the gap behind an MFMA is free, a second one costs 4-5 cycles, a third 8 more; ONE
    // ds_read_b128 costs 4, a second one 16 more.  So the tile body is written as 54 slots -- an MFMA and at most one other
    // instruction each, fenced by sched_barrier so the compiler cannot regroup them:
    //   * the MFMAs are asm so that every accumulation is IN PLACE (vDst = SrcC; left to the register allocator, dependent MFMAs
    //     got a destination different from their SrcC, which takes the result through the register file instead of the
    //     accumulate-forwarding path).  Hazards hipcc would pad for a builtin hold by construction: A comes from ds_reads (the
    //     compiler waits for asm inputs), B was loaded before the loop, a tile's first MFMAs take the literal 0 as SrcC, and the
    //     VALU reads an accumulator at the earliest 12 MFMAs after its last MFMA;
    //   * A operands flow through a RING-slot register ring, one ds_read_b128 per slot, PF taps ahead of their MFMAs, across tile
    //     boundaries (the tile loop is fully unrolled: all ring indices are static and the lgkmcnt waits are counted exactly);
    //   * all three partial products of a channel tile go into ONE accumulator (f32 either way), double-buffered by tile parity;
    //     the epilogue of tile j-1 (2x2 max, scale + bias, ReLU, hi/lo split, two stores) is 14 single instructions in the slots of taps 2-6.
    constexpr int PF = 5, RING = 6;
    uint4 ring_h[RING], ring_l[RING];
    // per-lane LDS offset of this wave's tile jj (window-in-tile, position dy/dx, k-group of the lane): loop-invariant, 7 registers
    unsigned tile_off[7];
#pragma unroll
    for (int jj = 0; jj < 7; jj++) {
        int g = 4 * (par + 2 * jj) + (c16 >> 2);
        if (g > 48) g = 48;
        const int wy = g / 7, wx = g - 7 * wy, sp = c16 & 3;
        tile_off[jj] = (unsigned)((2 * wy + (sp >> 1)) * ROW_STRIDE + (2 * wx + (sp & 1)) * POS_STRIDE + q * 16);
    }
#define SV_TAP_OFF(tap) (((tap) / 3) * ROW_STRIDE + ((tap) % 3) * POS_STRIDE)
#define SV_SLOT() __builtin_amdgcn_sched_barrier(0)
    const int ntile = (ablate & 2) ? 0 : (par ? 6 : 7);
    // Features leave this kernel already split into f16 pairs, in the order the fc kernels' A operands want them: a cell's 3136 values
    // (index = 64 window + channel) in groups of 8, each group 16 B of hi parts then 16 B of lo parts.  This lane's two channels of window
    // 4 (par + 2 jp) + q: group 64 jp + 32 par + 8 q + 4 np + c16 / 4, hi pair at byte 4 (c16 % 4) of it, lo pair 16 B behind.
    const unsigned out_off = (32 * par + 8 * q + 4 * np + (c16 >> 2)) * 32 + (c16 & 3) * 4;
    unsigned long long dbg_t[4] = {0, 0, 0, 0}, dbg_s = 0;
#define SV_STAMP(i) do { if (ablate & 16) { unsigned long long t_ = __builtin_amdgcn_s_memtime(); __builtin_amdgcn_s_waitcnt(0xC07F); dbg_t[i] += t_ - dbg_s; dbg_s = t_; } } while (0)
    if (ablate & 16) dbg_s = __builtin_amdgcn_s_memtime();
    for (long k = 0; k < ncell; k++) {
        SV_STAMP(0);
        const char *fcell = (const char *)(feat + (first + k * stride) * FEAT);  // wave-uniform: scalar base + 32-bit lane offsets
        const unsigned char *cbase = c1[k & 1];
        const unsigned char *ap = cbase + tile_off[0], *ap_next = ap;
#pragma unroll
        for (int st = 0; st < PF; st++) {                                       // the first tile of a cell cannot be fetched before the barrier
            ring_h[st % RING] = *(const uint4 *)(ap + SV_TAP_OFF(st));
            ring_l[st % RING] = *(const uint4 *)(ap + SV_TAP_OFF(st) + PLANE_B);
        }
        f32x4 acc[2][2];                                                        // [tile parity][t]
        float m0 = 0.f, m1 = 0.f, t0 = 0.f, t1 = 0.f;
        unsigned hp = 0, lp = 0;
        // one single-instruction piece of the epilogue of the tile whose sums sit in accumulator set `set` (tile index jp of this wave)
        auto epilogue = [&](int piece, int jp, int set) {
            const f32x4 &v0 = acc[set][0], &v1 = acc[set][1];
            switch (piece) {                                                    // (asm: fmaxf on asm-produced values gets a canonicalising v_max first)
            case 0: asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(m0) : "v"(v0[0]), "v"(v0[1]), "v"(v0[2])); break;
            case 1: asm volatile("v_max3_f32 %0, %1, %2, %3" : "=v"(m1) : "v"(v1[0]), "v"(v1[1]), "v"(v1[2])); break;
            case 2: asm volatile("v_max_f32 %0, %0, %1" : "+v"(m0) : "v"(v0[3])); break;
            case 3: asm volatile("v_max_f32 %0, %0, %1" : "+v"(m1) : "v"(v1[3])); break;
            case 4: asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(m0) : "s"(scale_inv), "v"(bias2_0)); break;
            case 5: asm volatile("v_fma_f32 %0, %0, %1, %2" : "+v"(m1) : "s"(scale_inv), "v"(bias2_1)); break;
            case 6: asm volatile("v_max_f32 %0, 0, %0" : "+v"(m0)); break;
            case 7: asm volatile("v_max_f32 %0, 0, %0" : "+v"(m1)); break;
            // split_h2 of both values in four instructions: both hi parts by one packed conversion, m - f32(hi) by v_fma_mix_f32 (the f16 half of
            // hp widened exactly, times -1, plus m: one rounding, the same as the subtraction), both lo parts by another packed conversion
            case 8: asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(hp) : "v"(m0), "v"(m1)); break;
            case 9: asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[0,0,0] op_sel_hi:[1,0,0]" : "=v"(t0) : "v"(hp), "v"(m0)); break;
            case 10: asm volatile("v_fma_mix_f32 %0, %1, -1.0, %2 op_sel:[1,0,0] op_sel_hi:[1,0,0]" : "=v"(t1) : "v"(hp), "v"(m1)); break;
            case 11: asm volatile("v_cvt_pk_f16_f32 %0, %1, %2" : "=v"(lp) : "v"(t0), "v"(t1)); break;
            case 12:                                                            // rows 4q..4q+3 = the 4 positions of window 4*(par + 2jp) + q
                if (jp < 6 || q == 0) *(unsigned *)(fcell + jp * 2048 + out_off) = hp;
                break;
            case 13:
                if (jp < 6 || q == 0) *(unsigned *)(fcell + jp * 2048 + out_off + 16) = lp;
                break;
            default: break;                                                     // (the slot structure offers 15 pieces)
            }
        };
        // 13 M tiles of 4 pooling windows; this wave takes tiles par, par+2, ... for its 32 channels
#pragma unroll
        for (int jj = 0; jj < 7; jj++) {
            if (jj >= ntile) break;
            const int set = jj & 1;
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                const int st = jj * 9 + tap;                                    // step number within the cell
                const bool pf_here = tap + PF < 9, pf_next = !pf_here && jj + 1 < ntile;
                const unsigned char *pa = pf_here ? ap + SV_TAP_OFF((tap + PF) % 9) : ap_next + SV_TAP_OFF((tap + PF) % 9);
                const h8 ah = __builtin_bit_cast(h8, ring_h[st % RING]), al = __builtin_bit_cast(h8, ring_l[st % RING]);
                const h8 bh0 = __builtin_bit_cast(h8, breg[tap][0][0]), bl0 = __builtin_bit_cast(h8, breg[tap][0][1]);
                const h8 bh1 = __builtin_bit_cast(h8, breg[tap][1][0]), bl1 = __builtin_bit_cast(h8, breg[tap][1][1]);
                const int e0 = (jj > 0 && tap >= 2 && tap <= 6 && 3 * (tap - 2) < 14) ? 3 * (tap - 2) : -1;   // epilogue pieces e0, e0+1, e0+2 in this tap
                // slot 1
                SV_SLOT();
                if (tap == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc[set][0]) : "v"(ah), "v"(bh0));
                else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[set][0]) : "v"(ah), "v"(bh0));
                if (pf_here || pf_next) ring_h[(st + PF) % RING] = *(const uint4 *)pa;
                SV_SLOT();
                // slot 2
                if (tap == 0) asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, 0" : "=&v"(acc[set][1]) : "v"(ah), "v"(bh1));
                else asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[set][1]) : "v"(ah), "v"(bh1));
                if (tap == 1 && jj + 1 < ntile) ap_next = cbase + tile_off[jj + 1 < 7 ? jj + 1 : 6];   // v_add: next tile's address
                SV_SLOT();
                // slot 3
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[set][0]) : "v"(ah), "v"(bl0));
                if (pf_here || pf_next) ring_l[(st + PF) % RING] = *(const uint4 *)(pa + PLANE_B);
                SV_SLOT();
                // slot 4
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[set][1]) : "v"(ah), "v"(bl1));
                if (e0 >= 0) epilogue(e0, jj - 1, set ^ 1);
                SV_SLOT();
                // slot 5
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[set][0]) : "v"(al), "v"(bh0));
                if (e0 >= 0) epilogue(e0 + 1, jj - 1, set ^ 1);
                SV_SLOT();
                // slot 6
                asm volatile("v_mfma_f32_16x16x32_f16 %0, %1, %2, %0" : "+v"(acc[set][1]) : "v"(al), "v"(bh1));
                if (e0 >= 0) epilogue(e0 + 2, jj - 1, set ^ 1);
                SV_SLOT();
            }
            ap = ap_next;
        }
        SV_STAMP(1);
        if (ntile > 0) {
            asm volatile("s_nop 15\n\ts_nop 15");                              // the last tile's accumulators: MFMA -> VALU read distance
#pragma unroll
            for (int piece = 0; piece < 14; piece++) epilogue(piece, ntile - 1, (ntile - 1) & 1);
        }
        SV_STAMP(2);
        __syncthreads();                                                        // (B_{k+1}) / (C)
        SV_STAMP(3);
    }
    if ((ablate & 16) && blockIdx.x < 2 && lane == 0)
        printf("blk %d wave %d ncell %ld: loop-top %llu tiles %llu epilogue %llu barrier %llu cycles/cell\n", (int)blockIdx.x, wave, ncell, dbg_t[0] / ncell, dbg_t[1] / ncell,
               dbg_t[2] / ncell, dbg_t[3] / ncell);
#undef SV_SLOT
#undef SV_TAP_OFF
}

// ---------------------------------------------------------------------------------------------------------------------
// k_fc_head_h2: 64 cells per 256-thread workgroup, one 16-cell M tile per wave, all 8 N tiles (128 hidden units), two workgroups
// per CU.  (Staging the 1.6 MB weight image through LDS is 3/4 of this kernel's time by ablation; a 96-cell, one-workgroup-per-CU
// form with three LDS stages ran at half the speed: the kernel lives on the second workgroup hiding the staging latency.)
// K = 3136 in 98 steps of 32; a stage = 2 steps of the weight image (2 x 8 N tiles x hi/lo x 1 KB = 32 KB), global ->
// registers -> LDS, double-buffered, one barrier per stage.  A: 16 consecutive f32 features per lane per stage straight from
// global memory (four dwordx4, issued a stage ahead), split into hi/lo in registers.
// ---------------------------------------------------------------------------------------------------------------------
__global__ __launch_bounds__(256, 2) void k_fc_head_h2(const float *__restrict__ feat, long B, const uint4 *__restrict__ w1img,
                                                       const float *__restrict__ b1, float scale_inv, const float *__restrict__ w2,
                                                       const float *__restrict__ b2, float *__restrict__ logits,
                                                       u8 *__restrict__ digits, float *__restrict__ conf, const int *__restrict__ run_if_clear)
{
    if (run_if_clear && *run_if_clear != 0) return;
    constexpr int SPS = 2, STAGE_V = SPS * 8 * 2 * 64, NSTAGE = 98 / SPS, WPT = STAGE_V / 256;   // uint4 per stage, per thread
    __shared__ __attribute__((aligned(16))) uint4 wt[2][STAGE_V];    // 2 x 32 KB; the hidden activations alias it after the K loop
    __shared__ float w2s[10][128];
    __shared__ float lg[4][16][12];
    float(*hs)[16][129] = (float(*)[16][129])wt;                      // [4][16][129] floats = 33 KB

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const long cell0 = (long)blockIdx.x * 64 + wave * 16;
    long crow = cell0 + r;
    if (crow >= B) crow = B - 1;
    // K is permuted so that a lane's operands of the SPS = 2 steps of a stage are 64 consecutive bytes (k-slot (q, j) of step
    // 2S + ss = feature 64S + 16q + 8ss + j; the weight image is packed to match; k_conv_features_h2 writes the features as f16
    // pairs, group of 8 by group of 8: 16 B of hi parts, 16 B of lo parts): the four q-lanes of a row read 256 contiguous bytes
    // per stage, and what arrives are the MFMA operands themselves
    const uint4 *ap = (const uint4 *)(feat + crow * FEAT + 16 * q);     // stage S: + 16*S uint4; [2 ss] = hi, [2 ss + 1] = lo

    for (int i = tid; i < 1280; i += 256) w2s[i >> 7][i & 127] = w2[i];

    f32x4 acc_h[8], acc_l[8];
#pragma unroll
    for (int t = 0; t < 8; t++) { acc_h[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc_l[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }

    uint4 wreg[WPT];
    uint4 areg[SPS][2];
#pragma unroll
    for (int j = 0; j < WPT; j++) wreg[j] = w1img[256 * j + tid];
#pragma unroll
    for (int s = 0; s < SPS; s++) { areg[s][0] = ap[2 * s]; areg[s][1] = ap[2 * s + 1]; }
#pragma unroll
    for (int j = 0; j < WPT; j++) wt[0][256 * j + tid] = wreg[j];
    __syncthreads();

    for (int st = 0; st < NSTAGE; st++) {
        const int cur = st & 1;
        uint4 a[SPS][2];
#pragma unroll
        for (int s = 0; s < SPS; s++) { a[s][0] = areg[s][0]; a[s][1] = areg[s][1]; }
        if (st + 1 < NSTAGE) {                                        // next stage: global -> registers while this one computes
#pragma unroll
            for (int j = 0; j < WPT; j++) wreg[j] = w1img[(long)(st + 1) * STAGE_V + 256 * j + tid];
#pragma unroll
            for (int s = 0; s < SPS; s++) { areg[s][0] = ap[16 * (st + 1) + 2 * s]; areg[s][1] = ap[16 * (st + 1) + 2 * s + 1]; }
        }
        __builtin_amdgcn_sched_barrier(0);     // (the compiler otherwise sinks these loads to their use at the end of the stage, exposing their latency)
#pragma unroll
        for (int s = 0; s < SPS; s++) {
            const h8 ah = __builtin_bit_cast(h8, a[s][0]), al = __builtin_bit_cast(h8, a[s][1]);
#pragma unroll
            for (int t = 0; t < 8; t++) {
                const h8 bh = __builtin_bit_cast(h8, wt[cur][((s * 8 + t) * 2 + 0) * 64 + lane]);
                const h8 bl = __builtin_bit_cast(h8, wt[cur][((s * 8 + t) * 2 + 1) * 64 + lane]);
                acc_h[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bh, acc_h[t], 0, 0, 0);
                acc_l[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah, bl, acc_l[t], 0, 0, 0);
                acc_l[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al, bh, acc_l[t], 0, 0, 0);
            }
        }
        if (st + 1 < NSTAGE) {
#pragma unroll
            for (int j = 0; j < WPT; j++) wt[cur ^ 1][256 * j + tid] = wreg[j];
        }
        __syncthreads();
    }

    // acc[t][reg]: cell row 4q + reg of this wave's tile, hidden unit 16t + r   (hs aliases the weight stages: all waves are
    // past the last barrier of the K loop, nobody reads wt any more)
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const float bias = b1[16 * t + r];
#pragma unroll
        for (int reg = 0; reg < 4; reg++) hs[wave][4 * q + reg][16 * t + r] = fmaxf((acc_h[t][reg] + acc_l[t][reg]) * scale_inv + bias, 0.f);
    }
    __syncthreads();
    for (int jj = 0; jj < 3; jj++) {                                  // fc2: lane (cell r, class group q) -> classes q, q+4, q+8
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


// ---------------------------------------------------------------------------------------------------------------------
// k_fc_head_h2p: the same arithmetic for large batches, one 768-thread workgroup per CU that streams the 1.6 MB weight image
// ONCE for all of its cells (k_fc_head_h2 streams it once per 64 cells: 324 times for 256 frames, 0.52 GB through the L2s).
// 12 waves = 6 M tiles (96 cells per pass) x 2 N halves; 3 waves per SIMD.
// What bounds an fc kernel of this shape is not the MFMA pipe (a quarter busy), the LDS or the L2s but the texture addresser:
// a wave-load of MFMA A fragments straight from the features touches 64 different 16-byte pieces in 32 lines and takes the
// TA ~55 cycles, and the waves queue up behind it in program order (tools/ubench_fc_read.hip: 62 us for the 260 MB read that
// way, 40 us with 16 adjacent lanes reading 256 contiguous bytes).  So BOTH operands go global -> LDS directly
// (global_load_lds_dwordx4, inline asm: hipcc would guard every ds_read after one with vmcnt(0)):
//   * weights: 32 pieces of 1 KB per 32-KB stage (waves 0-7, four each), ring of two stages, one in flight (L2 hits);
//   * features: 24 pieces per stage, each 4 rows x 256 B with lane-contiguous sources (two per wave), ring of three 24-KB
//     stages, two in flight (HBM).  The image is row-major, [96 rows][16 units of 16 B]; unit u of row r sits in slot
//     u ^ g(r), g(r) = (r & 3) << 2 | (r >> 2) & 3 -- the DMA cannot scatter, so the permutation is on the source address --
//     which makes the 16 lanes of every ds_read_b128 lane group of an A-fragment fetch hit 16 different bank slots.
// One counted wait and one raw barrier per stage: at the top of stage s a wave waits until only its two feature pieces of
// stage s + 1 are in flight (s_waitcnt vmcnt(2): its weight pieces of stage s, issued after the features of s and before
// those of s + 1, have landed), the barrier makes everybody's pieces visible and frees the slots that stage s - 1 read, and
// the pieces of weight stage s + 1 and feature stage s + 2 are issued into them.
// ---------------------------------------------------------------------------------------------------------------------
constexpr int FCP_WAVES = 12, FCP_MT = 6, FCP_CELLS = 16 * FCP_MT, FCP_NSTAGE = 49;
constexpr int FCP_W_STAGE = 32768, FCP_A_STAGE = FCP_CELLS * 256, FCP_OFF_A = 2 * FCP_W_STAGE;
constexpr int FCP_HS_LD = 129;

// one LDS-DMA piece: 64 lanes x 16 B from the lanes' own addresses to lds_dst + 16 * lane (M0 carries the LDS byte address)
__device__ __forceinline__ void fcp_glds(const void *gsrc, unsigned lds_dst)
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(64 * FCP_WAVES, 1) void k_fc_head_h2p(const float *__restrict__ feat, long B, long per, const uint4 *__restrict__ w1img,
                                                                  const float *__restrict__ b1, float scale_inv, const float *__restrict__ w2,
                                                                  const float *__restrict__ b2, float *__restrict__ logits, u8 *__restrict__ digits,
                                                                  float *__restrict__ conf, const int *__restrict__ run_if_clear)
{
    if (run_if_clear && *run_if_clear != 0) return;
    // one LDS object: [2 weight stages (the hidden activations alias them after the K loop)][3 feature stages][w2][logits]
    constexpr int OFF_W2 = FCP_OFF_A + 3 * FCP_A_STAGE, OFF_LG = OFF_W2 + 10 * 128 * 4, LDS_B = OFF_LG + FCP_MT * 16 * 12 * 4;
    static_assert(FCP_CELLS * FCP_HS_LD * 4 <= 2 * FCP_W_STAGE, "the hidden activations alias the weight ring");
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_B];
    float(*hs)[FCP_HS_LD] = (float(*)[FCP_HS_LD])lds;                                 // [96][129] floats = 49.5 KB
    float(*w2s)[128] = (float(*)[128])(lds + OFF_W2);
    float(*lg)[16][12] = (float(*)[16][12])(lds + OFF_LG);
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char *)lds;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int mt = wave % FCP_MT, nh = wave / FCP_MT;                                  // M tile, N half (hidden units 64 nh ..)
    const bool loader = wave < 8;
    for (int i = tid; i < 1280; i += 64 * FCP_WAVES) w2s[i >> 7][i & 127] = w2[i];

    // this lane's A fragments in a feature stage: row 16 mt + r; (step ss, part) = unit 4 q + 2 ss + part, in slot unit ^ g(r)
    const unsigned g_r = ((unsigned)(r & 3) << 2) | (unsigned)(r >> 2);
    unsigned a_off[4];
#pragma unroll
    for (int c = 0; c < 4; c++) a_off[c] = FCP_OFF_A + (16 * mt + r) * 256 + (((unsigned)(4 * q + c)) ^ g_r) * 16;
    // this lane's share of the wave's two feature pieces: row 8 wave + 4 i + lane / 16 of the pass, slot lane % 16 -> unit slot ^ g(row)
    const int prow = 8 * wave + (lane >> 4), pslot = lane & 15;

    const long base = (long)blockIdx.x * per, c_end = base + per < B ? base + per : B;       // per <= 96: one pass (svk_cnn_forward_h2)
    {
        const long cell0 = base + 16 * mt;
        const bool tile_live = cell0 < c_end;                            // (wave-uniform) an M tile with no cell does no arithmetic
        const unsigned char *asrc[2];
#pragma unroll
        for (int i = 0; i < 2; i++) {
            const int rl = prow + 4 * i;
            long row = base + rl;
            if (row >= c_end) row = c_end - 1;                                         // rows past the end: a valid address, results dropped
            const unsigned gr = ((unsigned)(rl & 3) << 2) | ((unsigned)(rl >> 2) & 3u);
            asrc[i] = (const unsigned char *)feat + row * (FEAT * 4) + (((unsigned)pslot) ^ gr) * 16;
        }
        const uint4 *wp = w1img + (wave & 7) * 4 * 64 + lane;                          // this wave's 4 pieces of a weight stage
        auto issue_w = [&](int st) {
            if (loader) {
                const unsigned dst = lds_base + (st & 1) * FCP_W_STAGE + (wave & 7) * 4096;
#pragma unroll
                for (int j = 0; j < 4; j++) fcp_glds(wp + (long)st * 2048 + 64 * j, dst + 1024 * j);
            }
        };
        auto issue_a = [&](int st) {
            const unsigned dst = lds_base + FCP_OFF_A + (st % 3) * FCP_A_STAGE + 2 * wave * 1024;
#pragma unroll
            for (int i = 0; i < 2; i++) fcp_glds(asrc[i] + 256 * st, dst + 1024 * i);
        };
        f32x4 acc_h[4], acc_l[4];
#pragma unroll
        for (int t = 0; t < 4; t++) { acc_h[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; acc_l[t] = (f32x4){0.f, 0.f, 0.f, 0.f}; }
        issue_w(0);
        issue_a(0);
        issue_a(1);

        // The stage's eight (step, N tile) pairs are a pipeline of their own: the hi/lo B fragments of pairs i + BD, i + BD + 1 are read from
        // the LDS while the six MFMAs of pairs i, i + 1 issue (two N tiles at a time: no MFMA reads the result of the one before it).
        auto compute = [&](int st) {
            constexpr int BD = 4;
            const unsigned char *wt = lds + (st & 1) * FCP_W_STAGE + lane * 16, *at = lds + (st % 3) * FCP_A_STAGE;
            uint4 fa[4], fb[8][2];
#pragma unroll
            for (int c = 0; c < 4; c++) fa[c] = *(const uint4 *)(at + a_off[c]);
            auto rd = [&](int i) {
                const int sstep = i >> 2, t = i & 3;
                fb[i][0] = *(const uint4 *)(wt + ((sstep * 8 + 4 * nh + t) * 2 + 0) * 1024);
                fb[i][1] = *(const uint4 *)(wt + ((sstep * 8 + 4 * nh + t) * 2 + 1) * 1024);
            };
#pragma unroll
            for (int i = 0; i < BD; i++) rd(i);
            __builtin_amdgcn_sched_barrier(0);
            const h8 ah[2] = {__builtin_bit_cast(h8, fa[0]), __builtin_bit_cast(h8, fa[2])}, al[2] = {__builtin_bit_cast(h8, fa[1]), __builtin_bit_cast(h8, fa[3])};
#pragma unroll
            for (int i = 0; i < 8; i += 2) {
                if (i + BD < 8) { rd(i + BD); rd(i + BD + 1); }
                __builtin_amdgcn_sched_barrier(0);
                const int sstep = i >> 2, t = i & 3;
                const h8 bh0 = __builtin_bit_cast(h8, fb[i][0]), bl0 = __builtin_bit_cast(h8, fb[i][1]);
                const h8 bh1 = __builtin_bit_cast(h8, fb[i + 1][0]), bl1 = __builtin_bit_cast(h8, fb[i + 1][1]);
                acc_h[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[sstep], bh0, acc_h[t], 0, 0, 0);
                acc_h[t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[sstep], bh1, acc_h[t + 1], 0, 0, 0);
                acc_l[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[sstep], bl0, acc_l[t], 0, 0, 0);
                acc_l[t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(ah[sstep], bl1, acc_l[t + 1], 0, 0, 0);
                acc_l[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[sstep], bh0, acc_l[t], 0, 0, 0);
                acc_l[t + 1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(al[sstep], bh1, acc_l[t + 1], 0, 0, 0);
                __builtin_amdgcn_sched_barrier(0);
            }
        };
        for (int st = 0; st < FCP_NSTAGE; st++) {
            if (st + 1 < FCP_NSTAGE) asm volatile("s_waitcnt vmcnt(2)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            __builtin_amdgcn_s_barrier();
            asm volatile("" ::: "memory");
            if (st + 1 < FCP_NSTAGE) issue_w(st + 1);
            if (st + 2 < FCP_NSTAGE) issue_a(st + 2);
            if (tile_live) compute(st);
        }
        __syncthreads();                                                                // everybody is done reading the rings: hs may overwrite them

        // acc[t][reg]: cell row 4q + reg of the wave's M tile, hidden unit 64 nh + 16 t + r
#pragma unroll
        for (int t = 0; t < 4; t++) {
            const float bias = b1[64 * nh + 16 * t + r];
#pragma unroll
            for (int reg = 0; reg < 4; reg++)
                hs[16 * mt + 4 * q + reg][64 * nh + 16 * t + r] = fmaxf((acc_h[t][reg] + acc_l[t][reg]) * scale_inv + bias, 0.f);
        }
        __syncthreads();
        if (nh == 0) {
            for (int jj = 0; jj < 3; jj++) {                                            // fc2: lane (cell r, class group q) -> classes q, q+4, q+8
                const int j = q + 4 * jj;
                if (j < 10) {
                    float s = b2[j];
                    for (int n = 0; n < 128; n++) s = __builtin_fmaf(hs[16 * mt + r][n], w2s[j][n], s);
                    lg[mt][r][j] = s;
                    if (cell0 + r < c_end) logits[(cell0 + r) * 10 + j] = s;
                }
            }
        }
        __syncthreads();
        if (nh == 0 && q == 0 && cell0 + r < c_end && (digits || conf)) {
            float best = lg[mt][r][0];
            int arg = 0;
            for (int j = 1; j < 10; j++)
                if (lg[mt][r][j] > best) { best = lg[mt][r][j]; arg = j; }
            if (digits) digits[cell0 + r] = (u8)arg;
            if (conf) {
                float den = 0.f;
                for (int j = 0; j < 10; j++) den += expf(lg[mt][r][j] - best);
                conf[cell0 + r] = 1.0f / den;
            }
        }
    }
}

}  // namespace

#ifdef SV_DEV
// development builds (make FLAGS+=-DSV_DEV): the ablation / stamp bits of k_conv_features_h2, see svk_cnn_forward_h2
extern "C" int sv_dev_set_h2_ablate(sv_ctx *ctx, int bits) { if (!ctx) return SV_ERR_BAD_ARG; ctx->dev_ablate = bits; return SV_OK; }
#endif

int svk_cnn_forward_h2(sv_ctx *ctx, const void *x, bool x_is_u8, long B, float *logits, u8 *digits, float *conf, const int *run_if_clear, hipStream_t s)
{
    const sv_weights &w = ctx->w;
    const int grid = (int)(B < (long)ctx->num_cus ? B : (long)ctx->num_cus);
#ifdef SV_DEV
    // development builds only: ctx->dev_ablate bit 0 skips the producers' conv1 tiles, bit 1 the consumers' conv2 tiles (wrong results); bit 4
    // makes the consumer waves of workgroups 0 and 1 print s_memtime stamps of their per-cell phases
#define SV_ABLATE_ARG , ctx->dev_ablate
#else
#define SV_ABLATE_ARG
#endif
    {
        sv_time_scope ts(ctx, SVK_CONV_FEATURES, s);
        if (x_is_u8)
            hipLaunchKernelGGL(k_conv_features_h2<true>, dim3(grid), dim3(512), 0, s, x, B, (const uint4 *)w.conv1_h2, w.conv1_b, w.conv1_h2_scale_inv, (const uint4 *)w.conv2_h2, w.conv2_b,
                               w.conv2_h2_scale_inv, ctx->features, run_if_clear SV_ABLATE_ARG);
        else
            hipLaunchKernelGGL(k_conv_features_h2<false>, dim3(grid), dim3(512), 0, s, x, B, (const uint4 *)w.conv1_h2, w.conv1_b, w.conv1_h2_scale_inv, (const uint4 *)w.conv2_h2, w.conv2_b,
                               w.conv2_h2_scale_inv, ctx->features, run_if_clear SV_ABLATE_ARG);
    }
#undef SV_ABLATE_ARG
    SV_LAUNCH_CHECK("k_conv_features_h2");
    sv_time_scope ts(ctx, SVK_FC_HEAD, s);
    // One workgroup per CU and one pass over the weight image as long as a CU's share of the cells fits a pass (96): faster than k_fc_head_h2 at
    // every such size (81 cells: 0.040 against 0.059 ms; 20,736: 0.069 against 0.094; tools/dev/fc_sweep.py).  Beyond that the share would take a
    // second full pass for a few cells, and two co-resident 64-cell workgroups per CU do better (32,768 cells: 0.115 against 0.136 ms).
    const long per = std::max<long>(16, (B + ctx->num_cus - 1) / ctx->num_cus);
    if (per <= FCP_CELLS) {
        hipLaunchKernelGGL(k_fc_head_h2p, dim3((unsigned)((B + per - 1) / per)), dim3(64 * FCP_WAVES), 0, s, ctx->features, B, per, (const uint4 *)w.fc1_h2, w.fc1_b,
                           w.fc1_h2_scale_inv, w.fc2_w, w.fc2_b, logits, digits, conf, run_if_clear);
        SV_LAUNCH_CHECK("k_fc_head_h2p");
        return SV_OK;
    }
    hipLaunchKernelGGL(k_fc_head_h2, dim3((unsigned)((B + 63) / 64)), dim3(256), 0, s, ctx->features, B, (const uint4 *)w.fc1_h2, w.fc1_b, w.fc1_h2_scale_inv,
                       w.fc2_w, w.fc2_b, logits, digits, conf, run_if_clear);
    SV_LAUNCH_CHECK("k_fc_head_h2");
    return SV_OK;
}
