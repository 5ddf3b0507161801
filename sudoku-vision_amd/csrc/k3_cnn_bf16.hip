// K3, bf16 configuration (BASELINE.json configs[4]): conv2 and fc1 on v_mfma_f32_16x16x32_bf16 with f32 accumulation.
// conv1 stays f32 on the VALU and rounds its ReLU/pool output to bf16; weights of conv2/fc1 are rounded to bf16 once on
// the host.  Parity target for this configuration: predicted digit indices (logits differ from f32 by ~1e-2).
//
//   k_conv_features_bf16 : 2 cells per workgroup iteration, 4 waves, 2 workgroups per CU.  conv1 output goes to LDS
//        channel-last (position-major, 32 bf16 channels = 64 B per position, rows padded to 80 B against bank conflicts)
//        so that one ds_read_b128 is one MFMA A operand (8 input channels of one 3x3 tap for one output position);
//        a wave keeps all 9 x 4 B operands (288 x 64 weights) in 144 VGPRs.  36 MFMAs per 16 positions x 64 channels
//        instead of 288 in f32.  N tiles are interleaved (column c of tile t = channel 4c + t) so that a lane ends up with
//        4 consecutive channels of one pooled window: one 8-byte store of bf16 features.
//   k_fc_head_bf16       : fc1 on the same MFMA (K = 3136 = 98 steps), fc2/argmax/softmax epilogue in f32 as in k_fc_head.
#include "sv_device.h"
#include "sv_internal.h"

typedef float f32x4 __attribute__((ext_vector_type(4)));
typedef float f32x2 __attribute__((ext_vector_type(2)));
typedef __bf16 bf16x8 __attribute__((ext_vector_type(8)));

namespace {

constexpr int POS_STRIDE = 80;                 // bytes per padded position (64 data + 16)
constexpr int C1B_CELL = 256 * POS_STRIDE;     // 16x16 padded positions
constexpr int IN_W = 30, IN_CELL = 900;
constexpr int FEAT = 3136;

__device__ __forceinline__ float glue_norm(u8 c)
{
    const float t = __fdiv_rn((float)(255 - (int)c), 255.0f);
    return __fdiv_rn(__fsub_rn(t, 0.5f), 0.5f);
}
__device__ __forceinline__ unsigned short bf16_bits(float x) { return __builtin_bit_cast(unsigned short, (__bf16)x); }

__global__ __launch_bounds__(256, 2) void k_conv_features_bf16(const u8 *__restrict__ cells, long B, const float *__restrict__ w1,
                                                               const float *__restrict__ b1, const uint4 *__restrict__ w2img,
                                                               const float *__restrict__ b2, unsigned short *__restrict__ feat)
{
    __shared__ __attribute__((aligned(16))) unsigned char c1b[2 * C1B_CELL];
    __shared__ float in_s[2 * IN_CELL];
    __shared__ float w1s[32 * 9 + 32];        // conv1 weights and biases (see the note at conv1 below)

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int c16 = lane & 15, q = lane >> 4;

    uint4 breg[9][4];                           // [tap][n tile]: B[k = 8q+j][col c16] = W2[oc = 4*c16 + t][ic = 8q + j][tap]
#pragma unroll
    for (int tap = 0; tap < 9; tap++)
#pragma unroll
        for (int t = 0; t < 4; t++) breg[tap][t] = w2img[(tap * 4 + t) * 64 + lane];
    float bias2[4];
#pragma unroll
    for (int t = 0; t < 4; t++) bias2[t] = b2[4 * c16 + t];

    for (int i = tid; i < 2 * C1B_CELL / 4; i += 256) ((unsigned *)c1b)[i] = 0;    // zero borders, for good
    for (int i = tid; i < 2 * IN_CELL; i += 256) in_s[i] = 0.f;
    for (int i = tid; i < 32 * 9 + 32; i += 256) w1s[i] = i < 288 ? w1[i] : b1[i - 288];
    __syncthreads();

    const long npairs = (B + 1) / 2;
    // Input staging in two phases: the 2 x 196 dwords of the NEXT pair of cells are loaded into registers (one or two per
    // thread) right after the current pair's have been written to LDS, so the load latency hides behind conv1 and conv2.
    unsigned sraw[2] = {0, 0};
    auto stage_load = [&](long pair) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int d = tid + 256 * k;
            if (d < 392) {
                const int cl = d >= 196 ? 1 : 0;
                long cg = pair * 2 + cl;
                if (cg >= B) cg = B - 1;
                sraw[k] = ((const unsigned *)(cells + cg * 784))[d - 196 * cl];
            }
        }
    };
    if ((long)blockIdx.x < npairs) stage_load(blockIdx.x);
    for (long pair = blockIdx.x; pair < npairs; pair += gridDim.x) {
#pragma unroll
        for (int k = 0; k < 2; k++) {
            const int d = tid + 256 * k;
            if (d < 392) {
                const int cl = d >= 196 ? 1 : 0, w = d - 196 * cl, y = w / 7, x = 4 * (w - 7 * y);
                float *dst = in_s + cl * IN_CELL + (y + 1) * IN_W + x + 1;
#pragma unroll
                for (int j = 0; j < 4; j++) dst[j] = glue_norm((u8)(sraw[k] >> (8 * j)));
            }
        }
        if (pair + gridDim.x < npairs) stage_load(pair + gridDim.x);
        __syncthreads();

        // conv1 + ReLU + pool (f32), output rounded to bf16, channel-last: wave w owns channels 8w..8w+7 = one 16-B store
        for (int rnd = 0; rnd < 7; rnd++) {
            const int idx = rnd * 64 + lane;
            if (idx < 392) {
                const int cl = idx / 196, pp = idx - cl * 196, py = pp / 14, px = pp - py * 14;
                f32x2 pr[4][3];                 // overlapping horizontal pairs of the 4x4 patch: one v_pk_fma_f32 = two outputs
                const float *src = in_s + cl * IN_CELL + (2 * py) * IN_W + 2 * px;
#pragma unroll
                for (int i = 0; i < 4; i++)
#pragma unroll
                    for (int j = 0; j < 3; j++) pr[i][j] = (f32x2){src[i * IN_W + j], src[i * IN_W + j + 1]};
                unsigned short o8[8];
#pragma unroll
                for (int o = 0; o < 8; o++) {
                    const int oc = wave * 8 + o;
                    // The weights come out of the LDS, i.e. in VGPRs: read from global memory they are wave-uniform, hipcc keeps them in SGPRs
                    // (s_load inside this loop) and feeds them to v_pk_fma_f32 as SGPR-pair operands, and with two waves per SIMD (two
                    // workgroups per CU) the results then differ from run to run in the last bits of a few per cent of the cells -- with
                    // scalar FMAs, with one workgroup per CU or with the weights in VGPRs they never do (round 3; the other kernels' packed
                    // FMAs take their SGPR operands from kernel arguments loaded once, and are bit-exact in every test).
                    const float *w = w1s + oc * 9;
                    const float bias = w1s[288 + oc];
                    f32x2 a0 = {bias, bias}, a1 = {bias, bias};
#pragma unroll
                    for (int ky = 0; ky < 3; ky++)
#pragma unroll
                        for (int kx = 0; kx < 3; kx++) {
                            const f32x2 wv = {w[ky * 3 + kx], w[ky * 3 + kx]};
                            a0 = __builtin_elementwise_fma(wv, pr[ky][kx], a0);
                            a1 = __builtin_elementwise_fma(wv, pr[ky + 1][kx], a1);
                        }
                    const float m = fmaxf(fmaxf(a0[0], a0[1]), fmaxf(a1[0], a1[1]));
                    o8[o] = bf16_bits(fmaxf(m, 0.f));
                }
                uint4 v;
                v.x = o8[0] | ((unsigned)o8[1] << 16); v.y = o8[2] | ((unsigned)o8[3] << 16);
                v.z = o8[4] | ((unsigned)o8[5] << 16); v.w = o8[6] | ((unsigned)o8[7] << 16);
                *(uint4 *)(c1b + cl * C1B_CELL + ((py + 1) * 16 + px + 1) * POS_STRIDE + wave * 16) = v;
            }
        }
        __syncthreads();

        // conv2: 25 tiles of 4 pooling windows
        for (int j = wave; j < 25; j += 4) {
            int g = 4 * j + (c16 >> 2);
            if (g > 97) g = 97;
            const int cl = g >= 49 ? 1 : 0, wl = g - 49 * cl, wy = wl / 7, wx = wl - 7 * wy, s = c16 & 3;
            const unsigned char *ap = c1b + cl * C1B_CELL + ((2 * wy + (s >> 1)) * 16 + 2 * wx + (s & 1)) * POS_STRIDE + q * 16;
            f32x4 acc[4];
#pragma unroll
            for (int t = 0; t < 4; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll
            for (int tap = 0; tap < 9; tap++) {
                const bf16x8 a = __builtin_bit_cast(bf16x8, *(const uint4 *)(ap + ((tap / 3) * 16 + tap % 3) * POS_STRIDE));
#pragma unroll
                for (int t = 0; t < 4; t++) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, breg[tap][t]), acc[t], 0, 0, 0);
            }
            const int gw = 4 * j + q;                      // rows 4q..4q+3 = the window's 4 positions; column c16 of tile t = channel 4*c16 + t
            if (gw < 98) {
                const int ocl = gw >= 49 ? 1 : 0, owl = gw - 49 * ocl;
                const long cg = pair * 2 + ocl;
                if (cg < B) {
                    unsigned short h[4];
#pragma unroll
                    for (int t = 0; t < 4; t++)
                        h[t] = bf16_bits(fmaxf(fmaxf(fmaxf(acc[t][0], acc[t][1]), fmaxf(acc[t][2], acc[t][3])) + bias2[t], 0.f));
                    uint2 v;
                    v.x = h[0] | ((unsigned)h[1] << 16);
                    v.y = h[2] | ((unsigned)h[3] << 16);
                    *(uint2 *)(feat + cg * FEAT + owl * 64 + 4 * c16) = v;
                }
            }
        }
        __syncthreads();
    }
}

__global__ __launch_bounds__(256) void k_fc_head_bf16(const unsigned short *__restrict__ feat, long B, const uint4 *__restrict__ w1img,
                                                      const float *__restrict__ b1, const float *__restrict__ w2, const float *__restrict__ b2,
                                                      float *__restrict__ logits, u8 *__restrict__ digits, float *__restrict__ conf)
{
    __shared__ float hs[4][16][129];
    __shared__ float w2s[10][128];
    __shared__ float lg[4][16][12];

    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int r = lane & 15, q = lane >> 4;
    const long cell0 = (long)blockIdx.x * 64 + wave * 16;
    long crow = cell0 + r;
    if (crow >= B) crow = B - 1;
    const uint4 *ap = (const uint4 *)(feat + crow * FEAT) + q;      // step s: + 4*s
    const uint4 *bp = w1img + lane;                                 // [98][8][64]

    for (int i = tid; i < 1280; i += 256) w2s[i >> 7][i & 127] = w2[i];

    f32x4 acc[8];
#pragma unroll
    for (int t = 0; t < 8; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
#pragma unroll 2
    for (int s = 0; s < 98; s++) {
        const bf16x8 a = __builtin_bit_cast(bf16x8, ap[4 * s]);
#pragma unroll
        for (int t = 0; t < 8; t++)
            acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(a, __builtin_bit_cast(bf16x8, bp[(s * 8 + t) * 64]), acc[t], 0, 0, 0);
    }
#pragma unroll
    for (int t = 0; t < 8; t++) {
        const float bias = b1[16 * t + r];
#pragma unroll
        for (int reg = 0; reg < 4; reg++) hs[wave][4 * q + reg][16 * t + r] = fmaxf(acc[t][reg] + bias, 0.f);
    }
    __syncthreads();
    for (int jj = 0; jj < 3; jj++) {
        const int j = q + 4 * jj;
        if (j < 10) {
            float sacc = b2[j];
            for (int n = 0; n < 128; n++) sacc = __builtin_fmaf(hs[wave][r][n], w2s[j][n], sacc);
            lg[wave][r][j] = sacc;
            if (cell0 + r < B) logits[(cell0 + r) * 10 + j] = sacc;
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


// k_fc_head_bf16p: the per-CU form of the fc head (k_fc_head_h2p in k3_cnn_h2.hip has the reasoning and the measurements): one 768-thread
// workgroup per CU, 12 waves = 6 M tiles x 2 N halves, the weight image streamed once per CU, both operands global -> LDS by DMA, one counted
// wait and one raw barrier per 64-k stage.  bf16 sizes: a feature row gives 128 B per stage, a DMA piece is 8 rows x 128 B (one per wave, ring of
// three 12-KB stages), the weight stage is 16 KB (waves 0-7, two pieces each, ring of two).  Feature image in the LDS: [96 rows][8 units of 16 B],
// unit u of row r in slot u ^ (r >> 1 & 7): two rows share a 256-B bank row, and with that the 16 lanes of a ds_read_b128 lane group hit 16 slots.
constexpr int BFP_WAVES = 12, BFP_MT = 6, BFP_CELLS = 16 * BFP_MT, BFP_NSTAGE = 49;
constexpr int BFP_W_STAGE = 16384, BFP_A_STAGE = BFP_CELLS * 128, BFP_OFF_A = 2 * BFP_W_STAGE, BFP_HS_LD = 129;

__device__ __forceinline__ void bfp_glds(const void *gsrc, unsigned lds_dst)          // 64 lanes x 16 B -> lds_dst + 16 * lane
{
    unsigned keep;
    asm volatile("s_mov_b32 %0, m0\n\ts_mov_b32 m0, %2\n\ts_nop 0\n\tglobal_load_lds_dwordx4 %1, off\n\ts_mov_b32 m0, %0" : "=&s"(keep) : "v"(gsrc), "s"(lds_dst) : "memory");
}

__global__ __launch_bounds__(64 * BFP_WAVES, 1) void k_fc_head_bf16p(const unsigned short *__restrict__ feat, long B, long per, const uint4 *__restrict__ w1img,
                                                                    const float *__restrict__ b1, const float *__restrict__ w2, const float *__restrict__ b2,
                                                                    float *__restrict__ logits, u8 *__restrict__ digits, float *__restrict__ conf)
{
    // one LDS object: [2 weight stages][3 feature stages] (the hidden activations alias both after the K loop) [w2][logits]
    constexpr int RINGS = BFP_OFF_A + 3 * BFP_A_STAGE, HS_B = BFP_CELLS * BFP_HS_LD * 4;
    constexpr int OFF_W2 = RINGS > HS_B ? RINGS : HS_B, OFF_LG = OFF_W2 + 10 * 128 * 4, LDS_B = OFF_LG + BFP_MT * 16 * 12 * 4;
    __shared__ __attribute__((aligned(16))) unsigned char lds[LDS_B];
    float(*hs)[BFP_HS_LD] = (float(*)[BFP_HS_LD])lds;
    float(*w2s)[128] = (float(*)[128])(lds + OFF_W2);
    float(*lg)[16][12] = (float(*)[16][12])(lds + OFF_LG);
    const unsigned lds_base = (unsigned)(unsigned long long)(__attribute__((address_space(3))) unsigned char *)lds;

    const int tid = threadIdx.x, lane = tid & 63;
    const int wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int r = lane & 15, q = lane >> 4;
    const int mt = wave % BFP_MT, nh = wave / BFP_MT;
    const bool loader = wave < 8;
    for (int i = tid; i < 1280; i += 64 * BFP_WAVES) w2s[i >> 7][i & 127] = w2[i];

    // this lane's A fragments in a feature stage: row 16 mt + r, step ss = unit 4 ss + q, in slot unit ^ (r >> 1 & 7)
    unsigned a_off[2];
#pragma unroll
    for (int ss = 0; ss < 2; ss++) a_off[ss] = BFP_OFF_A + (16 * mt + r) * 128 + (((unsigned)(4 * ss + q)) ^ ((unsigned)(r >> 1) & 7u)) * 16;

    const long base = (long)blockIdx.x * per, c_end = base + per < B ? base + per : B;          // per <= 96: one pass (svk_cnn_forward_bf16)
    const long cell0 = base + 16 * mt;
    const bool tile_live = cell0 < c_end;
    // this lane's share of the wave's feature piece: row 8 wave + lane / 8 of the pass, slot lane % 8 -> unit slot ^ (row >> 1 & 7)
    const int rl = 8 * wave + (lane >> 3);
    long srow = base + rl;
    if (srow >= c_end) srow = c_end - 1;
    const unsigned char *asrc = (const unsigned char *)(feat + srow * FEAT) + ((((unsigned)lane & 7u) ^ ((unsigned)(rl >> 1) & 7u)) * 16);
    const uint4 *wp = w1img + (wave & 7) * 2 * 64 + lane;                                       // this wave's 2 pieces of a weight stage
    auto issue_w = [&](int st) {
        if (loader) {
            const unsigned dst = lds_base + (st & 1) * BFP_W_STAGE + (wave & 7) * 2048;
            bfp_glds(wp + (long)st * 1024, dst);
            bfp_glds(wp + (long)st * 1024 + 64, dst + 1024);
        }
    };
    auto issue_a = [&](int st) { bfp_glds(asrc + 128 * st, lds_base + BFP_OFF_A + (st % 3) * BFP_A_STAGE + wave * 1024); };
    f32x4 acc[4];
#pragma unroll
    for (int t = 0; t < 4; t++) acc[t] = (f32x4){0.f, 0.f, 0.f, 0.f};
    issue_w(0);
    issue_a(0);
    issue_a(1);
    for (int st = 0; st < BFP_NSTAGE; st++) {
        // only this wave's feature piece of stage st + 1 stays in flight: its weight pieces of stage st were issued before it
        if (st + 1 < BFP_NSTAGE) asm volatile("s_waitcnt vmcnt(1)" ::: "memory"); else asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __builtin_amdgcn_s_barrier();
        asm volatile("" ::: "memory");
        if (st + 1 < BFP_NSTAGE) issue_w(st + 1);
        if (st + 2 < BFP_NSTAGE) issue_a(st + 2);
        if (tile_live) {
            const unsigned char *wt = lds + (st & 1) * BFP_W_STAGE + lane * 16, *at = lds + (st % 3) * BFP_A_STAGE;
            uint4 fa[2], fb[2][4];
#pragma unroll
            for (int ss = 0; ss < 2; ss++) fa[ss] = *(const uint4 *)(at + a_off[ss]);
#pragma unroll
            for (int ss = 0; ss < 2; ss++)
#pragma unroll
                for (int t = 0; t < 4; t++) fb[ss][t] = *(const uint4 *)(wt + (ss * 8 + 4 * nh + t) * 1024);
#pragma unroll
            for (int ss = 0; ss < 2; ss++)
#pragma unroll
                for (int t = 0; t < 4; t++)
                    acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_bf16(__builtin_bit_cast(bf16x8, fa[ss]), __builtin_bit_cast(bf16x8, fb[ss][t]), acc[t], 0, 0, 0);
        }
    }
    __syncthreads();                                                                           // everybody is done reading the rings: hs may overwrite them
#pragma unroll
    for (int t = 0; t < 4; t++) {
        const float bias = b1[64 * nh + 16 * t + r];
#pragma unroll
        for (int reg = 0; reg < 4; reg++) hs[16 * mt + 4 * q + reg][64 * nh + 16 * t + r] = fmaxf(acc[t][reg] + bias, 0.f);
    }
    __syncthreads();
    if (nh == 0) {
        for (int jj = 0; jj < 3; jj++) {
            const int j = q + 4 * jj;
            if (j < 10) {
                float sacc = b2[j];
                for (int n = 0; n < 128; n++) sacc = __builtin_fmaf(hs[16 * mt + r][n], w2s[j][n], sacc);
                lg[mt][r][j] = sacc;
                if (cell0 + r < c_end) logits[(cell0 + r) * 10 + j] = sacc;
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

}  // namespace

int svk_cnn_forward_bf16(sv_ctx *ctx, const u8 *cells, long B, float *logits, u8 *digits, float *conf, hipStream_t s)
{
    const sv_weights &w = ctx->w;
    const long npairs = (B + 1) / 2;
    const int grid = (int)(npairs < 2L * ctx->num_cus ? npairs : 2L * ctx->num_cus);
    if ((uintptr_t)cells & 3) return sv_fail(SV_ERR_UNSUPPORTED, "bf16 configuration: the 8-bit cell buffer must be 4-byte aligned (the kernel reads cells as dwords)");
    {
        sv_time_scope ts(ctx, SVK_CONV_FEATURES, s);
        hipLaunchKernelGGL(k_conv_features_bf16, dim3(grid), dim3(256), 0, s, cells, B, w.conv1_w, w.conv1_b, (const uint4 *)w.conv2_bf16, w.conv2_b,
                           (unsigned short *)ctx->features);
    }
    SV_LAUNCH_CHECK("k_conv_features_bf16");
    sv_time_scope ts(ctx, SVK_FC_HEAD, s);
    const long per = std::max<long>(16, (B + ctx->num_cus - 1) / ctx->num_cus);               // as in svk_cnn_forward_h2
    if (per <= BFP_CELLS) {
        hipLaunchKernelGGL(k_fc_head_bf16p, dim3((unsigned)((B + per - 1) / per)), dim3(64 * BFP_WAVES), 0, s, (const unsigned short *)ctx->features, B, per,
                           (const uint4 *)w.fc1_bf16, w.fc1_b, w.fc2_w, w.fc2_b, logits, digits, conf);
        SV_LAUNCH_CHECK("k_fc_head_bf16p");
        return SV_OK;
    }
    hipLaunchKernelGGL(k_fc_head_bf16, dim3((unsigned)((B + 63) / 64)), dim3(256), 0, s, (const unsigned short *)ctx->features, B, (const uint4 *)w.fc1_bf16,
                       w.fc1_b, w.fc2_w, w.fc2_b, logits, digits, conf);
    SV_LAUNCH_CHECK("k_fc_head_bf16");
    return SV_OK;
}
