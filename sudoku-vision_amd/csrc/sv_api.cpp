// Host side of libsudokuvision_hip.so: context, weight packing, fp64 homography, argument checks.
// Everything exported here is declared in include/sudoku_vision_hip.h.
#include <cmath>
#include <cstdlib>
#include <cstring>
#include <vector>

#include "sv_internal.h"

// ---- errors ---------------------------------------------------------------------------------------
static thread_local char g_err[512] = "";

int sv_fail(int code, const char *fmt, ...)
{
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
    return code;
}

extern "C" const char *sv_last_error(void) { return g_err; }
extern "C" int sv_version(void) { return 2; }

// ---- context --------------------------------------------------------------------------------------
extern "C" int sv_ctx_create(int device, sv_ctx **out)
{
    if (!out) return sv_fail(SV_ERR_BAD_ARG, "sv_ctx_create: out is NULL");
    int count = 0;
    SV_HIP(hipGetDeviceCount(&count));
    if (device < 0 || device >= count) return sv_fail(SV_ERR_BAD_ARG, "sv_ctx_create: device %d of %d", device, count);
    SV_HIP(hipSetDevice(device));
    hipDeviceProp_t prop;
    SV_HIP(hipGetDeviceProperties(&prop, device));
    sv_ctx *c = new sv_ctx();
    c->device = device;
    c->num_cus = prop.multiProcessorCount > 0 ? prop.multiProcessorCount : 256;
    *out = c;
    return SV_OK;
}

static void free_weights(sv_weights &w)
{
    float **ps[] = {&w.conv1_w, &w.conv1_b, &w.conv2_wreg, &w.conv2_wino, &w.conv2_b, &w.fc1_wreg, &w.fc1_b, &w.fc2_w, &w.fc2_b};
    for (float **p : ps) {
        if (*p) (void)hipFree(*p);
        *p = nullptr;
    }
    if (w.conv2_wsplit) (void)hipFree(w.conv2_wsplit);
    w.conv2_wsplit = nullptr;
    if (w.conv2_bf16) (void)hipFree(w.conv2_bf16);
    if (w.fc1_bf16) (void)hipFree(w.fc1_bf16);
    w.conv2_bf16 = w.fc1_bf16 = nullptr;
    if (w.conv1_h2) (void)hipFree(w.conv1_h2);
    w.conv1_h2 = nullptr;
    if (w.conv2_h2) (void)hipFree(w.conv2_h2);
    if (w.fc1_h2) (void)hipFree(w.fc1_h2);
    w.conv2_h2 = w.fc1_h2 = nullptr;
    w.loaded = false;
}

extern "C" int sv_ctx_destroy(sv_ctx *ctx)
{
    if (!ctx) return SV_OK;
    (void)hipSetDevice(ctx->device);
    free_weights(ctx->w);
    if (ctx->features) (void)hipFree(ctx->features);
    if (ctx->cells) (void)hipFree(ctx->cells);
    if (ctx->cells2) (void)hipFree(ctx->cells2);
    if (ctx->jpeg_planes) (void)hipFree(ctx->jpeg_planes);
    if (ctx->k1_list) (void)hipFree(ctx->k1_list);
    if (ctx->range_flag) (void)hipFree(ctx->range_flag);
    for (auto &t : ctx->timeline) { (void)hipEventDestroy(t.t0); (void)hipEventDestroy(t.t1); }
    for (auto &e : ctx->event_pool) (void)hipEventDestroy(e);
    delete ctx;
    return SV_OK;
}

int sv_ensure_scratch(sv_ctx *ctx, long cells)
{
    if (cells <= ctx->cap_cells) return SV_OK;
    SV_HIP(hipSetDevice(ctx->device));
    if (ctx->features) SV_HIP(hipFree(ctx->features));
    if (ctx->cells) SV_HIP(hipFree(ctx->cells));
    if (ctx->cells2) SV_HIP(hipFree(ctx->cells2));
    ctx->features = nullptr;
    ctx->cells = nullptr;
    ctx->cells2 = nullptr;
    ctx->cap_cells = 0;
    SV_HIP(hipMalloc((void **)&ctx->features, sizeof(float) * 3136 * (size_t)cells));
    SV_HIP(hipMalloc((void **)&ctx->cells, (size_t)SV_CELL_PX * (size_t)cells));
    SV_HIP(hipMalloc((void **)&ctx->cells2, (size_t)SV_CELL_PX * (size_t)cells));
    ctx->cap_cells = cells;
    return SV_OK;
}

extern "C" int sv_ctx_set_precision(sv_ctx *ctx, int precision)
{
    if (!ctx || (precision != SV_PREC_F32 && precision != SV_PREC_BF16)) return sv_fail(SV_ERR_BAD_ARG, "sv_ctx_set_precision: bad argument");
    ctx->precision = precision;
    return SV_OK;
}

extern "C" int sv_ctx_reserve(sv_ctx *ctx, long max_cells)
{
    if (!ctx || max_cells <= 0) return sv_fail(SV_ERR_BAD_ARG, "sv_ctx_reserve: bad argument");
    SV_HIP(hipSetDevice(ctx->device));
    if (!ctx->range_flag) SV_HIP(hipMalloc((void **)&ctx->range_flag, 2 * sizeof(int)));     // sv_cnn_forward_f32's per-call range flag: no hipMalloc after reserve
    return sv_ensure_scratch(ctx, max_cells);
}

// ---- per-kernel timing -------------------------------------------------------------------------------
sv_time_scope::sv_time_scope(sv_ctx *c, int kernel, hipStream_t st) : ctx(c), s(st)
{
    if (!ctx || !ctx->timing) return;
    hipEvent_t e[2];
    for (auto &ev : e) {
        if (!ctx->event_pool.empty()) { ev = ctx->event_pool.back(); ctx->event_pool.pop_back(); }
        else if (hipEventCreate(&ev) != hipSuccess) return;
    }
    (void)hipEventRecord(e[0], s);
    ctx->timeline.push_back({kernel, e[0], e[1]});
    idx = (int)ctx->timeline.size() - 1;
}

sv_time_scope::~sv_time_scope()
{
    if (idx >= 0) (void)hipEventRecord(ctx->timeline[idx].t1, s);
}

extern "C" int sv_timing_begin(sv_ctx *ctx)
{
    if (!ctx) return sv_fail(SV_ERR_BAD_ARG, "sv_timing_begin: NULL context");
    for (auto &t : ctx->timeline) { ctx->event_pool.push_back(t.t0); ctx->event_pool.push_back(t.t1); }
    ctx->timeline.clear();
    ctx->timing = true;
    return SV_OK;
}

extern "C" int sv_timing_end(sv_ctx *ctx, double *ms_total, long *launches, int n_kernels)
{
    if (!ctx || !ms_total || !launches || n_kernels < 0) return sv_fail(SV_ERR_BAD_ARG, "sv_timing_end: bad argument");
    ctx->timing = false;
    for (int k = 0; k < n_kernels; k++) { ms_total[k] = 0; launches[k] = 0; }
    for (auto &t : ctx->timeline) {
        SV_HIP(hipEventSynchronize(t.t1));
        float ms = 0;
        SV_HIP(hipEventElapsedTime(&ms, t.t0, t.t1));
        if (t.kernel < n_kernels) {                   // a caller built against an older header simply does not see the newer kernel ids
            ms_total[t.kernel] += ms;
            launches[t.kernel]++;
        }
        ctx->event_pool.push_back(t.t0);
        ctx->event_pool.push_back(t.t1);
    }
    ctx->timeline.clear();
    return SV_OK;
}

// ---- weights --------------------------------------------------------------------------------------
static int upload(float **dst, const std::vector<float> &v)
{
    SV_HIP(hipMalloc((void **)dst, v.size() * sizeof(float)));
    SV_HIP(hipMemcpy(*dst, v.data(), v.size() * sizeof(float), hipMemcpyHostToDevice));
    return SV_OK;
}

extern "C" int sv_load_weights_f32(sv_ctx *ctx, const float *blob)
{
    if (!ctx || !blob) return sv_fail(SV_ERR_BAD_ARG, "sv_load_weights_f32: NULL argument");
    SV_HIP(hipSetDevice(ctx->device));
    free_weights(ctx->w);
    const float *c1w = blob, *c1b = c1w + 288, *c2w = c1b + 32, *c2b = c2w + 18432, *f1w = c2b + 64,
                *f1b = f1w + 401408, *f2w = f1b + 128, *f2b = f2w + 1280;
    // conv2: [np][t][ks][lane]; lane -> oc = 32np + 16t + (lane&15), ic = icb + 8*(lane>>4); ks = tap*8 + icb
    std::vector<float> w2(2 * 2 * 72 * 64);
    for (int np = 0; np < 2; np++)
        for (int t = 0; t < 2; t++)
            for (int ks = 0; ks < 72; ks++)
                for (int lane = 0; lane < 64; lane++) {
                    const int oc = 32 * np + 16 * t + (lane & 15), tap = ks >> 3, ic = (ks & 7) + 8 * (lane >> 4);
                    w2[((np * 2 + t) * 72 + ks) * 64 + lane] = c2w[(oc * 32 + ic) * 9 + tap];
                }
    // fc1: [chunk][t][lane][e]; feature index k' = 16*chunk + 4*(lane>>4) + e = window*64 + oc;
    // the reference flattens NCHW: k = oc*49 + window (ml/model.py:38)
    std::vector<float> f1((size_t)196 * 8 * 64 * 4);
    for (int c = 0; c < 196; c++)
        for (int t = 0; t < 8; t++)
            for (int lane = 0; lane < 64; lane++)
                for (int e = 0; e < 4; e++) {
                    const int kp = 16 * c + 4 * (lane >> 4) + e, win = kp >> 6, oc = kp & 63, n = 16 * t + (lane & 15);
                    f1[(((size_t)c * 8 + t) * 64 + lane) * 4 + e] = f1w[(size_t)n * 3136 + oc * 49 + win];
                }
#ifdef SV_XCHECK
    // Winograd F(2x2,3x3) weights U = G g G^T (computed in double), as [nt][xi][ks][lane]: oc = 16nt + (lane&15), ic = 4ks + (lane>>4)
    std::vector<float> wino((size_t)4 * 16 * 8 * 64);
    // the same U split without error into three bf16 parts (each the next 8 mantissa bits, by truncation) for
    // k_conv_features_wsplit: [nt][xi][part][lane][j], oc = 16nt + (lane&15), ic = 8*(lane>>4) + j
    std::vector<uint16_t> wsplit((size_t)4 * 16 * 3 * 64 * 8);
    {
        const double G[4][3] = {{1, 0, 0}, {0.5, 0.5, 0.5}, {0.5, -0.5, 0.5}, {0, 0, 1}};
        for (int oc = 0; oc < 64; oc++)
            for (int ic = 0; ic < 32; ic++) {
                const float *g = c2w + (oc * 32 + ic) * 9;
                double Gg[4][3], U[4][4];
                for (int i = 0; i < 4; i++)
                    for (int j = 0; j < 3; j++) Gg[i][j] = G[i][0] * g[j] + G[i][1] * g[3 + j] + G[i][2] * g[6 + j];
                for (int i = 0; i < 4; i++)
                    for (int j = 0; j < 4; j++) U[i][j] = Gg[i][0] * G[j][0] + Gg[i][1] * G[j][1] + Gg[i][2] * G[j][2];
                const int nt = oc >> 4, lane = (oc & 15) + 16 * (ic & 3), ks = ic >> 2;
                for (int xi = 0; xi < 16; xi++) {
                    const float u = (float)U[xi >> 2][xi & 3];
                    wino[(((size_t)nt * 16 + xi) * 8 + ks) * 64 + lane] = u;
                    float rest = u;
                    for (int part = 0; part < 3; part++) {
                        uint32_t bits;
                        memcpy(&bits, &rest, 4);
                        bits &= 0xffff0000u;
                        float piece;
                        memcpy(&piece, &bits, 4);
                        rest -= piece;                                        // exact
                        wsplit[((((size_t)nt * 16 + xi) * 3 + part) * 64 + (oc & 15) + 16 * (ic >> 3)) * 8 + (ic & 7)] = (uint16_t)(bits >> 16);
                    }
                }
            }
    }
#endif
    // bf16 configuration: round-to-nearest-even images for v_mfma_f32_16x16x32_bf16.
    //   conv2 [tap][t][lane][j]: oc = 4*(lane&15) + t, ic = 8*(lane>>4) + j
    //   fc1   [step][t][lane][j]: n = 16t + (lane&15), feature k' = 32*step + 8*(lane>>4) + j = window*64 + oc
    auto bf16 = [](float f) -> uint16_t { uint32_t u; memcpy(&u, &f, 4); return (uint16_t)((u + 0x7FFFu + ((u >> 16) & 1u)) >> 16); };
    std::vector<uint16_t> w2b((size_t)9 * 4 * 64 * 8), fc1b((size_t)98 * 8 * 64 * 8);
    for (int tap = 0; tap < 9; tap++)
        for (int t = 0; t < 4; t++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    const int oc = 4 * (lane & 15) + t, ic = 8 * (lane >> 4) + j;
                    w2b[(((size_t)tap * 4 + t) * 64 + lane) * 8 + j] = bf16(c2w[(oc * 32 + ic) * 9 + tap]);
                }
    for (int st = 0; st < 98; st++)
        for (int t = 0; t < 8; t++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    const int kp = 32 * st + 8 * (lane >> 4) + j, win = kp >> 6, oc = kp & 63, n = 16 * t + (lane & 15);
                    fc1b[(((size_t)st * 8 + t) * 64 + lane) * 8 + j] = bf16(f1w[(size_t)n * 3136 + oc * 49 + win]);
                }
    // k3_cnn_h2.hip: w * 2^e = hi + lo, both f16 (round to nearest), e chosen so that max|w| * 2^e lies in [2^13, 2^14): hi is far
    // from f16's overflow (65504) and lo (~2^-11 of hi) stays a normal f16 for all but the tiniest weights
    auto pow2_scale = [](const float *v, size_t n) -> int {
        float m = 0.f;
        for (size_t i = 0; i < n; i++) m = std::fmax(m, std::fabs(v[i]));
        if (!(m > 0.f) || !std::isfinite(m)) return 0;
        const int e = 13 - std::ilogb(m);
        return e < -14 ? -14 : (e > 40 ? 40 : e);
    };
    auto split_h2 = [](float ws, uint16_t &hi, uint16_t &lo) {
        const _Float16 h = (_Float16)ws;
        const _Float16 l = (_Float16)(ws - (float)h);
        memcpy(&hi, &h, 2);
        memcpy(&lo, &l, 2);
    };
    const int e2 = pow2_scale(c2w, 18432), e1 = pow2_scale(f1w, 401408);
    const float s2 = std::ldexp(1.f, e2), s1 = std::ldexp(1.f, e1);
    ctx->w.conv2_h2_scale_inv = std::ldexp(1.f, -e2);
    ctx->w.fc1_h2_scale_inv = std::ldexp(1.f, -e1);
    std::vector<uint16_t> c2h((size_t)9 * 2 * 2 * 2 * 64 * 8), f1h((size_t)98 * 8 * 2 * 64 * 8);
    for (int tap = 0; tap < 9; tap++)
        for (int np = 0; np < 2; np++)
            for (int t = 0; t < 2; t++)
                for (int lane = 0; lane < 64; lane++)
                    for (int j = 0; j < 8; j++) {
                        const int oc = 32 * np + 2 * (lane & 15) + t, ic = 8 * (lane >> 4) + j;
                        const size_t base = ((((size_t)tap * 2 + np) * 2 + t) * 2) * 64 * 8;
                        split_h2(c2w[(oc * 32 + ic) * 9 + tap] * s2, c2h[base + (size_t)lane * 8 + j], c2h[base + 512 + (size_t)lane * 8 + j]);
                    }
    for (int st = 0; st < 98; st++)
        for (int t = 0; t < 8; t++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    // k-slot (q, j) of step st = feature 64*(st/2) + 16q + 8*(st%2) + j: a lane's operands of a 2-step stage are contiguous
                    const int kp = 64 * (st >> 1) + 16 * (lane >> 4) + 8 * (st & 1) + j, win = kp >> 6, oc = kp & 63, n = 16 * t + (lane & 15);
                    const size_t base = (((size_t)st * 8 + t) * 2) * 64 * 8;
                    split_h2(f1w[(size_t)n * 3136 + oc * 49 + win] * s1, f1h[base + (size_t)lane * 8 + j], f1h[base + 512 + (size_t)lane * 8 + j]);
                }
    // conv1 as a GEMM over the 4x4 input patch of a pooling window: k = 16*part + 4r + c (part 0: hi plane of the input, 1: lo plane),
    // column = channel 16*chalf + (lane&15) at conv position (dy, dx) of the window; B = w1[ch][r-dy][c-dx] (0 outside the 3x3).
    // MFMA 0 multiplies [xh | xl] by [wh | wh], MFMA 1 by [wl | 0]:  xh*wh + xl*wh + xh*wl.
    const int e0 = pow2_scale(c1w, 288);
    const float s0 = std::ldexp(1.f, e0);
    ctx->w.conv1_h2_scale_inv = std::ldexp(1.f, -e0);
    std::vector<uint16_t> c1h((size_t)2 * 4 * 2 * 64 * 8);
    for (int chalf = 0; chalf < 2; chalf++)
        for (int pos = 0; pos < 4; pos++)
            for (int lane = 0; lane < 64; lane++)
                for (int j = 0; j < 8; j++) {
                    const int kk = 8 * (lane >> 4) + j, part = kk >> 4, pp = kk & 15, r = pp >> 2, c = pp & 3;
                    const int ky = r - (pos >> 1), kx = c - (pos & 1), ch = 16 * chalf + (lane & 15);
                    uint16_t hi = 0, lo = 0;
                    if (ky >= 0 && ky < 3 && kx >= 0 && kx < 3) split_h2(c1w[ch * 9 + ky * 3 + kx] * s0, hi, lo);
                    const size_t base = (((size_t)chalf * 4 + pos) * 2) * 64 * 8 + (size_t)lane * 8 + j;
                    c1h[base] = hi;
                    c1h[base + 512] = part == 0 ? lo : 0;
                }
    SV_HIP(hipMalloc((void **)&ctx->w.conv1_h2, c1h.size() * 2));
    SV_HIP(hipMemcpy(ctx->w.conv1_h2, c1h.data(), c1h.size() * 2, hipMemcpyHostToDevice));
    SV_HIP(hipMalloc((void **)&ctx->w.conv2_h2, c2h.size() * 2));
    SV_HIP(hipMemcpy(ctx->w.conv2_h2, c2h.data(), c2h.size() * 2, hipMemcpyHostToDevice));
    SV_HIP(hipMalloc((void **)&ctx->w.fc1_h2, f1h.size() * 2));
    SV_HIP(hipMemcpy(ctx->w.fc1_h2, f1h.data(), f1h.size() * 2, hipMemcpyHostToDevice));
#ifdef SV_XCHECK
    SV_HIP(hipMalloc((void **)&ctx->w.conv2_wsplit, wsplit.size() * 2));
    SV_HIP(hipMemcpy(ctx->w.conv2_wsplit, wsplit.data(), wsplit.size() * 2, hipMemcpyHostToDevice));
    if (int rcx = upload(&ctx->w.conv2_wino, wino)) return rcx;
#endif
    SV_HIP(hipMalloc((void **)&ctx->w.conv2_bf16, w2b.size() * 2));
    SV_HIP(hipMemcpy(ctx->w.conv2_bf16, w2b.data(), w2b.size() * 2, hipMemcpyHostToDevice));
    SV_HIP(hipMalloc((void **)&ctx->w.fc1_bf16, fc1b.size() * 2));
    SV_HIP(hipMemcpy(ctx->w.fc1_bf16, fc1b.data(), fc1b.size() * 2, hipMemcpyHostToDevice));
    int rc;
    if ((rc = upload(&ctx->w.conv1_w, std::vector<float>(c1w, c1w + 288)))) return rc;
    if ((rc = upload(&ctx->w.conv1_b, std::vector<float>(c1b, c1b + 32)))) return rc;
    if ((rc = upload(&ctx->w.conv2_wreg, w2))) return rc;
    if ((rc = upload(&ctx->w.conv2_b, std::vector<float>(c2b, c2b + 64)))) return rc;
    if ((rc = upload(&ctx->w.fc1_wreg, f1))) return rc;
    if ((rc = upload(&ctx->w.fc1_b, std::vector<float>(f1b, f1b + 128)))) return rc;
    if ((rc = upload(&ctx->w.fc2_w, std::vector<float>(f2w, f2w + 1280)))) return rc;
    if ((rc = upload(&ctx->w.fc2_b, std::vector<float>(f2b, f2b + 10)))) return rc;
    // Range of the f16-pair kernels for these weights (k3_cnn_h2.hip carries inputs, conv1 activations and features as f16 pairs: each
    // must stay below f16's 65,504).  Worst case over inputs of magnitude <= xm:  |conv1| <= A1*xm + B1,  |features| <= A2*(A1*xm + B1) + B2
    // with A = the largest absolute row sum of a layer's weights, B = its largest |bias|.  h2_x_hi = the largest xm all three bounds allow;
    // 8-bit cells are in [-1, 1] after the glue, so the f16-pair kernels serve them iff h2_x_hi >= 1.  Below 2^-10 an input's low halves
    // are all f16-subnormal; such batches take the f32 kernels too.
    {
        auto row_sum_max = [](const float *wt, int rows, int cols) { double m = 0; for (int r = 0; r < rows; r++) { double a = 0; for (int c = 0; c < cols; c++) a += std::fabs((double)wt[(size_t)r * cols + c]); m = std::fmax(m, a); } return m; };
        auto abs_max = [](const float *v, int n) { double m = 0; for (int i = 0; i < n; i++) m = std::fmax(m, std::fabs((double)v[i])); return m; };
        const double LIM = 6.0e4, A1 = row_sum_max(c1w, 32, 9), B1 = abs_max(c1b, 32), A2 = row_sum_max(c2w, 64, 288), B2 = abs_max(c2b, 64);
        double hi = LIM;
        if (A1 > 0) hi = std::fmin(hi, (LIM - B1) / A1);
        else if (B1 > LIM) hi = -1;
        if (A2 > 0 && A1 > 0) hi = std::fmin(hi, ((LIM - B2) / A2 - B1) / A1);
        else if (A2 * B1 + B2 > LIM) hi = -1;
        if (!std::isfinite(A1) || !std::isfinite(A2) || !std::isfinite(B1) || !std::isfinite(B2) || !(hi == hi)) hi = -1;
        ctx->w.h2_x_hi = (float)hi;
        ctx->w.h2_x_lo = 0x1p-10f;
        ctx->w.h2_in_range = hi >= 1.0;
    }
    ctx->w.loaded = true;
    return SV_OK;
}

// ---- host math ------------------------------------------------------------------------------------
// cv2.getGaussianKernel(n, sigma<=0, CV_32F): fixed tables up to 7, else IEEE-double formula.
void sv_gaussian_taps_f32(int n, float *out)
{
    static const float t3[3] = {0.25f, 0.5f, 0.25f}, t5[5] = {0.0625f, 0.25f, 0.375f, 0.25f, 0.0625f},
                       t7[7] = {0.03125f, 0.109375f, 0.21875f, 0.28125f, 0.21875f, 0.109375f, 0.03125f};
    if (n == 1) { out[0] = 1.f; return; }
    if (n == 3) { memcpy(out, t3, sizeof t3); return; }
    if (n == 5) { memcpy(out, t5, sizeof t5); return; }
    if (n == 7) { memcpy(out, t7, sizeof t7); return; }
    const double sigma = std::fma((double)n, 0.15, 0.35);
    const double scale2x = -0.125 / (sigma * sigma);
    const int half = (n - 1) / 2;
    std::vector<double> v(half);
    double sum = 0.0;
    for (int i = 0, x = 1 - n; i < half; i++, x += 2) {
        v[i] = std::exp((double)(x * x) * scale2x);
        sum += v[i];
    }
    sum *= 2.0;
    sum += 1.0;
    const double inv = 1.0 / sum;
    for (int i = 0; i < half; i++) out[i] = out[n - 1 - i] = (float)(v[i] * inv);
    out[half] = (float)inv;
}

namespace {

// order_points, cv/grid.py:74-91: TL = argmin(x+y), BR = argmax(x+y), TR = argmin(y-x), BL = argmax(y-x)
void order_points(const float *p, float *o)
{
    int lo_s = 0, hi_s = 0, lo_d = 0, hi_d = 0;
    float s[4], d[4];
    for (int i = 0; i < 4; i++) { s[i] = p[2 * i] + p[2 * i + 1]; d[i] = p[2 * i + 1] - p[2 * i]; }
    for (int i = 1; i < 4; i++) {
        if (s[i] < s[lo_s]) lo_s = i;
        if (s[i] > s[hi_s]) hi_s = i;
        if (d[i] < d[lo_d]) lo_d = i;
        if (d[i] > d[hi_d]) hi_d = i;
    }
    const int pick[4] = {lo_s, lo_d, hi_s, hi_d};
    for (int i = 0; i < 4; i++) { o[2 * i] = p[2 * pick[i]]; o[2 * i + 1] = p[2 * pick[i] + 1]; }
}

// cv2.getPerspectiveTransform: 8x8 LU with partial pivoting in double, OpenCV's elimination order
bool perspective_transform(const float *src, const float *dst, double *M)
{
    double a[8][9];  // augmented [A | b]
    for (int i = 0; i < 4; i++) {
        const double sx = src[2 * i], sy = src[2 * i + 1], dx = dst[2 * i], dy = dst[2 * i + 1];
        const double r0[9] = {sx, sy, 1, 0, 0, 0, -sx * dx, -sy * dx, dx};
        const double r1[9] = {0, 0, 0, sx, sy, 1, -sx * dy, -sy * dy, dy};
        memcpy(a[i], r0, sizeof r0);
        memcpy(a[i + 4], r1, sizeof r1);
    }
    for (int i = 0; i < 8; i++) {
        int piv = i;
        for (int j = i + 1; j < 8; j++)
            if (std::fabs(a[j][i]) > std::fabs(a[piv][i])) piv = j;
        if (!(std::fabs(a[piv][i]) >= 2.220446049250313e-16 * 100)) return false;  // also rejects NaN
        if (piv != i)
            for (int j = i; j < 9; j++) std::swap(a[i][j], a[piv][j]);
        const double d = -1 / a[i][i];
        for (int j = i + 1; j < 8; j++) {
            const double alpha = a[j][i] * d;
            for (int k = i + 1; k < 9; k++) a[j][k] += alpha * a[i][k];
        }
    }
    for (int i = 7; i >= 0; i--) {
        double s = a[i][8];
        for (int k = i + 1; k < 8; k++) s -= a[i][k] * a[k][8];
        a[i][8] = s / a[i][i];
    }
    for (int i = 0; i < 8; i++) M[i] = a[i][8];
    M[8] = 1.0;
    return true;
}

// cv::invert for 3x3 doubles: cofactors scaled by 1/det
bool invert3(const double *S, double *D)
{
    double det = S[0] * (S[4] * S[8] - S[5] * S[7]) - S[1] * (S[3] * S[8] - S[5] * S[6]) + S[2] * (S[3] * S[7] - S[4] * S[6]);
    if (det == 0.) return false;
    det = 1. / det;
    D[0] = (S[4] * S[8] - S[5] * S[7]) * det;
    D[1] = (S[2] * S[7] - S[1] * S[8]) * det;
    D[2] = (S[1] * S[5] - S[2] * S[4]) * det;
    D[3] = (S[5] * S[6] - S[3] * S[8]) * det;
    D[4] = (S[0] * S[8] - S[2] * S[6]) * det;
    D[5] = (S[2] * S[3] - S[0] * S[5]) * det;
    D[6] = (S[3] * S[7] - S[4] * S[6]) * det;
    D[7] = (S[1] * S[6] - S[0] * S[7]) * det;
    D[8] = (S[0] * S[4] - S[1] * S[3]) * det;
    return true;
}

}  // namespace

static bool corners_to_minv_one(const float *corners, int out_size, float inset_ratio, double *minv)
{
    float o[8], in[8];
    order_points(corners, o);
    // float32 numpy arithmetic of cv/grid.py:113-121
    const float cx = (((o[0] + o[2]) + o[4]) + o[6]) / 4.f, cy = (((o[1] + o[3]) + o[5]) + o[7]) / 4.f;
    for (int i = 0; i < 4; i++) {
        const float dx = cx - o[2 * i], dy = cy - o[2 * i + 1];
        const float dist = std::sqrt(dx * dx + dy * dy);
        const float amt = dist * inset_ratio;
        in[2 * i] = o[2 * i] + (dx / dist) * amt;
        in[2 * i + 1] = o[2 * i + 1] + (dy / dist) * amt;
    }
    const float S = (float)(out_size - 1);
    const float dst[8] = {0, 0, S, 0, S, S, 0, S};
    double M[9];
    return perspective_transform(in, dst, M) && invert3(M, minv);
}

extern "C" int sv_corners_to_minv(const float *corners, int n, int out_size, float inset_ratio, double *minv)
{
    if (!corners || !minv || n <= 0 || out_size < 2) return sv_fail(SV_ERR_BAD_ARG, "sv_corners_to_minv: bad argument");
    for (int f = 0; f < n; f++)
        if (!corners_to_minv_one(corners + 8 * f, out_size, inset_ratio, minv + 9 * f))
            return sv_fail(SV_ERR_DEGENERATE, "sv_corners_to_minv: frame %d: corners do not define a homography", f);
    return SV_OK;
}

extern "C" int sv_corners_to_minv_batch(const float *corners, int n, int out_size, float inset_ratio, double *minv, uint8_t *ok)
{
    if (!corners || !minv || !ok || n <= 0 || out_size < 2) return sv_fail(SV_ERR_BAD_ARG, "sv_corners_to_minv_batch: bad argument");
    static const double ident[9] = {1, 0, 0, 0, 1, 0, 0, 0, 1};
    for (int f = 0; f < n; f++) {
        ok[f] = corners_to_minv_one(corners + 8 * f, out_size, inset_ratio, minv + 9 * f) ? 1 : 0;
        if (!ok[f]) memcpy(minv + 9 * f, ident, sizeof ident);
    }
    return SV_OK;
}

// ---- argument checks + dispatch ---------------------------------------------------------------------
#define REQUIRE(cond, what) \
    do { if (!(cond)) return sv_fail(SV_ERR_BAD_ARG, "%s: %s", __func__, what); } while (0)

static inline hipStream_t S(void *s) { return (hipStream_t)s; }

extern "C" int sv_gray_u8(sv_ctx *ctx, const uint8_t *bgr, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride, uint8_t *gray, void *stream)
{
    REQUIRE(ctx && bgr && gray, "NULL argument");
    REQUIRE(n > 0 && H > 0 && W > 0 && pitch >= 3 * (ptrdiff_t)W, "bad shape");
    return svk_gray(bgr, n, H, W, pitch, img_stride, gray, S(stream));
}

extern "C" int sv_blur_u8(sv_ctx *ctx, const uint8_t *src, int n, int H, int W, int ksize, uint8_t *dst, void *stream)
{
    REQUIRE(ctx && src && dst, "NULL argument");
    REQUIRE(n > 0 && H > 0 && W > 0, "bad shape");
    REQUIRE(ksize > 0 && (ksize & 1), "ksize must be odd and positive");
    if (ksize > 7) return sv_fail(SV_ERR_UNSUPPORTED, "sv_blur_u8: ksize %d (only 1,3,5,7 are restated bit-exactly)", ksize);
    return svk_blur(src, n, H, W, ksize, dst, S(stream));
}

extern "C" int sv_adaptive_threshold_u8(sv_ctx *ctx, const uint8_t *src, int n, int H, int W, int block, double c, int type_inv, uint8_t *dst, void *stream)
{
    REQUIRE(ctx && src && dst, "NULL argument");
    REQUIRE(n > 0 && H > 0 && W > 0, "bad shape");
    REQUIRE(block > 1 && (block & 1), "block_size must be odd and > 1");
    if (block > 31) return sv_fail(SV_ERR_UNSUPPORTED, "sv_adaptive_threshold_u8: block_size %d > 31", block);
    float taps[31];
    sv_gaussian_taps_f32(block, taps);
    const int idelta = type_inv ? (int)std::floor(c) : (int)std::ceil(c);
    return svk_adaptive_threshold(src, n, H, W, block, taps, idelta, type_inv ? 1 : 0, dst, S(stream));
}

extern "C" int sv_preprocess_u8(sv_ctx *ctx, const uint8_t *bgr, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride, uint8_t *binary, void *stream)
{
    REQUIRE(ctx && bgr && binary, "NULL argument");
    REQUIRE(n > 0 && n < 65536 && H > 0 && W > 0 && pitch >= 3 * (ptrdiff_t)W, "bad shape");
    return svk_preprocess(ctx, bgr, n, H, W, pitch, img_stride, binary, S(stream));
}

extern "C" int sv_preprocess_warp_cells_u8(sv_ctx *ctx, const uint8_t *bgr, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride, uint8_t *binary,
                                           const double *minv, uint8_t *cells, void *stream)
{
    REQUIRE(ctx && bgr && binary && minv && cells, "NULL argument");
    REQUIRE(n > 0 && n < 65536 && H > 0 && W > 0 && pitch >= 3 * (ptrdiff_t)W, "bad shape");
    return svk_preprocess_warp_fused(ctx, bgr, n, H, W, pitch, img_stride, binary, minv, cells, S(stream));
}

#ifdef SV_XCHECK   // the matrix-pipe formulation of K1: a second implementation, built into the test-only library
extern "C" int sv_preprocess_mm_u8(sv_ctx *ctx, const uint8_t *bgr, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride, uint8_t *binary, float *mean, void *stream)
{
    REQUIRE(ctx && bgr && binary, "NULL argument");
    REQUIRE(n > 0 && n < 65536 && H > 0 && W > 0 && pitch >= 3 * (ptrdiff_t)W, "bad shape");
    if (!svk_preprocess_mm_supported(bgr, H, W, pitch, img_stride, binary, false))
        return sv_fail(SV_ERR_UNSUPPORTED, "sv_preprocess_mm_u8: needs H, W >= 16, W %% 16 == 0, 4-byte aligned frames and a 16-byte aligned output");
    return svk_preprocess_mm(ctx, bgr, n, H, W, pitch, img_stride, binary, false, mean, S(stream));
}

extern "C" int sv_preprocess_stats(sv_ctx *ctx, unsigned *ambiguous, unsigned long *capacity)
{
    REQUIRE(ctx && ambiguous && capacity, "NULL argument");
    if (!ctx->k1_list) { *ambiguous = 0; *capacity = 0; return svk_preprocess_mm_enable_stats(ctx); }
    return svk_preprocess_mm_stats(ctx, ambiguous, capacity);
}

#endif  // SV_XCHECK

extern "C" int sv_preprocess_bits_u8(sv_ctx *ctx, const uint8_t *bgr, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride, uint32_t *bits, void *stream)
{
    REQUIRE(ctx && bgr && bits, "NULL argument");
    REQUIRE(n > 0 && n < 65536 && H > 0 && W > 0 && pitch >= 3 * (ptrdiff_t)W, "bad shape");
    REQUIRE(((uintptr_t)bits & 3) == 0, "bits must be 4-byte aligned");
    return svk_preprocess_bits(ctx, bgr, n, H, W, pitch, img_stride, bits, S(stream));
}

extern "C" int sv_despeckle_bits(sv_ctx *ctx, uint32_t *bits, int n, int H, int W, void *stream)
{
    REQUIRE(ctx && bits, "NULL argument");
    REQUIRE(n > 0 && H > 0 && W > 0 && W % 32 == 0, "bad shape (W must be a multiple of 32)");
    return svk_despeckle_bits(bits, n, H, W, S(stream));
}

extern "C" int sv_despeckle_u8(sv_ctx *ctx, const uint8_t *binary, int n, int H, int W, uint8_t *out, uint32_t *packed, void *stream)
{
    REQUIRE(ctx && binary && out, "NULL argument");
    REQUIRE(n > 0 && H > 0 && W > 0, "bad shape");
    REQUIRE(!packed || W % 32 == 0, "packed output needs W % 32 == 0");
    return svk_despeckle(binary, n, H, W, out, packed, S(stream));
}

extern "C" long sv_sparse_bits_record_bytes(int H, int W, long cap_values)
{
    if (H <= 0 || W <= 0 || (W & 31) || cap_values < 0) return -1;
    const long gpr = ((W >> 5) + 63) / 64;
    return (8 + 8 * gpr * H + 4 * cap_values + 15) / 16 * 16;
}

extern "C" int sv_pack_sparse_bits(sv_ctx *ctx, const uint32_t *bits, int n, int H, int W, uint8_t *records, long record_stride, void *stream)
{
    REQUIRE(ctx && bits && records, "NULL argument");
    REQUIRE(n > 0 && H > 0 && W > 0 && W % 32 == 0, "bad shape (W must be a multiple of 32)");
    const long G = (long)H * (((W >> 5) + 63) / 64);
    REQUIRE(G <= 16000, "frame too large for the sparse record (more than 16000 row groups)");
    REQUIRE(record_stride % 8 == 0 && record_stride >= 8 + 8 * G + 4 && ((uintptr_t)records & 7) == 0, "record stride must be a multiple of 8 with room for the masks");
    return svk_pack_sparse_bits(bits, n, H, W, records, record_stride, S(stream));
}

extern "C" int sv_copy_to_pinned_host(sv_ctx *ctx, const void *src, void *dst_host, size_t bytes, void *stream)
{
    REQUIRE(ctx && src && dst_host, "NULL argument");
    REQUIRE((((uintptr_t)src | (uintptr_t)dst_host) & 15) == 0, "source and destination must be 16-byte aligned");
    if (bytes == 0) return SV_OK;
    hipPointerAttribute_t attr;
    if (hipPointerGetAttributes(&attr, dst_host) != hipSuccess || attr.type != hipMemoryTypeHost) {
        (void)hipGetLastError();
        return sv_fail(SV_ERR_BAD_ARG, "sv_copy_to_pinned_host: destination is not pinned (hipHostMalloc / hipHostRegister) host memory");
    }
    return svk_copy_to_host(src, dst_host, bytes, S(stream));
}

extern "C" int sv_warp_perspective_u8(sv_ctx *ctx, const uint8_t *img, int H, int W, ptrdiff_t pitch, int channels, const double *minv, int out_size, uint8_t *dst, void *stream)
{
    REQUIRE(ctx && img && minv && dst, "NULL argument");
    REQUIRE(H > 0 && W > 0 && out_size > 0 && out_size < 32768, "bad shape");
    REQUIRE(channels == 1 || channels == 3, "channels must be 1 or 3");
    REQUIRE(pitch >= (ptrdiff_t)W * channels, "pitch < row bytes");
    return svk_warp_perspective(img, H, W, pitch, channels, minv, out_size, dst, S(stream));
}

extern "C" int sv_extract_cells_u8(sv_ctx *ctx, const uint8_t *grid, int h, int w, ptrdiff_t pitch, int channels, int cell_size, int margin_h, int margin_w, uint8_t *cells, void *stream)
{
    REQUIRE(ctx && grid && cells, "NULL argument");
    REQUIRE(h >= 9 && w >= 9 && cell_size > 0, "bad shape");
    REQUIRE(channels == 1 || channels == 3, "channels must be 1 or 3");
    REQUIRE(margin_h >= 0 && margin_w >= 0 && h / 9 - 2 * margin_h > 0 && w / 9 - 2 * margin_w > 0, "margin leaves an empty cell");
    return svk_extract_cells(grid, h, w, pitch, channels, cell_size, margin_h, margin_w, cells, S(stream));
}

extern "C" int sv_warp_cells_u8(sv_ctx *ctx, const uint8_t *frames, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t frame_stride, const double *minv, uint8_t *cells, void *stream)
{
    REQUIRE(ctx && frames && minv && cells, "NULL argument");
    REQUIRE(n > 0 && n < 65536 && H > 0 && W > 0 && pitch >= 3 * (ptrdiff_t)W, "bad shape");
    return svk_warp_cells(ctx, frames, n, H, W, pitch, frame_stride, minv, cells, S(stream));
}

static int cnn_common(sv_ctx *ctx, const void *x, bool u8in, int glue, long B, float *logits, uint8_t *digits, float *conf, void *stream)
{
    if (!ctx || !x || !logits) return sv_fail(SV_ERR_BAD_ARG, "sv_cnn_forward: NULL argument");
    if (B <= 0) return sv_fail(SV_ERR_BAD_ARG, "sv_cnn_forward: B = %ld", B);
    if (!ctx->w.loaded) return sv_fail(SV_ERR_NO_WEIGHTS, "sv_cnn_forward: call sv_load_weights_f32 first");
    int rc = sv_ensure_scratch(ctx, B);
    if (rc) return rc;
    if (glue != SV_GLUE_NORMALIZE && glue != SV_GLUE_RUNPY) return sv_fail(SV_ERR_BAD_ARG, "sv_cnn_forward: glue %d", glue);
    if (ctx->precision == SV_PREC_BF16) {
        if (!u8in) return sv_fail(SV_ERR_UNSUPPORTED, "sv_cnn_forward_f32: the bf16 configuration takes 8-bit cells (sv_cnn_forward_cells_u8 / sv_frames_to_digits)");
        const uint8_t *c = (const uint8_t *)x;
        if (glue == SV_GLUE_RUNPY) {
            if ((rc = svk_preprocess_cells(c, B, ctx->cells2, S(stream)))) return rc;
            c = ctx->cells2;
        }
        return svk_cnn_forward_bf16(ctx, c, B, logits, digits, conf, S(stream));
    }
    return svk_cnn_forward(ctx, x, u8in, glue, B, logits, digits, conf, S(stream));
}

extern "C" int sv_cnn_forward_f32(sv_ctx *ctx, const float *x, long B, float *logits, uint8_t *digits, float *conf, void *stream)
{
    return cnn_common(ctx, x, false, SV_GLUE_NORMALIZE, B, logits, digits, conf, stream);
}

extern "C" int sv_cnn_forward_cells_u8(sv_ctx *ctx, const uint8_t *cells, long B, int glue, float *logits, uint8_t *digits, float *conf, void *stream)
{
    return cnn_common(ctx, cells, true, glue, B, logits, digits, conf, stream);
}

extern "C" int sv_resize_linear_u8(sv_ctx *ctx, const uint8_t *src, int sh, int sw, ptrdiff_t pitch, uint8_t *dst, int dh, int dw, void *stream)
{
    REQUIRE(ctx && src && dst, "NULL argument");
    REQUIRE(sh > 0 && sw > 0 && dh > 0 && dw > 0 && dh < 65536 && pitch >= sw, "bad shape");
    return svk_resize_linear(src, sh, sw, pitch, dst, dh, dw, S(stream));
}

extern "C" int sv_cell_ink_ratio_u8(sv_ctx *ctx, const uint8_t *cells, long B, int cell_px, float *ratio, int *otsu, void *stream)
{
    REQUIRE(ctx && cells && ratio, "NULL argument");
    REQUIRE(B > 0 && B < 2147483647L && cell_px > 0, "bad shape");
    return svk_cell_ink_ratio(cells, B, cell_px, ratio, otsu, S(stream));
}

extern "C" int sv_preprocess_cells_u8(sv_ctx *ctx, const uint8_t *cells, long B, uint8_t *out, void *stream)
{
    REQUIRE(ctx && cells && out, "NULL argument");
    REQUIRE(B > 0, "B must be positive");
    return svk_preprocess_cells(cells, B, out, S(stream));
}

extern "C" int sv_softmax_topk_f32(sv_ctx *ctx, const float *logits, long B, int k, uint8_t *index, float *prob, void *stream)
{
    REQUIRE(ctx && logits && index && prob, "NULL argument");
    REQUIRE(B > 0 && k >= 1 && k <= SV_CLASSES, "need B > 0 and 1 <= k <= 10");
    return svk_softmax_topk(logits, B, k, index, prob, S(stream));
}

static int jpeg_info_ok(const sv_jpeg_info *info, ptrdiff_t pitch, const char *fn)
{
    const char *what = nullptr;
    const bool swap = info->orientation >= 5;
    if (!(info->width > 0 && info->height > 0 && info->width < 65536 && info->height < 65536)) what = "bad image size";
    else if (!(info->components == 1 || info->components == 3)) what = "components must be 1 or 3";
    else if (!((info->h_samp == 1 && info->v_samp == 1) || (info->h_samp == 2 && (info->v_samp == 1 || info->v_samp == 2)))) what = "sampling must be 1x1, 2x1 or 2x2";
    else if (!(info->orientation >= 1 && info->orientation <= 8)) what = "orientation must be 1..8";
    else if (!(info->out_width == (swap ? info->height : info->width) && info->out_height == (swap ? info->width : info->height))) what = "out_width/out_height do not match the orientation";
    else if (pitch < 3 * (ptrdiff_t)info->out_width) what = "pitch smaller than a row";
    return what ? sv_fail(SV_ERR_BAD_ARG, "%s: %s", fn, what) : SV_OK;
}

extern "C" int sv_jpeg_reconstruct_bgr_u8(sv_ctx *ctx, const sv_jpeg_info *info, const int16_t *coef, const uint16_t *quant, uint8_t *bgr, ptrdiff_t pitch, void *stream)
{
    REQUIRE(ctx && info && coef && quant && bgr, "NULL argument");
    const int rc = jpeg_info_ok(info, pitch, "sv_jpeg_reconstruct_bgr_u8");
    if (rc) return rc;
    REQUIRE(((uintptr_t)coef & 15) == 0 && ((uintptr_t)quant & 15) == 0, "coef and quant must be 16-byte aligned");
    return svk_jpeg_reconstruct(ctx, info, coef, nullptr, nullptr, nullptr, quant, bgr, pitch, S(stream));
}

extern "C" int sv_jpeg_reconstruct_sparse_bgr_u8(sv_ctx *ctx, const sv_jpeg_info *info, const uint64_t *masks, const uint32_t *offsets, const int16_t *values,
                                                 const uint16_t *quant, uint8_t *bgr, ptrdiff_t pitch, void *stream)
{
    REQUIRE(ctx && info && masks && offsets && values && quant && bgr, "NULL argument");
    const int rc = jpeg_info_ok(info, pitch, "sv_jpeg_reconstruct_sparse_bgr_u8");
    if (rc) return rc;
    REQUIRE(((uintptr_t)masks & 7) == 0 && ((uintptr_t)offsets & 3) == 0 && ((uintptr_t)values & 1) == 0 && ((uintptr_t)quant & 15) == 0, "misaligned argument");
    return svk_jpeg_reconstruct(ctx, info, nullptr, masks, offsets, values, quant, bgr, pitch, S(stream));
}

extern "C" int sv_frames_to_digits(sv_ctx *ctx, const uint8_t *frames, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t frame_stride, const double *minv, int glue, uint8_t *cells, float *logits, uint8_t *digits, float *conf, void *stream)
{
    REQUIRE(ctx && frames && minv && logits && digits, "NULL argument");
    REQUIRE(n > 0 && n < 65536 && H > 0 && W > 0 && pitch >= 3 * (ptrdiff_t)W, "bad shape");
    if (!ctx->w.loaded) return sv_fail(SV_ERR_NO_WEIGHTS, "sv_frames_to_digits: call sv_load_weights_f32 first");
    const long B = (long)n * SV_CELLS;
    int rc = sv_ensure_scratch(ctx, B);
    if (rc) return rc;
    uint8_t *c = cells ? cells : ctx->cells;
    if ((rc = svk_warp_cells(ctx, frames, n, H, W, pitch, frame_stride, minv, c, S(stream)))) return rc;
    return cnn_common(ctx, c, true, glue, B, logits, digits, conf, stream);
}
