// Internal declarations shared by the translation units of libsudokuvision_hip.so.
#pragma once
#include <hip/hip_runtime.h>
#include <cstdarg>
#include <cstdint>
#include <cstdio>
#include <vector>

#include "../../include/sudoku_vision_hip.h"
#ifdef SV_XCHECK
#include "../../include/sudoku_vision_xcheck.h"
#endif

typedef uint8_t u8;

// Packed DigitCNN weights as the kernels consume them (see k3_cnn.hip for the layouts).
struct sv_weights {
    float *conv1_w = nullptr;   // [32][9]
    float *conv1_b = nullptr;   // [32]
    float *conv2_wreg = nullptr;// [2 np][2 t][72 ks][64 lane]  MFMA B-operand register image
    float *conv2_wino = nullptr;// [4 nt][16 xi][8 ks][64 lane]  Winograd U = G g G^T as MFMA B-operand image
    float *conv2_b = nullptr;   // [64]
    float *fc1_wreg = nullptr;  // [196 chunk][8 t][64 lane][4 e] MFMA B-operand register image
    float *fc1_b = nullptr;     // [128]
    unsigned short *conv2_wsplit = nullptr; // [4 nt][16 xi][3 parts][64 lane][8] U split into three bf16 parts (k_conv_features_wsplit)
    unsigned short *conv2_bf16 = nullptr; // [9 tap][4 t][64 lane][8] bf16 MFMA B image (bf16 configuration)
    unsigned short *fc1_bf16 = nullptr;   // [98 step][8 t][64 lane][8] bf16
    // k3_cnn_h2.hip: weights x 2^e split into f16 hi + lo (w * 2^e = hi + lo to 22 bits); scale_inv = 2^-e
    unsigned short *conv2_h2 = nullptr;   // [9 tap][2 np][2 t][2 part][64 lane][8] f16: oc = 32np + 2*(lane&15) + t, ic = 8*(lane>>4) + j
    unsigned short *fc1_h2 = nullptr;     // [98 step][8 t][2 part][64 lane][8] f16: n = 16t + (lane&15), feature = 64*(step/2) + 16*(lane>>4) + 8*(step%2) + j
    unsigned short *conv1_h2 = nullptr;   // [2 chalf][4 pos][2 mfma][64 lane][8] f16: conv1 as a GEMM over the 4x4 patch of a pooling window (k3_cnn_h2.hip)
    float conv1_h2_scale_inv = 1.f, conv2_h2_scale_inv = 1.f, fc1_h2_scale_inv = 1.f;
    // range of the f16-pair kernels for THESE weights (sv_load_weights_f32): with inputs in [-1, 1] (8-bit cells after the glue) every activation
    // stays below the f16 range iff h2_in_range; an f32 input batch is inside it iff h2_x_lo <= max|x| <= h2_x_hi (0 > hi: never)
    bool h2_in_range = true;
    float h2_x_lo = 0.f, h2_x_hi = 0.f;
    float *fc2_w = nullptr;     // [10][128]
    float *fc2_b = nullptr;     // [10]
    bool loaded = false;
};

struct sv_ctx {
    int device = 0;
    int num_cus = 256;
    sv_weights w;
    // grow-only scratch
    float *features = nullptr;  // [cells][49][64] pooled conv2 output
    u8 *cells = nullptr;        // [cells][784]
    u8 *cells2 = nullptr;       // [cells][784] preprocess_cell output (SV_GLUE_RUNPY)
    long cap_cells = 0;
    u8 *jpeg_planes = nullptr;  // decoded component planes (MCU-padded) between the IDCT and the colour kernel
    size_t cap_jpeg = 0;
    void *k1_list = nullptr;    // k1_threshold_mm.hip: optional diagnostic counter (pixels decided by the exact evaluation), sv_preprocess_stats
    int precision = 0;          // SV_PREC_F32 / SV_PREC_BF16 (sv_ctx_set_precision)
    int cnn_kernels = 0;        // SV_CNN_AUTO / _F16PAIR / _F32MFMA (sv_ctx_set_cnn_kernels)
    int *range_flag = nullptr;  // [2] device ints: the per-call kernel choice for f32 inputs (k3_cnn.hip k_input_range)
    bool x_fc_frame = false;    // xcheck builds: k_fc_head_frame for large batches
    int dev_ablate = 0;         // SV_DEV builds: k3_cnn_h2.hip ablation / stamp bits
    // optional per-kernel timing (sv_timing_begin/sv_timing_end): hipEvents on the launch stream
    bool timing = false;
    struct timed { int kernel; hipEvent_t t0, t1; };
    std::vector<timed> timeline;
    std::vector<hipEvent_t> event_pool;
};

enum sv_kernel_id { SVK_PREPROCESS = 0, SVK_WARP_CELLS = 1, SVK_CONV_FEATURES = 2, SVK_FC_HEAD = 3, SVK_FUSED12 = 4, SVK_COUNT = 5 };

// RAII bracket: records an event before and after a launch when ctx->timing is on
struct sv_time_scope {
    sv_ctx *ctx; hipStream_t s; int idx = -1;
    sv_time_scope(sv_ctx *c, int kernel, hipStream_t st);
    ~sv_time_scope();
};

int sv_fail(int code, const char *fmt, ...);

#define SV_HIP(expr)                                                                         \
    do {                                                                                     \
        hipError_t e_ = (expr);                                                              \
        if (e_ != hipSuccess) return sv_fail(SV_ERR_HIP, "%s: %s", #expr, hipGetErrorString(e_)); \
    } while (0)

#define SV_LAUNCH_CHECK(name)                                                                \
    do {                                                                                     \
        hipError_t e_ = hipGetLastError();                                                   \
        if (e_ != hipSuccess) return sv_fail(SV_ERR_HIP, "launch %s: %s", name, hipGetErrorString(e_)); \
    } while (0)

int sv_ensure_scratch(sv_ctx *ctx, long cells);

// kernel launchers (one per .hip file)
int svk_gray(const u8 *bgr, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride, u8 *gray, hipStream_t s);
int svk_blur(const u8 *src, int n, int H, int W, int ksize, u8 *dst, hipStream_t s);
int svk_adaptive_threshold(const u8 *src, int n, int H, int W, int block, const float *taps, int idelta, int type_inv,
                           u8 *dst, hipStream_t s);
int svk_preprocess(sv_ctx *ctx, const u8 *bgr, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride, u8 *binary, hipStream_t s);
int svk_preprocess_bits(sv_ctx *ctx, const u8 *bgr, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride, uint32_t *bits, hipStream_t s);
int svk_despeckle_bits(uint32_t *bits, int n, int H, int W, hipStream_t s);
int svk_preprocess_mm_stats(sv_ctx *ctx, unsigned *ambiguous, unsigned long *capacity);
int svk_preprocess_mm_enable_stats(sv_ctx *ctx);
bool svk_preprocess_mm_supported(const u8 *bgr, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride, const void *out, bool bits);
int svk_preprocess_mm(sv_ctx *ctx, const u8 *bgr, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride, u8 *out, bool bits, float *mean_dbg, hipStream_t s);
int svk_preprocess_warp_fused(sv_ctx *ctx, const u8 *bgr, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride, u8 *binary, const double *minv, u8 *cells,
                              hipStream_t s);
int svk_warp_perspective(const u8 *img, int H, int W, ptrdiff_t pitch, int channels, const double *minv, int out_size,
                         u8 *dst, hipStream_t s);
int svk_extract_cells(const u8 *grid, int h, int w, ptrdiff_t pitch, int channels, int cell_size, int margin_h,
                      int margin_w, u8 *cells, hipStream_t s);
int svk_warp_cells(sv_ctx *ctx, const u8 *frames, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t frame_stride, const double *minv,
                   u8 *cells, hipStream_t s);
int svk_cnn_forward(sv_ctx *ctx, const void *x, bool x_is_u8, int glue, long B, float *logits, u8 *digits, float *conf,
                    hipStream_t s);

int svk_despeckle(const u8 *src, int n, int H, int W, u8 *dst, unsigned *packed, hipStream_t s);
int svk_pack_sparse_bits(const uint32_t *bits, int n, int H, int W, u8 *records, long stride, hipStream_t s);
int svk_copy_to_host(const void *src, void *dst_host, size_t bytes, hipStream_t s);
int svk_resize_linear(const u8 *src, int sh, int sw, ptrdiff_t pitch, u8 *dst, int dh, int dw, hipStream_t s);
int svk_cnn_forward_bf16(sv_ctx *ctx, const u8 *cells, long B, float *logits, u8 *digits, float *conf, hipStream_t s);
int svk_cnn_forward_h2(sv_ctx *ctx, const void *x, bool x_is_u8, long B, float *logits, u8 *digits, float *conf, const int *run_if_clear, hipStream_t s);
int svk_cell_ink_ratio(const u8 *cells, long B, int npx, float *ratio, int *otsu, hipStream_t s);
int svk_preprocess_cells(const u8 *cells, long B, u8 *out, hipStream_t s);
int svk_jpeg_reconstruct(sv_ctx *ctx, const sv_jpeg_info *info, const int16_t *coef, const uint64_t *masks, const uint32_t *offsets, const int16_t *values,
                         const uint16_t *quant, u8 *bgr, ptrdiff_t pitch, hipStream_t s);   // dense when coef != nullptr, else sparse
int svk_softmax_topk(const float *logits, long B, int k, u8 *index, float *prob, hipStream_t s);

// host helpers
void sv_gaussian_taps_f32(int n, float *out);
