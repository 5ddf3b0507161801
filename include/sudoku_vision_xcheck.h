/*
 * sudoku_vision_xcheck.h -- the extra entry points of libsudokuvision_xcheck.so, a TEST-ONLY superset build of libsudokuvision_hip.so
 * (csrc/Makefile: the product's sources with -DSV_XCHECK + csrc/k1_threshold_mm.hip).  It carries independent second implementations of
 * two stages so that tests/ and tests/fuzz_gpu.py can compare the product's kernels with them on the GPU; the product library does not
 * contain them and no Python module of the sudoku-vision_amd package loads this one.  Everything in sudoku_vision_hip.h is exported here as well.
 */
#ifndef SUDOKU_VISION_XCHECK_H
#define SUDOKU_VISION_XCHECK_H

#include "sudoku_vision_hip.h"

#ifdef __cplusplus
extern "C" {
#endif

/* Further selections for sv_ctx_set_cnn_kernels: round 1's conv2 implementations (ml/model.py:36-37).
 *   SV_CNN_X_WINOGRAD  Winograd F(2x2,3x3) stream on v_mfma_f32_16x16x4_f32 (k_conv_features_wstream)
 *   SV_CNN_X_WSPLIT    the same stream on bf16 MFMA with every f32 operand split into three bf16 parts (k_conv_features_wsplit) */
#define SV_CNN_X_WINOGRAD 102
#define SV_CNN_X_WSPLIT 103

/* fc1 of the f32-MFMA path by the one-workgroup-per-frame kernel (k_fc_head_frame) for batches of >= 64 frames; off by default. */
int svx_ctx_set_fc_frame_kernel(sv_ctx *ctx, int on);

/* preprocess_for_grid_detection, cv/preprocess.py:57-65, in its matrix-pipe formulation (csrc/k1_threshold_mm.hip: the four separable passes as
 * Toeplitz GEMMs on the f16 MFMA, the 11x11 float mean approximated and every pixel it cannot decide re-decided with cv2's exact
 * sequence).  Same output, bit for bit; 2x slower than the marching kernel on MI355X.  mean (optional, dev, n*H*W floats): the approximate
 * local mean per pixel, for the tests' error measurement.  Needs H, W >= 16, W % 16 == 0, 4-byte aligned frames, 16-byte aligned output:
 * SV_ERR_UNSUPPORTED otherwise. */
int sv_preprocess_mm_u8(sv_ctx *ctx, const uint8_t *bgr /*dev*/, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride,
                        uint8_t *binary /*dev, n*H*W*/, float *mean /*dev or NULL*/, void *stream);

/* Diagnostics for sv_preprocess_mm_u8: the number of pixels, since the previous call, whose approximate local mean was too close to the
 * threshold to decide and which were therefore decided with cv2's exact float sequence.  The first call on a context switches the
 * counter on and returns 0.  capacity: reserved (0).  Synchronises the device. */
int sv_preprocess_stats(sv_ctx *ctx, unsigned *ambiguous, unsigned long *capacity);

#ifdef __cplusplus
}
#endif
#endif /* SUDOKU_VISION_XCHECK_H */
