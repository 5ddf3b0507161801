/*
 * sudoku_vision_hip.h -- C ABI of libsudokuvision_hip.so, the MI355X (gfx950) implementation of the
 * sudoku-vision frame -> digits hot path.
 *
 * The reference (HueCodes/sudoku-vision) has no FFI: its boundary for this path is a set of Python
 * call signatures (cv/preprocess.py, cv/grid.py, cv/extract.py, ml/model.py) over cv2 / torch.
 * Each entry point below names the reference function (file:line under /root/reference) whose
 * arithmetic it replaces; sudoku-vision_amd/{cv,ml}/ re-expose them under the reference's names
 * through ctypes.
 *
 * Conventions
 *   - Plain pointers and sizes only.  Pointers marked [dev] are device (HBM) addresses valid on the
 *     context's device; [host] are host addresses.  The library never allocates or frees
 *     caller-visible memory; scratch and packed weights live inside the opaque context.
 *   - `stream` is a hipStream_t passed as void* (NULL = the default stream).  Every [dev] call is
 *     asynchronous on that stream; nothing synchronises the device.
 *   - Return value: SV_OK (0) or a negative sv_status.  sv_last_error() gives the message of the last
 *     failure on the calling thread.  No C++ exception crosses this boundary.
 *   - Images are 8-bit, row-major, `pitch` bytes between rows, BGR interleaved when 3-channel
 *     (what cv2.imread hands the reference, pipeline/run.py:250).
 *   - A context is bound to one device and is not thread-safe; use one per host thread/stream.
 */
#ifndef SUDOKU_VISION_HIP_H
#define SUDOKU_VISION_HIP_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

typedef struct sv_ctx sv_ctx;

typedef enum sv_status {
    SV_OK = 0,
    SV_ERR_BAD_ARG = -1,      /* null pointer, non-positive size, even kernel size, ... */
    SV_ERR_HIP = -2,          /* a HIP runtime call failed (message holds hipGetErrorString) */
    SV_ERR_NO_WEIGHTS = -3,   /* CNN entry point called before sv_load_weights_f32 */
    SV_ERR_UNSUPPORTED = -4,  /* parameter outside what this build restates (e.g. blur ksize 9) */
    SV_ERR_DEGENERATE = -5,   /* corners do not define a homography (singular system) */
    SV_ERR_BUFFER = -6        /* caller-provided output buffer too small (required sizes are reported) */
} sv_status;

#define SV_CNN_PARAMS 421642        /* ml/model.py: count_parameters(DigitCNN()) */
#define SV_CELLS 81
#define SV_CELL_PX 784              /* 28*28 */
#define SV_CLASSES 10

/* ---- library / context ------------------------------------------------------------------------ */

int sv_version(void);                               /* ABI version, currently 2 (2: sv_timing_end takes the array length; sv_ctx_set_cnn_kernels) */
const char *sv_last_error(void);                    /* thread-local, never NULL */

/* Creates a context on HIP device `device` (replaces nothing: the reference keeps no state except
 * the model object of pipeline/run.py:98-111). */
int sv_ctx_create(int device, sv_ctx **out);
int sv_ctx_destroy(sv_ctx *ctx);

/* Arithmetic of the CNN's conv2/fc1 (BASELINE.json configs[1] vs configs[4]).  SV_PREC_F32 (default): f32-grade results, logits
 * within 1e-4 of the PyTorch-CPU model (which kernels deliver them: sv_ctx_set_cnn_kernels below).  SV_PREC_BF16: bf16 operands, f32
 * accumulation (v_mfma_f32_16x16x32_bf16); parity target = predicted digit indices.  Applies to sv_cnn_forward_cells_u8 and
 * sv_frames_to_digits. */
#define SV_PREC_F32 0
#define SV_PREC_BF16 1
int sv_ctx_set_precision(sv_ctx *ctx, int precision);

/* Which kernels compute the SV_PREC_F32 forward.  Nothing in the process environment influences this.
 *   SV_CNN_AUTO (default)  csrc/k3_cnn_h2.hip -- every f32 operand as a pair of f16 halves (22 significant bits) on the f16 matrix pipe,
 *                          f32 accumulation, logits ~1e-6 from an exact evaluation -- whenever that is safe, else csrc/k3_cnn.hip's f32-MFMA
 *                          kernels.  "Safe" = inputs, conv1 activations and features stay inside f16's range (|v| < 65,504):
 *                          sv_load_weights_f32 bounds the activations from the weights (worst case over inputs in [-1, 1], which is what
 *                          8-bit cells become), and an f32 input batch (sv_cnn_forward_f32) is range-checked on the device per call -- out
 *                          of range, NaN/Inf or all below 2^-10, it takes the f32 kernels, with no host synchronisation either way.  So
 *                          the entry points accept what ml/model.py:34-42 accepts: any f32.
 *   SV_CNN_F16PAIR         the f16-pair kernels unconditionally (inputs/activations beyond 65,504 then overflow to inf)
 *   SV_CNN_F32MFMA         the f32-MFMA kernels unconditionally (true f32 throughout; about 4x slower) */
#define SV_CNN_AUTO 0
#define SV_CNN_F16PAIR 1
#define SV_CNN_F32MFMA 2
int sv_ctx_set_cnn_kernels(sv_ctx *ctx, int which);

/* Pre-sizes the context's scratch for batches of up to `max_cells` cells so that later calls do
 * no hipMalloc (needed before hipGraph capture). */
int sv_ctx_reserve(sv_ctx *ctx, long max_cells);

/* Measurement aid (no reference counterpart; pipeline/run.py:247-352 uses time.time()).  Between
 * sv_timing_begin and sv_timing_end every launch of the hot kernels is bracketed by hipEvents on
 * the stream it is launched on.  sv_timing_end waits for those events and returns, per kernel id
 * 0 = preprocess, 1 = warp_cells, 2 = conv_features, 3 = fc_head, 4 = the fused preprocess + warp_cells launch: total
 * milliseconds and launches. */
#define SV_TIMED_KERNELS 5
int sv_timing_begin(sv_ctx *ctx);
/* n_kernels = the length of the caller's two arrays (SV_TIMED_KERNELS of the header it was built against): ids >= n_kernels are not reported. */
int sv_timing_end(sv_ctx *ctx, double *ms_total /*host, n_kernels*/, long *launches /*host, n_kernels*/, int n_kernels);

/* Measurement aid: which conv/fc kernels sv_cnn_forward_cells_u8 / sv_frames_to_digits launch on this context with the weights
 * loaded (4 = f16 hi/lo operand pairs on the f16 matrix pipe, csrc/k3_cnn_h2.hip; 0 = f32 MFMA, csrc/k3_cnn.hip) and the matrix
 * instructions they issue per 28x28 cell: v_mfma_f32_16x16x4_f32 (2048 FLOP each) for conv2 and conv1 (0 = conv1 on the VALU) of the f32-MFMA
 * kernels, or v_mfma_f32_16x16x32_f16 (16384 FLOP each) for the conv kernel and the fc kernel of the default pair.  bench.py prices
 * a kernel's roofline fraction on the work it issues against the peak of the pipe it issues it on. */
int sv_conv_kernel_info(sv_ctx *ctx, int *algo, int *mfma_f32_conv2_per_cell, int *mfma_f32_conv1_per_cell,
                        int *mfma_f16_conv_per_cell, int *mfma_f16_fc_per_cell);

/* Loads DigitCNN weights: `blob` [host] is the state_dict flattened in key order
 * conv1.weight[32,1,3,3] conv1.bias[32] conv2.weight[64,32,3,3] conv2.bias[64] fc1.weight[128,3136]
 * fc1.bias[128] fc2.weight[10,128] fc2.bias[10] = SV_CNN_PARAMS floats.
 * Replaces model.load_state_dict(...) + model.to(device), pipeline/run.py:101-108.
 * Synchronous (packs on the host, copies, waits). */
int sv_load_weights_f32(sv_ctx *ctx, const float *blob);

/* ---- K1: preprocessing (cv/preprocess.py) ----------------------------------------------------- */

/* grayscale(), cv/preprocess.py:15-19 (cv2.cvtColor BGR2GRAY).  n images, `img_stride` bytes apart. */
int sv_gray_u8(sv_ctx *ctx, const uint8_t *bgr /*dev*/, int n, int H, int W, ptrdiff_t pitch,
               ptrdiff_t img_stride, uint8_t *gray /*dev, n*H*W*/, void *stream);

/* blur(), cv/preprocess.py:22-29 (cv2.GaussianBlur (k,k), sigma 0).  ksize in {1,3,5,7}. */
int sv_blur_u8(sv_ctx *ctx, const uint8_t *src /*dev, n*H*W*/, int n, int H, int W, int ksize,
               uint8_t *dst /*dev*/, void *stream);

/* threshold(), cv/preprocess.py:32-54 (cv2.adaptiveThreshold 255, GAUSSIAN_C).  block odd, 3..31.
 * type_inv: 1 = THRESH_BINARY_INV (preprocess.py:51), 0 = THRESH_BINARY (pipeline/run.py:91-93). */
int sv_adaptive_threshold_u8(sv_ctx *ctx, const uint8_t *src /*dev, n*H*W*/, int n, int H, int W,
                             int block, double c, int type_inv, uint8_t *dst /*dev*/, void *stream);

/* preprocess_for_grid_detection(), cv/preprocess.py:57-65: gray -> blur 5 -> threshold(11, 2, INV),
 * fused in one kernel.  n frames. */
int sv_preprocess_u8(sv_ctx *ctx, const uint8_t *bgr /*dev*/, int n, int H, int W, ptrdiff_t pitch,
                     ptrdiff_t img_stride, uint8_t *binary /*dev, n*H*W*/, void *stream);

/* BASELINE configs[4], "fused threshold/warp" for the device-only mode: preprocess_for_grid_detection (cv/preprocess.py:57-65) and
 * warp_perspective + extract_cells (cv/grid.py:94-133, cv/extract.py:13-56) of the same frames in ONE launch, for callers that know the
 * corners before thresholding (a tracker, the benchmark's generator corners).  Same outputs as sv_preprocess_u8 + sv_warp_cells_u8.
 * The grid places everything that reads frame f on the same XCD at the same time, so the two stages share the frame's bytes in L2.
 * Needs H, W >= 16, W % 4 == 0 and 4-byte aligned frames / binary: SV_ERR_UNSUPPORTED otherwise. */
int sv_preprocess_warp_cells_u8(sv_ctx *ctx, const uint8_t *bgr /*dev*/, int n, int H, int W, ptrdiff_t pitch, ptrdiff_t img_stride,
                                uint8_t *binary /*dev, n*H*W*/, const double *minv /*dev, n*9*/, uint8_t *cells /*dev, n*81*784*/, void *stream);

/* preprocess_for_grid_detection() with the binary written as 1 bit per pixel (bit = pixel != 0, LSB = leftmost, W/32
 * words per row, rows dense): the form the host corner search reads (sv_find_grid_corners_bits_batch) when nothing
 * else consumes the byte image.  Same pixels as sv_preprocess_u8.  Needs H, W >= 16, W % 32 == 0 and a 4-byte aligned
 * frame layout (pitch, img_stride, bgr), else SV_ERR_UNSUPPORTED -- use sv_preprocess_u8 + sv_despeckle_u8(packed). */
int sv_preprocess_bits_u8(sv_ctx *ctx, const uint8_t *bgr /*dev*/, int n, int H, int W, ptrdiff_t pitch,
                          ptrdiff_t img_stride, uint32_t *bits /*dev, n*H*W/32*/, void *stream);

/* sv_despeckle_u8 on a bit image, in place (same filter, same precondition, same result as its `packed` output). */
int sv_despeckle_bits(sv_ctx *ctx, uint32_t *bits /*dev, n*H*W/32*/, int n, int H, int W, void *stream);

/* Accelerator for the host corner search, not a reference stage: erases every connected component of a {0,255}
 * image that lies strictly inside a 64x64 tile (two offset tile grids).  Such components can neither be nor
 * influence the result of find_grid_contour (argument in csrc/k4_despeckle.hip), so
 * sv_find_grid_corners_u8(despeckled) == sv_find_grid_corners_u8(binary).  out may equal binary.
 * packed (optional, needs W % 32 == 0): the result as 1 bit per pixel (LSB = leftmost, W/32 words per row) for a
 * cheap D2H copy -- when given, `out` is scratch (first pass only) and `packed` holds the result; feed it to
 * sv_find_grid_corners_bits_batch.
 * PRECONDITION for the equality above: min_area_ratio * H * W > 61 * 61 (an erased component's bounding box is at most
 * 62x62 px, so its contour area is at most 61*61; with a smaller area floor, e.g. frames under ~193x193 at the default
 * ratio 0.1, the grid itself could be erased).  The library cannot check it (min_area_ratio belongs to the search);
 * callers must (sudoku-vision_amd/pipeline.py does). */
int sv_despeckle_u8(sv_ctx *ctx, const uint8_t *binary /*dev, n*H*W*/, int n, int H, int W,
                    uint8_t *out /*dev, n*H*W*/, uint32_t *packed /*dev, n*H*W/32, or NULL*/, void *stream);

/* Sparse form of sv_despeckle_u8's packed output for the D2H copy (a despeckled frame is mostly zero words: the hand-over to
 * the host search shrinks 3-4x).  Record of one frame, little endian:
 *   u32 n_values, u32 cap_values, u64 mask[H * gpr], u32 value[cap_values]        gpr = ceil(W/32/64)
 * mask[y * gpr + g] bit k = word 64g + k of row y is non-zero; the non-zero words follow in raster order.  n_values >
 * cap_values: the frame did not fit (values truncated) -- use the dense image for it.  cap_values is what fits in
 * record_stride (a multiple of 8; sv_sparse_bits_record_bytes gives the stride for a wanted capacity).
 * sv_find_grid_corners_sparse_batch / sv_sparse_bits_expand read records on the host. */
long sv_sparse_bits_record_bytes(int H, int W, long cap_values);
int sv_pack_sparse_bits(sv_ctx *ctx, const uint32_t *bits /*dev, n*H*W/32*/, int n, int H, int W,
                        uint8_t *records /*dev, n*record_stride*/, long record_stride, void *stream);

/* Device -> pinned host copy done by a kernel (16-byte stores into mapped host memory) instead of the DMA engine: the
 * hand-over of the binary to the host corner search (the reference hands cv2.findContours a numpy array, cv/grid.py:18;
 * here the array has to cross PCIe first).  ~55 GB/s against 22-30 for hipMemcpyAsync on the MI355X boxes measured
 * (tools/ubench_d2h.hip).  dst_host must be pinned host memory (hipHostMalloc, hipHostRegister, torch pin_memory) --
 * SV_ERR_BAD_ARG otherwise; both pointers 16-byte aligned.  Stream-ordered like a hipMemcpyAsync. */
int sv_copy_to_pinned_host(sv_ctx *ctx, const void *src /*dev*/, void *dst_host /*pinned host*/, size_t bytes, void *stream);

/* ---- host corner search (cv/grid.py:16-71; stays on the CPU, no context, no GPU) ----------------- */

/* find_grid_contour(binary, min_area_ratio), cv/grid.py:37-71, with approximate_polygon's
 * epsilon_ratio (:24-34): external contours (cv2.findContours RETR_EXTERNAL/CHAIN_APPROX_SIMPLE), largest
 * first, the first one >= min_area_ratio*H*W whose cv2.approxPolyDP(epsilon_ratio*perimeter) has 4
 * vertices.  binary [host].  Returns 1 and corners[8] = (x,y)*4 in approxPolyDP order, 0 if none
 * (the reference returns None), or a negative sv_status. */
int sv_find_grid_corners_u8(const uint8_t *binary /*host*/, int H, int W, ptrdiff_t pitch,
                            double min_area_ratio, double epsilon_ratio, int *corners /*host, 8*/);

/* The same for n images on `threads` host threads; found[i] = 1/0. */
int sv_find_grid_corners_batch_u8(const uint8_t *binary /*host*/, int n, int H, int W, ptrdiff_t pitch,
                                  ptrdiff_t img_stride, double min_area_ratio, double epsilon_ratio,
                                  int *corners /*host, n*8*/, uint8_t *found /*host, n*/, int threads);

/* sv_find_grid_corners_batch_u8 on bit-packed images (layout of sv_despeckle_u8's `packed`). */
int sv_find_grid_corners_bits_batch(const uint32_t *bits /*host, n*H*W/32*/, int n, int H, int W,
                                    double min_area_ratio, double epsilon_ratio,
                                    int *corners /*host, n*8*/, uint8_t *found /*host, n*/, int threads);

/* Restrict the library's host worker threads (the batch searches above, the JPEG entropy decoder) to the given CPUs --
 * normally the CPUs of the NUMA node the GPU's pinned buffers live on: on a two-socket MI355X host the search runs 10-20 %
 * slower and with 40-ms outliers when its threads wander to the other socket.  n = 0: no restriction for workers started
 * later.  The thread that calls a batch function works too and keeps its own mask. */
int sv_host_pool_set_affinity(const int *cpus /*host, n*/, int n);

/* The same on sparse records (sv_pack_sparse_bits).  found[i] = 2: record i overflowed, search its dense image instead. */
int sv_find_grid_corners_sparse_batch(const uint8_t *records /*host, n*record_stride*/, long record_stride, int n, int H, int W,
                                      double min_area_ratio, double epsilon_ratio,
                                      int *corners /*host, n*8*/, uint8_t *found /*host, n*/, int threads);

/* One sparse record -> the dense bit image (H*W/32 words).  SV_ERR_BUFFER if the record overflowed. */
int sv_sparse_bits_expand(const uint8_t *record /*host*/, int H, int W, uint32_t *bits /*host, H*W/32*/);

/* find_contours(), cv/grid.py:16-21.  Contours in cv2's order, concatenated: points = (x,y) pairs,
 * sizes[i] = vertices of contour i.  If a buffer is too small (or NULL) returns SV_ERR_BUFFER with
 * the required counts in n_points / n_contours. */
int sv_find_contours_u8(const uint8_t *binary /*host*/, int H, int W, ptrdiff_t pitch,
                        int *points /*host, cap_points*2*/, long cap_points, int *sizes /*host*/,
                        int cap_contours, long *n_points, int *n_contours);

/* The same on a bit-packed image (1 bit per pixel, LSB = leftmost, W/32 words per row -- sv_despeckle_u8's packed output; W % 32 == 0):
 * the scanner sv_find_grid_corners_bits_batch uses, which never expands the image to bytes.  Same contours, same order. */
int sv_find_contours_bits(const uint32_t *bits /*host*/, int H, int W, int *points, long cap_points, int *sizes, int cap_contours,
                          long *n_points, int *n_contours);

/* cv2.contourArea / cv2.arcLength / cv2.approxPolyDP on int32 (x,y) vertices (cv/grid.py:31-33,58,61). */
int sv_contour_area_i32(const int *xy /*host*/, int n, double *area);
int sv_arc_length_i32(const int *xy /*host*/, int n, int closed, double *length);
int sv_approx_poly_dp_i32(const int *xy /*host*/, int n, double epsilon, int closed,
                          int *out /*host, n*2*/, int *n_out);

/* ---- solver (host; scope row N4) -------------------------------------------------------------------- */

/* solve_sudoku(), solver/src/sudoku.c:72-81, as an in-process call instead of pipeline/run.py:163-202's
 * subprocess + /tmp files.  grid/solution: 81 digits row-major, 0 = empty.  *result: 1 solved, 0 no solution,
 * -1 invalid input (solver/include/sudoku.h:13-16); solution = grid unless solved. */
int sv_solve_sudoku(const uint8_t *grid /*host, 81*/, uint8_t *solution /*host, 81*/, int *result);

/* ---- K2: perspective warp + cell extraction (cv/grid.py, cv/extract.py) ------------------------- */

/* Host, fp64.  order_points + inset + cv2.getPerspectiveTransform to (0,0)..(S-1,S-1) + the inverse
 * warpPerspective takes, cv/grid.py:74-91,111-130.  corners: n*8 floats (x,y)*4 in any order;
 * minv: n*9 doubles (destination -> source). */
int sv_corners_to_minv(const float *corners /*host*/, int n, int out_size, float inset_ratio,
                       double *minv /*host*/);

/* The same for a batch in which single frames may be degenerate (order_points picks one point twice for a quad rotated
 * near 45 degrees, cv/grid.py:79-91: the 8x8 system is then singular): ok[f] = 1 and minv[f] filled, or ok[f] = 0 and
 * minv[f] = identity.  Never returns SV_ERR_DEGENERATE; the caller masks the frames with ok[f] = 0. */
int sv_corners_to_minv_batch(const float *corners /*host*/, int n, int out_size, float inset_ratio,
                             double *minv /*host*/, uint8_t *ok /*host, n*/);

/* cv2.warpPerspective(image, M, (S,S)), cv/grid.py:131: bilinear, 1/32-px coordinates, 15-bit
 * weights, constant-0 border.  channels 1 or 3.  One image. */
int sv_warp_perspective_u8(sv_ctx *ctx, const uint8_t *img /*dev*/, int H, int W, ptrdiff_t pitch,
                           int channels, const double *minv /*dev, 9*/, int out_size,
                           uint8_t *dst /*dev, S*S*channels*/, void *stream);

/* extract_cells(), cv/extract.py:13-56: 9x9 split, margin crop, BGR2GRAY, cv2.resize to
 * cell_size^2.  margin_h/margin_w are the caller's int(cell_h*margin_ratio), int(cell_w*...). */
int sv_extract_cells_u8(sv_ctx *ctx, const uint8_t *grid /*dev*/, int h, int w, ptrdiff_t pitch,
                        int channels, int cell_size, int margin_h, int margin_w,
                        uint8_t *cells /*dev, 81*cell_size^2*/, void *stream);

/* Fused warp_perspective(frame, corners) -> extract_cells(warped) with the reference defaults
 * (450, inset 0, 28, 0.1): only the 81 40x40 crops are ever warped.  n frames; minv n*9 doubles. */
int sv_warp_cells_u8(sv_ctx *ctx, const uint8_t *frames /*dev*/, int n, int H, int W, ptrdiff_t pitch,
                     ptrdiff_t frame_stride, const double *minv /*dev, n*9*/,
                     uint8_t *cells /*dev, n*81*784*/, void *stream);

/* ---- K3: DigitCNN forward (ml/model.py) --------------------------------------------------------- */

/* DigitCNN.forward, ml/model.py:34-42 (eval mode): x f32 [B,1,28,28] -> logits f32 [B,10].
 * digits (argmax, pipeline/run.py:142) and conf (softmax[argmax], :141-143) may be NULL.
 * Any finite f32 input is accepted: a batch outside the range the default kernels carry exactly is computed by the f32-MFMA kernels
 * (sv_ctx_set_cnn_kernels).  A cell holding NaN/Inf does not disturb the other cells of its batch; its own logits are unspecified (the
 * reference yields NaN there; ReLU and max-pool here are IEEE maxNum, which drops a NaN). */
int sv_cnn_forward_f32(sv_ctx *ctx, const float *x /*dev*/, long B, float *logits /*dev, B*10*/,
                       uint8_t *digits /*dev, B, or NULL*/, float *conf /*dev, B, or NULL*/,
                       void *stream);

/* The glue pipeline/run.py:122-136 puts between extract_cells and the model, fused into the CNN's input stage. */
typedef enum sv_glue {
    SV_GLUE_NORMALIZE = 0,  /* x = ((255 - cell)/255 - 0.5)/0.5                      (run.py:126-135 without preprocess_cell) */
    SV_GLUE_RUNPY = 1       /* preprocess_cell first: CLAHE(2.0,(4,4)) + adaptiveThreshold(GAUSSIAN_C, BINARY, 11, 2),
                               run.py:73-95, then the same invert + normalise -- exactly what run.py feeds the model */
} sv_glue;

/* cv2.resize(img, (dw, dh)), default INTER_LINEAR, on an 8-bit gray image (cv/extract.py:52 and :93). */
int sv_resize_linear_u8(sv_ctx *ctx, const uint8_t *src /*dev*/, int sh, int sw, ptrdiff_t pitch,
                        uint8_t *dst /*dev, dh*dw*/, int dh, int dw, void *stream);

/* is_cell_empty(), cv/extract.py:59-79, batched: per cell the Otsu threshold (cv2.threshold THRESH_OTSU) and
 * ratio = countNonZero(BINARY_INV image) / pixels; the caller compares ratio < threshold (default 0.02).
 * cells: B images of cell_px pixels each.  otsu may be NULL. */
int sv_cell_ink_ratio_u8(sv_ctx *ctx, const uint8_t *cells /*dev, B*cell_px*/, long B, int cell_px,
                         float *ratio /*dev, B*/, int *otsu /*dev, B, or NULL*/, void *stream);

/* preprocess_cell(), pipeline/run.py:73-95, on B 28x28 gray cells: CLAHE(2.0,(4,4)) then
 * adaptiveThreshold(GAUSSIAN_C, THRESH_BINARY, 11, 2).  out: u8 {0,255}, B*784. */
int sv_preprocess_cells_u8(sv_ctx *ctx, const uint8_t *cells /*dev, B*784*/, long B,
                           uint8_t *out /*dev, B*784*/, void *stream);

/* DigitCNN.forward on 8-bit cells with the glue fused in. */
int sv_cnn_forward_cells_u8(sv_ctx *ctx, const uint8_t *cells /*dev, B*784*/, long B, int glue,
                            float *logits /*dev*/, uint8_t *digits /*dev or NULL*/,
                            float *conf /*dev or NULL*/, void *stream);

/* F.softmax(output, dim=1) then probs.topk(top_k), pipeline/run_v2.py:165-178 (predict_cells_with_alternatives):
 * per cell the k most probable classes, most probable first (index[.,0] = the predicted digit, prob[.,0] = its
 * confidence, the rest = run_v2's `alternatives`).  Equal probabilities: lower class index first.  1 <= k <= 10. */
int sv_softmax_topk_f32(sv_ctx *ctx, const float *logits /*dev, B*10*/, long B, int k,
                        uint8_t *index /*dev, B*k*/, float *prob /*dev, B*k*/, void *stream);

/* ---- N4: JPEG front end -- what cv2.imread does before the path starts (pipeline/run.py:250, pipeline/run_v2.py:267,
 * tests/test_integration.py:126).  Baseline / extended-sequential Huffman JPEG, 8-bit, gray or YCbCr 4:4:4 / 4:2:2 / 4:2:0,
 * restart intervals, EXIF orientation applied as imread applies it.  The serial Huffman bit stream is decoded on the host
 * (threads: restart intervals of one image, or images of a batch); dequantisation, the inverse DCT (libjpeg's JDCT_ISLOW),
 * "fancy" chroma up-sampling, YCbCr -> BGR and the orientation run on the GPU, so the frame is born in HBM where K1 and K2
 * read it.  Progressive, arithmetic-coded, 12-bit, CMYK files: SV_ERR_UNSUPPORTED. */
typedef struct sv_jpeg_info {
    int width, height;              /* as stored */
    int out_width, out_height;      /* after the EXIF orientation: the shape imread returns */
    int components;                 /* 1 (gray) or 3 (YCbCr) */
    int h_samp, v_samp;             /* luma sampling factors: 1x1, 2x1 or 2x2 */
    int orientation;                /* EXIF tag 0x0112, 1..8 (1 when absent) */
    int restart_interval;           /* MCUs, 0 = none */
    long coef_count;                /* int16 values sv_jpeg_entropy_decode writes (64 per block) */
    long sparse_capacity;           /* values sv_jpeg_entropy_decode_sparse may need room for (bound from the file size) */
} sv_jpeg_info;

int sv_jpeg_parse(const uint8_t *data /*host*/, size_t size, sv_jpeg_info *info);

/* Huffman decoding -> coefficient blocks: per component, blocks row-major over the MCU-padded block grid, 64 values per
 * block in natural (row-major, de-zigzagged) order; quant: 3 x 64 quantiser steps, natural order, per component. */
int sv_jpeg_entropy_decode(const uint8_t *data /*host*/, size_t size, int16_t *coef /*host, coef_count*/,
                           uint16_t *quant /*host, 192*/, int threads);
/* The same in the compact form that crosses PCIe: per block (same block order) a 64-bit mask over ZIGZAG positions and the
 * index of the block's first value; `values` receives the non-zero coefficients, zigzag order, block after block.  A q90
 * 1080p frame is ~2 MB this way instead of 6.3 MB.  values_used: every index the masks/offsets refer to is below it (copy
 * that prefix).  SV_ERR_BUFFER when values_cap < info.sparse_capacity turns out too small. */
int sv_jpeg_entropy_decode_sparse(const uint8_t *data /*host*/, size_t size, uint64_t *masks /*host, coef_count/64*/,
                                  uint32_t *offsets /*host, coef_count/64*/, int16_t *values /*host, values_cap*/,
                                  long values_cap, long *values_used, uint16_t *quant /*host, 192*/, int threads);

/* A batch of files over `threads` host threads.  Dense output when coefs != NULL (masks..values_used ignored), sparse output
 * otherwise.  status[i] = per-image sv_status; returns the first failure. */
int sv_jpeg_entropy_decode_batch(const uint8_t *const *datas, const size_t *sizes, int n, int16_t *const *coefs /*host or NULL*/,
                                 uint64_t *const *masks, uint32_t *const *offsets, int16_t *const *values,
                                 const long *values_cap, long *values_used, uint16_t *quants /*host, n*192*/,
                                 int threads, int *status /*n*/);

/* Device half: coefficients -> BGR frame (out_height x out_width x 3, row pitch `pitch` bytes), asynchronous on `stream`. */
int sv_jpeg_reconstruct_bgr_u8(sv_ctx *ctx, const sv_jpeg_info *info, const int16_t *coef /*dev*/,
                               const uint16_t *quant /*dev, 192*/, uint8_t *bgr /*dev*/, ptrdiff_t pitch, void *stream);
int sv_jpeg_reconstruct_sparse_bgr_u8(sv_ctx *ctx, const sv_jpeg_info *info, const uint64_t *masks /*dev*/,
                                      const uint32_t *offsets /*dev*/, const int16_t *values /*dev*/,
                                      const uint16_t *quant /*dev, 192*/, uint8_t *bgr /*dev*/, ptrdiff_t pitch, void *stream);

/* ---- the whole device-resident path ----------------------------------------------------------- */

/* frames + homographies -> 81 digits per frame: K2 then K3 on `stream`, no host sync.
 * cells may be NULL (then context scratch is used). */
int sv_frames_to_digits(sv_ctx *ctx, const uint8_t *frames /*dev*/, int n, int H, int W,
                        ptrdiff_t pitch, ptrdiff_t frame_stride, const double *minv /*dev, n*9*/, int glue,
                        uint8_t *cells /*dev n*81*784 or NULL*/, float *logits /*dev, n*81*10*/,
                        uint8_t *digits /*dev, n*81*/, float *conf /*dev n*81 or NULL*/, void *stream);

#ifdef __cplusplus
}
#endif
#endif /* SUDOKU_VISION_HIP_H */
