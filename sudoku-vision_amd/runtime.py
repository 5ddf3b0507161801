"""Device context and the device-resident entry point of the hot path.

One `Context` per (process, GPU): it owns the packed CNN weights and the scratch the kernels use.
Tensors are PyTorch-ROCm tensors; the library sees raw pointers and the current HIP stream.
"""
import ctypes as C
import threading

import numpy as np
import torch

from . import _native

_KEYS = ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")
_SHAPES = ((32, 1, 3, 3), (32,), (64, 32, 3, 3), (64,), (128, 3136), (128,), (10, 128), (10,))


def _require_gpu():
    if not torch.cuda.is_available():
        raise _native.NativeError("no ROCm GPU visible: the sudoku-vision hot path runs on MI355X only (no CPU fallback)")


def _stream_ptr():
    return C.c_void_p(torch.cuda.current_stream().cuda_stream)


def _ptr(t):
    return C.c_void_p(t.data_ptr())


def _frame_layout(frames):
    """frames u8 [n,H,W,3]: (tensor, row pitch, frame stride) in bytes.  Row padding and gaps between frames are passed through to
    the library (camera buffers are rarely dense); anything else is made contiguous first."""
    if frames.dim() != 4 or frames.shape[3] != 3 or frames.dtype != torch.uint8:
        raise TypeError("expected a uint8 tensor of shape [n,H,W,3]")
    n, H, W, _ = frames.shape
    st = frames.stride()
    ok = st[3] == 1 and st[2] == 3 and st[1] >= 3 * W and (n == 1 or st[0] >= st[1] * (H - 1) + 3 * W)
    if not ok:
        frames = frames.contiguous()
        st = frames.stride()
    return frames, st[1], (st[0] if n > 1 else st[1] * H)


class Context:
    def __init__(self, device=None, library=None):
        """library: a handle from _native (default: the product library).  Tests pass _native.lib_xcheck() to run the cross-check kernels of
        the test-only build through the same methods."""
        _require_gpu()
        self.device = torch.device("cuda", torch.cuda.current_device() if device is None else torch.device(device).index or 0)
        self._lib = library or _native.lib()
        self._h = C.c_void_p()
        self._check(self._lib.sv_ctx_create(self.device.index, C.byref(self._h)), "sv_ctx_create")
        self._weights_key = None

    def _check(self, rc, what):
        _native.check(rc, what, self._lib)

    def _out(self, t, shape, dtype, name):
        """A caller-provided output tensor: the kernels write shape-many elements through its raw pointer, so it has to be exactly that."""
        if not isinstance(t, torch.Tensor) or tuple(t.shape) != tuple(shape) or t.dtype != dtype or not t.is_contiguous() or t.device != self.device:
            raise TypeError(f"{name} must be a contiguous {dtype} tensor of shape {list(shape)} on {self.device}")
        return t

    def close(self):
        if self._h:
            self._lib.sv_ctx_destroy(self._h)
            self._h = C.c_void_p()

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    # ---- weights ------------------------------------------------------------------------------
    def load_state_dict(self, sd, key=None):
        """sd: DigitCNN state_dict (ml/model.py) -- tensors or arrays, any device."""
        parts = []
        for k, shape in zip(_KEYS, _SHAPES):
            v = sd[k]
            v = v.detach().to("cpu", torch.float32).numpy() if isinstance(v, torch.Tensor) else np.asarray(v, np.float32)
            if tuple(v.shape) != shape:
                raise ValueError(f"{k}: shape {tuple(v.shape)} != {shape}")
            parts.append(np.ascontiguousarray(v).reshape(-1))
        blob = np.concatenate(parts)
        assert blob.size == 421642
        self._check(self._lib.sv_load_weights_f32(self._h, blob.ctypes.data_as(C.c_void_p)), "sv_load_weights_f32")
        self._weights_key = key

    PREC_F32, PREC_BF16 = 0, 1

    def set_precision(self, precision):
        """PREC_F32 (default, logits within 1e-4 of the reference model) or PREC_BF16 (bf16 MFMA, digit-index parity)."""
        self._check(self._lib.sv_ctx_set_precision(self._h, int(precision)), "sv_ctx_set_precision")

    CNN_AUTO, CNN_F16PAIR, CNN_F32MFMA = 0, 1, 2
    CNN_X_WINOGRAD, CNN_X_WSPLIT = 102, 103          # cross-check selections of the test-only library (include/sudoku_vision_xcheck.h)

    def set_cnn_kernels(self, which):
        """CNN_AUTO (default: the f16-pair kernels whenever the weights / inputs are inside their range, else the f32-MFMA kernels),
        CNN_F16PAIR or CNN_F32MFMA (sv_ctx_set_cnn_kernels)."""
        self._check(self._lib.sv_ctx_set_cnn_kernels(self._h, int(which)), "sv_ctx_set_cnn_kernels")

    def reserve(self, max_cells):
        self._check(self._lib.sv_ctx_reserve(self._h, int(max_cells)), "sv_ctx_reserve")

    # ---- per-kernel timing (hipEvents on the launch stream, inside the library) -------------------
    KERNELS = ("k_preprocess", "k_warp_cells", "k_conv_features", "k_fc_head", "k_preprocess_warp_fused")

    def timing_begin(self):
        self._check(self._lib.sv_timing_begin(self._h), "sv_timing_begin")

    def timing_end(self):
        """-> {kernel name: (total ms, launches)}; waits for the recorded events."""
        ms = (C.c_double * len(self.KERNELS))()
        cnt = (C.c_long * len(self.KERNELS))()
        self._check(self._lib.sv_timing_end(self._h, ms, cnt, len(self.KERNELS)), "sv_timing_end")
        return {k: (ms[i], cnt[i]) for i, k in enumerate(self.KERNELS)}

    CONV_ALGO_NAMES = {0: "k_conv_features_pc + k_fc_head (direct implicit GEMM, f32 MFMA)",
                       2: "k_conv_features_wstream (Winograd F(2x2,3x3), f32 MFMA; cross-check build)",
                       3: "k_conv_features_wsplit (Winograd, bf16 MFMA with 3-way operand split; cross-check build)",
                       4: "k_conv_features_h2 + k_fc_head_h2p (f16 hi/lo operand pairs, f16 MFMA, f32 accumulation)"}

    def conv_kernel_info(self):
        """Which conv/fc kernels this process launches and the matrix instructions they issue per cell (sv_conv_kernel_info)."""
        v = [C.c_int() for _ in range(5)]
        self._check(self._lib.sv_conv_kernel_info(self._h, *[C.byref(x) for x in v]), "sv_conv_kernel_info")
        a = v[0].value
        return {"algo": a, "name": self.CONV_ALGO_NAMES.get(a, str(a)), "mfma_conv2": v[1].value, "mfma_conv1": v[2].value,
                "mfma_f16_conv": v[3].value, "mfma_f16_fc": v[4].value}

    # ---- K1 -----------------------------------------------------------------------------------
    def gray(self, bgr):
        bgr, pitch, fstride = _frame_layout(bgr)
        n, H, W = bgr.shape[0], bgr.shape[1], bgr.shape[2]
        out = torch.empty((n, H, W), dtype=torch.uint8, device=self.device)
        self._check(self._lib.sv_gray_u8(self._h, _ptr(bgr), n, H, W, pitch, fstride, _ptr(out), _stream_ptr()), "sv_gray_u8")
        return out

    def blur(self, gray, ksize):
        n, H, W = gray.shape
        out = torch.empty_like(gray)
        self._check(self._lib.sv_blur_u8(self._h, _ptr(gray), n, H, W, int(ksize), _ptr(out), _stream_ptr()), "sv_blur_u8")
        return out

    def adaptive_threshold(self, gray, block_size, c, inv=True):
        n, H, W = gray.shape
        out = torch.empty_like(gray)
        self._check(self._lib.sv_adaptive_threshold_u8(self._h, _ptr(gray), n, H, W, int(block_size), float(c), int(bool(inv)),
                                                             _ptr(out), _stream_ptr()), "sv_adaptive_threshold_u8")
        return out

    def preprocess(self, frames, out=None):
        """frames u8 [n,H,W,3] on device (rows may be padded, frames may have gaps) -> binary u8 [n,H,W]
        (preprocess_for_grid_detection).  out: optional contiguous u8 [n,H,W] tensor to write into."""
        frames, pitch, fstride = _frame_layout(frames)
        n, H, W = frames.shape[0], frames.shape[1], frames.shape[2]
        if out is None:
            out = torch.empty((n, H, W), dtype=torch.uint8, device=self.device)
        else:
            self._out(out, (n, H, W), torch.uint8, "out")
        self._check(self._lib.sv_preprocess_u8(self._h, _ptr(frames), n, H, W, pitch, fstride, _ptr(out), _stream_ptr()), "sv_preprocess_u8")
        return out

    def preprocess_and_warp_cells(self, frames, minv_dev, binary=None, cells=None):
        """K1 and K2 of the same frames in one launch (sv_preprocess_warp_cells_u8; BASELINE configs[4]'s fused threshold/warp for callers
        that know the corners beforehand) -> (binary u8 [n,H,W], cells u8 [n,81,28,28])."""
        frames, pitch, fstride = _frame_layout(frames)
        n, H, W = frames.shape[0], frames.shape[1], frames.shape[2]
        binary = torch.empty((n, H, W), dtype=torch.uint8, device=self.device) if binary is None else self._out(binary, (n, H, W), torch.uint8, "binary")
        cells = torch.empty((n, 81, 28, 28), dtype=torch.uint8, device=self.device) if cells is None else self._out(cells, (n, 81, 28, 28), torch.uint8, "cells")
        self._check(self._lib.sv_preprocess_warp_cells_u8(self._h, _ptr(frames), n, H, W, pitch, fstride, _ptr(binary), _ptr(minv_dev), _ptr(cells),
                                                                _stream_ptr()), "sv_preprocess_warp_cells_u8")
        return binary, cells

    def preprocess_mm(self, frames, want_mean=False):
        """preprocess() through the matrix-pipe formulation of K1 (sv_preprocess_mm_u8): the same binary, an independent implementation.
        Only on a Context of the test-only library (Context(library=_native.lib_xcheck())).
        want_mean: also return the kernel's approximate local mean per pixel (f32 [n,H,W])."""
        frames, pitch, fstride = _frame_layout(frames)
        n, H, W = frames.shape[0], frames.shape[1], frames.shape[2]
        out = torch.empty((n, H, W), dtype=torch.uint8, device=self.device)
        mean = torch.zeros((n, H, W), dtype=torch.float32, device=self.device) if want_mean else None
        self._check(self._lib.sv_preprocess_mm_u8(self._h, _ptr(frames), n, H, W, pitch, fstride, _ptr(out), _ptr(mean) if want_mean else None,
                                                        _stream_ptr()), "sv_preprocess_mm_u8")
        return (out, mean) if want_mean else out

    def preprocess_stats(self):
        """(pixels decided by the exact evaluation in preprocess_mm launches since the last call, 0); the first call switches the counter on.
        Synchronises."""
        a, c = C.c_uint(), C.c_ulong()
        self._check(self._lib.sv_preprocess_stats(self._h, C.byref(a), C.byref(c)), "sv_preprocess_stats")
        return a.value, c.value

    def preprocess_bits(self, frames, out=None):
        """K1 with the binary as 1 bit per pixel: frames u8 [n,H,W,3] -> int32 [n,H,W//32] (LSB = leftmost pixel).  Needs W % 32 == 0 and
        4-byte aligned rows (NativeError SV_ERR_UNSUPPORTED otherwise: use preprocess + despeckle(packed=...))."""
        frames, pitch, fstride = _frame_layout(frames)
        n, H, W = frames.shape[0], frames.shape[1], frames.shape[2]
        if W % 32:
            raise ValueError("preprocess_bits needs W % 32 == 0")
        out = torch.empty((n, H, W // 32), dtype=torch.int32, device=self.device) if out is None else self._out(out, (n, H, W // 32), torch.int32, "out")
        self._check(self._lib.sv_preprocess_bits_u8(self._h, _ptr(frames), n, H, W, pitch, fstride, _ptr(out), _stream_ptr()), "sv_preprocess_bits_u8")
        return out

    def despeckle_bits(self, bits):
        """despeckle on a bit image int32 [n,H,W//32], in place."""
        n, H, wpr = bits.shape
        self._check(self._lib.sv_despeckle_bits(self._h, _ptr(bits), n, H, wpr * 32, _stream_ptr()), "sv_despeckle_bits")
        return bits

    def despeckle(self, binary, out=None, packed=None):
        """binary u8 [n,H,W] in {0,255} -> the same with every component that fits strictly inside a 64x64 tile erased
        (find_grid_contour-equivalent; used only to make the host corner search cheaper).  packed: optional int32 [n,H,W//32]
        tensor receiving the result as 1 bit per pixel (then `out` is scratch)."""
        n, H, W = binary.shape
        out = torch.empty_like(binary) if out is None else out
        self._check(self._lib.sv_despeckle_u8(self._h, _ptr(binary), n, H, W, _ptr(out), _ptr(packed) if packed is not None else None,
                                                    _stream_ptr()), "sv_despeckle_u8")
        return out if packed is None else packed

    def pack_sparse_bits(self, bits, records):
        """bits int32 [n,H,W//32] (despeckle's packed output) -> records uint8 [n,stride] (device): per frame the row masks of its
        non-zero words and those words (sv_pack_sparse_bits; layout in include/sudoku_vision_hip.h).  A third to a quarter of the
        dense image for the D2H copy; host.find_grid_corners_sparse_batch reads it."""
        n, H, wpr = bits.shape
        if records.dtype != torch.uint8 or records.dim() != 2 or records.shape[0] < n or not records.is_contiguous() or not bits.is_contiguous():
            raise TypeError("records must be a contiguous uint8 [>=n, stride] device tensor")
        self._check(self._lib.sv_pack_sparse_bits(self._h, _ptr(bits), n, H, wpr * 32, _ptr(records), records.shape[1], _stream_ptr()),
                      "sv_pack_sparse_bits")
        return records[:n]

    def copy_to_pinned(self, src, dst):
        """src (device tensor) -> dst (pinned host tensor of the same byte size) by a copy kernel on the current stream: about twice
        the PCIe rate of the DMA engine tensor.copy_(non_blocking=True) uses (sv_copy_to_pinned_host).  Stream-ordered; synchronise
        (an event, the stream) before reading dst."""
        if not dst.is_pinned() or not dst.is_contiguous() or not src.is_contiguous():
            raise TypeError("copy_to_pinned needs a contiguous device tensor and a contiguous pinned host tensor")
        nbytes = src.numel() * src.element_size()
        if nbytes != dst.numel() * dst.element_size():
            raise ValueError("size mismatch")
        self._check(self._lib.sv_copy_to_pinned_host(self._h, _ptr(src), _ptr(dst), nbytes, _stream_ptr()), "sv_copy_to_pinned_host")
        return dst

    # ---- K2 -----------------------------------------------------------------------------------
    @staticmethod
    def corners_to_minv(corners, output_size=450, inset_ratio=0.0):
        """corners [n,4,2] (host, any order) -> float64 [n,3,3] destination->source homographies (host)."""
        c = np.ascontiguousarray(np.asarray(corners, dtype=np.float32).reshape(-1, 8))
        out = np.empty((c.shape[0], 3, 3), np.float64)
        _native.check(_native.lib().sv_corners_to_minv(c.ctypes.data_as(C.c_void_p), c.shape[0], int(output_size), float(inset_ratio),
                                                       out.ctypes.data_as(C.c_void_p)), "sv_corners_to_minv")
        return out

    @staticmethod
    def corners_to_minv_batch(corners, output_size=450, inset_ratio=0.0):
        """As corners_to_minv, but a degenerate quad does not raise: -> (minv float64 [n,3,3], ok bool [n]); minv[f] is the
        identity where ok[f] is False."""
        c = np.ascontiguousarray(np.asarray(corners, dtype=np.float32).reshape(-1, 8))
        out = np.empty((c.shape[0], 3, 3), np.float64)
        ok = np.empty(c.shape[0], np.uint8)
        _native.check(_native.lib().sv_corners_to_minv_batch(c.ctypes.data_as(C.c_void_p), c.shape[0], int(output_size), float(inset_ratio),
                                                             out.ctypes.data_as(C.c_void_p), ok.ctypes.data_as(C.c_void_p)), "sv_corners_to_minv_batch")
        return out, ok.astype(bool)

    def minv_to_device(self, minv):
        return torch.from_numpy(np.ascontiguousarray(minv, np.float64)).to(self.device)

    def warp_perspective(self, img, minv_dev, output_size):
        H, W = img.shape[0], img.shape[1]
        ch = 1 if img.dim() == 2 else img.shape[2]
        shape = (output_size, output_size) if img.dim() == 2 else (output_size, output_size, ch)
        out = torch.empty(shape, dtype=torch.uint8, device=self.device)
        self._check(self._lib.sv_warp_perspective_u8(self._h, _ptr(img), H, W, W * ch, ch, _ptr(minv_dev), int(output_size), _ptr(out),
                                                           _stream_ptr()), "sv_warp_perspective_u8")
        return out

    def extract_cells(self, grid, cell_size, margin_h, margin_w):
        h, w = grid.shape[0], grid.shape[1]
        ch = 1 if grid.dim() == 2 else grid.shape[2]
        out = torch.empty((81, cell_size, cell_size), dtype=torch.uint8, device=self.device)
        self._check(self._lib.sv_extract_cells_u8(self._h, _ptr(grid), h, w, w * ch, ch, int(cell_size), int(margin_h), int(margin_w),
                                                        _ptr(out), _stream_ptr()), "sv_extract_cells_u8")
        return out

    def warp_cells(self, frames, minv_dev):
        frames, pitch, fstride = _frame_layout(frames)
        n, H, W = frames.shape[0], frames.shape[1], frames.shape[2]
        out = torch.empty((n, 81, 28, 28), dtype=torch.uint8, device=self.device)
        self._check(self._lib.sv_warp_cells_u8(self._h, _ptr(frames), n, H, W, pitch, fstride, _ptr(minv_dev), _ptr(out), _stream_ptr()),
                      "sv_warp_cells_u8")
        return out

    # ---- K3 -----------------------------------------------------------------------------------
    GLUE_NORMALIZE, GLUE_RUNPY = 0, 1

    def resize_linear(self, img, dsize):
        """cv2.resize(img, dsize=(w, h)) INTER_LINEAR on a gray u8 image."""
        dw, dh = dsize
        out = torch.empty((dh, dw), dtype=torch.uint8, device=self.device)
        self._check(self._lib.sv_resize_linear_u8(self._h, _ptr(img), img.shape[0], img.shape[1], img.shape[1], _ptr(out), dh, dw, _stream_ptr()),
                      "sv_resize_linear_u8")
        return out

    def cell_ink_ratio(self, cells):
        """cells u8 [B,h,w] -> (ratio f32 [B], otsu i32 [B]): is_cell_empty's Otsu ink share, batched."""
        B = cells.shape[0]
        ratio = torch.empty((B,), dtype=torch.float32, device=self.device)
        otsu = torch.empty((B,), dtype=torch.int32, device=self.device)
        self._check(self._lib.sv_cell_ink_ratio_u8(self._h, _ptr(cells), B, int(cells[0].numel()), _ptr(ratio), _ptr(otsu), _stream_ptr()),
                      "sv_cell_ink_ratio_u8")
        return ratio, otsu

    # ---- JPEG front end (scope row N4) -----------------------------------------------------------------
    def _jpeg_staging(self, nbytes):
        """Two pinned + device staging sets used alternately, so the H2D copy and reconstruction of one call overlap the
        host Huffman decoding of the next.  Returns (pinned u8 tensor, device u8 tensor) of at least nbytes."""
        if not hasattr(self, "_jpeg_sets"):
            self._jpeg_sets, self._jpeg_turn = [None, None], 0
        self._jpeg_turn ^= 1
        cur = self._jpeg_sets[self._jpeg_turn]
        if cur is not None:
            cur[2].synchronize()                                       # the copy that last read this set has finished
        if cur is None or cur[0].numel() < nbytes:
            cap = int(nbytes * 1.25) + 4096
            cur = [torch.empty(cap, dtype=torch.uint8).pin_memory(), torch.empty(cap, dtype=torch.uint8, device=self.device), torch.cuda.Event()]
            self._jpeg_sets[self._jpeg_turn] = cur
        return cur

    def imdecode_batch(self, datas, threads=16, dense=False):
        """A batch of JPEG files -> uint8 CUDA tensor [n,H,W,3] when all share a shape, else a list of [H,W,3] tensors.
        Images are Huffman-decoded on `threads` host threads (sv_jpeg_entropy_decode_batch) into pinned memory in the compact
        mask + values form (dense=True: plain int16 blocks), cross PCIe one image per copy, and are reconstructed on the GPU
        back to back.  Returns after the last launch; the next call's host decoding overlaps this call's copies and kernels."""
        from . import host
        n = len(datas)
        datas = [bytes(d) for d in datas]
        infos = [host.jpeg_parse(d) for d in datas]
        lay, total = [], 0                                             # per image: (base, masks, offsets, values, capacity) byte offsets
        for info in infos:
            nb = int(info.coef_count) // 64
            base = total
            if dense:
                lay.append((base, 0, 0, base, int(info.coef_count)))
                total += (2 * int(info.coef_count) + 63) // 64 * 64
            else:
                lay.append((base, base, base + 8 * nb, base + 12 * nb, int(info.sparse_capacity)))
                total += (12 * nb + 2 * int(info.sparse_capacity) + 63) // 64 * 64
        qoff = total
        total += 384 * n
        pin, dev, ev = self._jpeg_staging(total)
        pb, db = pin.data_ptr(), dev.data_ptr()
        VP = C.c_void_p * n
        bufs = (C.c_char_p * n)(*datas)
        sizes = (C.c_size_t * n)(*[len(d) for d in datas])
        status = (C.c_int * n)()
        used = (C.c_long * n)()
        if dense:
            args = (VP(*[pb + l[3] for l in lay]), None, None, None, None, None)
        else:
            args = (None, VP(*[pb + l[1] for l in lay]), VP(*[pb + l[2] for l in lay]), VP(*[pb + l[3] for l in lay]),
                    (C.c_long * n)(*[l[4] for l in lay]), used)
        self._check(self._lib.sv_jpeg_entropy_decode_batch(bufs, sizes, n, *args, C.c_void_p(pb + qoff), int(threads), status),
                      "sv_jpeg_entropy_decode_batch")
        same = all((i.out_height, i.out_width) == (infos[0].out_height, infos[0].out_width) for i in infos)
        if same:
            out = torch.empty((n, infos[0].out_height, infos[0].out_width, 3), dtype=torch.uint8, device=self.device)
            outs = [out[i] for i in range(n)]
        else:
            outs = [torch.empty((i.out_height, i.out_width, 3), dtype=torch.uint8, device=self.device) for i in infos]
        dev[qoff:qoff + 384 * n].copy_(pin[qoff:qoff + 384 * n], non_blocking=True)
        lib, stream = self._lib, _stream_ptr()
        for i, (info, l) in enumerate(zip(infos, lay)):
            end = l[3] + 2 * (int(info.coef_count) if dense else used[i])
            dev[l[0]:end].copy_(pin[l[0]:end], non_blocking=True)
            q = C.c_void_p(db + qoff + 384 * i)
            if dense:
                rc = lib.sv_jpeg_reconstruct_bgr_u8(self._h, C.byref(info), C.c_void_p(db + l[3]), q, _ptr(outs[i]), outs[i].stride(0), stream)
            else:
                rc = lib.sv_jpeg_reconstruct_sparse_bgr_u8(self._h, C.byref(info), C.c_void_p(db + l[1]), C.c_void_p(db + l[2]), C.c_void_p(db + l[3]), q,
                                                           _ptr(outs[i]), outs[i].stride(0), stream)
            self._check(rc, "sv_jpeg_reconstruct")
        ev.record(torch.cuda.current_stream(self.device))
        self._jpeg_last_bytes = sum((l[3] - l[0]) + 2 * (int(i.coef_count) if dense else used[k]) for k, (i, l) in enumerate(zip(infos, lay))) / max(n, 1)
        return out if same else outs

    def imdecode(self, data: bytes, threads=1, out=None, dense=False):
        """cv2.imdecode / cv2.imread of a baseline JPEG -> BGR uint8 CUDA tensor [H,W,3] (EXIF orientation applied).
        Huffman decoding on the host (csrc/host_jpeg.cpp; `threads` work on restart intervals when the file has them) into
        pinned staging memory, everything after it on the GPU."""
        if out is None and threads <= 1:
            return self.imdecode_batch([data], 1, dense=dense)[0]
        from . import host
        data = bytes(data)
        info = host.jpeg_parse(data)
        nb, ncoef, cap = int(info.coef_count) // 64, int(info.coef_count), int(info.sparse_capacity)
        voff = 0 if dense else 12 * nb
        qoff = (voff + 2 * (ncoef if dense else cap) + 63) // 64 * 64
        pin, dev, ev = self._jpeg_staging(qoff + 384)
        pb, db = pin.data_ptr(), dev.data_ptr()
        lib = self._lib
        if dense:
            self._check(lib.sv_jpeg_entropy_decode(data, len(data), C.c_void_p(pb), C.c_void_p(pb + qoff), int(threads)), "sv_jpeg_entropy_decode")
            end = 2 * ncoef
        else:
            used = C.c_long()
            self._check(lib.sv_jpeg_entropy_decode_sparse(data, len(data), C.c_void_p(pb), C.c_void_p(pb + 8 * nb), C.c_void_p(pb + voff), cap, C.byref(used),
                                                            C.c_void_p(pb + qoff), int(threads)), "sv_jpeg_entropy_decode_sparse")
            end = voff + 2 * used.value
        dev[:end].copy_(pin[:end], non_blocking=True)
        dev[qoff:qoff + 384].copy_(pin[qoff:qoff + 384], non_blocking=True)
        if out is None:
            out = torch.empty((info.out_height, info.out_width, 3), dtype=torch.uint8, device=self.device)
        if dense:
            rc = lib.sv_jpeg_reconstruct_bgr_u8(self._h, C.byref(info), C.c_void_p(db), C.c_void_p(db + qoff), _ptr(out), out.stride(0), _stream_ptr())
        else:
            rc = lib.sv_jpeg_reconstruct_sparse_bgr_u8(self._h, C.byref(info), C.c_void_p(db), C.c_void_p(db + 8 * nb), C.c_void_p(db + voff), C.c_void_p(db + qoff),
                                                       _ptr(out), out.stride(0), _stream_ptr())
        self._check(rc, "sv_jpeg_reconstruct")
        ev.record(torch.cuda.current_stream(self.device))
        return out

    def softmax_topk(self, logits, k=3):
        """F.softmax(logits, 1).topk(k) (pipeline/run_v2.py:165-178): (index u8 [B,k], prob f32 [B,k]), best first."""
        logits = logits.reshape(-1, 10).contiguous()
        B = logits.shape[0]
        index = torch.empty((B, k), dtype=torch.uint8, device=self.device)
        prob = torch.empty((B, k), dtype=torch.float32, device=self.device)
        self._check(self._lib.sv_softmax_topk_f32(self._h, _ptr(logits), B, int(k), _ptr(index), _ptr(prob), _stream_ptr()), "sv_softmax_topk_f32")
        return index, prob

    def preprocess_cells(self, cells):
        """run.py's preprocess_cell on u8 cells [B,28,28] -> u8 {0,255} [B,28,28]."""
        out = torch.empty_like(cells)
        self._check(self._lib.sv_preprocess_cells_u8(self._h, _ptr(cells), cells.shape[0], _ptr(out), _stream_ptr()), "sv_preprocess_cells_u8")
        return out

    def cnn_forward(self, x, want_digits=False, glue=0):
        """x f32 [B,1,28,28], or u8 [B,28,28] cells with the run.py glue fused in (glue=GLUE_NORMALIZE: invert+normalise;
        GLUE_RUNPY: preprocess_cell (CLAHE + adaptive threshold) first) -> logits [B,10] (, digits, conf)."""
        B = x.shape[0]
        logits = torch.empty((B, 10), dtype=torch.float32, device=self.device)
        digits = torch.empty((B,), dtype=torch.uint8, device=self.device) if want_digits else None
        conf = torch.empty((B,), dtype=torch.float32, device=self.device) if want_digits else None
        dg, cf = (_ptr(digits) if want_digits else None), (_ptr(conf) if want_digits else None)
        if x.dtype == torch.uint8:
            rc = self._lib.sv_cnn_forward_cells_u8(self._h, _ptr(x), B, int(glue), _ptr(logits), dg, cf, _stream_ptr())
        else:
            rc = self._lib.sv_cnn_forward_f32(self._h, _ptr(x), B, _ptr(logits), dg, cf, _stream_ptr())
        self._check(rc, "sv_cnn_forward")
        return (logits, digits, conf) if want_digits else logits

    # ---- whole path ---------------------------------------------------------------------------
    def frames_to_digits(self, frames, minv_dev, out=None, keep_cells=False, glue=0):
        """frames u8 [n,H,W,3], minv_dev f64 [n,3,3] on device -> dict(logits [n,81,10], digits [n,81], conf [n,81])."""
        frames, pitch, fstride = _frame_layout(frames)
        n, H, W = frames.shape[0], frames.shape[1], frames.shape[2]
        if out is None:
            out = {"logits": torch.empty((n, 81, 10), dtype=torch.float32, device=self.device),
                   "digits": torch.empty((n, 81), dtype=torch.uint8, device=self.device),
                   "conf": torch.empty((n, 81), dtype=torch.float32, device=self.device)}
            if keep_cells:
                out["cells"] = torch.empty((n, 81, 28, 28), dtype=torch.uint8, device=self.device)
        cells = out.get("cells")
        self._check(self._lib.sv_frames_to_digits(self._h, _ptr(frames), n, H, W, pitch, fstride, _ptr(minv_dev), int(glue),
                                                        _ptr(cells) if cells is not None else None, _ptr(out["logits"]), _ptr(out["digits"]),
                                                        _ptr(out["conf"]), _stream_ptr()), "sv_frames_to_digits")
        return out


_default = {}
_lock = threading.Lock()


def default_context(device=None) -> Context:
    _require_gpu()
    idx = torch.cuda.current_device() if device is None else (torch.device(device).index or 0)
    with _lock:
        if idx not in _default:
            _default[idx] = Context(torch.device("cuda", idx))
        return _default[idx]


def frames_to_digits(frames, corners, state_dict=None, ctx=None):
    """Device-resident hot path: frames u8 [n,H,W,3] (cuda tensor), corners [n,4,2] (host) -> dict of tensors."""
    ctx = ctx or default_context(frames.device)
    if state_dict is not None:
        ctx.load_state_dict(state_dict)
    minv = ctx.minv_to_device(Context.corners_to_minv(corners))
    return ctx.frames_to_digits(frames, minv)
