"""GPU (-m gpu): the JPEG front end (scope row N4; cv2.imread, pipeline/run.py:250) through the C ABI -- host Huffman
decoding + HIP reconstruction -- bit-exact against the oracle AND against Pillow's libjpeg-turbo decode of the same bytes."""
import os

import numpy as np
import pytest
import torch
from PIL import Image

import cnn_oracle
import sv_oracle as o
from test_jpeg import CASES, encode, pil_bgr, synth_image

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("h,w,kw", CASES)
def test_imdecode_bit_exact(ctx, h, w, kw):
    data = encode(synth_image(h, w, h * 131 + w), **kw)
    want = pil_bgr(data)
    assert (want == o.imdecode(data)).all()
    for dense in (False, True):                                   # compact (mask + values) and plain int16 transport
        for threads in (1, 2):
            assert (ctx.imdecode(data, threads=threads, dense=dense).cpu().numpy() == want).all(), (dense, threads)


def test_imdecode_gray_and_orientations(ctx):
    data = encode(synth_image(40, 56, 5, gray=True), quality=88)
    assert (ctx.imdecode(data).cpu().numpy() == pil_bgr(data)).all()
    for orient in range(1, 9):
        for sub in (0, 1, 2):
            exif = Image.Exif()
            exif[0x0112] = orient
            data = encode(synth_image(41, 73, orient), quality=90, subsampling=sub, exif=exif)
            got = ctx.imdecode(data).cpu().numpy()
            assert got.shape == ((73, 41, 3) if orient >= 5 else (41, 73, 3))
            assert (got == pil_bgr(data)).all(), (orient, sub)
            assert (got == o.imdecode(data)).all(), (orient, sub)


def test_imread_photo_then_recognise(ctx, golden_dir):
    """The reference photo straight from its JPEG bytes: imread (this build) == Pillow == oracle, 2736x3648x3; then the run.py
    call order on the decoded frame without it ever leaving the GPU."""
    from sudoku_vision_amd import imgcodecs
    from sudoku_vision_amd.pipeline import recognize_image
    path = os.path.join(golden_dir, "sample_4.jpg")
    img = imgcodecs.imread(path)
    data = open(path, "rb").read()
    assert img.dtype == np.uint8 and img.shape == (3648, 2736, 3)
    assert (img == pil_bgr(data)).all()
    assert (img == o.imdecode(data)).all()
    g2 = np.load(os.path.join(golden_dir, "cnn_coreml_fp16.npz"))
    ctx.load_state_dict({k: torch.from_numpy(g2[k.replace(".", "_")].astype(np.float32)) for k in cnn_oracle.KEYS})
    a = recognize_image(img, ctx=ctx)
    b = recognize_image(imgcodecs.imread(path, device=True), ctx=ctx)
    assert a is not None and (a["digits"] == b["digits"]).all() and (a["corners"] == b["corners"]).all()


def test_imread_conventions(ctx, tmp_path):
    """cv2.imread returns None for a missing or undecodable file; unsupported JPEG flavours raise instead of guessing."""
    import sudoku_vision_amd as sva
    from sudoku_vision_amd import imgcodecs
    assert imgcodecs.imread(tmp_path / "nope.jpg") is None
    (tmp_path / "junk.jpg").write_bytes(b"not a jpeg at all")
    assert imgcodecs.imread(tmp_path / "junk.jpg") is None
    (tmp_path / "prog.jpg").write_bytes(encode(synth_image(32, 32, 1), quality=80, progressive=True))
    with pytest.raises(sva._native.NativeError, match="progressive"):
        imgcodecs.imread(tmp_path / "prog.jpg")


def test_imdecode_back_to_back(ctx):
    """The two pinned staging sets alternate: consecutive decodes of different sizes must not trample each other."""
    datas = [encode(synth_image(h, w, h + w), quality=85, subsampling=2) for h, w in ((200, 300), (64, 64), (333, 222), (90, 500), (512, 512))]
    outs = [ctx.imdecode(d) for d in datas]
    for d, t in zip(datas, outs):
        assert (t.cpu().numpy() == pil_bgr(d)).all()


def test_imdecode_batch(ctx):
    """Batch entry: same-shape files -> one [n,H,W,3] tensor, mixed shapes -> a list; both transports; a bad file fails loudly."""
    import sudoku_vision_amd as sva
    same = [encode(synth_image(144, 256, s), quality=60 + 10 * s, subsampling=2) for s in range(4)]
    for dense in (False, True):
        out = ctx.imdecode_batch(same, threads=3, dense=dense)
        assert tuple(out.shape) == (4, 144, 256, 3)
        for d, t in zip(same, out):
            assert (t.cpu().numpy() == pil_bgr(d)).all()
    mixed = [encode(synth_image(h, w, h), quality=80, subsampling=sub) for h, w, sub in ((50, 70, 0), (33, 97, 1), (128, 64, 2))]
    outs = ctx.imdecode_batch(mixed, threads=2)
    assert isinstance(outs, list)
    for d, t in zip(mixed, outs):
        assert (t.cpu().numpy() == pil_bgr(d)).all()
    with pytest.raises(sva._native.NativeError):
        ctx.imdecode_batch([same[0], b"\xff\xd8 this is not a jpeg"], threads=2)


def test_run_pipeline_counterpart(ctx, golden_dir, tmp_path):
    """pipeline/run.py:244-355's run_pipeline on the MI355X path: same result fields, same error strings."""
    import sudoku_vision_amd as sva
    from sudoku_vision_amd.pipeline import PipelineResult, recognize_image, run_pipeline, run_solver, check_constraints
    g2 = np.load(os.path.join(golden_dir, "cnn_coreml_fp16.npz"))
    sd = {k: torch.from_numpy(g2[k.replace(".", "_")].astype(np.float32)) for k in cnn_oracle.KEYS}
    path = os.path.join(golden_dir, "sample_4.jpg")
    res = run_pipeline(path, state_dict=sd, ctx=ctx)
    assert isinstance(res, PipelineResult)
    assert res.original_image.shape == (3648, 2736, 3) and res.warped_grid.shape == (450, 450, 3)
    assert len(res.cells) == 81 and all(c.shape == (28, 28) and c.dtype == np.uint8 for c in res.cells)
    assert len(res.predictions) == 81 and [(p.row, p.col) for p in res.predictions] == [(i // 9, i % 9) for i in range(81)]
    # the staged cells (warp, then extract) are the fused kernel's cells; the grid is recognize_image's
    ref = recognize_image(res.original_image, ctx=ctx)
    assert res.recognized_grid == ref["grid"]
    img = res.original_image
    corners = sva.host.find_grid_corners(ctx.preprocess(torch.from_numpy(img).cuda()[None])[0].cpu().numpy())
    assert (res.warped_grid == o.warp_perspective(img, corners.astype(np.float32))).all()
    assert (np.stack(res.cells) == o.warp_cells(img, corners.astype(np.float32))).all()
    assert res.constraint_violations == check_constraints(res.recognized_grid)
    ok, sol = run_solver(res.recognized_grid)
    assert res.success == ok and res.solution == sol
    if ok:
        assert res.error is None and all(p.digit == sol[p.row][p.col] for p in res.predictions)
        assert all(p.is_original == (res.recognized_grid[p.row][p.col] != 0) for p in res.predictions)
    else:
        assert res.error == "Solver failed: puzzle may be invalid or have recognition errors" and res.solution == res.recognized_grid
    assert all(0 <= r < 9 and 0 <= c < 9 and cf < 0.7 for r, c, cf in res.low_confidence_cells)
    assert res.time_total >= res.time_cv > 0 and res.time_ml > 0
    # error conventions
    miss = run_pipeline(tmp_path / "missing.jpg", ctx=ctx)
    assert not miss.success and miss.error == f"Failed to load image: {tmp_path / 'missing.jpg'}"
    blank = tmp_path / "blank.jpg"
    Image.fromarray(np.full((480, 640, 3), 200, np.uint8)).save(blank, "JPEG", quality=90)
    nog = run_pipeline(blank, ctx=ctx)
    assert not nog.success and nog.error == "Grid detection failed: no quadrilateral found" and nog.original_image.shape == (480, 640, 3)


def test_imdecode_noninterleaved_scans(ctx):
    """A baseline file with one scan per component (re-encoded from a Pillow file in tests/test_jpeg.py) through both transports."""
    from test_jpeg import _encode_noninterleaved
    for h, w, sub in ((61, 83, 2), (50, 37, 1), (40, 40, 0)):
        data = _encode_noninterleaved(encode(synth_image(h, w, 7 * h + w), quality=85, subsampling=sub))
        want = pil_bgr(data)
        for dense in (False, True):
            assert (ctx.imdecode(data, dense=dense).cpu().numpy() == want).all(), (h, w, sub, dense)
