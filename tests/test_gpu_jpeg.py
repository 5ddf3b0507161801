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
    got = ctx.imdecode(data, threads=2).cpu().numpy()
    assert (got == o.imdecode(data)).all()
    assert (got == pil_bgr(data)).all()


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
    """The pinned staging buffer is reused: consecutive decodes of different sizes must not trample each other."""
    datas = [encode(synth_image(h, w, h + w), quality=85, subsampling=2) for h, w in ((200, 300), (64, 64), (333, 222))]
    outs = [ctx.imdecode(d) for d in datas]
    for d, t in zip(datas, outs):
        assert (t.cpu().numpy() == pil_bgr(d)).all()
