"""Scope row N4 (JPEG decode for real-image feeds; replaces cv2.imread, pipeline/run.py:250).

CPU part: the oracle (oracle/sv_jpeg_oracle.c) is pinned bit-for-bit against Pillow's libjpeg-turbo decoder (the same
library family and defaults cv2.imread decodes with; cv2 itself is absent from this image), and the product's host-side
entropy decoder (csrc/host_jpeg.cpp, runs without a GPU) is checked against the oracle's coefficients.
"""
import glob
import io
import os
import sys

import numpy as np
import pytest
from PIL import Image, ImageOps

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import sv_oracle as o  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
REF_PHOTOS = sorted(glob.glob("/root/reference/data/test_images/sample_*.jpg"))


def synth_image(h, w, seed, gray=False):
    """Smooth structure + edges + noise, so that every AC band and the chroma planes carry signal."""
    rs = np.random.RandomState(seed)
    yy, xx = np.mgrid[0:h, 0:w]
    chans = []
    for c in range(1 if gray else 3):
        base = 128 + 90 * np.sin(xx / (7.0 + 3 * c) + seed) * np.cos(yy / (5.0 + 2 * c))
        base += 60 * (((xx // 9) + (yy // 7) + c) % 2) + rs.randint(-20, 21, (h, w))
        chans.append(np.clip(base, 0, 255).astype(np.uint8))
    return chans[0] if gray else np.stack(chans, -1)


def encode(img, **kw):
    b = io.BytesIO()
    Image.fromarray(img).save(b, "JPEG", **kw)
    return b.getvalue()


def pil_bgr(data):
    """What the checker of record says: Pillow's decode, EXIF orientation applied as cv2.imread does, RGB -> BGR."""
    im = ImageOps.exif_transpose(Image.open(io.BytesIO(data)))
    return np.asarray(im.convert("RGB"))[..., ::-1]


CASES = [  # (h, w, kwargs)
    (64, 80, dict(quality=90, subsampling=0)),
    (64, 80, dict(quality=90, subsampling=1)),
    (64, 80, dict(quality=90, subsampling=2)),
    (61, 83, dict(quality=75, subsampling=2)),            # odd sizes: partial MCUs, odd chroma edge
    (50, 37, dict(quality=30, subsampling=1)),
    (17, 16, dict(quality=95, subsampling=2)),
    (8, 8, dict(quality=85, subsampling=2)),
    (3, 4, dict(quality=85, subsampling=2)),              # chroma width 2: libjpeg falls back to replication
    (5, 3, dict(quality=85, subsampling=1)),
    (1, 1, dict(quality=85, subsampling=2)),
    (9, 5, dict(quality=85, subsampling=2)),              # chroma width 3: smallest fancy case
    (120, 200, dict(quality=100, subsampling=0)),         # quality 100: large coefficients, clamping
    (120, 200, dict(quality=5, subsampling=2)),           # quality 5: 16-bit-ish quantisers, heavy ringing
    (96, 144, dict(quality=80, subsampling=2, restart_marker_blocks=3)),
    (96, 144, dict(quality=80, subsampling=0, restart_marker_rows=1)),
    (70, 70, dict(quality=80, subsampling=2, optimize=True)),     # optimised (non-default) Huffman tables
]


@pytest.mark.parametrize("h,w,kw", CASES)
def test_oracle_matches_pillow_synthetic(h, w, kw):
    data = encode(synth_image(h, w, h * 131 + w), **kw)
    assert (o.imdecode(data) == pil_bgr(data)).all()


def test_oracle_matches_pillow_gray():
    for h, w in ((40, 56), (33, 21)):
        data = encode(synth_image(h, w, 5, gray=True), quality=88)
        assert o.jpeg_info(data).components == 1
        assert (o.imdecode(data) == pil_bgr(data)).all()


@pytest.mark.parametrize("orient", range(1, 9))
def test_oracle_exif_orientation(orient):
    """The 8 EXIF orientations against PIL.ImageOps.exif_transpose (the standard mapping cv2.imread applies as well)."""
    img = synth_image(40, 72, 11)
    exif = Image.Exif()
    exif[0x0112] = orient
    data = encode(img, quality=90, subsampling=2, exif=exif)
    info = o.jpeg_info(data)
    assert info.orientation == orient and (info.out_height, info.out_width) == ((72, 40) if orient >= 5 else (40, 72))
    assert (o.imdecode(data) == pil_bgr(data)).all()


def test_oracle_matches_pillow_fixture_photo():
    """tests/golden/sample_4.jpg = the reference's data/test_images/sample_4.jpg (data fixture), 2736x3648 4:2:0."""
    data = open(os.path.join(GOLDEN, "sample_4.jpg"), "rb").read()
    info = o.jpeg_info(data)
    assert (info.width, info.height, info.components, info.h_samp, info.v_samp) == (2736, 3648, 3, 2, 2)
    assert (o.imdecode(data) == pil_bgr(data)).all()


@pytest.mark.skipif(not REF_PHOTOS, reason="reference tree not present (GPU box)")
def test_oracle_matches_pillow_all_reference_photos():
    for f in REF_PHOTOS:
        data = open(f, "rb").read()
        assert (o.imdecode(data) == pil_bgr(data)).all(), f


def test_oracle_rejects_what_it_does_not_restate():
    prog = encode(synth_image(32, 32, 1), quality=80, progressive=True)
    with pytest.raises(ValueError):
        o.imdecode(prog)
    with pytest.raises(ValueError):
        o.imdecode(b"\x89PNG\r\n\x1a\n" + b"\0" * 64)
    cmyk = io.BytesIO()
    Image.fromarray(synth_image(16, 16, 2)).convert("CMYK").save(cmyk, "JPEG")
    with pytest.raises(ValueError):
        o.imdecode(cmyk.getvalue())


# ---- product, host half (no GPU needed): csrc/host_jpeg.cpp vs the oracle ------------------------------------------
@pytest.fixture(scope="module")
def host():
    import __graft_entry__ as g
    import sudoku_vision_amd as sva
    if not os.path.exists(sva._native.LIB_PATH):
        g.build()
    return sva.host


def _same_info(a, b):
    """product sv_jpeg_info vs the oracle's (which has no sparse_capacity)"""
    return all(getattr(a, f) == getattr(b, f) for f, _ in type(b)._fields_)


_ZZ = [0, 1, 8, 16, 9, 2, 3, 10, 17, 24, 32, 25, 18, 11, 4, 5, 12, 19, 26, 33, 40, 48, 41, 34, 27, 20, 13, 6, 7, 14, 21, 28,
       35, 42, 49, 56, 57, 50, 43, 36, 29, 22, 15, 23, 30, 37, 44, 51, 58, 59, 52, 45, 38, 31, 39, 46, 53, 60, 61, 54, 47, 55, 62, 63]


def _densify(info, masks, offs, vals):
    """The compact transport form back to dense natural-order blocks (what the GPU kernel does by popcount rank)."""
    out = np.zeros(info.coef_count, np.int16)
    for b in range(len(masks)):
        m, k = int(masks[b]), int(offs[b])
        for z in range(64):
            if (m >> z) & 1:
                out[b * 64 + _ZZ[z]] = vals[k]
                k += 1
    return out


@pytest.mark.parametrize("h,w,kw", CASES)
def test_host_entropy_decode_matches_oracle(host, h, w, kw):
    data = encode(synth_image(h, w, h * 131 + w), **kw)
    info, coef, quant = host.jpeg_entropy_decode(data, threads=3)          # threads only matter with restart intervals
    assert _same_info(info, o.jpeg_info(data))
    oc, oq = o.jpeg_coefficients(data)
    assert (coef == oc).all() and (quant == oq).all()
    # the compact (mask + values) form carries the same coefficients, within the capacity the header promised
    info2, masks, offs, vals, quant2 = host.jpeg_entropy_decode_sparse(data, threads=3)
    assert len(vals) <= info2.sparse_capacity <= info2.coef_count and (quant2 == oq).all()
    assert (_densify(info2, masks, offs, vals) == oc).all()
    assert int(sum(bin(int(m)).count("1") for m in masks)) == int(np.count_nonzero(oc))


def test_host_entropy_decode_gray_orientation_photo(host):
    exif = Image.Exif()
    exif[0x0112] = 6
    for data in (encode(synth_image(40, 56, 5, gray=True), quality=88),
                 encode(synth_image(40, 72, 11), quality=90, exif=exif),
                 open(os.path.join(GOLDEN, "sample_4.jpg"), "rb").read()):
        info, coef, quant = host.jpeg_entropy_decode(data)
        assert _same_info(info, o.jpeg_info(data))
        oc, oq = o.jpeg_coefficients(data)
        assert (coef == oc).all() and (quant[:info.components] == oq).all()


def test_host_entropy_decode_restart_intervals_threaded(host):
    """Restart intervals are independent bit streams: decoded on several threads, same coefficients."""
    data = encode(synth_image(480, 640, 3), quality=85, subsampling=2, restart_marker_rows=1)
    assert o.jpeg_info(data).restart_interval == 40
    one = host.jpeg_entropy_decode(data, threads=1)[1]
    many = host.jpeg_entropy_decode(data, threads=8)[1]
    assert (one == many).all() and (one == o.jpeg_coefficients(data)[0]).all()


def test_host_jpeg_errors(host):
    import sudoku_vision_amd as sva
    with pytest.raises(sva._native.NativeError, match="SV_ERR_UNSUPPORTED"):
        host.jpeg_parse(encode(synth_image(32, 32, 1), quality=80, progressive=True))
    with pytest.raises(sva._native.NativeError, match="SV_ERR_BAD_ARG"):
        host.jpeg_parse(b"\x89PNG\r\n\x1a\n" + b"\0" * 64)
    good = encode(synth_image(64, 64, 1), quality=80)
    with pytest.raises(sva._native.NativeError, match="SV_ERR_BAD_ARG"):
        host.jpeg_entropy_decode(good[:len(good) // 4])                    # cut inside the tables / before the scan
    cmyk = io.BytesIO()
    Image.fromarray(synth_image(16, 16, 2)).convert("CMYK").save(cmyk, "JPEG")
    with pytest.raises(sva._native.NativeError, match="SV_ERR_UNSUPPORTED"):
        host.jpeg_parse(cmyk.getvalue())


# ---- non-interleaved (one scan per component) baseline files ---------------------------------------------------------
# Pillow only writes interleaved scans, so the test re-encodes a Pillow file: same tables and coefficients, three scans.
def _parse_tables(data):
    """(dqt segments raw, {(class, id): (counts[16], symbols)}, sof payload, component table selectors from the SOS)"""
    pos, dqt, dht, sof, sel, app = 2, [], {}, None, None, []
    while True:
        assert data[pos] == 0xFF
        m, L = data[pos + 1], int.from_bytes(data[pos + 2:pos + 4], "big")
        seg = data[pos + 4:pos + 2 + L]
        if m == 0xDB:
            dqt.append(seg)
        elif m == 0xC4:
            s = seg
            while s:
                tc, th = s[0] >> 4, s[0] & 15
                counts = list(s[1:17])
                tot = sum(counts)
                dht[(tc, th)] = (counts, list(s[17:17 + tot]))
                s = s[17 + tot:]
        elif m == 0xC0:
            sof = seg
        elif m in (0xE0, 0xE1):
            app.append(data[pos:pos + 2 + L])
        elif m == 0xDA:
            ns = seg[0]
            sel = {seg[1 + 2 * i]: (seg[2 + 2 * i] >> 4, seg[2 + 2 * i] & 15) for i in range(ns)}
            return dqt, dht, sof, sel, app
        pos += 2 + L


def _huff_codes(counts, symbols):
    codes, code, k = {}, 0, 0
    for ln in range(1, 17):
        for _ in range(counts[ln - 1]):
            codes[symbols[k]] = (code, ln)
            code += 1
            k += 1
        code <<= 1
    return codes


class _BitWriter:
    def __init__(self):
        self.out, self.acc, self.n = bytearray(), 0, 0

    def put(self, value, length):
        self.acc = (self.acc << length) | (value & ((1 << length) - 1))
        self.n += length
        while self.n >= 8:
            b = (self.acc >> (self.n - 8)) & 0xFF
            self.out.append(b)
            if b == 0xFF:
                self.out.append(0)
            self.n -= 8

    def flush(self):
        if self.n:
            self.put((1 << (8 - self.n)) - 1, 8 - self.n)
        return bytes(self.out)


def _encode_noninterleaved(data):
    """Re-encodes a Pillow (interleaved, 3-component) baseline JPEG with one scan per component."""
    info = o.jpeg_info(data)
    coef, _ = o.jpeg_coefficients(data)
    dqt, dht, sof, sel, app = _parse_tables(data)
    H, W = info.height, info.width
    hs = [info.h_samp, 1, 1]
    vs = [info.v_samp, 1, 1]
    mcux, mcuy = -(-W // (8 * info.h_samp)), -(-H // (8 * info.v_samp))
    out = bytearray(b"\xff\xd8")
    for a in app:
        out += a
    for q in dqt:
        out += b"\xff\xdb" + (len(q) + 2).to_bytes(2, "big") + q
    out += b"\xff\xc0" + (len(sof) + 2).to_bytes(2, "big") + sof
    for (tc, th), (counts, symbols) in dht.items():
        body = bytes([tc << 4 | th]) + bytes(counts) + bytes(symbols)
        out += b"\xff\xc4" + (len(body) + 2).to_bytes(2, "big") + body
    off = 0
    zz = _ZZ
    for c in range(3):
        cid = sof[6 + 3 * c]
        td, ta = sel[cid]
        dc_codes, ac_codes = _huff_codes(*dht[(0, td)]), _huff_codes(*dht[(1, ta)])
        bw, bh = mcux * hs[c], mcuy * vs[c]                                   # padded grid the coefficients are stored on
        rw = -(-(-(-W * hs[c] // info.h_samp)) // 8)                          # real block grid of a non-interleaved scan
        rh = -(-(-(-H * vs[c] // info.v_samp)) // 8)
        out += b"\xff\xda" + (8).to_bytes(2, "big") + bytes([1, cid, td << 4 | ta, 0, 63, 0])
        bwr, pred = _BitWriter(), 0
        for by in range(rh):
            for bx in range(rw):
                blk = coef[off + (by * bw + bx) * 64: off + (by * bw + bx) * 64 + 64]
                diff = int(blk[0]) - pred
                pred = int(blk[0])
                size = abs(diff).bit_length()
                bwr.put(*dc_codes[size])
                if size:
                    bwr.put(diff if diff > 0 else diff + (1 << size) - 1, size)
                run = 0
                last = max([k for k in range(1, 64) if blk[zz[k]] != 0], default=0)
                for k in range(1, last + 1):
                    v = int(blk[zz[k]])
                    if v == 0:
                        run += 1
                        continue
                    while run > 15:
                        bwr.put(*ac_codes[0xF0])
                        run -= 16
                    size = abs(v).bit_length()
                    bwr.put(*ac_codes[run << 4 | size])
                    bwr.put(v if v > 0 else v + (1 << size) - 1, size)
                    run = 0
                if last < 63:
                    bwr.put(*ac_codes[0x00])
        out += bwr.flush()
        off += bw * bh * 64
    return bytes(out + b"\xff\xd9")


@pytest.mark.parametrize("h,w,sub", [(64, 80, 2), (61, 83, 2), (50, 37, 1), (40, 40, 0), (17, 9, 2)])
def test_noninterleaved_scans(host, h, w, sub):
    """One scan per component: MCU = one block of the component's own (unpadded) block grid.  The oracle, Pillow and the
    product's host decoder agree on a file re-encoded that way, and it decodes to the same pixels as the interleaved original."""
    original = encode(synth_image(h, w, 7 * h + w), quality=85, subsampling=sub)     # standard (complete) Huffman tables
    data = _encode_noninterleaved(original)
    assert data != original and data.count(b"\xff\xda") >= 3
    want = pil_bgr(data)
    assert (want == pil_bgr(original)).all()                                 # the re-encoding is lossless
    assert (o.imdecode(data) == want).all()
    info, coef, quant = host.jpeg_entropy_decode(data)
    oc, oq = o.jpeg_coefficients(data)
    assert (coef == oc).all() and (quant == oq).all()
    info2, masks, offs, vals, _ = host.jpeg_entropy_decode_sparse(data)
    assert (_densify(info2, masks, offs, vals) == oc).all()


def test_exif_orientation_big_endian(host):
    """EXIF written Motorola-order ("MM"), as many cameras do: same orientation handling as the little-endian form Pillow writes."""
    base = encode(synth_image(24, 40, 3), quality=90, subsampling=2)
    for orient in (3, 6, 8):
        tiff = b"MM\x00\x2a\x00\x00\x00\x08" + b"\x00\x01" + b"\x01\x12\x00\x03\x00\x00\x00\x01" + orient.to_bytes(2, "big") + b"\x00\x00" + b"\x00\x00\x00\x00"
        seg = b"Exif\x00\x00" + tiff
        data = base[:2] + b"\xff\xe1" + (len(seg) + 2).to_bytes(2, "big") + seg + base[2:]
        assert Image.open(io.BytesIO(data)).getexif().get(0x0112) == orient     # Pillow reads the tag we wrote
        assert o.jpeg_info(data).orientation == orient and host.jpeg_parse(data).orientation == orient
        assert (o.imdecode(data) == pil_bgr(data)).all()


def test_random_files_oracle_pillow_host(host):
    """Seeded sweep over sizes, qualities, sub-samplings, restart intervals and table optimisation: the oracle equals Pillow
    pixel for pixel, and the product's host decoder (dense and compact form) equals the oracle coefficient for coefficient."""
    rs = np.random.RandomState(2024)
    for it in range(40):
        h, w = int(rs.randint(1, 120)), int(rs.randint(1, 160))
        gray = it % 6 == 0
        kw = dict(quality=int(rs.randint(1, 101)))
        if not gray:
            kw["subsampling"] = int(rs.randint(0, 3))
        if it % 5 == 0:
            kw["restart_marker_blocks"] = int(rs.randint(1, 7))
        if it % 4 == 0:
            kw["optimize"] = True
        data = encode(synth_image(h, w, it, gray=gray), **kw)
        assert (o.imdecode(data) == pil_bgr(data)).all(), (h, w, kw)
        oc, oq = o.jpeg_coefficients(data)
        info, coef, quant = host.jpeg_entropy_decode(data, threads=1 + it % 3)
        assert (coef == oc).all() and (quant[:info.components] == oq).all(), (h, w, kw)
        info2, masks, offs, vals, _ = host.jpeg_entropy_decode_sparse(data, threads=1 + it % 2)
        assert (_densify(info2, masks, offs, vals) == oc).all(), (h, w, kw)


# ---- damaged files: no write past a buffer, libjpeg's insufficient-data behaviour ----------------------------
def _pil_truncated(data):
    """Pillow's decode of a cut file (LOAD_TRUNCATED_IMAGES): the rows libjpeg-turbo completed, then mid-grey."""
    from PIL import ImageFile
    ImageFile.LOAD_TRUNCATED_IMAGES = True
    try:
        return pil_bgr(data)
    finally:
        ImageFile.LOAD_TRUNCATED_IMAGES = False


@pytest.mark.parametrize("kw", [dict(quality=90, subsampling=2), dict(quality=85, subsampling=0),
                                dict(quality=85, subsampling=2, restart_marker_rows=1), dict(quality=85, subsampling=1, restart_marker_blocks=5)])
def test_truncated_files_stay_inside_their_buffers(host, kw):
    """A file cut anywhere after its first scan header decodes without error into: every block the remaining bits hold, then zero
    blocks (libjpeg: insufficient_data).  The sparse form never uses more values than the capacity sv_jpeg_parse promised (the
    round-1 build wrote 7.0 M values into a 2.4 M buffer here).  Product == oracle; rows above the cut == Pillow; the tail is grey."""
    h, w = 240, 320
    data = encode(synth_image(h, w, 77), **kw)
    full = pil_bgr(data)
    sos = data.index(b"\xff\xda")
    body = len(data) - sos
    for frac in (0.15, 0.4, 0.5, 0.77, 0.98):
        cut = data[:sos + 14 + int(frac * (body - 14))]
        if kw.get("restart_marker_rows") or kw.get("restart_marker_blocks"):
            # with restart markers a cut file has too few intervals: rejected cleanly (a decoder may also resynchronise; cv2 shows grey)
            import sudoku_vision_amd as sva
            try:
                info, masks, offs, vals, quant = host.jpeg_entropy_decode_sparse(cut, threads=2)
            except sva._native.NativeError as e:
                assert "SV_ERR_BAD_ARG" in str(e)
                continue
        else:
            info, masks, offs, vals, quant = host.jpeg_entropy_decode_sparse(cut, threads=2)
        assert len(vals) <= info.sparse_capacity
        _, coef, _ = host.jpeg_entropy_decode(cut)
        oc, oq = o.jpeg_coefficients(cut)
        assert (coef == oc).all() and (_densify(info, masks, offs, vals) == oc).all()
        img = o.imdecode(cut)
        pil = _pil_truncated(cut)
        good = int(np.argmin((pil == full).all(axis=(1, 2))))          # first row Pillow could not complete
        assert good > 0 and (img[:max(good - 16, 0)] == full[:max(good - 16, 0)]).all()
        if frac < 0.9:                                                   # (at 0.98 the cut is inside the last MCU row)
            assert (img[-8:] == 128).all() and (pil[-8:] == 128).all()


def test_truncated_photo_capacity(host):
    """The advisor's reproduction: the fixture photo cut to 605,593 and 205,593 bytes."""
    data = open(os.path.join(GOLDEN, "sample_4.jpg"), "rb").read()
    for keep in (605593, 205593):
        info, masks, offs, vals, quant = host.jpeg_entropy_decode_sparse(data[:keep], threads=1)
        assert len(vals) <= info.sparse_capacity
        assert (_densify(info, masks, offs, vals) == o.jpeg_coefficients(data[:keep])[0]).all()


def test_oversubscribed_dht_is_rejected(host):
    """SOI + a DHT whose 255 one-bit codes pass the total <= 256 check: rejected before the look-up table is indexed with them
    (the round-1 build wrote ~130 k entries into a 1024-entry table)."""
    import sudoku_vision_amd as sva
    counts = bytes([255] + [0] * 15)
    seg = bytes([0x00]) + counts + bytes(range(255))
    evil = b"\xff\xd8" + b"\xff\xc4" + (len(seg) + 2).to_bytes(2, "big") + seg + b"\xff\xd9"
    with pytest.raises(sva._native.NativeError, match="SV_ERR_BAD_ARG"):
        host.jpeg_parse(evil)
    for c in ([0, 5] + [0] * 14, [3] + [0] * 15, [1, 1, 1, 1, 1, 1, 1, 1, 1, 4] + [0] * 6):      # 5 two-bit codes, 3 one-bit codes, 4 after a full prefix chain
        tot = sum(c)
        seg = bytes([0x10]) + bytes(c) + bytes(range(tot))
        evil = b"\xff\xd8" + b"\xff\xc4" + (len(seg) + 2).to_bytes(2, "big") + seg + b"\xff\xd9"
        with pytest.raises(sva._native.NativeError, match="SV_ERR_BAD_ARG"):
            host.jpeg_parse(evil)


def test_repeated_scan_component_is_rejected(host):
    """A non-interleaved file whose second scan repeats the first component never finishes `covered` by counting: rejected."""
    import sudoku_vision_amd as sva
    data = _encode_noninterleaved(encode(synth_image(40, 40, 3), quality=85, subsampling=0))
    host.jpeg_entropy_decode(data)                                       # the well-formed file decodes
    first = data.index(b"\xff\xda")
    second = data.index(b"\xff\xda", first + 2)
    ln = int.from_bytes(data[second + 2:second + 4], "big")
    seg = bytearray(data[second:second + 2 + ln])
    seg[5] = data[first + 5]                                             # component selector of scan 2 := that of scan 1
    bad = data[:second] + bytes(seg) + data[second + 2 + ln:]
    with pytest.raises(sva._native.NativeError, match="SV_ERR_BAD_ARG"):
        host.jpeg_entropy_decode(bad)
