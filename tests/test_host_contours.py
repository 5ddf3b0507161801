"""CPU: the host corner search (cv/grid.py:16-71).  Product = csrc/host_contours.cpp through the C ABI;
checker = oracle/sv_oracle.c; independent cross-checks = scipy.ndimage.label and known cv2 conventions."""
import numpy as np
import pytest
from scipy import ndimage

import sv_oracle as o


@pytest.fixture(scope="module")
def host():
    import sudoku_vision_amd as sva
    sva._native.lib()
    return sva.host


def _blobs(seed, h, w, thr=0.55):
    rs = np.random.RandomState(seed)
    return ((ndimage.gaussian_filter(rs.uniform(size=(h, w)), 2.5) > np.quantile(ndimage.gaussian_filter(rs.uniform(size=(h, w)), 2.5), thr)) * 255).astype(np.uint8)


def test_oracle_known_conventions():
    img = np.zeros((8, 8), np.uint8)
    img[2:6, 2:6] = 255
    # cv2 lists a filled square counter-clockwise from its top-left pixel: TL, BL, BR, TR
    assert [c.reshape(-1, 2).tolist() for c in o.find_contours(img)] == [[[2, 2], [2, 5], [5, 5], [5, 2]]]
    img[3:5, 3:5] = 0                       # a hole is not an external contour
    assert len(o.find_contours(img)) == 1
    img[:] = 0
    img[0, 0] = img[2, 3] = img[7, 7] = 255
    got = [c.reshape(-1, 2).tolist() for c in o.find_contours(img)]
    assert got == [[[7, 7]], [[3, 2]], [[0, 0]]]          # last found first; single pixels; image-border pixels kept
    assert o.find_contours(np.zeros((5, 5), np.uint8)) == []
    full = o.find_contours(np.full((4, 6), 255, np.uint8))
    assert [c.reshape(-1, 2).tolist() for c in full] == [[[0, 0], [0, 3], [5, 3], [5, 0]]]


def test_oracle_area_length_approx():
    sq = np.array([[2, 2], [2, 5], [5, 5], [5, 2]])
    assert o.contour_area(sq) == 9.0 and o.arc_length(sq) == 12.0 and o.arc_length(sq, closed=False) == 9.0
    assert o.contour_area(np.zeros((0, 2))) == 0.0
    # a noisy rectangle collapses to its 4 corners
    t = np.arange(0, 100)
    top = np.stack([t, (t % 2)], 1)
    right = np.stack([100 + (t % 2), t], 1)
    bottom = np.stack([100 - t, 100 - (t % 2)], 1)
    left = np.stack([(t % 2), 100 - t], 1)
    poly = np.concatenate([top, right, bottom, left])
    ap = o.approx_poly_dp(poly, 0.02 * o.arc_length(poly)).reshape(-1, 2)
    assert len(ap) == 4
    want = np.array([[0, 0], [0, 100], [100, 0], [100, 100]])
    assert np.abs(ap[np.lexsort((ap[:, 1], ap[:, 0]))] - want).max() <= 1
    assert len(o.approx_poly_dp(poly, 0.0)) > 100          # epsilon 0 keeps the zig-zag


@pytest.mark.parametrize("seed", [0, 1, 2])
def test_oracle_external_contours_vs_labels(seed):
    """Every external contour belongs to exactly one 8-connected component and starts at that component's
    first pixel in raster order; components enclosed by another component's hole are not reported."""
    img = _blobs(seed, 90, 120)
    lab, n = ndimage.label(img > 0, structure=np.ones((3, 3)))
    cs = o.find_contours(img)
    seen = set()
    for c in cs:
        p = c.reshape(-1, 2)
        ids = {lab[y, x] for x, y in p}
        assert len(ids) == 1 and 0 not in ids
        cid = ids.pop()
        assert cid not in seen
        seen.add(cid)
        ys, xs = np.nonzero(lab == cid)
        assert (p[0] == [xs[ys == ys.min()].min(), ys.min()]).all()
    # components not reported must be enclosed: filling the holes of the reported ones covers them
    filled = ndimage.binary_fill_holes(np.isin(lab, list(seen)), structure=np.ones((3, 3)))  # holes w.r.t. 4-connected background
    missing = set(range(1, n + 1)) - seen
    for cid in missing:
        assert filled[lab == cid].all()


@pytest.mark.parametrize("seed", [3, 4, 5, 6])
def test_product_equals_oracle_on_blobs(host, seed):
    img = _blobs(seed, 150, 200, thr=0.45 + 0.05 * (seed % 3))
    a, b = host.find_contours(img), o.find_contours(img)
    assert len(a) == len(b) and all(x.shape == y.shape and (x == y).all() for x, y in zip(a, b))
    for c in a[:50]:
        assert host.contour_area(c) == o.contour_area(c)
        assert host.arc_length(c) == o.arc_length(c)
        for er in (0.0, 0.005, 0.02, 0.1):
            eps = er * o.arc_length(c)
            assert (host.approx_poly_dp(c, eps) == o.approx_poly_dp(c, eps)).all()
            assert (host.approx_poly_dp(c, eps, closed=False) == o.approx_poly_dp(c, eps, closed=False)).all()
    for mar in (0.1, 0.01, 0.0):
        g, go = host.find_grid_corners(img, mar), o.find_grid_contour(img, mar)
        assert (g is None) == (go is None) and (g is None or (g == go).all())


@pytest.mark.parametrize("seed", range(8))
def test_bit_scanner_equals_byte_scanner(host, seed):
    """The scanner the batched pipeline uses works on the bit-packed image without expanding it (labels in two side bit planes,
    the raster scan visits only label pixels and run starts and jumps from a positive label to the next right-exit).  Contour by
    contour -- points, order -- it gives what the byte-image scanner gives: blobs at several densities (nested components,
    touching image borders, 1-pixel specks), widths 32..416, a frame-like case with a big outline full of clutter, and the
    reused per-thread planes (two different images back to back)."""
    rs = np.random.RandomState(seed)
    for h, w, thr in ((64, 32, 0.5), (97, 160, 0.45), (200, 416, 0.6), (130, 96, 0.3)):
        img = _blobs(seed * 7 + h, h, w, thr)
        img[rs.randint(0, h, 30), rs.randint(0, w, 30)] = 255             # specks, some on the border
        if seed % 2:
            img[5:h - 5, 4] = 255; img[5:h - 5, w - 5] = 255; img[5, 4:w - 4] = 255; img[h - 6, 4:w - 4] = 255   # an outline around the clutter
        bits = np.packbits(img > 0, axis=1, bitorder="little").view(np.uint32)
        a, b = host.find_contours(img), host.find_contours_bits(bits, h, w)
        assert len(a) == len(b)
        for ca, cb in zip(a, b):
            assert ca.shape == cb.shape and (ca == cb).all()
        got_c, got_f = host.find_grid_corners_bits_batch(bits[None], h, w, 0.1, 0.02, 1)
        want = host.find_grid_corners(img)
        assert bool(got_f[0]) == (want is not None) and (want is None or (got_c[0] == want).all())


def test_synthetic_frames_corners(host):
    from sudoku_vision_amd.synth import synth_frames
    frames, corners, _ = synth_frames(3, 540, 960, seed=12)
    bins = np.stack([o.preprocess_for_grid_detection(f) for f in frames.numpy()])
    got, found = host.find_grid_corners_batch(bins, threads=2)
    assert found.all()
    for i in range(3):
        assert (got[i] == o.find_grid_contour(bins[i])).all()
        assert got[i].dtype == np.int32 and got[i].shape == (4, 2)
        ordered = o.order_points(got[i].astype(np.float32))
        assert np.abs(ordered - corners[i]).max() <= 6          # outer edge of the border line, ~half a line width out
    assert host.find_grid_corners(np.zeros((64, 64), np.uint8)) is None        # "not found" is None, never an exception
    noise = (np.random.RandomState(0).uniform(size=(128, 128)) > 0.7).astype(np.uint8) * 255
    assert host.find_grid_corners(noise) is None and o.find_grid_contour(noise) is None


def test_reference_named_functions(host):
    """find_contours / approximate_polygon / find_grid_contour as cv/grid.py exposes them."""
    from sudoku_vision_amd.cv.grid import find_contours, approximate_polygon, find_grid_contour, order_points
    img = np.zeros((200, 300), np.uint8)
    img[20:180, 40:260] = 255
    img[30:170, 50:250] = 0
    cs = find_contours(img)
    assert len(cs) == 1 and cs[0].dtype == np.int32 and cs[0].shape[1:] == (1, 2)
    ap = approximate_polygon(cs[0])
    assert ap.shape == (4, 1, 2)
    g = find_grid_contour(img)
    assert g.shape == (4, 2) and order_points(g).tolist() == [[40, 20], [259, 20], [259, 179], [40, 179]]
    assert find_grid_contour(img, min_area_ratio=0.9) is None


def test_buffer_protocol_and_bad_args(host):
    import ctypes as C
    import sudoku_vision_amd as sva
    lib = sva._native.lib()
    img = np.zeros((16, 16), np.uint8)
    img[4:8, 4:8] = 255
    n, m = C.c_long(), C.c_int()
    assert lib.sv_find_contours_u8(img.ctypes.data_as(C.c_void_p), 16, 16, 16, None, 0, None, 0, C.byref(n), C.byref(m)) == -6
    assert (n.value, m.value) == (4, 1)
    assert lib.sv_find_grid_corners_u8(None, 16, 16, 16, 0.1, 0.02, None) == -1
    with pytest.raises(TypeError):
        host.find_grid_corners(np.zeros((4, 4), np.float32))


def test_reference_photos_success_rate(host):
    """The reference prints 'CV success rate on test images: 4/5' (tests/test_integration.py:261).  The restated pipeline
    (oracle K1 + host corner search) finds a grid on exactly 4 of the 5 photos.  Needs the reference tree (skipped elsewhere)."""
    import os
    from PIL import Image
    d = "/root/reference/data/test_images"
    if not os.path.isdir(d):
        pytest.skip("reference tree not present")
    found = []
    for i in range(1, 6):
        img = np.asarray(Image.open(os.path.join(d, f"sample_{i}.jpg")).convert("RGB"))[..., ::-1].copy()
        b = o.preprocess_for_grid_detection(img)
        g = host.find_grid_corners(b)
        go = o.find_grid_contour(b)
        assert (g is None) == (go is None) and (g is None or (g == go).all())
        found.append(g is not None)
    assert sum(found) == 4


def _pack_sparse_np(host, bits, cap):
    """numpy statement of the sparse record sv_pack_sparse_bits writes (include/sudoku_vision_hip.h)"""
    H, wpr = bits.shape
    gpr = (wpr + 63) // 64
    nz = bits != 0
    masks = np.zeros((H, gpr), np.uint64)
    for g in range(gpr):
        seg = nz[:, 64 * g:64 * g + 64]
        masks[:, g] = (seg.astype(np.uint64) << np.arange(seg.shape[1], dtype=np.uint64)).sum(1, dtype=np.uint64)
    vals = bits[nz].astype(np.uint32)
    stride = host.sparse_bits_record_bytes(H, wpr * 32, cap)
    cap = (stride - 8 - 8 * H * gpr) // 4                      # what fits in the (16-byte rounded) record is the capacity
    rec = np.zeros(stride, np.uint8)
    rec[:8] = np.array([vals.size, cap], np.uint32).view(np.uint8)
    rec[8:8 + 8 * H * gpr] = masks.reshape(-1).view(np.uint8)
    k = min(vals.size, cap)
    rec[8 + 8 * H * gpr:8 + 8 * H * gpr + 4 * k] = vals[:k].view(np.uint8)
    return rec


@pytest.mark.parametrize("H,W", [(64, 64), (120, 1920), (40, 2752), (33, 4128)])
def test_sparse_record_roundtrip_and_search(host, H, W):
    """Host half of the sparse hand-over: a record expands to the dense image it was made from, the search on records equals the search on the
    dense images, and a record that overflowed is reported (found == 2), not searched."""
    rng = np.random.RandomState(H + W)
    imgs = []
    for density in (0.0, 0.02, 0.3):
        b = (rng.rand(H, W) < density)
        b[H // 8:H - H // 8, W // 8:W // 8 + 3] = True               # something large enough to be a contour
        b[H // 8:H // 8 + 3, W // 8:W - W // 8] = True
        imgs.append(np.packbits(b, axis=1, bitorder="little").view(np.uint32))
    words = H * (W // 32)
    recs = np.stack([_pack_sparse_np(host, b, words) for b in imgs])
    for b, r in zip(imgs, recs):
        assert np.array_equal(host.sparse_bits_expand(r, H, W), b)
    c1, f1 = host.find_grid_corners_sparse_batch(recs, H, W, 0.01, 0.02, 2)
    c2, f2 = host.find_grid_corners_bits_batch(np.stack(imgs), H, W, 0.01, 0.02, 2)
    assert np.array_equal(f1.astype(bool), f2) and np.array_equal(c1, c2)
    small = np.stack([_pack_sparse_np(host, b, 8) for b in imgs])
    _, f3 = host.find_grid_corners_sparse_batch(small, H, W, 0.01, 0.02, 2)
    assert (f3 == 2).all()
    with pytest.raises(Exception):
        host.sparse_bits_expand(small[0], H, W)


def test_pool_affinity(host):
    """sv_host_pool_set_affinity pins the library's worker threads (visible in /proc/self/task) and the search still gives the same answer;
    the cpulist parser of pipeline.gpu_local_cpus."""
    import os
    from sudoku_vision_amd.pipeline import _parse_cpulist
    assert _parse_cpulist("0-3,8,10-11\n") == {0, 1, 2, 3, 8, 10, 11} and _parse_cpulist("") == set()
    if not hasattr(os, "sched_getaffinity"):
        pytest.skip("no sched_getaffinity")
    allowed = sorted(os.sched_getaffinity(0))
    imgs = np.stack([_blobs(s, 96, 128) for s in range(6)])
    ref = host.find_grid_corners_batch(imgs, 0.01, 0.02, 3)

    def masks():
        out = []
        for t in os.listdir("/proc/self/task"):
            for ln in open(f"/proc/self/task/{t}/status"):
                if ln.startswith("Cpus_allowed_list:"):
                    out.append(ln.split()[1])
        return out

    try:
        host.set_pool_affinity([allowed[-1]])
        got = host.find_grid_corners_batch(imgs, 0.01, 0.02, 3)
        assert masks().count(str(allowed[-1])) >= 2                     # the two workers beside the calling thread
        assert np.array_equal(got[0], ref[0]) and np.array_equal(got[1], ref[1])
    finally:
        host.set_pool_affinity(allowed)
    assert host._native.lib().sv_host_pool_set_affinity(None, 3) != 0          # SV_ERR_BAD_ARG


def test_bit_scanner_shapes_with_equal_word_count(host):
    """480x640 and 640x480 have the same number of bit-plane words (9600); the per-thread label planes and
    per-row dirty spans of the bit scanner must be rebuilt when H or words-per-row change, not only when their product does.
    One thread, landscape -> portrait -> landscape, every result against the byte scanner."""
    def frame(seed, h, w):
        b = _blobs(seed, h, w, thr=0.6)
        b[h // 6:h - h // 6, w // 6:w // 6 + 4] = 255
        b[h // 6:h - h // 6, w - w // 6 - 4:w - w // 6] = 255
        b[h // 6:h // 6 + 4, w // 6:w - w // 6] = 255
        b[h - h // 6 - 4:h - h // 6, w // 6:w - w // 6] = 255
        return b
    for rep, (h, w) in enumerate([(480, 640), (640, 480), (480, 640), (96, 3200), (3200, 96), (640, 480)]):
        img = frame(rep, h, w)
        bits = np.packbits(img != 0, axis=1, bitorder="little").view(np.uint32)
        cb, fb = host.find_grid_corners_bits_batch(bits[None], h, w, 0.05, 0.02, 1)
        cu, fu = host.find_grid_corners_batch(img[None], 0.05, 0.02, 1)
        assert np.array_equal(fb.astype(bool), fu.astype(bool)) and np.array_equal(cb, cu), (h, w)


def test_sparse_record_validation(host):
    """a stale, torn or foreign sparse record must be rejected, never expanded past the row or the record."""
    H, W = 40, 2752                                               # wpr = 86: the second mask word of a row has 22 valid bits
    rng = np.random.RandomState(5)
    b = rng.rand(H, W) < 0.05
    bits = np.packbits(b, axis=1, bitorder="little").view(np.uint32)
    good = _pack_sparse_np(host, bits, H * (W // 32))
    assert np.array_equal(host.sparse_bits_expand(good, H, W), bits)
    gpr = 2
    # (1) a mask bit beyond the last word of the row
    bad = good.copy()
    m = bad[8:8 + 8 * H * gpr].view(np.uint64)
    m[1] |= np.uint64(1) << np.uint64(40)
    with pytest.raises(Exception):
        host.sparse_bits_expand(bad, H, W)
    # (2) more mask bits than n_values says (a torn record: masks of one frame, header of another)
    bad = good.copy()
    bad[:4] = np.array([3], np.uint32).view(np.uint8)
    with pytest.raises(Exception):
        host.sparse_bits_expand(bad, H, W)
    # (3) a capacity that does not fit the stride
    bad = good.copy()
    bad[:8] = np.array([10, 1 << 30], np.uint32).view(np.uint8)
    _, f = host.find_grid_corners_sparse_batch(bad[None], H, W, 0.01, 0.02, 1)
    assert f[0] == 2
    # the batch search reports all three as "search the dense image" and never crashes; the good one is searched
    recs = np.stack([good, bad])
    _, f = host.find_grid_corners_sparse_batch(recs, H, W, 0.01, 0.02, 1)
    assert f[0] in (0, 1) and f[1] == 2
    # (4) a stride smaller than the header + masks
    import ctypes as C
    lib = host._native.lib()
    out = np.zeros(8, np.int32); fo = np.zeros(1, np.uint8)
    assert lib.sv_find_grid_corners_sparse_batch(good.ctypes.data_as(C.c_void_p), C.c_long(64), 1, H, W, C.c_double(0.01), C.c_double(0.02),
                                                 out.ctypes.data_as(C.c_void_p), fo.ctypes.data_as(C.c_void_p), 1) == -1


@pytest.mark.parametrize("H,W", [(96, 128), (70, 2080), (130, 4128)])
def test_sparse_search_sequences_on_one_thread(host, H, W):
    """The per-thread scratch of the bit scanner survives from frame to frame (the image is rewritten chunk by chunk where the old or the new
    frame has words, labels are cleared through a touched list): sequences of very different frames on ONE thread -- dense, empty, shifted,
    sparse, the same again -- through the sparse entry and the dense entry in turn, every answer against the byte scanner; and the
    contour lists of the bit and byte scanners on random images of many densities."""
    rng = np.random.RandomState(H * W)

    def frame(kind, seed):
        rs = np.random.RandomState(seed)
        if kind == "empty":
            return np.zeros((H, W), bool)
        if kind == "full":
            return np.ones((H, W), bool)
        b = rs.rand(H, W) < {"sparse": 0.002, "mid": 0.08, "dense": 0.45}[kind]
        if seed % 2:
            y0, x0 = rs.randint(2, H // 3), rs.randint(2, W // 3)
            b[y0:H - 3, x0:x0 + 3] = True
            b[y0:H - 3, W - 8:W - 5] = True
            b[y0:y0 + 3, x0:W - 5] = True
            b[H - 6:H - 3, x0:W - 5] = True
        return b
    kinds = ["dense", "empty", "sparse", "mid", "full", "sparse", "empty", "mid", "dense", "mid"]
    words = H * (W // 32)
    for i, kind in enumerate(kinds * 2):
        b = frame(kind, i)
        img = (b * 255).astype(np.uint8)
        bits = np.ascontiguousarray(np.packbits(b, axis=1, bitorder="little").view(np.uint32))
        want = host.find_grid_corners(img, 0.05)
        if i % 3 == 2:                                            # the dense entry in between: it shares the scratch with the sparse one
            c, f = host.find_grid_corners_bits_batch(bits[None], H, W, 0.05, 0.02, 1)
        else:
            rec = _pack_sparse_np(host, bits, words)
            c, f = host.find_grid_corners_sparse_batch(rec[None], H, W, 0.05, 0.02, 1)
            assert f[0] in (0, 1)
        assert bool(f[0]) == (want is not None) and (want is None or (c[0] == want).all()), (i, kind)
    for density in (0.001, 0.01, 0.1, 0.3, 0.5, 0.7, 0.97):
        b = rng.rand(H, W) < density
        a = host.find_contours((b * 255).astype(np.uint8))
        g = host.find_contours_bits(np.ascontiguousarray(np.packbits(b, axis=1, bitorder="little").view(np.uint32)), H, W)
        assert len(a) == len(g) and all(np.array_equal(x, y) for x, y in zip(a, g)), density
