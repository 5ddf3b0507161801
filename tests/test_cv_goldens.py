"""Committed golden checksums of the unpinned CV stages (tests/golden/cv_goldens.json, photo_goldens.npz; made by
tests/golden/make_cv_goldens.py from the C oracle) and oracle-independent property tests of the contour routines.

CPU part (this file, no GPU): the live oracle and the product's host corner search (csrc/host_contours.cpp, runs without a GPU)
reproduce the committed values, so an edit that changes oracle and product together is caught.  The GPU part is
tests/test_gpu_goldens.py.  PARITY CAVEAT (every report carries it): these goldens come from a restatement of OpenCV, not from
cv2 -- cv2 is absent from the image and the reference's tests hold no cv2 outputs (cv/test_pipeline.py:163)."""
import hashlib
import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import sv_oracle as o  # noqa: E402

GOLDEN = os.path.join(ROOT, "tests", "golden")
G = json.load(open(os.path.join(GOLDEN, "cv_goldens.json")))


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@pytest.fixture(scope="module")
def host():
    import __graft_entry__ as g
    import sudoku_vision_amd as sva
    if not os.path.exists(sva._native.LIB_PATH):
        g.build()
    return sva.host


def synthetic_frames(rec):
    from sudoku_vision_amd.synth import synth_frames
    frames, corners, _ = synth_frames(rec["n"], rec["H"], rec["W"], seed=rec["seed"], noise="int")
    return frames[rec["index"]].numpy(), corners[rec["index"]]


@pytest.mark.parametrize("rec", G["synthetic"], ids=lambda r: f"{r['H']}x{r['W']}-s{r['seed']}-{r['index']}")
def test_oracle_and_host_search_reproduce_synthetic_goldens(host, rec):
    import sudoku_vision_amd as sva
    f, corners = synthetic_frames(rec)
    assert sha(f) == rec["frame_sha256"], "synth_frames(noise='int') is no longer bit-stable: regenerate goldens knowingly"
    binary = o.preprocess_for_grid_detection(f)
    assert sha(binary) == rec["binary_sha256"]
    assert sha(o.corners_to_minv(corners)) == rec["minv_sha256"]
    assert sha(sva.Context.corners_to_minv(corners[None])[0]) == rec["minv_sha256"]        # product homography (host fp64)
    assert sha(o.warp_cells(f, corners)) == rec["cells_sha256"]
    got = host.find_grid_corners(binary)                                                  # product corner search (C++, host)
    assert (None if got is None else got.tolist()) == rec["found_corners"]


@pytest.mark.parametrize("rec", G["photos"], ids=lambda r: r["file"])
def test_oracle_and_host_search_reproduce_photo_goldens(host, rec):
    """All five data/test_images photos (data fixtures).  sample_2 is the reference's "1 of 5" without a grid."""
    from PIL import Image
    img = np.asarray(Image.open(os.path.join(GOLDEN, rec["file"])).convert("RGB"))[..., ::-1].copy()
    assert list(img.shape) == rec["shape"] and sha(img) == rec["frame_sha256"]
    assert (o.imdecode(open(os.path.join(GOLDEN, rec["file"]), "rb").read()) == img).all()  # the JPEG oracle agrees with Pillow
    binary = o.preprocess_for_grid_detection(img)
    assert sha(binary) == rec["binary_sha256"]
    got = host.find_grid_corners(binary)
    assert (None if got is None else got.tolist()) == rec["corners"]
    if rec["corners"] is not None:
        P = np.load(os.path.join(GOLDEN, "photo_goldens.npz"))
        key = rec["file"].split(".")[0]
        cells = o.warp_cells(img, np.asarray(rec["corners"], np.float32))
        assert sha(cells) == rec["cells_sha256"] and (cells == P[key + "_cells"]).all()


# ---- approxPolyDP / arcLength / contourArea: properties that do not use the oracle ------------------------------------------
def _dist_to_line(p, a, b):
    ab = (b - a).astype(np.float64)
    n = np.hypot(*ab)
    if n == 0:
        return float(np.hypot(*(p - a)))
    return float(abs((p[0] - a[0]) * ab[1] - (p[1] - a[1]) * ab[0]) / n)


def _random_blob_contour(rs, size=160):
    """outer contour of a random smooth blob, by the product's own border following"""
    import sudoku_vision_amd as sva
    yy, xx = np.mgrid[0:size, 0:size]
    img = np.zeros((size, size), bool)
    for _ in range(rs.randint(2, 6)):
        cx, cy, r = rs.uniform(40, size - 40), rs.uniform(40, size - 40), rs.uniform(12, 38)
        img |= (xx - cx) ** 2 / rs.uniform(0.5, 1.5) + (yy - cy) ** 2 < r * r
    cs = sva.host.find_contours((img * 255).astype(np.uint8))
    return max(cs, key=len).reshape(-1, 2)


def _check_subsequence(out, src):
    """out is a subsequence of src in cyclic order (a closed curve may start anywhere)"""
    idx = []
    for p in out:
        hits = np.nonzero((src == p).all(1))[0]
        assert len(hits) >= 1
        idx.append(hits)
    # some rotation of the chosen indices must be increasing: greedy from each candidate start
    n = len(src)
    for s0 in idx[0]:
        cur, ok = s0, True
        for k in range(1, len(idx)):
            nxt = [h for h in ((idx[k] - s0) % n) if h > (cur - s0) % n]
            if not nxt:
                ok = False
                break
            cur = s0 + min(nxt)
        if ok:
            return True
    return False


@pytest.mark.parametrize("seed", range(12))
def test_approx_poly_dp_properties(host, seed):
    """cv2.approxPolyDP on a closed contour (cv/grid.py:24-34), checked without the oracle:
       * the result's vertices are input vertices, in the input's cyclic order;
       * every input vertex lies within (1 + sqrt(1/2)) * eps of the line through the two output vertices that bracket it
         (Douglas-Peucker keeps eps to the chord; the final pass may drop a vertex that is within eps/sqrt(2) of its neighbours'
         chord, which is where the second term comes from);
       * translating the input translates the output; scaling coordinates and eps by 2 or 4 scales it (all arithmetic is on
         integer differences, exact under powers of two);
       * growing eps never yields more vertices."""
    rs = np.random.RandomState(seed)
    c = _random_blob_contour(rs)
    per = host.arc_length(c, True)
    prev = None
    for ratio in (0.005, 0.02, 0.05):
        eps = ratio * per
        out = host.approx_poly_dp(c, eps, True).reshape(-1, 2)
        assert 1 <= len(out) <= len(c)
        assert _check_subsequence(out, c)
        if len(out) >= 2:
            # bracket every input vertex between consecutive output vertices
            pos = []
            start = int(np.nonzero((c == out[0]).all(1))[0][0])
            rolled = np.roll(c, -start, axis=0)
            j = 0
            for p in out:
                while not (rolled[j] == p).all():
                    j += 1
                pos.append(j)
            pos.append(len(c))
            bound = (1 + np.sqrt(0.5)) * eps + 1e-9
            for k in range(len(out)):
                a, b = out[k], out[(k + 1) % len(out)]
                for v in rolled[pos[k]:pos[k + 1]]:
                    assert _dist_to_line(v, a, b) <= bound
        t = np.array([37, -11])
        assert (host.approx_poly_dp(c + t, eps, True).reshape(-1, 2) == out + t).all()
        for k in (2, 4):
            assert (host.approx_poly_dp(c * k, eps * k, True).reshape(-1, 2) == out * k).all()
        if prev is not None:
            assert len(out) <= prev
        prev = len(out)


@pytest.mark.parametrize("seed", range(10))
def test_rasterised_quad_returns_its_corners(host, seed):
    """A filled convex quadrilateral, rasterised (the staircase is the noise): find_contours -> approxPolyDP(0.02 * perimeter)
    returns 4 vertices, each within eps of a true corner, and contourArea / arcLength agree with the polygon's area and
    perimeter to rasterisation accuracy -- the whole of find_grid_contour's geometry (cv/grid.py:24-34,53-69) against ground
    truth instead of against the oracle."""
    rs = np.random.RandomState(100 + seed)
    S = 400
    base = np.array([[80, 70], [320, 60], [330, 310], [70, 330]], np.float64)
    q = base + rs.uniform(-35, 35, (4, 2))
    yy, xx = np.mgrid[0:S, 0:S]
    inside = np.ones((S, S), bool)
    for i in range(4):
        a, b = q[i], q[(i + 1) % 4]
        inside &= (b[0] - a[0]) * (yy - a[1]) - (b[1] - a[1]) * (xx - a[0]) >= 0
    img = (inside * 255).astype(np.uint8)
    cs = host.find_contours(img)
    assert len(cs) == 1
    c = cs[0].reshape(-1, 2)
    per = host.arc_length(c, True)
    true_per = sum(np.hypot(*(q[(i + 1) % 4] - q[i])) for i in range(4))
    true_area = 0.5 * abs(sum(q[i][0] * q[(i + 1) % 4][1] - q[(i + 1) % 4][0] * q[i][1] for i in range(4)))
    assert abs(host.contour_area(c) - true_area) <= 0.02 * true_area
    assert true_per * 0.98 <= per <= true_per * 1.20                     # a chain-code perimeter over-estimates slanted edges (at most sqrt(2))
    eps = 0.02 * per
    out = host.approx_poly_dp(c, eps, True).reshape(-1, 2)
    assert len(out) == 4
    for p in out:
        assert min(np.hypot(*(p - qq)) for qq in q) <= eps
    got = host.find_grid_corners(img)
    assert got is not None and sorted(map(tuple, got.tolist())) == sorted(map(tuple, out.tolist()))


def test_arc_length_and_area_known_values(host):
    """Closed forms: axis-aligned rectangle w x h traced on pixel centres has perimeter 2(w+h) and area w*h; a diamond with
    diagonal steps has perimeter 4*r*sqrt(2) (float32 accumulation in OpenCV: tolerance 1e-3) and area 2 r^2."""
    rect = np.array([[10, 10], [10, 50], [90, 50], [90, 10]])
    assert host.arc_length(rect, True) == pytest.approx(2 * (80 + 40), abs=1e-9) and host.contour_area(rect) == 80 * 40
    assert host.arc_length(rect, False) == pytest.approx(40 + 80 + 40, abs=1e-9)
    r = 25
    dia = np.array([[50, 50 - r], [50 - r, 50], [50, 50 + r], [50 + r, 50]])
    assert host.arc_length(dia, True) == pytest.approx(4 * r * np.sqrt(2), abs=1e-3) and host.contour_area(dia) == 2 * r * r
