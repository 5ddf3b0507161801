#!/usr/bin/env python3
"""(Not collected by pytest; it lives here because it uses the oracle.)  One-off differential fuzz of the HIP path against the oracle / Pillow on random shapes (run on a GPU box):
K1 on random H x W noise and structured images, K2 on random quadrilaterals (some partly outside the frame), the JPEG
front end on random sizes / qualities / sub-samplings.  Prints a summary; exits non-zero on the first mismatch."""
import io
import os
import sys

import numpy as np
import torch
from PIL import Image

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import sudoku_vision_amd as sva  # noqa: E402
import sv_oracle as o  # noqa: E402

ctx = sva.default_context()
xctx = sva.Context(library=sva._native.lib_xcheck())                         # the test-only library: K1's matrix-pipe second implementation
rs = np.random.RandomState(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
n1 = n2 = n3 = 0
for it in range(60):                                                      # K1
    H, W = int(rs.randint(1, 260)), int(rs.randint(1, 420))
    if it % 3 == 0:
        W = (W + 3) // 4 * 4                                              # the marching kernel's aligned path
    kind = it % 4
    if kind == 0:
        img = rs.randint(0, 256, (H, W, 3))
    elif kind == 1:
        img = np.full((H, W, 3), rs.randint(0, 256)) + rs.randint(-3, 4, (H, W, 3))
    elif kind == 2:
        yy, xx = np.mgrid[0:H, 0:W]
        img = (128 + 100 * np.sin(xx / 3.0) * np.cos(yy / 5.0))[..., None] + rs.randint(-10, 11, (H, W, 3))
    else:
        img = (rs.randint(0, 2, (H // 6 + 1, W // 6 + 1)).repeat(6, 0).repeat(6, 1)[:H, :W] * 200 + 20)[..., None] + rs.randint(-5, 6, (H, W, 3))
    img = np.clip(img, 0, 255).astype(np.uint8)
    got = ctx.preprocess(torch.from_numpy(img[None]).cuda())[0].cpu().numpy()
    if not (got == o.preprocess_for_grid_detection(img)).all():
        print("K1 MISMATCH", H, W, kind); sys.exit(1)
    n1 += 1
n1b = 0
for it in range(40):                                                      # K1's other forms: bit image, in-place bit despeckle, matrix pipe, fused launch
    H, W = int(rs.randint(16, 300)), 16 * int(rs.randint(1, 30))
    n = int(rs.randint(1, 4))
    img = rs.randint(0, 256, (n, H, W, 3))
    if it % 2:
        img = (img * rs.uniform(0.02, 0.3) + rs.randint(0, 180)).astype(np.int64)     # low contrast: many means near the decision boundary
    d = torch.from_numpy(np.clip(img, 0, 255).astype(np.uint8)).cuda()
    ref = ctx.preprocess(d)
    want = np.stack([o.preprocess_for_grid_detection(f) for f in d.cpu().numpy()])
    if not (ref.cpu().numpy() == want).all() or not torch.equal(xctx.preprocess_mm(d), ref):
        print("K1 matrix-pipe MISMATCH", n, H, W); sys.exit(1)
    if W % 32 == 0:
        bits = ctx.preprocess_bits(d)
        if not np.array_equal(bits.cpu().numpy().view(np.uint32), np.packbits(want > 0, axis=2, bitorder="little").view(np.uint32).reshape(n, H, W // 32)):
            print("K1 bit image MISMATCH", n, H, W); sys.exit(1)
        packed = torch.empty_like(bits)
        ctx.despeckle(ref, out=torch.empty_like(ref), packed=packed)
        if not torch.equal(ctx.despeckle_bits(bits), packed):
            print("despeckle bits MISMATCH", n, H, W); sys.exit(1)
    side = 0.8 * min(H, W)
    corners = np.stack([np.array([[W / 2 - side / 2, H / 2 - side / 2], [W / 2 + side / 2, H / 2 - side / 2], [W / 2 + side / 2, H / 2 + side / 2],
                                  [W / 2 - side / 2, H / 2 + side / 2]]) + rs.uniform(-0.1, 0.1, (4, 2)) * side for _ in range(n)]).astype(np.float32)
    minv = ctx.minv_to_device(sva.Context.corners_to_minv(corners).reshape(n, 9))
    b2, c2 = ctx.preprocess_and_warp_cells(d, minv)
    if not torch.equal(b2, ref) or not torch.equal(c2, ctx.warp_cells(d, minv)):
        print("fused K1+K2 MISMATCH", n, H, W); sys.exit(1)
    n1b += 1
for it in range(60):                                                      # K2
    H, W = int(rs.randint(60, 500)), int(rs.randint(60, 700))
    img = rs.randint(0, 256, (H, W, 3)).astype(np.uint8)
    side = rs.uniform(0.3, 1.3) * min(H, W)
    cx, cy = rs.uniform(0.2, 0.8) * W, rs.uniform(0.2, 0.8) * H
    base = np.array([[cx - side / 2, cy - side / 2], [cx + side / 2, cy - side / 2], [cx + side / 2, cy + side / 2], [cx - side / 2, cy + side / 2]])
    corners = np.round(base + rs.uniform(-0.15, 0.15, (4, 2)) * side).astype(np.float32)
    minv = ctx.minv_to_device(sva.Context.corners_to_minv(corners[None]))
    d = torch.from_numpy(img).cuda()
    cells = ctx.warp_cells(d[None], minv)[0].cpu().numpy()
    if not (cells == o.warp_cells(img, corners)).all():
        print("K2 cells MISMATCH", H, W, corners.tolist()); sys.exit(1)
    warped = ctx.warp_perspective(d, minv[0], 450).cpu().numpy()
    if not (warped == o.warp_perspective(img, corners)).all():
        print("K2 warp MISMATCH", H, W, corners.tolist()); sys.exit(1)
    n2 += 1
import cnn_oracle  # noqa: E402
n4 = 0
for it in range(12):                                                      # K3: weights of very different magnitudes through the f16-pair kernels
    sd = cnn_oracle.random_state_dict(1000 + it + 100 * (int(sys.argv[1]) if len(sys.argv) > 1 else 0))
    for k in sd:
        if k.endswith("weight"):
            sd[k] = sd[k] * float(10 ** rs.uniform(-1.5, 1.0))
    ctx.load_state_dict(sd)
    B = int(rs.randint(1, 700))
    x = torch.from_numpy(rs.uniform(-1, 1, (B, 1, 28, 28)).astype(np.float32))
    want = cnn_oracle.forward(sd, x).numpy()
    got = ctx.cnn_forward(x.cuda()).cpu().numpy()
    tol = max(1e-4, 2e-5 * float(np.abs(want).max()))
    if np.abs(got - want).max() > tol:
        print("K3 MISMATCH", it, B, np.abs(got - want).max(), tol); sys.exit(1)
    cells = torch.from_numpy(rs.randint(0, 256, (B, 28, 28)).astype(np.uint8))
    want = cnn_oracle.forward(sd, torch.from_numpy(o.cells_to_input(cells.numpy())[:, None])).numpy()
    got = ctx.cnn_forward(cells.cuda()).cpu().numpy()
    if np.abs(got - want).max() > max(1e-4, 2e-5 * float(np.abs(want).max())):
        print("K3 (cells) MISMATCH", it, B, np.abs(got - want).max()); sys.exit(1)
    n4 += 1
for it in range(120):                                                     # JPEG
    H, W = int(rs.randint(1, 200)), int(rs.randint(1, 300))
    gray = it % 7 == 0
    yy, xx = np.mgrid[0:H, 0:W]
    img = np.stack([128 + 90 * np.sin(xx / (3.0 + c) + it) * np.cos(yy / (4.0 + c)) + rs.randint(-25, 26, (H, W)) for c in range(1 if gray else 3)], -1)
    img = np.clip(img, 0, 255).astype(np.uint8)
    b = io.BytesIO()
    kw = dict(quality=int(rs.randint(1, 101)))
    if not gray:
        kw["subsampling"] = int(rs.randint(0, 3))
    if it % 5 == 0:
        kw["restart_marker_blocks"] = int(rs.randint(1, 9))
    if it % 4 == 0:
        kw["optimize"] = True
    Image.fromarray(img[..., 0] if gray else img).save(b, "JPEG", **kw)
    data = b.getvalue()
    want = np.asarray(Image.open(io.BytesIO(data)).convert("RGB"))[..., ::-1]
    for dense in (False, True):
        got = ctx.imdecode(data, threads=1 + it % 3, dense=dense).cpu().numpy()
        if not (got == want).all():
            print("JPEG MISMATCH", H, W, kw, dense); sys.exit(1)
    if not (o.imdecode(data) == want).all():
        print("JPEG ORACLE MISMATCH", H, W, kw); sys.exit(1)
    n3 += 1
print(f"fuzz ok: K1 {n1} shapes + {n1b} in its other forms (bits, matrix pipe, fused launch), K2 {n2} quads, K3 {n4} weight sets, JPEG {n3} files (seed {sys.argv[1] if len(sys.argv) > 1 else 0})")
