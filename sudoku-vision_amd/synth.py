"""Synthetic camera frames of a sudoku sheet: seeded, generated with torch ops on
whatever device is asked for (GPU for bench.py, CPU for the small oracle tests).

Each frame: paper-like background (per-frame level 170-230, a low-frequency illumination gradient,
N(0,4) noise), a dark-lined 9x9 grid whose corners are a random perspective jitter (+-8 % of the side)
of a square of 0.76*H centred in the frame (always fully inside it), ~30 of 81 cells filled with 5x7 block glyphs.
Returns the generator's ground truth corners (TL,TR,BR,BL) so the device-only metric does not depend
on the host corner search.
"""
import numpy as np
import torch

_FONT = {
    1: ["..#..", ".##..", "..#..", "..#..", "..#..", "..#..", ".###."],
    2: [".###.", "#...#", "....#", "...#.", "..#..", ".#...", "#####"],
    3: [".###.", "#...#", "....#", "..##.", "....#", "#...#", ".###."],
    4: ["...#.", "..##.", ".#.#.", "#..#.", "#####", "...#.", "...#."],
    5: ["#####", "#....", "####.", "....#", "....#", "#...#", ".###."],
    6: [".###.", "#....", "#....", "####.", "#...#", "#...#", ".###."],
    7: ["#####", "....#", "...#.", "..#..", "..#..", ".#...", ".#..."],
    8: [".###.", "#...#", "#...#", ".###.", "#...#", "#...#", ".###."],
    9: [".###.", "#...#", "#...#", ".####", "....#", "....#", ".###."],
}


def _font_table():
    t = np.zeros((10, 7, 5), np.uint8)
    for d, rows in _FONT.items():
        for y, row in enumerate(rows):
            for x, ch in enumerate(row):
                t[d, y, x] = ch == "#"
    return t


def _homography(src, dst):
    """src,dst float64 [n,4,2] -> [n,3,3] with H*src ~ dst (numpy, host)."""
    n = src.shape[0]
    A = np.zeros((n, 8, 8))
    b = np.zeros((n, 8))
    for i in range(4):
        sx, sy, dx, dy = src[:, i, 0], src[:, i, 1], dst[:, i, 0], dst[:, i, 1]
        A[:, i, 0], A[:, i, 1], A[:, i, 2] = sx, sy, 1
        A[:, i, 6], A[:, i, 7] = -sx * dx, -sy * dx
        A[:, i + 4, 3], A[:, i + 4, 4], A[:, i + 4, 5] = sx, sy, 1
        A[:, i + 4, 6], A[:, i + 4, 7] = -sx * dy, -sy * dy
        b[:, i], b[:, i + 4] = dx, dy
    h = np.linalg.solve(A, b[..., None])[..., 0]
    return np.concatenate([h, np.ones((n, 1))], 1).reshape(n, 3, 3)


def synth_frames(n, H=1080, W=1920, seed=1234, device="cpu", chunk=8, noise="torch"):
    """-> frames u8 [n,H,W,3] (BGR, on `device`), corners f32 [n,4,2] (numpy), puzzle u8 [n,9,9] (numpy).
    noise="torch": N(0,4) / N(0,3) from the device's generator (fast; the values depend on the device and the torch build).
    noise="int": integer noise (sums of three uniform integers, same spread) from numpy's RandomState -- every operation that
    follows is a single correctly-rounded IEEE operation, so with device="cpu" the frames are bit-identical on every machine;
    this is what the committed golden checksums (tests/golden/make_cv_goldens.py) are made from."""
    if noise not in ("torch", "int"):
        raise ValueError("noise must be 'torch' or 'int'")
    rs = np.random.RandomState(seed)
    side = 0.76 * min(H, W)
    cx, cy = W / 2 + rs.uniform(-0.03, 0.03, n) * W, H / 2 + rs.uniform(-0.02, 0.02, n) * H
    base = np.stack([np.stack([cx - side / 2, cy - side / 2], 1), np.stack([cx + side / 2, cy - side / 2], 1),
                     np.stack([cx + side / 2, cy + side / 2], 1), np.stack([cx - side / 2, cy + side / 2], 1)], 1)
    corners = base + rs.uniform(-0.08, 0.08, (n, 4, 2)) * side
    corners = np.round(corners)  # the reference's corner search yields integer pixel corners
    grid_pts = np.tile(np.array([[0, 0], [9, 0], [9, 9], [0, 9]], np.float64), (n, 1, 1))
    Hm = _homography(corners.astype(np.float64), grid_pts)
    puzzle = np.where(rs.uniform(size=(n, 9, 9)) < 30 / 81, rs.randint(1, 10, (n, 9, 9)), 0).astype(np.uint8)
    level = rs.uniform(170, 230, n)
    grad = rs.uniform(-20, 20, (n, 2))
    tint = rs.uniform(-6, 6, (n, 3))

    dev = torch.device(device)
    gen = torch.Generator(device=dev)
    gen.manual_seed(seed)
    font = torch.from_numpy(_font_table()).to(dev)
    ys, xs = torch.meshgrid(torch.arange(H, device=dev, dtype=torch.float32), torch.arange(W, device=dev, dtype=torch.float32), indexing="ij")
    frames = torch.empty((n, H, W, 3), dtype=torch.uint8, device=dev)
    for s in range(0, n, chunk):
        e = min(n, s + chunk)
        m = e - s
        h = torch.from_numpy(Hm[s:e]).to(dev, torch.float32).reshape(m, 9, 1, 1)
        den = h[:, 6] * xs + h[:, 7] * ys + h[:, 8]
        u = (h[:, 0] * xs + h[:, 1] * ys + h[:, 2]) / den
        v = (h[:, 3] * xs + h[:, 4] * ys + h[:, 5]) / den
        inside = (u >= -0.05) & (u <= 9.05) & (v >= -0.05) & (v <= 9.05)
        ru, rv = torch.round(u), torch.round(v)
        wu = torch.where(torch.remainder(ru, 3) == 0, 0.05, 0.022)
        wv = torch.where(torch.remainder(rv, 3) == 0, 0.05, 0.022)
        line = ((u - ru).abs() < wu) | ((v - rv).abs() < wv)
        cu, cv_ = u.floor().clamp(0, 8).long(), v.floor().clamp(0, 8).long()
        fu, fv = u - cu, v - cv_
        pz = torch.from_numpy(puzzle[s:e]).to(dev).long()
        dig = pz.reshape(m, 81).gather(1, (cv_ * 9 + cu).reshape(m, -1)).reshape(m, H, W)
        bx = ((fu - 0.25) / 0.5 * 5).floor().long()
        by = ((fv - 0.15) / 0.7 * 7).floor().long()
        ing = (bx >= 0) & (bx < 5) & (by >= 0) & (by < 7)
        bit = font[dig, by.clamp(0, 6), bx.clamp(0, 4)].bool() & ing
        ink = inside & (line | bit)
        lv = torch.from_numpy(level[s:e]).to(dev, torch.float32).reshape(m, 1, 1)
        g = torch.from_numpy(grad[s:e]).to(dev, torch.float32)
        paper = lv + g[:, 0].reshape(m, 1, 1) * (xs / W - 0.5) + g[:, 1].reshape(m, 1, 1) * (ys / H - 0.5)
        if noise == "int":
            n1 = torch.from_numpy(rs.randint(-4, 5, (3, m, H, W)).sum(0).astype(np.float32)).to(dev)      # sd 4.5
            n2 = torch.from_numpy(rs.randint(-3, 4, (3, m, H, W)).sum(0).astype(np.float32)).to(dev)      # sd 3.5
        else:
            n1 = 4.0 * torch.randn((m, H, W), generator=gen, device=dev)
            n2 = 3.0 * torch.randn((m, H, W), generator=gen, device=dev)
        paper = paper + n1
        val = torch.where(ink, paper * 0.0 + 45.0 + n2, paper)
        t = torch.from_numpy(tint[s:e]).to(dev, torch.float32).reshape(m, 1, 1, 3)
        frames[s:e] = (val[..., None] + t).round().clamp(0, 255).to(torch.uint8)
    return frames, corners.astype(np.float32), puzzle


_CNN_KEYS = ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias", "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")
_CNN_SHAPES = ((32, 1, 3, 3), (32,), (64, 32, 3, 3), (64,), (128, 3136), (128,), (10, 128), (10,))


def random_state_dict(seed: int):
    """Random-init weights of the DigitCNN architecture (ml/model.py:22-32) for benchmarks: nn.Conv2d / nn.Linear default
    ranges, U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights and biases, drawn from numpy's RandomState so the values do not
    depend on the torch version.  state_dict keys and shapes as the reference's."""
    rs = np.random.RandomState(seed)
    sd, bound = {}, 1.0
    for key, shape in zip(_CNN_KEYS, _CNN_SHAPES):
        if key.endswith("weight"):
            bound = 1.0 / np.sqrt(float(np.prod(shape[1:])))
        sd[key] = torch.from_numpy(rs.uniform(-bound, bound, size=shape).astype(np.float32))
    return sd
