#!/usr/bin/env python3
"""Generates tests/golden/cv_goldens.json + photo_goldens.npz: outputs of the CPU ORACLE (oracle/sv_oracle.c -- a restatement
of OpenCV's arithmetic, PARITY UNPINNED: cv2 is absent from this image and the reference's tests hold no cv2 output) for

  * seeded synthetic frames (fixture 2; synth_frames(..., device="cpu", noise="int"): bit-identical on every
    machine, and their SHA-256 is stored so that a test notices if that ever stops being true): SHA-256 of the K1 binary, of the homography bytes (fp64 Minv) and of the
    81 cells, plus the corners the oracle's contour search finds on that binary;
  * the reference's five photos data/test_images/sample_{1..5}.jpg (copied next to this file as data fixtures), PIL-decoded:
    SHA-256 of the K1 binary, the corners or null ("CV success rate on test images: 4/5", tests/test_integration.py:261), the
    u8[81,28,28] cells, and the digit indices + logits of run.py's glue (preprocess_cell, pipeline/run.py:73-95) + DigitCNN
    with the trained weights the reference ships (tests/golden/cnn_coreml_fp16.npz).

Why committed values and not only the live oracle: a change that edits oracle and kernel together would pass a live comparison;
it cannot pass these.  Both the CPU suite (oracle vs file) and the GPU suite (HIP path vs file) read them.

Run here (the reference tree supplies the photos):   python tests/golden/make_cv_goldens.py
"""
import hashlib
import json
import os
import shutil
import sys

import numpy as np
import torch
from PIL import Image

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, "oracle"))

import cnn_oracle  # noqa: E402
import sv_oracle as o  # noqa: E402
from sudoku_vision_amd.synth import synth_frames  # noqa: E402

SYNTH = [(270, 480, 3, 2), (540, 960, 4, 2), (1080, 1920, 1234, 3), (97, 131, 97, 1)]      # (H, W, seed, frames)
PHOTOS = [f"sample_{i}.jpg" for i in range(1, 6)]


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


def main():
    torch.set_num_threads(1)
    out = {"synthetic": [], "photos": []}
    for H, W, seed, n in SYNTH:
        frames, corners, _ = synth_frames(n, H, W, seed=seed, noise="int")
        for i in range(n):
            f = frames[i].numpy()
            binary = o.preprocess_for_grid_detection(f)
            found = o.find_grid_contour(binary)
            out["synthetic"].append({"H": H, "W": W, "seed": seed, "n": n, "index": i, "frame_sha256": sha(f), "binary_sha256": sha(binary),
                                     "minv_sha256": sha(o.corners_to_minv(corners[i])), "cells_sha256": sha(o.warp_cells(f, corners[i])),
                                     "found_corners": None if found is None else np.asarray(found).reshape(4, 2).tolist()})
    g2 = np.load(os.path.join(HERE, "cnn_coreml_fp16.npz"))
    sd = {k: torch.from_numpy(g2[k.replace(".", "_")].astype(np.float32)) for k in cnn_oracle.KEYS}
    arrays = {}
    for name in PHOTOS:
        dst = os.path.join(HERE, name)
        if not os.path.exists(dst):
            shutil.copyfile(os.path.join("/root/reference/data/test_images", name), dst)
        img = np.asarray(Image.open(dst).convert("RGB"))[..., ::-1].copy()
        binary = o.preprocess_for_grid_detection(img)
        found = o.find_grid_contour(binary)
        rec = {"file": name, "shape": list(img.shape), "frame_sha256": sha(img), "binary_sha256": sha(binary),
               "corners": None if found is None else np.asarray(found).reshape(4, 2).tolist()}
        if found is not None:
            c = np.asarray(found, np.float32).reshape(4, 2)
            cells = o.warp_cells(img, c)
            logits, digits, _ = cnn_oracle.predict(sd, o.cells_to_input(o.preprocess_cells(cells))[:, None])
            rec["cells_sha256"], rec["minv_sha256"] = sha(cells), sha(o.corners_to_minv(c))
            key = name.split(".")[0]
            arrays[key + "_cells"], arrays[key + "_digits"], arrays[key + "_logits"] = cells, digits.numpy(), logits.numpy()
        out["photos"].append(rec)
    assert sum(r["corners"] is not None for r in out["photos"]) == 4           # the reference's own note: 4/5
    with open(os.path.join(HERE, "cv_goldens.json"), "w") as f:
        json.dump(out, f, indent=1)
    np.savez_compressed(os.path.join(HERE, "photo_goldens.npz"), **arrays)
    print("wrote cv_goldens.json, photo_goldens.npz;", [(r["file"], r["corners"] is not None) for r in out["photos"]])


if __name__ == "__main__":
    main()
