#!/usr/bin/env python3
"""Generates tests/golden/cnn_*.npz by IMPORTING THE REFERENCE (ml/model.py) in the build container.

Run here only (the reference never travels):   python tests/golden/make_goldens.py
Outputs (data only -- inputs, weights that the reference tree ships as a data file, expected outputs):
  cnn_random_seed1234.npz  : seed, x_seed, logits of reference DigitCNN with numpy-seeded weights
  cnn_coreml_fp16.npz      : the trained DigitCNN weights the reference ships as fp16 in
                             ios/.../DigitClassifier.mlpackage/.../weight.bin (a data blob), the
                             inputs, and the reference module's logits with those weights (fp32 upcast)
"""
import os
import struct
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
sys.path.insert(0, "/root/reference/ml")

from model import DigitCNN, count_parameters  # the reference itself  # noqa: E402
import cnn_oracle  # noqa: E402

torch.set_num_threads(1)


def ref_logits(sd, x):
    m = DigitCNN()
    m.load_state_dict(sd)
    m.eval()
    with torch.no_grad():
        return m(torch.from_numpy(x)).numpy()


def coreml_weights():
    p = "/root/reference/ios/SudokuVision/Resources/DigitClassifier.mlpackage/Data/com.apple.CoreML/weights/weight.bin"
    b = open(p, "rb").read()
    off, out = 64, {}
    for key, shape in zip(cnn_oracle.KEYS, cnn_oracle.SHAPES):
        magic, dtype, size, dataoff = struct.unpack("<IIQQ", b[off:off + 24])
        assert magic == 0xDEADBEEF and dtype == 1 and size == 2 * int(np.prod(shape))
        out[key] = np.frombuffer(b, np.float16, int(np.prod(shape)), dataoff).reshape(shape).copy()
        off = (dataoff + size + 63) // 64 * 64
    return out


def main():
    assert count_parameters(DigitCNN()) == 421642
    # 1. seeded random weights
    seed, x_seed = 1234, 99
    sd = cnn_oracle.random_state_dict(seed)
    x = cnn_oracle.golden_inputs(x_seed, 81)
    logits = ref_logits(sd, x)
    np.savez_compressed(os.path.join(HERE, "cnn_random_seed1234.npz"), seed=seed, x_seed=x_seed,
                        logits=logits, digits=logits.argmax(1).astype(np.uint8))
    # 2. trained fp16 weights shipped in the reference tree
    w16 = coreml_weights()
    sd2 = {k: torch.from_numpy(v.astype(np.float32)) for k, v in w16.items()}
    x2 = cnn_oracle.golden_inputs(7, 162)
    logits2 = ref_logits(sd2, x2)
    np.savez_compressed(os.path.join(HERE, "cnn_coreml_fp16.npz"), x_seed=7, logits=logits2,
                        digits=logits2.argmax(1).astype(np.uint8), **{k.replace(".", "_"): v for k, v in w16.items()})
    # restatement vs reference, here and now
    for s, xx, ll in ((sd, x, logits), (sd2, x2, logits2)):
        d = np.abs(cnn_oracle.forward(s, xx).numpy() - ll).max()
        print("oracle vs reference max abs diff", d)
        assert d <= 1e-6


if __name__ == "__main__":
    main()
