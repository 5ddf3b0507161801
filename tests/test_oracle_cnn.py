"""CPU: the torch restatement of DigitCNN (oracle/cnn_oracle.py) against goldens captured from the
reference's own ml/model.py (tests/golden/make_goldens.py)."""
import os

import numpy as np
import torch

import cnn_oracle


def _coreml_sd(g):
    return {k: torch.from_numpy(g[k.replace(".", "_")].astype(np.float32)) for k in cnn_oracle.KEYS}


def test_random_weights_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "cnn_random_seed1234.npz"))
    sd = cnn_oracle.random_state_dict(int(g["seed"]))
    x = cnn_oracle.golden_inputs(int(g["x_seed"]), 81)
    logits = cnn_oracle.forward(sd, x).numpy()
    assert np.abs(logits - g["logits"]).max() <= 1e-5
    assert (logits.argmax(1) == g["digits"]).all()


def test_trained_weights_golden(golden_dir):
    g = np.load(os.path.join(golden_dir, "cnn_coreml_fp16.npz"))
    sd = _coreml_sd(g)
    assert sum(v.numel() for v in sd.values()) == 421642
    x = cnn_oracle.golden_inputs(int(g["x_seed"]), 162)
    logits, digits, conf = cnn_oracle.predict(sd, x)
    assert np.abs(logits.numpy() - g["logits"]).max() <= 1e-4
    assert (digits.numpy() == g["digits"]).all()
    assert ((conf > 0) & (conf <= 1)).all()


def test_batch_independence():
    sd = cnn_oracle.random_state_dict(1)
    x = cnn_oracle.golden_inputs(2, 16)
    a = cnn_oracle.forward(sd, x).numpy()
    b = np.concatenate([cnn_oracle.forward(sd, x[i:i + 1]).numpy() for i in range(16)])
    assert np.abs(a - b).max() <= 1e-5
