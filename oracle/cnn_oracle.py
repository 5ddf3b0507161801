"""CPU oracle for the digit CNN forward (row A11).  TEST INFRASTRUCTURE ONLY.

Restates /root/reference/ml/model.py:19-42 (DigitCNN) with plain torch.nn.functional calls on CPU
fp32 tensors, taking the weights as a state_dict-shaped mapping:
    conv1.weight [32,1,3,3]  conv1.bias [32]   conv2.weight [64,32,3,3]  conv2.bias [64]
    fc1.weight [128,3136]    fc1.bias [128]    fc2.weight [10,128]       fc2.bias [10]
Dropout (model.py:31,40) is identity in eval mode and is therefore absent here.

Pinned: tests/golden/cnn_*.npz hold logits produced by importing the reference module itself
(tests/golden/make_goldens.py); tests/test_oracle_cnn.py checks this restatement against them.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import this module.
"""
import numpy as np
import torch
import torch.nn.functional as F

KEYS = ("conv1.weight", "conv1.bias", "conv2.weight", "conv2.bias",
        "fc1.weight", "fc1.bias", "fc2.weight", "fc2.bias")
SHAPES = ((32, 1, 3, 3), (32,), (64, 32, 3, 3), (64,), (128, 3136), (128,), (10, 128), (10,))


def random_state_dict(seed: int):
    """Deterministic PyTorch-default-like init from numpy's RandomState (stable across versions):
    U(-1/sqrt(fan_in), 1/sqrt(fan_in)) for weights and biases (model.py uses nn.Conv2d/nn.Linear defaults)."""
    rs = np.random.RandomState(seed)
    sd = {}
    for key, shape in zip(KEYS, SHAPES):
        if key.endswith("weight"):
            fan_in = int(np.prod(shape[1:]))
        bound = 1.0 / np.sqrt(fan_in)
        sd[key] = torch.from_numpy(rs.uniform(-bound, bound, size=shape).astype(np.float32))
    return sd


def golden_inputs(seed, n):
    """n synthetic cells in the glue's value range [-1,1]: noisy background + a dark blob each."""
    rs = np.random.RandomState(seed)
    x = rs.uniform(-1, 1, size=(n, 1, 28, 28)).astype(np.float32)
    u8 = rs.randint(0, 256, size=(n, 1, 28, 28)).astype(np.uint8)     # exact glue values
    x[n // 2:] = ((255 - u8[n // 2:]).astype(np.float32) / np.float32(255.0) - np.float32(0.5)) / np.float32(0.5)
    return x


def forward(sd, x):
    """x: float32 [B,1,28,28] (tensor or ndarray) -> logits float32 [B,10] (tensor, CPU)."""
    x = torch.as_tensor(x, dtype=torch.float32, device="cpu")
    with torch.no_grad():
        x = F.max_pool2d(F.relu(F.conv2d(x, sd["conv1.weight"], sd["conv1.bias"], padding=1)), 2, 2)
        x = F.max_pool2d(F.relu(F.conv2d(x, sd["conv2.weight"], sd["conv2.bias"], padding=1)), 2, 2)
        x = x.reshape(x.size(0), -1)            # NCHW flatten: c*49 + y*7 + x
        x = F.relu(F.linear(x, sd["fc1.weight"], sd["fc1.bias"]))
        return F.linear(x, sd["fc2.weight"], sd["fc2.bias"])


def predict(sd, x):
    """Reference glue pipeline/run.py:139-143: pred = argmax(logits), conf = softmax(logits)[pred]."""
    logits = forward(sd, x)
    probs = torch.softmax(logits, dim=1)
    pred = logits.argmax(dim=1)
    conf = probs.gather(1, pred[:, None])[:, 0]
    return logits, pred.to(torch.uint8), conf
