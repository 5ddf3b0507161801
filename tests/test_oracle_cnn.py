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


def _h2(x, scale=1.0):
    """x * scale as an f16 pair (hi, lo): hi = f16(x), lo = f16(x - hi), both returned as float64"""
    x = np.asarray(x, np.float32) * np.float32(scale)
    hi = x.astype(np.float16).astype(np.float32)
    lo = (x - hi).astype(np.float16).astype(np.float32)
    return hi.astype(np.float64), lo.astype(np.float64)


def test_f16_pair_arithmetic_is_f32_grade(golden_dir):
    """The default GPU kernels (csrc/k3_cnn_h2.hip) carry each f32 operand of conv2 and fc1 as an f16 pair hi + lo (weights
    pre-scaled by a power of two) and form a product as ah*wh + ah*wl + al*wh.  This simulates exactly that operand treatment
    (exact products, exact sums) on the golden inputs with the trained weights and checks the claim the kernel's header makes:
    its logits are as close to an exact (float64) evaluation of the model as PyTorch-CPU's own f32 forward is -- i.e. what the
    scheme gives up (operands cut to 22 bits, the lo*lo term) is below f32's own rounding noise, two orders inside the 1e-4
    contract."""
    import torch.nn.functional as F
    g = np.load(os.path.join(golden_dir, "cnn_coreml_fp16.npz"))
    sd = _coreml_sd(g)
    x = torch.from_numpy(cnn_oracle.golden_inputs(int(g["x_seed"]), 162))

    def head(feats64, mode):
        w1 = sd["fc1.weight"].numpy()
        if mode == "exact":
            h = feats64 @ torch.from_numpy(w1.astype(np.float64)).T
        else:
            e = 13 - int(np.floor(np.log2(np.abs(w1).max())))
            ah, al = _h2(feats64.numpy().astype(np.float32))
            wh, wl = _h2(w1, 2.0 ** e)
            h = (torch.from_numpy(ah) @ torch.from_numpy(wh).T + torch.from_numpy(ah) @ torch.from_numpy(wl).T
                 + torch.from_numpy(al) @ torch.from_numpy(wh).T) * 2.0 ** -e
        h = F.relu(h + sd["fc1.bias"].double()).float()
        return (h @ sd["fc2.weight"].T + sd["fc2.bias"]).numpy()

    with torch.no_grad():
        c1 = F.max_pool2d(F.relu(F.conv2d(x, sd["conv1.weight"], sd["conv1.bias"], padding=1)), 2, 2)      # f32, as the kernel
        w2 = sd["conv2.weight"].numpy()
        y_exact = F.conv2d(c1.double(), sd["conv2.weight"].double(), sd["conv2.bias"].double(), padding=1)
        e = 13 - int(np.floor(np.log2(np.abs(w2).max())))
        ah, al = (torch.from_numpy(t) for t in _h2(c1.numpy()))
        wh, wl = (torch.from_numpy(t) for t in _h2(w2, 2.0 ** e))
        y_pair = (F.conv2d(ah, wh, None, padding=1) + F.conv2d(ah, wl, None, padding=1) + F.conv2d(al, wh, None, padding=1)) * 2.0 ** -e
        y_pair = y_pair + sd["conv2.bias"].double().view(1, -1, 1, 1)
        f_exact = F.max_pool2d(F.relu(y_exact), 2, 2).reshape(162, -1)
        f_pair = F.max_pool2d(F.relu(y_pair), 2, 2).float().double().reshape(162, -1)                    # features are stored as f32
        exact = head(f_exact, "exact")
        pair = head(f_pair, "pair")
    torch_f32 = cnn_oracle.forward(sd, x).numpy()
    err_pair, err_torch = np.abs(pair - exact).max(), np.abs(torch_f32 - exact).max()
    assert err_pair <= 5e-6 and err_pair <= 2 * err_torch + 1e-7, (err_pair, err_torch)
    assert np.abs(pair - torch_f32).max() <= 1e-5
    assert (pair.argmax(1) == g["digits"]).all()
