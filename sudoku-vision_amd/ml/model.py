"""MI355X drop-in for the reference's ml/model.py: `DigitCNN` keeps the attributes, state_dict keys
and call protocol pipeline/run.py:98-143 relies on; forward() runs the hand-written HIP kernels of
csrc/k3_cnn.hip (fp32, MFMA implicit-GEMM conv2 + fc1).  Inference only, GPU only."""
import os
import sys

import torch
import torch.nn as nn

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from _bootstrap import package  # noqa: E402
sys.path.pop(0)
_rt = package().runtime


class DigitCNN(nn.Module):
    """Simple CNN for digit classification (0-9, where 0 = empty) -- reference ml/model.py:19-42."""

    def __init__(self, num_classes: int = 10):
        super().__init__()
        if num_classes != 10:
            raise NotImplementedError("the HIP forward is specialised for the reference's 10 classes")
        self.conv1 = nn.Conv2d(1, 32, kernel_size=3, padding=1)
        self.conv2 = nn.Conv2d(32, 64, kernel_size=3, padding=1)
        self.pool = nn.MaxPool2d(2, 2)
        self.fc1 = nn.Linear(64 * 7 * 7, 128)
        self.dropout = nn.Dropout(0.5)
        self.fc2 = nn.Linear(128, num_classes)

    def _weights_key(self):
        return tuple((p.data_ptr(), p._version) for p in self.parameters())

    def forward(self, x: torch.Tensor) -> torch.Tensor:
        if self.training:
            raise NotImplementedError("DigitCNN (MI355X): inference only -- call .eval() (dropout is identity in eval mode)")
        if not x.is_cuda:
            raise RuntimeError("DigitCNN (MI355X): input must be a CUDA tensor; there is no CPU fallback")
        if x.dim() != 4 or tuple(x.shape[1:]) != (1, 28, 28):
            raise ValueError(f"expected input of shape (batch, 1, 28, 28), got {tuple(x.shape)}")
        ctx = _rt.default_context(x.device)
        key = (id(self), self._weights_key())
        if ctx._weights_key != key:
            ctx.load_state_dict(self.state_dict(), key=key)
        return ctx.cnn_forward(x.to(torch.float32).contiguous())


def count_parameters(model: nn.Module) -> int:
    """Count trainable parameters (reference ml/model.py:45-47)."""
    return sum(p.numel() for p in model.parameters() if p.requires_grad)
