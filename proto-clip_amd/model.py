"""Query adapters with the reference's constructor signatures and state-dict key names
(reference model.py:12-95; shipped checkpoints pin the names) whose forward runs the fused
gfx950 kernels of csrc/pclip_adapter.hip instead of eager conv/LayerNorm modules.

`nn.Conv2d` / `nn.LayerNorm` objects are kept purely as parameter containers so that
`.parameters()`, `.state_dict()`, `.load_state_dict()`, `.half()`, `.cuda()` behave exactly as in the
reference (including conv-2x's unused conv2/bn2 parameters).  Under autograd (grad mode on, trainable parameters:
the reference's training loop, main.py:267-309) `forward` records a node whose backward runs the explicit backward kernels
(proto_clip_amd/autograd.py); under torch.no_grad() it is the fused inference kernel alone."""
import math

import torch
import torch.nn as nn

from . import autograd as pag
from . import ops
from ._lib import PclipError


def _needs_tape(x, module):
    """True when the call must record an autograd node: grad mode on and the input or a parameter requires grad — the reference's
    training loop (`adapter(zq_imgs).float()` ... `train_loss.backward()`, main.py:267-309).  Under torch.no_grad() — every
    evaluation path — the fused inference kernel runs alone."""
    return torch.is_grad_enabled() and (x.requires_grad or any(p.requires_grad for p in module.parameters()))


class Adapter(nn.Module):
    """conv-2x / conv-3x feature adapter (reference model.py:12-78).  No ReLU is applied (the reference
    constructs one but never calls it, model.py:47 vs 49-78)."""

    def __init__(self, c_in, c_type, width=16, dtype=None):
        super().__init__()
        if width != 16:
            raise PclipError("the gfx950 adapter kernel is specialised for width=16 (the reference's value)")
        if c_type not in ("conv-3x", "conv-2x"):
            raise PclipError(f"unknown adapter type {c_type!r}")
        self.c_in = c_in
        self.c_type = c_type
        size = int(math.ceil(math.sqrt(self.c_in)))
        self.conv1 = nn.Conv2d(1, width, kernel_size=1, stride=1, bias=False, dtype=dtype)
        self.bn1 = nn.LayerNorm([width, size, size], dtype=dtype)
        self.conv2 = nn.Conv2d(width, width, kernel_size=3, stride=1, padding=1, bias=False, dtype=dtype)
        self.bn2 = nn.LayerNorm([width, size, size], dtype=dtype)
        self.conv3 = nn.Conv2d(width, 1, kernel_size=1, stride=1, bias=False, dtype=dtype)
        self.bn3 = nn.LayerNorm([1, size, size], dtype=dtype)
        self.relu = nn.ReLU(inplace=True)   # parity with the reference's attribute list; never applied

    def forward(self, x, l2norm_out: bool = False):
        """x [B, c_in] fp16 -> [B, c_in] fp16.  l2norm_out=True additionally fuses the row normalise
        that every caller applies next (main.py:408-409) — an extension, default off."""
        if _needs_tape(x, self):
            if l2norm_out:
                raise PclipError("l2norm_out is an inference-only fusion; under autograd normalise with torch ops as the reference does")
            return pag.AdapterConvFn.apply(x, self.c_type == "conv-3x", self.conv1.weight, self.bn1.weight, self.bn1.bias,
                                           self.conv2.weight, self.bn2.weight, self.bn2.bias, self.conv3.weight, self.bn3.weight,
                                           self.bn3.bias)
        return ops.adapter_conv(
            x, self.c_type == "conv-3x", self.conv1.weight, self.bn1.weight, self.bn1.bias, self.conv2.weight,
            self.bn2.weight, self.bn2.bias, self.conv3.weight, self.bn3.weight, self.bn3.bias, l2norm_out=l2norm_out)


class Adapter_FC(nn.Module):
    """Linear -> LN -> Linear -> LN with a 0.2/0.8 residual blend (reference model.py:81-95)."""

    def __init__(self, c_in, reduction=4, dtype=None):
        super().__init__()
        self.fc = nn.Sequential(
            nn.Linear(c_in, c_in // reduction, bias=False, dtype=dtype),
            nn.LayerNorm(c_in // reduction, dtype=dtype),
            nn.Linear(c_in // reduction, c_in, bias=False, dtype=dtype),
            nn.LayerNorm(c_in, dtype=dtype),
        )

    def forward(self, image_features, l2norm_out: bool = False):
        fc = self.fc
        if _needs_tape(image_features, self):
            if l2norm_out:
                raise PclipError("l2norm_out is an inference-only fusion; under autograd normalise with torch ops as the reference does")
            return pag.AdapterFcFn.apply(image_features, fc[0].weight, fc[1].weight, fc[1].bias, fc[2].weight, fc[3].weight, fc[3].bias)
        return ops.adapter_fc(image_features, fc[0].weight, fc[1].weight, fc[1].bias, fc[2].weight, fc[3].weight,
                              fc[3].bias, ratio=0.2, l2norm_out=l2norm_out)
