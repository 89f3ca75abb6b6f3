#!/usr/bin/env python3
"""The narrow 3x3 convolutions of the ModifiedResNet tower: csrc/pclip_conv_strip.hip against the implicit-GEMM kernel it replaces (us per launch, TFLOP/s)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import ops

SHAPES = [("stem conv2", 112, 112, 32, 32), ("stem conv3", 112, 112, 32, 64), ("layer1 conv2", 56, 56, 64, 64)]


def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


for B in (256, 1024):
    for name, H, W, Cin, Cout in SHAPES:
        g = torch.Generator(device="cuda").manual_seed(1)
        x = (torch.randn(B * H * W, Cin, device="cuda", generator=g) * 0.7).half()
        w = (torch.randn(Cout, (9 * Cin + 63) // 64 * 64, device="cuda", generator=g) * (9 * Cin) ** -0.5).half()
        if 9 * Cin % 64: w[:, 9 * Cin:] = 0
        sc, sh = torch.ones(Cout, device="cuda"), torch.zeros(Cout, device="cuda")
        fl = 2.0 * B * H * W * Cout * 9 * Cin
        with ops.conv_strip(False):
            old = t(lambda: ops.conv3x3_bn(x, w, sc, sh, B, H, W, Cin))
        new = t(lambda: ops.conv3x3_bn(x, w, sc, sh, B, H, W, Cin))
        print(f"B={B:5d} {name:13s} {Cin}->{Cout} {H}x{W}: implicit GEMM {old:7.1f} us ({fl / old * 1e-6:6.0f} TFLOP/s)   strip kernel {new:7.1f} us ({fl / new * 1e-6:6.0f} TFLOP/s)   x{old / new:.2f}", flush=True)
