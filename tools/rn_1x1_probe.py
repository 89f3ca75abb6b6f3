#!/usr/bin/env python3
"""The 1x1 convolutions of RN50 at 1024 images (BatchNorm-epilogue GEMMs): us per launch and GB/s of algorithmic traffic per layer shape.
PCLIP_GEMM_BN_CFG=<cfg> forces a tile configuration (A/B of the cost model)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import ops
B = int(os.environ.get("IMAGES", "1024"))
# (name, pixels per image, K, N, residual)
SHAPES = [("l1 conv1 256->64", 3136, 256, 64, False), ("l1 conv3 64->256 +id", 3136, 64, 256, True), ("l2 conv1 512->128", 784, 512, 128, False),
          ("l2 conv3 128->512 +id", 784, 128, 512, True), ("l3 conv1 1024->256", 196, 1024, 256, False), ("l3 conv3 256->1024 +id", 196, 256, 1024, True),
          ("l4 conv3 512->2048 +id", 49, 512, 2048, True), ("l2.0 downsample 256->512", 784, 256, 512, False)]


def t(fn, n=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e3


tot = 0.0
for name, px, K, N, res in SHAPES:
    M = B * px
    a = (torch.randn(M, K, device="cuda") * 0.5).half()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    sc, sh = torch.ones(N, device="cuda"), torch.zeros(N, device="cuda")
    r = torch.randn(M, N, device="cuda").half() if res else None
    us = t((lambda: ops.gemm_bn_res_relu(a, w, sc, sh, r)) if res else (lambda: ops.gemm_bn(a, w, sc, sh, relu=True)))
    byts = M * K * 2 + M * N * 2 * (2 if res else 1)
    tot += us
    print(f"{name:28s} M={M:8d}: {us:8.1f} us  {byts / us * 1e-3:7.0f} GB/s algorithmic  {2.0 * M * N * K / us * 1e-6:6.0f} TFLOP/s", flush=True)
    del a, w, r
print(f"sum {tot:.0f} us   (PCLIP_GEMM_BN_CFG={os.environ.get('PCLIP_GEMM_BN_CFG')})")
