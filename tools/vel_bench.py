#!/usr/bin/env python3
"""vit_embed_ln (class token + patches + positional embedding -> ln_pre -> ln_1, one pass) on the bench shape: time, bytes per second, and bit-identity of the
whole-batch form (affine vectors from LDS) against the same rows pushed through in small batches (the global-load form)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import ops
from kernel_bench import timeit
for B, G2, W in ((1024, 196, 768), (256, 256, 1024)):
    g = torch.Generator(device="cuda").manual_seed(W)
    patch = torch.randn(B * G2, W, device="cuda", generator=g).half()
    cls, pos = torch.randn(W, device="cuda", generator=g).half(), (torch.randn(G2 + 1, W, device="cuda", generator=g) * 0.1).half()
    g0, b0, g1, b1 = (1 + 0.2 * torch.randn(W, device="cuda", generator=g), 0.1 * torch.randn(W, device="cuda", generator=g),
                      1 + 0.2 * torch.randn(W, device="cuda", generator=g), 0.1 * torch.randn(W, device="cuda", generator=g))
    f = lambda: ops.vit_embed_ln(patch, cls, pos, B, G2, W, g0, b0, g1, b1)
    t = timeit(f, iters=20, warm=3)
    x0, h = f()
    nb = 64                                     # small batches take the global-load kernel
    xs, hs = zip(*[ops.vit_embed_ln(patch[i * G2:(i + nb) * G2], cls, pos, nb, G2, W, g0, b0, g1, b1) for i in range(0, B, nb)])
    same = torch.equal(torch.cat(xs), x0) and torch.equal(torch.cat(hs), h)
    R = B * (G2 + 1)
    print(f"vit_embed_ln B={B} G2={G2} W={W}: {t * 1e6:7.1f} us  ({(2.0 * B * G2 * W + 4.0 * R * W) / t / 1e12:.2f} TB/s)  whole batch == small batches: {same}", flush=True)
