#!/usr/bin/env python3
"""A/B of the two attention kernels (PCLIP_ATT_VARIANT is read once per process -> one subprocess per variant):
bitwise comparison of the outputs and interleaved-free timing per shape."""
import os, subprocess, sys
if len(sys.argv) > 1:
    import torch
    sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
    from proto_clip_amd import ops
    from kernel_bench import timeit
    for name, B, L, H, causal in (("ViT-B/16", 1024, 197, 12, False), ("ViT-L/14", 256, 257, 16, False), ("ViT-B/32", 1024, 50, 12, False),
                                  ("text", 7000, 77, 8, True), ("small", 64, 197, 12, False)):
        g = torch.Generator(device="cuda").manual_seed(L)
        qkv = torch.randn(B * L, 3 * H * 64, device="cuda", generator=g).half()
        out = ops.attention(qkv, B, L, H, causal=causal)
        t = sorted(timeit(lambda: ops.attention(qkv, B, L, H, causal=causal, out=out), iters=10, warm=2) for _ in range(3))[1]
        flops = 4.0 * B * H * L * L * 64 * (0.5 if causal else 1.0)
        print(f"{name:9s} {t*1e6:8.1f} us {flops/t/1e12:6.0f} TF  checksum {out.float().sum().item():.6f} {out.view(torch.int16).to(torch.int64).sum().item()}")
else:
    for v in ("0", "1"):
        print("variant", v, flush=True)
        subprocess.run([sys.executable, __file__, "run"], env=dict(os.environ, PCLIP_ATT_VARIANT=v))
