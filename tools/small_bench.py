#!/usr/bin/env python3
"""Kernel-only timing of BASELINE configs[1] (EuroSAT 16-shot ViT-B/32: prototype build + classification) and the other
small-N datasets: these are launch-latency / HBM bound (8.35 MB => 1.3 us at 6.3 TB/s)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import ops
from kernel_bench import timeit
for name, N, K, D, Q in (("EuroSAT C2", 10, 16, 512, 8100), ("Caltech-101", 100, 16, 1024, 2465), ("FewSOL-198", 198, 16, 768, 666),
                         ("ImageNet", 1000, 16, 512, 50000)):
    mem = torch.nn.functional.normalize(torch.randn(N * K, D, device="cuda"), dim=-1).half()
    q = torch.nn.functional.normalize(torch.randn(Q, D, device="cuda"), dim=-1).half()
    zt = torch.nn.functional.normalize(torch.randn(N, D, device="cuda"), dim=-1).half()
    zi, zi_sq = ops.proto_build(mem, N, K, want_sq=True)
    t_pb = timeit(lambda: ops.proto_build(mem, N, K), iters=50)
    t_cl = timeit(lambda: ops.classify(q, zi, zt, 0.5, 12.0, want_p=False, want_argmax=True), iters=50)
    byts = Q * D * 2 + 2 * N * D * 2 + Q * 4
    print(f"{name:12s} N={N:4d} D={D:4d} Q={Q:5d}: proto_build {t_pb*1e6:6.1f} us | classify(argmax) {t_cl*1e6:7.1f} us "
          f"= {Q/t_cl/1e6:7.1f} M queries/s, {byts/t_cl/1e9:7.1f} GB/s of {byts/1e6:.2f} MB algorithmic", flush=True)
