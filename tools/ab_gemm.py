#!/usr/bin/env python3
"""Interleaved A/B timing of GEMM epilogue variants on one shape (guide §5.4 rule 24: within-probe rounds)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import ops
from kernel_bench import timeit
m, n, k = 50432, int(os.environ.get("N", 2304)), int(os.environ.get("K", 768))
a = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
bias = torch.randn(n, device="cuda").half(); out = torch.empty(m, n, device="cuda", dtype=torch.float16)
variants = {"plain": lambda: ops.gemm(a, w, None, 0, None, out), "bias": lambda: ops.gemm(a, w, bias, 0, None, out),
            "bias+gelu": lambda: ops.gemm(a, w, bias, 1, None, out), "gelu": lambda: ops.gemm(a, w, None, 1, None, out)}
res = {v: [] for v in variants}
for r in range(6):
    for v, f in variants.items():
        res[v].append(timeit(f, iters=15, warm=2) * 1e6)
for v, t in res.items():
    t = sorted(t)
    print(f"{v:10s} min {t[0]:7.1f} med {t[len(t)//2]:7.1f} max {t[-1]:7.1f} us   ({2.0*m*n*k/t[len(t)//2]/1e6:6.0f} TF)")
