#!/usr/bin/env python3
"""LayerNorm pass + linear against the folded form (row statistics + pclip_gemm_ln_f16) on the bench's shapes, same process,
interleaved: time of [layernorm; gemm] vs [row_stats; gemm_ln] for in_proj (bias) and c_fc (bias + QuickGELU)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kernel_bench import timeit
from proto_clip_amd import ops
M, K = 201728, 768
x = (torch.randn(M, K, device="cuda") * 1.3).half()
g, be = torch.ones(K, device="cuda") + 0.1 * torch.randn(K, device="cuda"), 0.1 * torch.randn(K, device="cuda")
for name, N, act in (("in_proj", 2304, 0), ("c_fc", 3072, 1)):
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).half(); b = (0.1 * torch.randn(N, device="cuda")).half()
    wf, cs, bf = ops.ln_fold_weights(w, b, g, be)
    out = torch.empty(M, N, device="cuda", dtype=torch.float16); h = torch.empty_like(x)
    parts = {"layernorm": lambda: ops.layernorm(x, g, be, out=h), "gemm": lambda: ops.gemm(h, w, b, act=act, out=out),
             "row_stats": lambda: ops.row_stats(x), "gemm_ln": None}
    st = ops.row_stats(x)
    parts["gemm_ln"] = lambda: ops.gemm_ln(x, st, wf, cs, bf, act=act, out=out)
    res = {k: [] for k in parts}
    for r in range(5):
        for k in (list(parts) if r % 2 == 0 else list(parts)[::-1]):
            res[k].append(timeit(parts[k], iters=6, warm=2) * 1e6)
    med = {k: sorted(v)[len(v) // 2] for k, v in res.items()}
    print(f"{name} N={N}: layernorm {med['layernorm']:.1f} + gemm {med['gemm']:.1f} = {med['layernorm'] + med['gemm']:.1f} us | "
          f"row_stats {med['row_stats']:.1f} + gemm_ln {med['gemm_ln']:.1f} = {med['row_stats'] + med['gemm_ln']:.1f} us", flush=True)
