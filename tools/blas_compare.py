#!/usr/bin/env python3
"""Context for the roofline fraction: the vendor BLAS behind torch (hipBLASLt / rocBLAS) on the bench's four linear shapes, beside pclip_gemm_f16.
torch.nn.functional.linear = GEMM + bias only (no QuickGELU, no residual add: those are extra elementwise passes there); a measurement tool, not a product path."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import ops
from kernel_bench import timeit
print("torch", torch.__version__, "hip", torch.version.hip, flush=True)
for m, n, k, what in [(201728, 3072, 768, "c_fc"), (201728, 2304, 768, "in_proj"), (201728, 768, 768, "out_proj"), (201728, 768, 3072, "c_proj"), (8192, 8192, 8192, "square")]:
    a = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half(); b = torch.randn(n, device="cuda").half()
    out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    t_blas = timeit(lambda: torch.nn.functional.linear(a, w, b), iters=10, warm=3)
    t_mm = timeit(lambda: torch.matmul(a, w.t(), out=out), iters=10, warm=3)
    t_ours = timeit(lambda: ops.gemm(a, w, b, 0, None, out), iters=10, warm=3)
    fl = 2.0 * m * n * k
    print(f"{what:9s} {m}x{n}x{k}: torch linear (bias) {t_blas * 1e6:7.1f} us ({fl / t_blas / 1e12:5.0f} TF) | torch matmul {t_mm * 1e6:7.1f} us ({fl / t_mm / 1e12:5.0f} TF) | "
          f"pclip_gemm_f16 (bias) {t_ours * 1e6:7.1f} us ({fl / t_ours / 1e12:5.0f} TF)", flush=True)
