#!/usr/bin/env python3
"""Interleaved A/B timing of the GEMM tile configurations (env PCLIP_GEMM_CFG read per call) on encoder shapes,
with a correctness check of every configuration against configuration 1."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import ops
from kernel_bench import timeit
cfgs = os.environ.get("CFGS", "0,1,2").split(",")
shapes = [tuple(int(x) for x in os.environ["SHAPE"].split("x"))] if "SHAPE" in os.environ else [(50432, 2304, 768), (50432, 768, 768), (50432, 3072, 768), (50432, 768, 3072), (4096, 4096, 4096), (8192, 8192, 8192), (1000, 768, 128), (777, 512, 192)]
for m, n, k in shapes:
    a = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
    if os.environ.get("DATA") == "zero": a.zero_(); w.zero_()
    if os.environ.get("DATA") == "small": a.mul_(1e-3); w.mul_(1e-3)
    bias = torch.randn(n, device="cuda").half(); out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    for name, f in {"plain": lambda: ops.gemm(a, w, None, 0, None, out), "bias": lambda: ops.gemm(a, w, bias, 0, None, out), "bias+gelu": lambda: ops.gemm(a, w, bias, 1, None, out)}.items():
        os.environ["PCLIP_GEMM_CFG"] = "1"; os.environ["PCLIP_GEMM_M16"] = "0"; ref = f().clone()
        res = {c: [] for c in cfgs}
        for c in cfgs:
            os.environ["PCLIP_GEMM_CFG"] = c.rstrip("m"); os.environ["PCLIP_GEMM_M16"] = "1" if c.endswith("m") else "0"
            out.zero_(); got = f(); torch.cuda.synchronize()
            bad = (got != ref).sum().item()
            if bad: print(f"  cfg {c} {name} {m}x{n}x{k}: {bad} mismatching elements, max diff {(got.float()-ref.float()).abs().max().item():.4g}")
        for r in range(5):
            for c in cfgs:
                os.environ["PCLIP_GEMM_CFG"] = c.rstrip("m"); os.environ["PCLIP_GEMM_M16"] = "1" if c.endswith("m") else "0"
                res[c].append(timeit(f, iters=12, warm=2) * 1e6)
        line = f"{m}x{n}x{k} {name:9s}"
        for c in cfgs:
            t = sorted(res[c])[len(res[c]) // 2]
            line += f" | cfg{c} {t:7.1f} us {2.0*m*n*k/t/1e6:6.0f} TF"
        print(line, flush=True)
