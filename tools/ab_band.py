#!/usr/bin/env python3
"""Tile order of the persistent linear kernel: column tiles fastest (band 0) vs bands of `band` column tiles with the row blocks
fastest inside a band (PCLIP_GEMM_BAND), same process, interleaved, bitwise comparison.  VERDICT r2 item 4: the c_fc launch reads
1.8 GB over the fabric for 0.31 GB of operands because its 4.7 MB weight matrix does not stay in the 4 MiB L2 of an XCD.
    python tools/ab_band.py            timing;     PCLIP_GEMM_BAND=6 python tools/ab_band.py once   (a few launches, for rocprofv3 --pmc)"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kernel_bench import timeit
os.environ["PCLIP_GEMM_CFG_LIVE"] = "1"
from proto_clip_amd import ops
once = len(sys.argv) > 1 and sys.argv[1] == "once"
shapes = [("c_fc", 201728, 3072, 768, 1), ("qkv", 201728, 2304, 768, 0)]
for name, m, n, k, act in shapes:
    a = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half(); bias = torch.randn(n, device="cuda").half()
    if once:
        y = torch.empty(m, n, device="cuda", dtype=torch.float16)
        for _ in range(4): ops.gemm(a, w, bias, act, None, y)
        torch.cuda.synchronize()
        continue
    bands = [0, 3, 4, 6] if n == 3072 else [0, 3, 5]
    out = {b: torch.empty(m, n, device="cuda", dtype=torch.float16) for b in bands}
    def call(b):
        os.environ["PCLIP_GEMM_BAND"] = str(b)
        ops.gemm(a, w, bias, act, None, out[b])
    res = {b: [] for b in bands}
    for r in range(6):
        for b in (bands if r % 2 == 0 else bands[::-1]):
            res[b].append(timeit(lambda: call(b), iters=6, warm=2) * 1e6)
    print(f"{name} {m}x{n}x{k}: " + " | ".join(f"band {b}: {sorted(res[b])[3]:7.1f} us ({2.0 * m * n * k / sorted(res[b])[3] / 1e6:5.0f} TF{'' if torch.equal(out[b], out[0]) else ' DIFF'})" for b in bands), flush=True)
