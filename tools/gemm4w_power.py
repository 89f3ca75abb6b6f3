#!/usr/bin/env python3
"""Is the four-wave GEMM power-bound, and what does its epilogue cost in ENERGY?  (VERDICT r5 #1.)  For in_proj and c_fc (the bench's shapes) the product loop
(variant 0) and the same loop WITHOUT its epilogue (variant 6: nothing converted, staged or stored) each run back to back for ~1.5 s while the socket power and the
shader clock are sampled (proto_clip_amd.telemetry, amdsmi, 50 Hz).  Reading: if both run at the socket's power cap, the time of a launch is its energy divided
by the cap — overlapping the epilogue with MFMA work (hiding it) cannot shorten the launch, only removing ENERGY can; the clock each variant holds tells how much
power its instruction mix draws per cycle."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import ops
from proto_clip_amd.telemetry import Sampler

SHAPES = [("in_proj", 201728, 2304, 768, 0), ("c_fc", 201728, 3072, 768, 1)]
for name, M, N, K, act in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(M, K, device="cuda", generator=g).half()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    bias = torch.randn(N, device="cuda", generator=g).half()
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    res = {}
    for var in (0, 6, 0, 6):
        for _ in range(200):                                   # ~0.15 s of warm-up: the DVFS loop settles
            ops.gemm4w(a, w, bias, act, None, out, var)
        torch.cuda.synchronize()
        n = 1600
        with Sampler(period=0.02, skip_s=0.2) as s:
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(n):
                ops.gemm4w(a, w, bias, act, None, out, var)
            e1.record()
            torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / n * 1e3
        sm = s.summary()
        res.setdefault(var, []).append((us, (sm.get("sclk_mhz") or {}).get("median"), (sm.get("power_w") or {}).get("median")))
    for var, rows in res.items():
        for us, clk, pw in rows:
            tiles = ((M + 255) // 256) * (N // 256) / 256.0
            cyc = us * (clk or 0) / tiles if clk else float("nan")
            print(f"{name:8s} variant {var} ({'product loop' if var == 0 else 'no epilogue '}): {us:7.1f} us per launch, sclk {clk} MHz, socket {pw} W, "
                  f"{us * (pw or 0) * 1e-6:.3f} J per launch, {cyc / 1e3:.1f} k shader cycles per tile", flush=True)
