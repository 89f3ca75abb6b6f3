#!/usr/bin/env python3
"""Device time (hipGraph replay) of the encoder linears at serving sizes: persistent kernel vs split-K (ops.low_latency)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import ops
from kernel_bench import timeit


def gpu_time(fn, reps=20):
    fn(); torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    return timeit(g.replay, iters=10) / reps * 1e6


widths = [int(x) for x in os.environ.get("WIDTHS", "768,1024").split(",")]
for W in widths:
    for M in (1, 8, 197, 788, 1576, 3152):
        line = f"W={W} M={M:5d}:"
        for name, N, K, act in (("qkv", 3 * W, W, 0), ("out", W, W, 0), ("c_fc", 4 * W, W, 1), ("c_proj", W, 4 * W, 0)):
            a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
            b = torch.randn(N, device="cuda").half(); out = torch.empty(M, N, device="cuda", dtype=torch.float16)
            t0 = gpu_time(lambda: ops.gemm(a, w, b, act, None, out))
            ok = ops._lib.load().pclip_gemm_splitk_workspace(M, N, K) > 0
            if ok:
                with ops.low_latency():
                    t1 = gpu_time(lambda: ops.gemm(a, w, b, act, None, out))
            line += f"  {name} {t0:5.1f}" + (f" -> {t1:5.1f}" if ok else "    --   ")
        print(line, flush=True)
