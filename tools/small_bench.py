#!/usr/bin/env python3
"""Kernel-only timing of BASELINE configs[1] (EuroSAT 16-shot ViT-B/32: prototype build + classification) and the other
small-N datasets: these are launch-latency / HBM bound (8.35 MB => 1.3 us at 6.3 TB/s)."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import ops
from kernel_bench import timeit
for name, N, K, D, Q in (("EuroSAT C2", 10, 16, 512, 8100), ("OxfordPets", 37, 16, 512, 3669), ("DTD", 47, 16, 512, 1692), ("N=64", 64, 16, 512, 20000),
                         ("N=24 D=1024", 24, 16, 1024, 20000), ("Caltech-101", 100, 16, 1024, 2465), ("FewSOL-198", 198, 16, 768, 666),
                         ("ImageNet", 1000, 16, 512, 50000)):
    mem = torch.nn.functional.normalize(torch.randn(N * K, D, device="cuda"), dim=-1).half()
    q = torch.nn.functional.normalize(torch.randn(Q, D, device="cuda"), dim=-1).half()
    zt = torch.nn.functional.normalize(torch.randn(N, D, device="cuda"), dim=-1).half()
    zi, zi_sq = ops.proto_build(mem, N, K, want_sq=True)
    def gpu_time(fn, reps=20):
        """Device time per call: `reps` calls recorded into one hipGraph, so host launch overhead (~25 us per Python call) is out."""
        fn(); torch.cuda.synchronize()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(reps):
                fn()
        return timeit(g.replay, iters=20) / reps
    t_pb = gpu_time(lambda: ops.proto_build(mem, N, K))
    t_cl = gpu_time(lambda: ops.classify(q, zi, zt, 0.5, 12.0, want_p=False, want_argmax=True))
    byts = Q * D * 2 + 2 * N * D * 2 + Q * 4
    print(f"{name:12s} N={N:4d} D={D:4d} Q={Q:5d}: proto_build {t_pb*1e6:6.1f} us | classify(argmax) {t_cl*1e6:7.1f} us "
          f"= {Q/t_cl/1e6:7.1f} M queries/s, {byts/t_cl/1e9:7.1f} GB/s of {byts/1e6:.2f} MB algorithmic", flush=True)
    if N > 32:
        with ops.classify_two_stage():
            t_2s = gpu_time(lambda: ops.classify(q, zi, zt, 0.5, 12.0, want_p=False, want_argmax=True), reps=5)
        print(f"{'':12s} two-stage path (sqdist + fuse_probs): {t_2s*1e6:7.1f} us -> fused row panels {t_cl*1e6:7.1f} us ({t_2s / t_cl:4.2f} x); "
              f"{4.0 * Q * N * D * 2 / t_cl / 1e12:6.0f} TFLOP/s executed (two passes x two banks)", flush=True)
