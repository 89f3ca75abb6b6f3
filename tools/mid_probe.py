#!/usr/bin/env python3
"""Device time of the one-launch mid-N classification (csrc/pclip_classify_mid.hip) against the two stages it replaces, hipGraph replay of 20 calls; the Q = 16 row is ONE
workgroup on the whole chip: the time a CU needs to stream both banks."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import ops
from kernel_bench import timeit
nrm = torch.nn.functional.normalize
for name, N, D, Q in (("OxfordPets", 37, 512, 3669), ("Caltech-101", 100, 1024, 2465), ("FewSOL-198", 198, 768, 666), ("FewSOL-198 Q=16", 198, 768, 16), ("N=256 D=1024", 256, 1024, 4096),
                      ("N=24 D=1024", 24, 1024, 20000), ("EuroSAT", 10, 512, 8100), ("N=32 D=512", 32, 512, 4000), ("N=16 D=512", 16, 512, 4000), ("N=10 D=1024", 10, 1024, 8100), ("N=17 D=512", 17, 512, 300)):
    q = nrm(torch.randn(Q, D, device="cuda"), dim=-1).half()
    zi = nrm(torch.randn(N, D, device="cuda"), dim=-1).half()
    zt = nrm(torch.randn(N, D, device="cuda"), dim=-1).half()
    out = []
    for mode in (2, 0):                                              # 2: the one-launch mid-N kernel for every shape it can run; 0: what else the shape takes (N <= 32: classify_small, otherwise the two stages)
        def fn():
            with ops.classify_mid(mode):
                return ops.classify(q, zi, zt, 0.5, 12.0, want_p=False, want_argmax=True)
        fn(); torch.cuda.synchronize()
        side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
        g = torch.cuda.CUDAGraph()
        with torch.cuda.graph(g):
            for _ in range(20):
                fn()
        out.append(f"{'mid-N kernel' if mode else 'other route'}: {timeit(g.replay, iters=20) / 20 * 1e6:6.1f} us")
    print(f"{name:18s} N={N:4d} D={D:4d} Q={Q:5d}: " + " | ".join(out), flush=True)
