#!/usr/bin/env python3
"""Ablation of the persistent GEMM from compile-time variants of the library (proto-clip_amd/libpclip_abl<N>.so built with
-DPCLIP_ABL=N: 1 no LDS-DMA inside the K-loop, 2 no MFMAs, 4 no epilogue; guide §5.4: ablate before optimising).
A run-time switch was tried first and is useless: the extra branches split the K-loop's basic blocks and the kernel ran 2.3x slower."""
import ctypes, glob, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kernel_bench import timeit
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "proto-clip_amd")
names = ["libpclip.so"] + sorted(os.path.basename(p) for p in glob.glob(os.path.join(root, "libpclip_abl*.so")))
libs = {n.replace("libpclip", "").replace(".so", "") or "full": ctypes.CDLL(os.path.join(root, n)) for n in names}
P = ctypes.c_void_p
for l in libs.values():
    l.pclip_gemm_f16.argtypes = [P, ctypes.c_int, P, ctypes.c_int, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, ctypes.c_int, P, P]
shapes = [(201728, 3072, 768), (201728, 768, 768), (201728, 768, 3072)]
for m, n, k in shapes:
    a = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
    bias = torch.randn(n, device="cuda").half(); out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    st = P(torch.cuda.current_stream().cuda_stream)
    for act in (0, 1):
        res = {x: [] for x in libs}
        for r in range(3):
            for x in libs:
                def call():
                    assert libs[x].pclip_gemm_f16(P(a.data_ptr()), k, P(w.data_ptr()), k, P(out.data_ptr()), n, m, n, k, P(bias.data_ptr()), act, None, st) == 0
                res[x].append(timeit(call, iters=6, warm=2) * 1e6)
        print(f"{m}x{n}x{k} act {act}: " + " | ".join(f"{x} {sorted(t)[1]:7.1f}" for x, t in res.items()), flush=True)
