#!/usr/bin/env python3
"""Interleaved (ABBA) A/B of pclip_attention_f16 between two builds of libpclip (libpclip.so vs libpclip_old.so), same process,
same tensors, bitwise comparison of the outputs."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kernel_bench import timeit
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "proto-clip_amd")
libs = {n: ctypes.CDLL(os.path.join(root, f)) for n, f in (("new", "libpclip.so"), ("old", "libpclip_old.so"))}
P = ctypes.c_void_p
for l in libs.values():
    l.pclip_attention_f16.argtypes = [P, P] + [ctypes.c_int] * 5 + [P]
for name, B, L, H, causal in (("ViT-B/16", 1024, 197, 12, 0), ("ViT-L/14", 256, 257, 16, 0), ("ViT-B/32", 1024, 50, 12, 0), ("text", 7000, 77, 8, 1),
                              ("L=288", 64, 288, 12, 0), ("L=33 causal", 100, 33, 8, 1)):
    g = torch.Generator(device="cuda").manual_seed(L)
    qkv = torch.randn(B * L, 3 * H * 64, device="cuda", generator=g).half()
    out = {x: torch.zeros(B * L, H * 64, device="cuda", dtype=torch.float16) for x in libs}
    st = P(torch.cuda.current_stream().cuda_stream)
    def call(x):
        assert libs[x].pclip_attention_f16(P(qkv.data_ptr()), P(out[x].data_ptr()), B, L, H, 64, causal, st) == 0
    res = {x: [] for x in libs}
    for r in range(6):
        for x in (list(libs) if r % 2 == 0 else list(libs)[::-1]):
            res[x].append(timeit(lambda: call(x), iters=8, warm=2) * 1e6)
    print(f"{name:12s} " + " | ".join(f"{x} {sorted(res[x])[len(res[x]) // 2]:7.1f} us" for x in libs) + f" | identical {torch.equal(out['new'], out['old'])}", flush=True)
