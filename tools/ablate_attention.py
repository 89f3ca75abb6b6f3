#!/usr/bin/env python3
"""Ablation of the persistent attention kernel from compile-time variants (proto-clip_amd/libpclip_att<N>.so built with
-DPCLIP_ATT_ABL=N: 1 no prefetch DMA inside the loop, 2 no compute, 4 no stores; tools/build_att_abl.sh).  Prints the time of
pclip_attention_f16 in its persistent mode per variant, and of the one-workgroup-per-item kernel, on the bench's shape."""
import ctypes, glob, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kernel_bench import timeit
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "proto-clip_amd")
names = ["libpclip.so"] + sorted(os.path.basename(p) for p in glob.glob(os.path.join(root, "libpclip_att*.so")))
libs = {n.replace("libpclip", "").replace(".so", "") or "full": ctypes.CDLL(os.path.join(root, n)) for n in names}
P = ctypes.c_void_p
for l in libs.values():
    l.pclip_attention_f16.argtypes = [P, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P]
    l.pclip_attention_config.argtypes = [ctypes.c_int, ctypes.c_int]
shapes = [tuple(int(v) for v in x.split("x")) for x in os.environ.get("SHAPES", "1024x197x12,1024x50x12,1024x77x8").split(",")]
for B, L, H in shapes:
    qkv = torch.randn(B * L, 3 * H * 64, device="cuda").half()
    out = torch.empty(B * L, H * 64, device="cuda", dtype=torch.float16)
    st = P(torch.cuda.current_stream().cuda_stream)
    row = []
    for x, lib in libs.items():
        for mode in ((0, 1) if x == "full" else (1,)):
            lib.pclip_attention_config(mode, 0)
            def call():
                assert lib.pclip_attention_f16(P(qkv.data_ptr()), P(out.data_ptr()), B, L, H, 64, 0, st) == 0
            ts = sorted(timeit(call, iters=10, warm=2) * 1e6 for _ in range(3))
            row.append(f"{x}{'/per-item' if mode == 0 else ''} {ts[1]:7.1f}")
    print(f"B={B} L={L} H={H}: " + " | ".join(row), flush=True)
