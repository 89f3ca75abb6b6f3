#!/usr/bin/env python3
"""Residual GEMM with and without the row-statistics epilogue (act 6 vs act 9) on the bench's two residual shapes; any
proto-clip_amd/libpclip_r9_<tag>.so next to the library is timed as a further act-9 variant (the ablation builds behind
profiles/r02_ab_res_stats_ablation.txt — no partial stores / no statistics arithmetic — were compile-time edits of put_partials)."""
import ctypes, glob, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kernel_bench import timeit
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "proto-clip_amd")
names = ["libpclip.so"] + sorted(os.path.basename(p) for p in glob.glob(os.path.join(root, "libpclip_r9_*.so")))
libs = {n.replace("libpclip", "").replace(".so", "") or "full": ctypes.CDLL(os.path.join(root, n)) for n in names}
P = ctypes.c_void_p
for l in libs.values():
    l.pclip_gemm_f16.argtypes = [P, ctypes.c_int, P, ctypes.c_int, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, ctypes.c_int, P, P]
    l.pclip_gemm_res_stats_f16.argtypes = [P, ctypes.c_int, P, ctypes.c_int, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, P, P, P]
M, N = 201728, 768
for K in (768, 3072):
    a = torch.randn(M, K, device="cuda").half(); w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    b = torch.randn(N, device="cuda").half(); x = torch.randn(M, N, device="cuda").half()
    part = torch.empty(M, N // 64, 2, device="cuda"); st = P(torch.cuda.current_stream().cuda_stream)
    cases = {}
    for nm, lib in libs.items():
        if nm == "full":
            cases["act6"] = lambda lib=lib: lib.pclip_gemm_f16(P(a.data_ptr()), K, P(w.data_ptr()), K, P(x.data_ptr()), N, M, N, K, P(b.data_ptr()), 0, P(x.data_ptr()), st)
        cases["act9" + nm.replace("full", "")] = lambda lib=lib: lib.pclip_gemm_res_stats_f16(P(a.data_ptr()), K, P(w.data_ptr()), K, P(x.data_ptr()), N, M, N, K, P(b.data_ptr()), P(x.data_ptr()), P(part.data_ptr()), st)
    res = {k: [] for k in cases}
    for r in range(5):
        for k in (list(cases) if r % 2 == 0 else list(cases)[::-1]):
            x.normal_()
            res[k].append(timeit(cases[k], iters=6, warm=2) * 1e6)
    print(f"K={K}: " + " | ".join(f"{k} {sorted(v)[len(v) // 2]:7.1f}" for k, v in res.items()), flush=True)
