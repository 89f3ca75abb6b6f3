#!/usr/bin/env python3
"""Interleaved A/B of pclip_gemm_f16 between two builds of libpclip (proto-clip_amd/libpclip.so vs libpclip_old.so), same
process, same tensors: the only reliable way to see +-2 % on this pool (boxes differ by +-3 %)."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from kernel_bench import timeit
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "proto-clip_amd")
libs = {n: ctypes.CDLL(os.path.join(root, f)) for n, f in (("new", "libpclip.so"), ("old", "libpclip_old.so"))}
P = ctypes.c_void_p
for l in libs.values():
    l.pclip_gemm_f16.argtypes = [P, ctypes.c_int, P, ctypes.c_int, P, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int, P, ctypes.c_int, P, P]
shapes = [tuple(int(v) for v in x.split("x")) for x in os.environ["SHAPES"].split(",")] if "SHAPES" in os.environ else [(201728, 3072, 768), (201728, 2304, 768), (201728, 768, 768), (201728, 768, 3072)]
for m, n, k in shapes:
    a = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
    bias = torch.randn(n, device="cuda").half(); out = {x: torch.empty(m, n, device="cuda", dtype=torch.float16) for x in libs}
    st = P(torch.cuda.current_stream().cuda_stream)
    resid = torch.randn(m, n, device="cuda").half() if n <= 1024 else None
    cases = {"bias": (bias, 0, None), "bias+gelu": (bias, 1, None)}
    if resid is not None:
        cases["bias+res"] = (bias, 0, resid)                    # x + linear(a): the old build takes its generic kernel for this
    for name, (b, act, rs) in cases.items():
        def call(x):
            rc = libs[x].pclip_gemm_f16(P(a.data_ptr()), k, P(w.data_ptr()), k, P(out[x].data_ptr()), n, m, n, k, P(b.data_ptr()), act,
                                        P(rs.data_ptr()) if rs is not None else None, st)
            assert rc == 0
        res = {x: [] for x in libs}
        for r in range(6):                                     # ABBA order: whichever build runs first in a round pays the clock ramp
            for x in (list(libs) if r % 2 == 0 else list(libs)[::-1]):
                res[x].append(timeit(lambda: call(x), iters=6, warm=2) * 1e6)
        same = torch.equal(out["new"], out["old"])
        med = {x: sorted(res[x])[len(res[x]) // 2] for x in libs}
        print(f"{m}x{n}x{k} {name:9s} " + " | ".join(f"{x} {med[x]:7.1f} us ({2.0 * m * n * k / med[x] / 1e6:5.0f} TF)" for x in libs) + f" | identical {same}", flush=True)
