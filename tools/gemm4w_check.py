#!/usr/bin/env python3
"""Four-wave asm-loop GEMM (pclip_gemm4w_f16) against the eight-wave persistent kernel (pclip_gemm_f16, PCLIP_GEMM_4W=0):
bit-identity on the bench shapes and the edge cases of the ring (every tail variant of the K-loop, ragged M), then interleaved
timing rounds of both kernels on the bench's four linears.  PCLIP_RACE_STRESS=1 runs the jittered build of the same loop."""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import _lib  # noqa: E402
from proto_clip_amd import ops  # noqa: E402

_lib.load().pclip_gemm4w_config(0)                    # ops.gemm = the eight-wave reference inside this tool
VAR = 1 if os.environ.get("PCLIP_RACE_STRESS") == "1" else 0


def gemm4w(a, w, bias, act, residual, out, var=None):
    return ops.gemm4w(a, w, bias, act, residual, out, VAR if var is None else var)


def case(M, N, K, act, use_bias, use_res, seed=0):
    g = torch.Generator(device="cuda").manual_seed(seed)
    a = torch.randn(M, K, device="cuda", generator=g).half()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    bias = torch.randn(N, device="cuda", generator=g).half() if use_bias else None
    res = torch.randn(M, N, device="cuda", generator=g).half() if use_res else None
    return a, w, bias, res


def check(M, N, K, act, use_bias, use_res):
    a, w, bias, res = case(M, N, K, act, use_bias, use_res)
    ref = torch.empty(M, N, device="cuda", dtype=torch.float16)
    out = torch.full((M + 1, N), 7.0, device="cuda", dtype=torch.float16)      # a guard row behind the last one
    ops.gemm(a, w, bias, act, res, ref)
    gemm4w(a, w, bias, act, res, out[:M])
    torch.cuda.synchronize()
    same = torch.equal(ref, out[:M]) and bool((out[M] == 7.0).all())
    err = (ref.float() - out[:M].float()).abs().max().item()
    # independent check of the reference itself (fp32 matmul) so that "identical" is not "identically wrong"
    sl = slice(0, min(M, 512))
    y = a[sl].float() @ w.float().t()
    if bias is not None: y = y + bias.float()
    y = y.half().float()
    if act == 1: y = (y * torch.sigmoid(1.702 * y))
    if res is not None: y = (y + res[sl].float())
    rel = ((y - out[sl].float()).abs().max() / y.abs().max()).item()
    print(f"check M={M:6d} N={N:4d} K={K:4d} act={act} bias={int(use_bias)} res={int(use_res)}: identical={same} max|d|={err:.3g} rel-to-fp32={rel:.2e}", flush=True)
    return same and rel < 2e-2


def timeit(fn, iters):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e-3


def variants(args):
    vs = [0] + [int(v) for v in args.variants.split(",")]
    M = args.imgs * 197
    shapes = {"in_proj": (M, 2304, 768, 0, True, False), "out_proj": (M, 768, 768, 0, True, True), "c_fc": (M, 3072, 768, 1, True, False),
              "c_proj": (M, 768, 3072, 0, True, True), "sq8192": (8192, 8192, 8192, 0, True, False)}
    for name, (m, n, k, act, ub, ur) in shapes.items():
        a, w, bias, res = case(m, n, k, act, ub, ur)
        outs = {v: torch.empty(m, n, device="cuda", dtype=torch.float16) for v in vs}
        def run(v):
            gemm4w(a, w, bias, act, res, outs[v], v)
        for v in vs:
            run(v); run(v)
        ts = {v: [] for v in vs}
        for _ in range(args.rounds):
            for v in vs:
                ts[v].append(timeit(lambda: run(v), 10))
        med = {v: sorted(ts[v])[len(ts[v]) // 2] for v in vs}
        same = all(torch.equal(outs[0], outs[v]) for v in vs if v != 6)         # variant 6 stores nothing (epilogue ablation)
        fl = 2.0 * m * n * k
        print(f"variants {name:9s}: " + " | ".join(f"V{v} {med[v] * 1e6:7.1f} us {fl / med[v] / 1e12:5.0f} TF ({med[0] / med[v]:5.3f})" for v in vs) + f" identical={same}", flush=True)
    return 0


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--no-bench", action="store_true")
    ap.add_argument("--rounds", type=int, default=5)
    ap.add_argument("--imgs", type=int, default=1024)
    ap.add_argument("--variants", default="", help="comma list of PCLIP_GEMM4W_VAR values timed against variant 0 (same process, interleaved rounds)")
    args = ap.parse_args()
    if args.variants:
        return variants(args)
    ok = True
    # ring edge cases: nt = 3, 4, 5, 6 (zero to three steady iterations), ragged M, every epilogue
    for K in (192, 256, 320, 384, 768):
        for (M, N) in ((256, 256), (1000, 512), (777, 768)):
            for act, ub, ur in ((0, False, False), (0, True, False), (1, True, False), (0, True, True)):
                ok &= check(M, N, K, act, ub, ur)
    # the ring runs on from output tile to output tile: every ring phase (2 nt mod 5) with several tiles per workgroup
    for K in (192, 256, 320, 448, 832):
        ok &= check(40000, 768, K, 0, True, True)
        ok &= check(33333, 1024, K, 1, True, False)
    # more tiles than CUs (persistent rounds), bench widths
    for (M, N, K, act, ub, ur) in ((257, 256, 768, 0, True, False), (1000, 3072, 768, 1, True, False), (66000, 256, 768, 0, True, False), (131072, 512, 768, 1, True, False),
                                   (20000, 768, 768, 0, True, True), (20000, 2304, 768, 0, True, False), (20000, 3072, 768, 1, True, False),
                                   (20000, 768, 3072, 0, True, True), (70001, 768, 768, 0, True, True)):
        ok &= check(M, N, K, act, ub, ur)
    print("ALL IDENTICAL" if ok else "MISMATCH", flush=True)
    if args.no_bench:
        return 0 if ok else 1
    M = args.imgs * 197
    shapes = {"in_proj": (M, 2304, 768, 0, True, False), "out_proj": (M, 768, 768, 0, True, True), "c_fc": (M, 3072, 768, 1, True, False),
              "c_proj": (M, 768, 3072, 0, True, True), "sq8192": (8192, 8192, 8192, 0, False, False), "sq4096": (4096, 4096, 4096, 0, False, False)}
    for name, (m, n, k, act, ub, ur) in shapes.items():
        a, w, bias, res = case(m, n, k, act, ub, ur)
        o8 = torch.empty(m, n, device="cuda", dtype=torch.float16)
        o4 = torch.empty(m, n, device="cuda", dtype=torch.float16)
        f8 = lambda: ops.gemm(a, w, bias, act, res, o8)
        f4 = lambda: gemm4w(a, w, bias, act, res, o4)
        for _ in range(3):
            f8(); f4()
        t8, t4 = [], []
        for _ in range(args.rounds):
            t8.append(timeit(f8, 10))
            t4.append(timeit(f4, 10))
        same = torch.equal(o8, o4)
        fl = 2.0 * m * n * k
        m8, m4 = sorted(t8)[len(t8) // 2], sorted(t4)[len(t4) // 2]
        print(f"bench {name:9s} {m}x{n}x{k}: 8-wave {m8 * 1e6:8.1f} us ({fl / m8 / 1e12:6.0f} TF) | 4-wave asm {m4 * 1e6:8.1f} us ({fl / m4 / 1e12:6.0f} TF) "
              f"| ratio {m8 / m4:5.3f} identical={same}", flush=True)
    return 0 if ok else 1


if __name__ == "__main__":
    sys.exit(main())
