#!/usr/bin/env python3
"""Tile order of the four-wave GEMM: column tiles fastest (band 0) against bands of `b` column tiles with the row panels fastest inside a band
(PCLIP_GEMM_BAND / PCLIP_GEMM_BAND_N, re-read per call under PCLIP_GEMM_CFG_LIVE), same process, interleaved rounds, bit-identical outputs.
    python tools/ab_band4w.py            timing table
    PMC_BAND=<b> python tools/ab_band4w.py pmc     three launches per shape with that band (under rocprofv3 --pmc FETCH_SIZE)"""
import os, sys, torch
os.environ["PCLIP_GEMM_CFG_LIVE"] = "1"
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import ops

SHAPES = [("in_proj", 201728, 2304, 768, 0, False), ("c_fc", 201728, 3072, 768, 1, False), ("out_proj", 201728, 768, 768, 0, True), ("c_proj", 201728, 768, 3072, 0, True)]
BANDS = {"in_proj": (0, 3, 5), "c_fc": (0, 3, 4, 6), "out_proj": (0, 1, 2), "c_proj": (0, 1, 2)}


def setband(name, b):
    os.environ["PCLIP_GEMM_BAND"] = str(b)           # >= 8 column tiles (c_fc: 12, in_proj: 9)
    os.environ["PCLIP_GEMM_BAND_N"] = str(b)         # narrower launches


def operands(M, N, K, res):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = torch.randn(M, K, device="cuda", generator=g).half()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    bias = torch.randn(N, device="cuda", generator=g).half()
    r = torch.randn(M, N, device="cuda", generator=g).half() if res else None
    return a, w, bias, r, torch.empty(M, N, device="cuda", dtype=torch.float16)


def timeit(fn, iters=8):
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


if len(sys.argv) > 1 and sys.argv[1] == "pmc":
    b = int(os.environ.get("PMC_BAND", "0"))
    for name, M, N, K, act, res in SHAPES[:2]:
        a, w, bias, r, out = operands(M, N, K, res)
        setband(name, b)
        for _ in range(3):
            ops.gemm(a, w, bias, act, r, out)
        torch.cuda.synchronize()
    sys.exit(0)

for name, M, N, K, act, res in SHAPES:
    a, w, bias, r, out = operands(M, N, K, res)
    setband(name, 0)
    ref = ops.gemm(a, w, bias, act, r).clone()
    med, same = {}, {}
    for b in BANDS[name]:
        setband(name, b)
        ops.gemm(a, w, bias, act, r, out)
        same[b] = torch.equal(out, ref)
        med[b] = []
    for rnd in range(5):
        for b in (BANDS[name] if rnd % 2 == 0 else BANDS[name][::-1]):
            setband(name, b)
            ops.gemm(a, w, bias, act, r, out)
            med[b].append(timeit(lambda: ops.gemm(a, w, bias, act, r, out)))
    fl = 2.0 * M * N * K
    print(f"{name:9s} {M}x{N}x{K}: " + " | ".join(f"band {b}: {sorted(v)[len(v) // 2]:7.1f} us ({fl / sorted(v)[len(v) // 2] / 1e6:5.0f} TF) identical={same[b]}" for b, v in med.items()), flush=True)
