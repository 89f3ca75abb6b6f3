#!/usr/bin/env python3
"""Per-phase time stamps of ONE output tile of the four-wave GEMM ("the record must show per-phase cycle stamps of one tile").  Loop variant 8 = the
product loop in a kernel whose waves keep eight s_memrealtime stamps (100 MHz: 10 ns) per tile for their first eight tiles (csrc/pclip_gemm4w.hip); the chip is loaded
as in the bench (the launch is the bench's own shape, every CU busy).  Prints, per shape, the median over tiles 1 .. 6 of workgroups 0 / G/2, waves 0 / 3, of
    init      tile loop top -> accumulators initialised + barrier           (256 v_accvgpr_write + descriptors + one barrier)
    k_loop    the asm statement
    drain     statement done -> epilogue's first barrier passed            (MFMA results back, residual requests, LDS barrier)
    slab k    barrier opening interval k -> next                            (stage slab k+1 | read back + store slab k)
    tail      last interval (stores of slab 3, B'(1) request) -> next tile's loop top
in microseconds and as a fraction of the tile, + the sum against the launch's time per tile."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import _lib, ops

lib = _lib.load()
buf = torch.zeros(260, dtype=torch.int32, device="cuda")
lib.pclip_gemm4w_stamp_buffer(_lib.ptr(buf))
SHAPES = [("in_proj", 201728, 2304, 768, 0, False), ("c_fc", 201728, 3072, 768, 1, False), ("out_proj", 201728, 768, 768, 0, True), ("c_proj", 201728, 768, 3072, 0, True)]
for name, M, N, K, act, res in SHAPES:
    g = torch.Generator(device="cuda").manual_seed(1)
    a = torch.randn(M, K, device="cuda", generator=g).half()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    bias = torch.randn(N, device="cuda", generator=g).half()
    r = torch.randn(M, N, device="cuda", generator=g).half() if res else None
    out = torch.empty(M, N, device="cuda", dtype=torch.float16)
    for _ in range(300):                                         # ~0.2 s: the DVFS loop settles at the load's clock (a cold first shape read 13 % slow)
        ops.gemm4w(a, w, bias, act, r, out, 0)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.gemm4w(a, w, bias, act, r, out, 0)
    e1.record()
    torch.cuda.synchronize()
    t_prod = e0.elapsed_time(e1) / 5 * 1e3
    buf.zero_()
    for _ in range(50):
        ops.gemm4w(a, w, bias, act, r, out, 8)
    e0.record()
    for _ in range(5):
        ops.gemm4w(a, w, bias, act, r, out, 8)
    e1.record()
    torch.cuda.synchronize()
    t_stamp = e0.elapsed_time(e1) / 5 * 1e3
    st = buf.cpu().numpy().astype("int64") & 0xffffffff
    tiles_per_cu = ((M + 255) // 256) * (N // 256) / 256.0
    names = ["init", "k_loop", "drain", "slab 0 staged", "interval 0", "interval 1", "interval 2", "tail (interval 3)"]
    acc = {n: [] for n in names}
    for wv in range(4):
        s = st[wv * 64:(wv + 1) * 64].reshape(8, 8)
        for t in range(1, 6):                                   # tiles 1 .. 5: steady state (tile 0 carries the cold start)
            d = [(int(s[t][p + 1]) - int(s[t][p])) & 0xffffffff for p in range(7)] + [(int(s[t + 1][0]) - int(s[t][7])) & 0xffffffff]
            for n, v in zip(names, d):
                acc[n].append(v * 0.01)                        # 100 MHz ticks -> us
    med = {n: sorted(v)[len(v) // 2] for n, v in acc.items()}
    tot = sum(med.values())
    print(f"{name:9s} {M}x{N}x{K}: product loop {t_prod:7.1f} us, stamped build {t_stamp:7.1f} us ({tiles_per_cu:.1f} tiles per CU -> {t_prod / tiles_per_cu:.2f} us per tile); stamps sum {tot:.2f} us per tile")
    print("          " + " | ".join(f"{n} {med[n]:.2f} ({100 * med[n] / tot:.0f} %)" for n in names), flush=True)
    epi = tot - med["k_loop"] - med["init"]
    print(f"          epilogue (drain .. tail) {epi:.2f} us = {100 * epi / tot:.0f} % of the tile; init {100 * med['init'] / tot:.0f} %; K-loop {100 * med['k_loop'] / tot:.0f} %", flush=True)
