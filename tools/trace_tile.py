#!/usr/bin/env python3
"""Tile-level timing of the persistent GEMM (libpclip_trace.so, -DPCLIP_TRACE=1): s_memtime ticks per output tile spent in the K-loop, in the
requests of the next tile's first K-tile, and in the four phases of the LDS-staged epilogue (both slabs), for wave 0 and wave 7."""
import ctypes, os, sys, torch
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "proto-clip_amd")
l = ctypes.CDLL(os.path.join(root, "libpclip_trace.so"))
P, I = ctypes.c_void_p, ctypes.c_int
l.pclip_gemm_f16.argtypes = [P, I, P, I, P, I, I, I, I, P, I, P, P]
l.pclip_debug_trace2.argtypes = [P, I]
buf = (ctypes.c_ulonglong * 16)()
names = ["slab hook + barrier", "staging writes", "barrier", "row-major reads + stores"]
for (m, n, k), act, res in (((201728, 2304, 768), 0, False), ((201728, 3072, 768), 1, False), ((201728, 768, 768), 0, True), ((201728, 768, 3072), 0, True)):
    a = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
    bias = torch.randn(n, device="cuda").half(); out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    rs = torch.randn(m, n, device="cuda").half() if res else None
    st = P(torch.cuda.current_stream().cuda_stream)
    call = lambda: l.pclip_gemm_f16(P(a.data_ptr()), k, P(w.data_ptr()), k, P(out.data_ptr()), n, m, n, k, P(bias.data_ptr()), act, P(rs.data_ptr()) if res else None, st)
    for _ in range(3): call()
    torch.cuda.synchronize(); l.pclip_debug_trace2(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); call(); e1.record(); torch.cuda.synchronize()
    l.pclip_debug_trace2(ctypes.cast(buf, P), 0)
    print(f"{m}x{n}x{k} act {act}{' + residual' if res else ''}: {e0.elapsed_time(e1) * 1e3:.0f} us (instrumented)")
    for wv, off in (("wave 0", 0), ("wave 7", 8)):
        nt = max(buf[off + 7], 1)
        print(f"  {wv}: per tile: K-loop {buf[off + 4] / nt:.0f} ticks, next-tile requests {buf[off + 5] / nt:.0f}, epilogue {buf[off + 6] / nt:.0f} = " +
              ", ".join(f"{names[i]} {buf[off + i] / nt:.0f}" for i in range(4)) + " (two slabs)")
