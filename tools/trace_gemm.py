#!/usr/bin/env python3
"""Where do the waves of the persistent GEMM wait?  libpclip_trace.so (-DPCLIP_TRACE=1) stamps s_memtime around every wait of the K-loop
and adds the per-wave totals of waves 0 and 7 of every workgroup; printed as a share of the K-loop's cycles."""
import ctypes, os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "proto-clip_amd")
l = ctypes.CDLL(os.path.join(root, "libpclip_trace.so"))
P, I = ctypes.c_void_p, ctypes.c_int
l.pclip_gemm_f16.argtypes = [P, I, P, I, P, I, I, I, I, P, I, P, P]
l.pclip_debug_trace.argtypes = [P, I]
buf = (ctypes.c_ulonglong * 18)()
names = ["top vmcnt wait", "top barrier", "X1 lgkmcnt wait", "X1 barrier", "X2 lgkmcnt wait", "X2 barrier", "K-loop total", "K-tiles"]
for m, n, k in ((201728, 3072, 768), (201728, 768, 768), (201728, 768, 3072)):
    a = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
    bias = torch.randn(n, device="cuda").half(); out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    st = P(torch.cuda.current_stream().cuda_stream)
    call = lambda: l.pclip_gemm_f16(P(a.data_ptr()), k, P(w.data_ptr()), k, P(out.data_ptr()), n, m, n, k, P(bias.data_ptr()), 0, None, st)
    for _ in range(3): call()
    torch.cuda.synchronize(); l.pclip_debug_trace(None, 1)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record(); call(); e1.record(); torch.cuda.synchronize()
    l.pclip_debug_trace(ctypes.cast(buf, P), 0)
    us = e0.elapsed_time(e1) * 1e3
    print(f"{m}x{n}x{k}: {us:.0f} us (instrumented); longest workgroup span {buf[17]} s_memtime ticks = {buf[17] / us:.1f} ticks per us")
    for wv, off in (("wave 0", 0), ("wave 7", 8)):
        tot, kt = buf[off + 6], max(buf[off + 7], 1)
        print(f"  {wv}: K-loop {tot / kt:.0f} cycles per K-tile; " + ", ".join(f"{names[i]} {100.0 * buf[off + i] / tot:.1f} %" for i in range(6)))
