#!/usr/bin/env python3
"""Phase timing of the ping-pong K-loop (pgemm::mainloop_pp): libpclip_trace.so (-DPCLIP_TRACE=1) stamps s_memtime at the phase
boundaries of wave 0 (group 0: M, barrier, C, publish wait, barrier) and wave 7 (group 1: M, waits, barrier, C, barrier)."""
import ctypes, os, sys, torch
root = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "proto-clip_amd")
l = ctypes.CDLL(os.path.join(root, "libpclip_" + (sys.argv[1] if len(sys.argv) > 1 else "trace") + ".so"))
P, I = ctypes.c_void_p, ctypes.c_int
l.pclip_gemm_f16.argtypes = [P, I, P, I, P, I, I, I, I, P, I, P, P]
l.pclip_debug_trace.argtypes = [P, I]
buf = (ctypes.c_ulonglong * 18)()
n0 = ["M: reads + DMA issue + landing of the reads", "lgkm + barrier (end M)", "C: 32 MFMAs", "publish wait (vmcnt)", "barrier (end C)"]
n1 = ["M: reads + DMA issue + landing of the reads", "lgkm + publish wait", "barrier (end M)", "C: 32 MFMAs", "barrier (end C)"]
for m, n, k in ((201728, 3072, 768), (201728, 768, 3072)):
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
    print(f"{m}x{n}x{k}: {us:.0f} us (instrumented); longest workgroup span {buf[17]} ticks")
    for wv, off, names in (("wave 0 (group 0)", 0, n0), ("wave 7 (group 1)", 8, n1)):
        tot, kt = buf[off + 6], max(buf[off + 7], 1)
        print(f"  {wv}: {tot / kt:.0f} ticks per K-step (two phases); " + ", ".join(f"{names[i]} {buf[off + i] / kt:.0f}" for i in range(5)))
