#!/usr/bin/env python3
"""Does s_memtime tick at the shader clock, and what IS the shader clock under the c_fc GEMM?  (tools/probe/clock_probe.hip.)
The probe wave (a dependent v_fma_f32 chain of fixed length) runs (a) on an idle chip, (b) while a train of c_fc-shaped GEMM
launches (M = 201 728, N = 3072, K = 768, the bench's act-8 shape) on N(0,1) operands occupies the other CUs, (c) the same on
zero-filled operands.  Printed per case: s_memtime ticks and 100 MHz reference ticks per instruction, their ratio (= the
s_memtime rate in MHz), the SMI shader clock / socket power sampled meanwhile, and the GEMM rate.
    python tools/clock_probe.py > gpurun_out/clock_probe.txt"""
import ctypes, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from proto_clip_amd import ops
from proto_clip_amd.telemetry import Sampler

lib = ctypes.CDLL(os.path.join(ROOT, "tools", "probe", "libclock_probe.so"))
lib.clock_probe_launch.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_void_p]
M, N, K = 201728, 3072, 768
probe_stream, gemm_stream = torch.cuda.Stream(), torch.cuda.Stream()
out = torch.zeros(4, dtype=torch.int64, device="cuda")


def probe(iters):
    with torch.cuda.stream(probe_stream):
        rc = lib.clock_probe_launch(ctypes.c_void_p(out.data_ptr()), iters, ctypes.c_void_p(probe_stream.cuda_stream))
        assert rc == 0


def report(tag, extra=""):
    probe_stream.synchronize()
    t, r, n, _ = out.tolist()
    print(f"{tag:34s} s_memtime {t / n:7.4f} ticks/instr | 100 MHz ref {r / n * 10:7.4f} ns/instr | s_memtime rate {t / (r / 100.0):8.1f} MHz {extra}", flush=True)
    return t / n, r / n * 10


# calibration of the probe length: ~30 ms of chain on an idle chip
probe(20000); probe_stream.synchronize()
ITERS = 1_000_000
for rep in range(3):
    with Sampler(period=0.02) as s:
        probe(ITERS)
        probe_stream.synchronize()
    sm = s.summary()
    report(f"idle chip (rep {rep})", f"| smi {sm.get('sclk_mhz')} {sm.get('power_w')}")

for data in ("normal", "zero", "normal"):
    a = torch.randn(M, K, device="cuda").half()
    w = (torch.randn(N, K, device="cuda") * K ** -0.5).half()
    bias = torch.randn(N, device="cuda").half()
    if data == "zero":
        a.zero_(); w.zero_(); bias.zero_()
    y = torch.empty(M, N, device="cuda", dtype=torch.float16)
    with torch.cuda.stream(gemm_stream):
        for _ in range(5):
            ops.gemm(a, w, bias, 1, None, y)
    gemm_stream.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    NL = 120                                       # ~130 ms of GEMM launches; the probe (~30-40 ms) starts after the first few
    with Sampler(period=0.02) as s:
        with torch.cuda.stream(gemm_stream):
            e0.record()
            for i in range(NL):
                ops.gemm(a, w, bias, 1, None, y)
                if i == 10:
                    probe(ITERS)                    # enqueued on its own stream while the GEMM queue is ~10 launches deep
            e1.record()
        gemm_stream.synchronize(); probe_stream.synchronize()
    ms = e0.elapsed_time(e1)
    sm = s.summary()
    report(f"beside c_fc GEMM train ({data})", f"| GEMM {2.0 * M * N * K * NL / ms / 1e9:6.0f} TFLOP/s, {ms / NL * 1e3:6.0f} us/launch | smi {sm.get('sclk_mhz')} {sm.get('power_w')}")
    del a, w, y
