#!/usr/bin/env python3
"""Shader clock + socket power (proto_clip_amd.telemetry, >= 10 Hz) over (a) 5 s of back-to-back c_fc-shaped GEMM launches and
(b) the bench's step loop, on N(0,1) operands / images and on the bench's own data.
    python tools/power_trace.py [steps] > gpurun_out/power_trace.txt"""
import json, os, sys, time, torch
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import bench
from proto_clip_amd import ops
from proto_clip_amd.telemetry import Sampler

steps = int(sys.argv[1]) if len(sys.argv) > 1 else 200
dev = torch.device("cuda", 0)
st = bench.build_state(dev, 0, 1)
M, N, K = 1024 * 197, 3072, 768

# operands of the c_fc launch: N(0,1) and what the encoder really feeds it (the residual stream after block 5 of the bench model)
cases = {}
cases["c_fc, N(0,1) activations x N(0, 1/K) weights"] = (torch.randn(M, K, device=dev).half(), (torch.randn(N, K, device=dev) * K ** -0.5).half())
with torch.no_grad():
    blk = st["model"].visual.transformer.resblocks[5]
    real = ops.gemm
    grabbed = {}
    def grab(a, w, bias=None, act=0, residual=None, out=None):
        if a.shape == (M, K) and w.shape[0] == N and "a" not in grabbed: grabbed["a"] = a.clone()
        return real(a, w, bias, act, residual, out)
    ops.gemm = grab
    try:
        bench.step(st)
    finally:
        ops.gemm = real
    torch.cuda.synchronize()
if "a" in grabbed:
    cases["c_fc, the bench's own operands (residual stream x folded c_fc weight of block 0)"] = (grabbed["a"], grabbed.get("w", blk.mlp.c_fc.weight))
bias = torch.randn(N, device=dev).half()
y = torch.empty(M, N, device=dev, dtype=torch.float16)
for name, (a, w) in cases.items():
    for _ in range(5): ops.gemm(a, w, bias, 1, None, y)
    torch.cuda.synchronize()
    n = 0
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    with Sampler(period=0.05, skip_s=1.0) as s:
        t0 = time.perf_counter(); e0.record()
        while time.perf_counter() - t0 < 5.0:
            for _ in range(40): ops.gemm(a, w, bias, 1, None, y)
            torch.cuda.synchronize(); n += 40
        e1.record(); torch.cuda.synchronize()
    ms = e0.elapsed_time(e1)
    print(json.dumps({"case": name, "us_per_launch": ms / n * 1e3, "tflops": 2.0 * M * N * K * n / ms / 1e9, **s.summary()}), flush=True)

# the bench step loop on its own images and on N(0,1) images
for name, imgs in (("bench step loop, bench images", st["images"]), ("bench step loop, N(0,1) images", torch.randn_like(st["images"]))):
    st2 = dict(st, images=imgs)
    for _ in range(3): bench.step(st2)
    torch.cuda.synchronize()
    with Sampler(period=0.05, skip_s=1.0) as s:
        t0 = time.perf_counter()
        for _ in range(steps): bench.step(st2)
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    print(json.dumps({"case": name, "steps": steps, "ms_per_step": dt / steps * 1e3, "img_per_s": steps * bench.BATCH / dt, **s.summary()}), flush=True)
