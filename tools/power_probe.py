#!/usr/bin/env python3
"""Samples GPU clock / power (rocm-smi) while the big-tile GEMM runs back-to-back on zero-filled, N(0,1) and
encoder-like operands: shows whether the fp16 MFMA loop is clock-throttled by data-dependent power."""
import os, subprocess, sys, threading, time, re, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import ops

def sample(stop, out):
    while not stop.is_set():
        try:
            txt = subprocess.run(["rocm-smi", "-c", "-P", "--showuse"], capture_output=True, text=True, timeout=5).stdout
            sclk = re.search(r"sclk clock level:?\s*\d*:?\s*\(?(\d+)Mhz", txt)
            pw = re.search(r"Power \(W\):\s*([\d.]+)", txt)
            out.append((int(sclk.group(1)) if sclk else -1, float(pw.group(1)) if pw else -1.0))
        except Exception as e:
            out.append((-2, -2.0))
        time.sleep(0.2)

m = n = k = 8192
for data in ("idle", "zero", "normal", "normal_x1e-3"):
    a = torch.randn(m, k, device="cuda").half(); w = torch.randn(n, k, device="cuda").half()
    if data == "zero": a.zero_(); w.zero_()
    if data == "normal_x1e-3": a.mul_(1e-3); w.mul_(1e-3)
    out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    stop, samples = threading.Event(), []
    th = threading.Thread(target=sample, args=(stop, samples)); th.start()
    t0 = time.time(); it = 0
    if data == "idle":
        time.sleep(3)
    else:
        while time.time() - t0 < 4:
            for _ in range(50): ops.gemm(a, w, None, 0, None, out)
            torch.cuda.synchronize(); it += 50
    dt = time.time() - t0
    stop.set(); th.join()
    good = [s for s in samples if s[0] > 0]
    tf = 2.0 * m * n * k * it / dt / 1e12
    print(f"{data:13s} {tf:7.0f} TFLOP/s  sclk MHz {[s[0] for s in good][-8:]}  power W {[s[1] for s in good][-8:]}", flush=True)
