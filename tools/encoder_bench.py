#!/usr/bin/env python3
"""encode_image throughput of every backbone the reference lists (random-init weights, synthetic images)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd.clip.model import BACKBONES, build_model, random_state_dict
GF = {"ViT-B/32": 8.8, "ViT-B/16": 35.1, "ViT-L/14": 162.0, "RN50": 12.2, "RN101": 19.6}     # GFLOP per image
for name, B in (("ViT-B/32", 1024), ("ViT-B/16", 1024), ("ViT-L/14", 512), ("RN50", 256), ("RN50", 1024), ("RN101", 256), ("RN101", 1024)):
    kw = BACKBONES[name]
    model = build_model(random_state_dict(seed=1, **kw)).cuda()
    x = torch.randn(B, 3, kw["image_resolution"], kw["image_resolution"], device="cuda")
    with torch.no_grad():
        for _ in range(3): model.encode_image(x)          # (one warm-up pass was not enough: RN50 once read 9.7 ms instead of 6.9, profiles/r04_ab_attention16.txt)
        torch.cuda.synchronize()
        t0 = time.perf_counter(); n = 10
        for _ in range(n): model.encode_image(x)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
    print(f"{name:9s} batch {B:5d}: {1e3*dt:8.1f} ms  {B/dt:9.0f} img/s  {B/dt*GF[name]/1e3:7.0f} TFLOP/s-equivalent", flush=True)
    del model, x; torch.cuda.empty_cache()
