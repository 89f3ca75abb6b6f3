#!/usr/bin/env python3
"""ModifiedResNet image tower: images per pass (PCLIP_RN_CHUNK) against throughput, 1024 images per call."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd.clip.model import BACKBONES, build_model, random_state_dict
for name in ("RN50", "RN101"):
    kw = BACKBONES[name]
    model = build_model(random_state_dict(seed=1, **kw)).cuda()
    x = torch.randn(1024, 3, kw["image_resolution"], kw["image_resolution"], device="cuda")
    for chunk in (128, 256, 512, 1024):
        model.visual.chunk = chunk
        with torch.no_grad():
            for _ in range(2): model.encode_image(x)
            torch.cuda.synchronize()
            t0 = time.perf_counter(); n = 5
            for _ in range(n): model.encode_image(x)
            torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / n
        print(f"{name:6s} 1024 images in passes of {chunk:5d}: {1e3 * dt:7.1f} ms  {1024 / dt:8.0f} img/s", flush=True)
    del model, x; torch.cuda.empty_cache()
