#!/usr/bin/env python3
"""Conv adapter forward on cached features (the reference's per-epoch validation pass): rows/s at ImageNet size."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd.model import Adapter
for D, kind in ((512, "conv-3x"), (512, "conv-2x"), (768, "conv-3x"), (1024, "conv-3x")):
    torch.manual_seed(0)
    ad = Adapter(D, kind, dtype=torch.half).cuda()
    x = torch.nn.functional.normalize(torch.randn(50000, D, device="cuda"), dim=-1).half()
    with torch.no_grad():
        y = ad(x, l2norm_out=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5): y = ad(x, l2norm_out=True)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 5
    print(f"D={D} {kind}: 50000 rows {dt*1e3:6.2f} ms = {50000/dt/1e6:5.2f} M rows/s  checksum {y.float().sum().item():.4f} {y.view(torch.int16).to(torch.int64).sum().item()}", flush=True)
