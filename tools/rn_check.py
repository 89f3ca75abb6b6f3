import os, sys, time, torch
sys.path.insert(0, "/root/repo")
from proto_clip_amd.clip.model import BACKBONES, build_model, random_state_dict
for name, B in (("RN50", 256), ("RN101", 256), ("RN50", 256)):
    kw = BACKBONES[name]
    model = build_model(random_state_dict(seed=1, **kw)).cuda()
    x = torch.randn(B, 3, 224, 224, device="cuda")
    with torch.no_grad():
        for _ in range(3): model.encode_image(x)
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            t0 = time.perf_counter(); model.encode_image(x); torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(name, "ms per pass:", " ".join(f"{1e3*t:.2f}" for t in ts), flush=True)
