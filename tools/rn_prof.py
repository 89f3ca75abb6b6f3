import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd.clip.model import BACKBONES, build_model, random_state_dict
name = os.environ.get("BACKBONE", "RN50")
kw = BACKBONES[name]
model = build_model(random_state_dict(seed=1, **kw)).cuda()
x = torch.randn(int(os.environ.get("IMAGES", "256")), 3, kw["image_resolution"], kw["image_resolution"], device="cuda")
with torch.no_grad():
    for _ in range(3): model.encode_image(x)
torch.cuda.synchronize()
