import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from proto_clip_amd import ops
for name, B, L, H, causal in (("ViT-B/16", 1024, 197, 12, False), ("text", 7000, 77, 8, True)):
    qkv = torch.randn(B * L, 3 * H * 64, device="cuda").half()
    for _ in range(3): out = ops.attention(qkv, B, L, H, causal=causal)
torch.cuda.synchronize()
