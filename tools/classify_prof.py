import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import ops
nrm = torch.nn.functional.normalize
N, K, D, Q = 1000, 16, 512, 50000
g = torch.Generator(device="cuda").manual_seed(1)
cen = torch.randn(N, D, device="cuda", generator=g)
y = torch.randint(0, N, (Q,), device="cuda", generator=g)
q = nrm(cen[y] + 0.8 * torch.randn(Q, D, device="cuda", generator=g), dim=-1).half()
zi = ops.proto_build(nrm(cen.repeat_interleave(K, 0) + 0.8 * torch.randn(N * K, D, device="cuda", generator=g), dim=-1).half(), N, K)
zt = nrm(cen + 0.5 * torch.randn(N, D, device="cuda", generator=g), dim=-1).half()
for _ in range(12):
    ops.classify(q, zi, zt, 0.5, 12.0, want_p=False, want_argmax=True)
with ops.classify_two_stage():
    for _ in range(12):
        ops.classify(q, zi, zt, 0.5, 12.0, want_p=False, want_argmax=True)
torch.cuda.synchronize()
