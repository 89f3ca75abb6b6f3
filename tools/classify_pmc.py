#!/usr/bin/env python3
"""The ImageNet test split through the fused row-panel classification and through the two stages it replaces, a few times each (rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE:
tools/gpu_r5.sh) — algorithmic traffic 53.4 MB."""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import ops
N, D, Q = 1000, 512, 50000
nrm = torch.nn.functional.normalize
zi = nrm(torch.randn(N, D, device="cuda"), dim=-1).half()
zt = nrm(torch.randn(N, D, device="cuda"), dim=-1).half()
q = nrm(torch.randn(Q, D, device="cuda"), dim=-1).half()
for _ in range(3):
    ops.classify(q, zi, zt, 0.5, 12.0, want_p=False, want_argmax=True)
    with ops.classify_two_stage():
        ops.classify(q, zi, zt, 0.5, 12.0, want_p=False, want_argmax=True)
torch.cuda.synchronize()
