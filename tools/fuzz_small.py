#!/usr/bin/env python3
"""Differential fuzz of the one-launch classifications (N <= 32: classify_small; 32 < N <= 256: classify_mid, forced for every shape it can run): random
(Q, N, K, D, alpha, beta) and data regimes against the oracle's P (p within 1e-5, argmax equal unless the oracle's top two tie to 1e-6) and against the two
stages (pclip_sqdist_f16 + pclip_fuse_probs: p within 2e-6) where those apply (D % 64 == 0).
    python tools/fuzz_small.py [cases] [seed]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import ops
from oracle import proto_oracle as po

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 60
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
nrm = torch.nn.functional.normalize
bad = 0
for it in range(cases):
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    N, K, Q = (ri(1, 32) if it % 2 else ri(33, 256)), ri(1, 20), ri(1, 9000 if it % 7 == 0 else 700)
    D = 32 * ri(1, 32) if N <= 32 else 128 * ri(1, 8)                      # N > 32: the one-launch kernel takes D % 128 == 0 (other widths: two stages, D % 64 == 0)
    regime = ri(0, 3)
    cen = torch.randn(N, D, generator=g)
    y = torch.randint(0, N, (Q,), generator=g)
    mem = cen.repeat_interleave(K, 0) + 0.8 * torch.randn(N * K, D, generator=g)
    zt = cen + 0.5 * torch.randn(N, D, generator=g)
    q = cen[y] + 0.8 * torch.randn(Q, D, generator=g) if regime != 1 else torch.randn(Q, D, generator=g)
    if regime != 2: zt, q = nrm(zt, dim=-1), nrm(q, dim=-1)
    if regime == 3 and N > 2:
        mem[K:2 * K] = mem[:K]; zt[1] = zt[0]                               # classes 0 and 1 identical: exact ties
    mem, zt, q = mem.half(), zt.half(), q.half()
    alpha = [0.0, 1.0, 0.5, float(torch.rand(1, generator=g))][ri(0, 3)]
    beta = [0.1, 1.0, 12.0, 20.0, float(20 * torch.rand(1, generator=g))][ri(0, 4)]
    k = min(3, N)
    zi = ops.proto_build(mem.cuda(), N, K)
    with ops.classify_mid(2):
        p, am, _, _ = ops.classify(q.cuda(), zi, zt.cuda(), alpha, beta, want_p=True, want_argmax=True)
    same = True
    if D % 64 == 0:
        with ops.classify_two_stage():
            p1, am1, _, _ = ops.classify(q.cuda(), zi, zt.cuda(), alpha, beta, want_p=True, want_argmax=True)
        same = (p - p1).abs().max().item() <= (2e-6 if regime != 2 else 1e-5)              # un-normalised rows: larger norms, larger fp32 cancellation
    p_or = po.P(q, zi.cpu(), zt, alpha, beta)
    err = (p.cpu() - p_or).abs().max().item()
    top2 = p_or.double().topk(min(2, N), dim=1).values
    clear = (top2[:, 0] - top2[:, -1] > 1e-6) if N > 1 else torch.ones(Q, dtype=torch.bool)
    arg_ok = torch.equal(am.cpu().long()[clear], p_or.max(1)[1][clear]) and torch.equal(p.cpu().max(1)[1], am.cpu().long())
    if not (same and err <= 1e-5 and arg_ok):
        bad += 1
        print(f"case {it}: N={N} K={K} D={D} Q={Q} regime={regime} alpha={alpha:.3f} beta={beta:.3f}: one launch ~ two stages {same}, max|p - oracle| {err:.2e}, argmax ok {arg_ok}", flush=True)
print(f"{cases} cases, failures in {bad}")
sys.exit(1 if bad else 0)
