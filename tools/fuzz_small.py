#!/usr/bin/env python3
"""Differential fuzz of the small-class-count path (N <= 32: classify_small, one launch) and of the prototype-build + classification launch: random (Q, N, K, D, alpha, beta)
and data regimes against the oracle's P (p within 1e-5, argmax equal unless the oracle's top two tie to 1e-6) and against proto_build + classify (bit for bit).
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
    N, K, D, Q = ri(1, 32), ri(1, 20), 32 * ri(1, 32), ri(1, 9000 if it % 7 == 0 else 700)
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
    p, am, tp, ti = ops.classify(q.cuda(), zi, zt.cuda(), alpha, beta, want_p=True, want_argmax=True, topk=k)
    z1, p1, am1, tp1, ti1 = ops.proto_classify(mem.cuda(), N, K, q.cuda(), zt.cuda(), alpha, beta, want_p=True, want_argmax=True, topk=k, one_launch=True)
    same = torch.equal(zi, z1) and torch.equal(p, p1) and torch.equal(am, am1) and torch.equal(tp, tp1) and torch.equal(ti, ti1)
    p_or = po.P(q, zi.cpu(), zt, alpha, beta)
    err = (p.cpu() - p_or).abs().max().item()
    top2 = p_or.double().topk(min(2, N), dim=1).values
    clear = (top2[:, 0] - top2[:, -1] > 1e-6) if N > 1 else torch.ones(Q, dtype=torch.bool)
    arg_ok = torch.equal(am.cpu().long()[clear], p_or.max(1)[1][clear]) and torch.equal(p.cpu().max(1)[1], am.cpu().long())
    if not (same and err <= 1e-5 and arg_ok):
        bad += 1
        print(f"case {it}: N={N} K={K} D={D} Q={Q} regime={regime} alpha={alpha:.3f} beta={beta:.3f}: one launch == two calls {same}, max|p - oracle| {err:.2e}, argmax ok {arg_ok}", flush=True)
print(f"{cases} cases, failures in {bad}")
sys.exit(1 if bad else 0)
