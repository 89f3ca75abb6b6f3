#!/usr/bin/env python3
"""Differential fuzz of the argmax classification routings: fused row panels (one pass + candidate proof, forced for every shape it can run) against the two stages
(sqdist + fuse_probs) over random shapes, (alpha, beta), data regimes (class-structured, structureless, un-normalised, duplicated prototypes, duplicated queries, tiny
and huge scales).  A difference is accepted only where the two-stage p of that query ties its top two classes to 1e-6 (proven per query).  Prints one line per case that
differs anywhere and a summary; exit status 1 on an unproven difference.    python tools/fuzz_classify.py [cases] [seed]"""
import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import ops

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 120
g = torch.Generator().manual_seed(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
nrm = torch.nn.functional.normalize
bad = second = panels = 0
for it in range(cases):
    ri = lambda lo, hi: int(torch.randint(lo, hi + 1, (1,), generator=g))
    N = ri(33, 1100)
    D = 64 * ri(2, 16)
    Q = ri(1, 3000)
    regime = ri(0, 5)
    cen = torch.randn(N, D, generator=g)
    y = torch.randint(0, N, (Q,), generator=g)
    if regime == 1:                                                  # structureless
        zi, zt, q = torch.randn(N, D, generator=g), torch.randn(N, D, generator=g), torch.randn(Q, D, generator=g)
    else:
        zi, zt, q = cen + 0.3 * torch.randn(N, D, generator=g), cen + 0.5 * torch.randn(N, D, generator=g), cen[y] + 0.8 * torch.randn(Q, D, generator=g)
    if regime != 2:                                                  # 2: un-normalised features
        zi, zt, q = nrm(zi, dim=-1), nrm(zt, dim=-1), nrm(q, dim=-1)
    if regime == 3:                                                  # duplicated prototypes / queries: exact ties
        for _ in range(ri(1, 8)):
            a, b = ri(0, N - 1), ri(0, N - 1)
            zi[b] = zi[a]
            if ri(0, 1): zt[b] = zt[a]
        if Q > 4: q[Q // 2:] = q[:Q - Q // 2].clone()
    if regime == 4: zi, zt, q = zi * 0.01, zt * 0.01, q * 0.01         # tiny scale: every distance ~ 0
    if regime == 5: zi, zt, q = zi * 30.0, zt * 30.0, q * 30.0         # huge scale: the exponentials of all but the nearest underflow
    zi, zt, q = zi.half().cuda(), zt.half().cuda(), q.half().cuda()
    alpha = [0.0, 1.0, 0.5, float(torch.rand(1, generator=g))][ri(0, 3)]
    beta = [0.0, 0.1, 1.0, 12.0, 20.0, float(20 * torch.rand(1, generator=g))][ri(0, 5)]
    with ops.classify_fused():
        ops.classify_panel_stats(reset=True)
        am = ops.classify(q, zi, zt, alpha, beta, want_p=False, want_argmax=True)[1]
        st = ops.classify_panel_stats()
    with ops.classify_two_stage():
        am2 = ops.classify(q, zi, zt, alpha, beta, want_p=False, want_argmax=True)[1]
        p2 = ops.classify(q, zi, zt, alpha, beta, want_p=True, want_argmax=False)[0]
    panels += st[0]; second += st[1]
    diff = (am != am2).nonzero().flatten()
    if len(diff):
        top2 = p2[diff].double().topk(2, dim=1).values
        margin = (top2[:, 0] - top2[:, 1]).max().item()
        ok = margin < 1e-6
        bad += not ok
        print(f"case {it}: N={N} D={D} Q={Q} regime={regime} alpha={alpha:.3f} beta={beta:.3f}: {len(diff)} queries differ, largest top-2 margin {margin:.2e} -> {'proven ties' if ok else 'UNPROVEN'}", flush=True)
print(f"{cases} cases, {panels} panels ({second} through the second pass), unproven differences in {bad} cases")
sys.exit(1 if bad else 0)
