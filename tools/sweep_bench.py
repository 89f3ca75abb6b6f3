import os, sys, torch, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import ops
from oracle import proto_oracle as po
import numpy as np
for Q, N, D in ((50000, 1000, 512), (8100, 10, 512), (666, 198, 768)):
    q = torch.nn.functional.normalize(torch.randn(Q, D, device="cuda"), dim=-1).half()
    zi = torch.nn.functional.normalize(torch.randn(N, D, device="cuda"), dim=-1).half()
    zt = torch.nn.functional.normalize(torch.randn(N, D, device="cuda"), dim=-1).half()
    labels = torch.randint(0, N, (Q,), device="cuda")
    al, be = po.hp_grid()
    def run():
        d2i, d2t, _ = ops.sqdist(q, zi, zt)
        return ops.hp_sweep(d2i, d2t, N, labels, al, be)
    run(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(5): c = run()
    torch.cuda.synchronize()
    print(f"Q={Q} N={N}: grid search {len(al)}x{len(be)} pairs: {(time.perf_counter()-t0)/5*1e3:.2f} ms")
