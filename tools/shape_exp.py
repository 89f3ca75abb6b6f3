import os, sys, torch
sys.path.insert(0, os.environ.get("GRAFT_REPO_ROOT", "/root/repo"))
from proto_clip_amd import ops
sys.path.insert(0, os.path.join(os.environ.get("GRAFT_REPO_ROOT", "/root/repo"), "tools"))
from kernel_bench import timeit
for (m, n, k) in [(4096, 2304, 768), (8192, 2304, 768), (16384, 2304, 768), (50432, 2304, 768), (50432, 2304, 1536), (50432, 2304, 3072),
                  (8192, 2304, 3072), (8192, 2304, 8192), (50432, 2304, 8192), (2048, 2048, 768), (2048, 2048, 8192)]:
    a = torch.randn(m, k, device="cuda").half(); w = (torch.randn(n, k, device="cuda") * k ** -0.5).half()
    out = torch.empty(m, n, device="cuda", dtype=torch.float16)
    t = timeit(lambda: ops.gemm(a, w, None, 0, None, out), iters=30)
    print(f"M={m} N={n} K={k}: {t*1e6:8.1f} us {2.0*m*n*k/t/1e12:7.1f} TF  (A {m*k*2/1e6:.0f} MB)", flush=True)
