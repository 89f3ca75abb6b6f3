import os, sys, torch
sys.path.insert(0, os.getcwd())
from proto_clip_amd import ops
sys.path.insert(0, os.path.join(os.getcwd(), "tools"))
from kernel_bench import timeit
nrm = torch.nn.functional.normalize
def gpu_time(fn, reps=10):
    fn(); torch.cuda.synchronize()
    side = torch.cuda.Stream(); side.wait_stream(torch.cuda.current_stream())
    with torch.cuda.stream(side):
        fn()
    torch.cuda.current_stream().wait_stream(side); torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    return timeit(g.replay, iters=20) / reps
for name, N, K, D, Q in (("OxfordPets", 37, 16, 512, 3669), ("DTD", 47, 16, 512, 1692), ("Caltech-101", 100, 16, 1024, 2465), ("Food-101", 101, 16, 512, 30300), ("Cars-196", 196, 16, 512, 8041),
                         ("FewSOL-198", 198, 16, 768, 666), ("SUN397", 397, 16, 512, 19850), ("ImageNet 10k", 1000, 16, 512, 10000), ("ImageNet 25k", 1000, 16, 512, 25000)):
    g = torch.Generator(device="cuda").manual_seed(1)
    cen = torch.randn(N, D, device="cuda", generator=g)
    y = torch.randint(0, N, (Q,), device="cuda", generator=g)
    q = nrm(cen[y] + 0.8 * torch.randn(Q, D, device="cuda", generator=g), dim=-1).half()
    zi = ops.proto_build(nrm(cen.repeat_interleave(K, 0) + 0.8 * torch.randn(N * K, D, device="cuda", generator=g), dim=-1).half(), N, K)
    zt = nrm(cen + 0.5 * torch.randn(N, D, device="cuda", generator=g), dim=-1).half()
    f = lambda: ops.classify(q, zi, zt, 0.5, 12.0, want_p=False, want_argmax=True)
    with ops.classify_two_stage():
        t2 = gpu_time(f)
    with ops.classify_fused():
        ops.classify_panel_stats(reset=True); f(); st = ops.classify_panel_stats()
        tf = gpu_time(f)
    td = gpu_time(f)
    print(f"{name:14s} N={N:4d} D={D:4d} Q={Q:6d}: two stages {t2*1e6:7.1f} us | fused forced {tf*1e6:7.1f} us (second pass {st[1]}/{st[0]}) | default routing {td*1e6:7.1f} us", flush=True)
