"""C2 (EuroSAT 16-shot) latency probe: prototype build + classification, eager and under hipGraph replay.
Run under `rocprofv3 --kernel-trace --stats` to split device time into kernel durations and launch gaps."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from proto_clip_amd import ops


def main():
    dev = torch.device("cuda:0")
    N, K, D, Q = 10, 16, 512, 8100
    g = torch.Generator(device=dev).manual_seed(3)
    nrm = torch.nn.functional.normalize
    mem = nrm(torch.randn(N * K, D, device=dev, generator=g), dim=-1).half()
    q = nrm(torch.randn(Q, D, device=dev, generator=g), dim=-1).half()
    zt = nrm(torch.randn(N, D, device=dev, generator=g), dim=-1).half()
    zi = ops.proto_build(mem, N, K)

    def both():
        z = ops.proto_build(mem, N, K)
        return ops.classify(q, z, zt, 1.0, 0.7, want_p=False, want_argmax=True)[1]

    one = q[:1].contiguous()
    fns = {"trivial (row_sqnorm of one row)": lambda: ops.row_sqnorm(one),
           "proto_build": lambda: ops.proto_build(mem, N, K),
           "classify": lambda: ops.classify(q, zi, zt, 1.0, 0.7, want_p=False, want_argmax=True),
           "both": both}
    for name, fn in fns.items():
        fn()
        torch.cuda.synchronize()
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            fn()
        torch.cuda.current_stream().wait_stream(side)
        torch.cuda.synchronize()
        gr = torch.cuda.CUDAGraph()
        with torch.cuda.graph(gr):
            for _ in range(20):
                fn()
        gr.replay()
        torch.cuda.synchronize()
        ts = []
        for _ in range(7):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                gr.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1) * 1e3 / 400)
        ts.sort()
        print(f"{name:32s} graph replay: median {ts[3]:7.2f} us   min {ts[0]:7.2f}   max {ts[-1]:7.2f}", flush=True)


if __name__ == "__main__":
    main()
