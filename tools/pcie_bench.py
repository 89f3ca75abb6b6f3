#!/usr/bin/env python3
"""The bench step with its 1024 pre-processed fp32 images arriving from PINNED HOST memory every step (616 MB): copy and compute
serialised on one stream, and double-buffered on a copy stream.  bench.py itself times device-resident inputs (its contract)."""
import os, sys, time, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench

dev = torch.device("cuda", 0)
st = bench.build_state(dev, 0, 1)
host = [st["images"].cpu().pin_memory() for _ in range(2)]
bufs = [torch.empty_like(st["images"]) for _ in range(2)]
for _ in range(2):
    bench.step(st)
torch.cuda.synchronize()
K = 8
t0 = time.perf_counter()
for i in range(K):
    bench.step(st)
torch.cuda.synchronize()
t_res = (time.perf_counter() - t0) / K
t0 = time.perf_counter()
for i in range(K):
    bufs[0].copy_(host[i & 1], non_blocking=True)
    st["images"] = bufs[0]
    bench.step(st)
torch.cuda.synchronize()
t_ser = (time.perf_counter() - t0) / K
copy_s = torch.cuda.Stream()
ev_copied = [torch.cuda.Event() for _ in range(2)]
ev_used = [torch.cuda.Event() for _ in range(2)]
with torch.cuda.stream(copy_s):
    bufs[0].copy_(host[0], non_blocking=True); ev_copied[0].record()
t0 = time.perf_counter()
for i in range(K):
    cur, nxt = i & 1, (i + 1) & 1
    with torch.cuda.stream(copy_s):                       # prefetch the next batch while this one is encoded
        if i >= 1: copy_s.wait_event(ev_used[nxt])
        bufs[nxt].copy_(host[nxt], non_blocking=True); ev_copied[nxt].record()
    torch.cuda.current_stream().wait_event(ev_copied[cur])
    st["images"] = bufs[cur]
    bench.step(st)
    ev_used[cur].record()
torch.cuda.synchronize()
t_ovl = (time.perf_counter() - t0) / K
B = bench.BATCH
print(f"resident {B / t_res:8.0f} img/s ({t_res * 1e3:.1f} ms) | host fp32 batch copied then encoded {B / t_ser:8.0f} img/s ({t_ser * 1e3:.1f} ms, "
      f"copy {616.6 / (t_ser - t_res) / 1e3:.1f} GB/s) | copy of the next batch under the encode {B / t_ovl:8.0f} img/s ({t_ovl * 1e3:.1f} ms)")
