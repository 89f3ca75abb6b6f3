#!/usr/bin/env python3
"""Times the episodic training step (proto_clip_amd/train.py) at ImageNet / FewSOL sizes on synthetic banks:
one epoch of the reference's episode schedule, per-step wall time and the split of one step by kernel family."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import synth
from proto_clip_amd.main import make_adapter
from proto_clip_amd.train import ProtoClipTrainer, sample_epoch

for name, N, K, D, kind in (("ImageNet conv-3x", 1000, 16, 512, "conv-3x"), ("ImageNet fc", 1000, 16, 512, "fc"),
                             ("FewSOL-198 ViT-L fc", 198, 16, 768, "fc")):
    split = synth.make_split(N, K, D, 8, 8, seed=1, sigma=4.0)
    cfg = dict(shots=K, lr=1e-3, train_epoch=1, adapter=kind, train_vis_mem_only=False, losses=["L1", "L2", "L3"])
    torch.manual_seed(1)
    tr = ProtoClipTrainer(cfg, split.visual_memory_keys.cuda(), split.textual_memory_bank.cuda(), make_adapter(cfg, D), 0.5, 12.0)
    eps = [(qi, ql) for _, qi, ql in sample_epoch(N, K, np.random.RandomState(1))]
    tr.step(*eps[0]); torch.cuda.synchronize()
    dts = []
    for _ in range(3):                                   # wall clock of a 4-step epoch: the median of three (one host stall used to decide the line)
        t0 = time.perf_counter()
        for qi, ql in eps:
            tr.step(qi, ql)
        torch.cuda.synchronize()
        dts.append(time.perf_counter() - t0)
    dt = sorted(dts)[1]
    q = sum(len(ql) for _, ql in eps)
    print(f"{name:22s} {len(eps)} episodes/epoch, {q} queries: {1e3 * dt / len(eps):7.2f} ms/step, {q / dt:9.0f} queries/s", flush=True)
