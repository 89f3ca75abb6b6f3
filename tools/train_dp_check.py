#!/usr/bin/env python3
"""Runs a few training steps under torch.distributed (launch with torchrun, any world size that fits the box): checks that the
data-parallel step leaves every rank with identical parameters and, for world 1, identical results to the non-distributed
path.  `python -m torch.distributed.run --nproc-per-node 1 --master-addr 127.0.0.1 --master-port 29611 tools/train_dp_check.py`"""
import os, sys
import numpy as np, torch, torch.distributed as dist
REPO = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, REPO); sys.path.insert(0, os.path.join(REPO, "tests"))
from golden.spec import train_inputs
from proto_clip_amd.main import make_adapter
from proto_clip_amd.train import ProtoClipTrainer, sample_epoch

def run(steps=4):
    split, cfg = train_inputs("T_fc")
    torch.manual_seed(1)
    ad = make_adapter(cfg, split.visual_memory_keys.shape[0])
    tr = ProtoClipTrainer(cfg, split.visual_memory_keys.cuda(), split.textual_memory_bank.cuda(), ad, cfg["alpha"], cfg["beta"])
    rng = np.random.RandomState(1)
    losses = []
    for i, (_, qi, ql) in enumerate(sample_epoch(tr.N, tr.K, rng)):
        if i == steps: break
        losses.append(tr.step(qi, ql)[1].item())
    return tr, losses

single, l0 = run()
dist.init_process_group("nccl", device_id=torch.device("cuda", int(os.environ.get("LOCAL_RANK", 0))))
torch.cuda.set_device(int(os.environ.get("LOCAL_RANK", 0)))
tr, l1 = run()
v = tr.visual.float()
ref = v.clone(); dist.broadcast(ref, 0)
same = torch.equal(ref, v)
if dist.get_rank() == 0:
    print("losses single", l0); print("losses dist  ", l1)
    print("ranks identical:", same, "| world", dist.get_world_size(), "| max |dist - single| on the visual bank:",
          (tr.visual.float() - single.visual.float()).abs().max().item())
dist.destroy_process_group()
