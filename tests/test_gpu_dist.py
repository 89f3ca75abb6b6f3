"""The N>1 path over RCCL (torch.distributed backend "nccl"): one process per GPU runs `dist.sharded_prototypes` (per-rank
fp32 class sums -> ONE all-gather -> rank-ordered combine) and the data-parallel training step (queries sharded, ONE flat
all-reduce of the query-dependent gradients).  World 1 always runs (RCCL initialises and carries the collectives on one GPU);
world 2 needs two GPUs and is skipped on a one-GPU box — RCCL refuses two ranks on one device.  The CPU twin of this test
(tests/test_dist_cpu.py, gloo, world 2 / 3) covers the decomposition logic everywhere."""
import os
import sys

import pytest
import torch
import torch.multiprocessing as mp

from conftest import REPO

pytestmark = pytest.mark.gpu


def _worker(rank, world, port, ret):
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK=str(rank))
    os.environ.setdefault("HSA_ENABLE_IPC_MODE_LEGACY", "0")
    import numpy as np
    import torch.distributed as dist
    torch.cuda.set_device(rank)
    from golden.spec import train_inputs
    from proto_clip_amd import ops, synth
    from proto_clip_amd.dist import allreduce_counts, shard_bounds, sharded_prototypes
    from proto_clip_amd.main import make_adapter
    from proto_clip_amd.train import ProtoClipTrainer, sample_epoch

    def train(steps=3):
        split, cfg = train_inputs("T_fc")
        torch.manual_seed(1)
        ad = make_adapter(cfg, split.visual_memory_keys.shape[0])
        tr = ProtoClipTrainer(cfg, split.visual_memory_keys.cuda(), split.textual_memory_bank.cuda(), ad, cfg["alpha"], cfg["beta"])
        losses = []
        for i, (_, qi, ql) in enumerate(sample_epoch(tr.N, tr.K, np.random.RandomState(1))):
            if i == steps:
                break
            losses.append(tr.step(qi, ql)[1].item())
        return tr, losses

    single, l_single = train()                               # before init_process_group: the non-distributed step
    N, K, D = 1000, 16, 512
    split = synth.make_split(N, K, D, 8, 8, seed=4, sigma=3.0)
    rows = (split.visual_memory_keys.t().contiguous().float() * 1.3).half().cuda()
    labels = torch.arange(N).repeat_interleave(K).int().cuda()
    proto_single = ops.proto_build(rows, N, K)
    dist.init_process_group("nccl", rank=rank, world_size=world, device_id=torch.device("cuda", rank))
    try:
        lo, hi = shard_bounds(N * K, rank, world)
        proto = sharded_prototypes(rows[lo:hi], labels[lo:hi], N)
        ref = proto.clone()
        dist.broadcast(ref, 0)
        # the side-stream exchange of bench.py's step (dist.PrototypeExchange): launched, other work on the main stream, joined
        from proto_clip_amd.dist import PrototypeExchange
        ex = PrototypeExchange()
        side_ok = True
        for _ in range(3):
            ex.launch(rows[lo:hi], labels[lo:hi], N)
            busy = ops.l2norm_rows(rows)                    # main-stream work while the all-gather is in flight
            side_ok = side_ok and torch.equal(ex.result(), proto)
        del busy
        tot, n = allreduce_counts(torch.tensor([[rank + 1, 2]], dtype=torch.int32, device="cuda"), 10 + rank)
        tr, l_dist = train()
        v = tr.visual.clone()
        dist.broadcast(v, 0)
        flat = torch.cat([p.detach().reshape(-1).float() for p in tr.adapter.parameters()])
        f0 = flat.clone()
        dist.broadcast(f0, 0)
        ok = torch.tensor([int(torch.equal(ref, proto)), int(torch.equal(proto, proto_single)), int(torch.equal(v, tr.visual)),
                           int(torch.equal(f0, flat)), int(side_ok)], device="cuda")
        dist.all_reduce(ok, op=dist.ReduceOp.MIN)
        if rank == 0:
            ret["world"] = dist.get_world_size()
            ret["backend"] = dist.get_backend()
            ret["ranks_identical_protos"], ret["equal_single_gpu"], ret["ranks_identical_bank"], ret["ranks_identical_adapter"], ret["side_stream_exchange"] = [bool(x) for x in ok.tolist()]
            ret["counts"], ret["n"] = tot.tolist(), n
            ret["losses"] = (l_single, l_dist)
            ret["bank_diff"] = (tr.visual.float() - single.visual.float()).abs().max().item()
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [1, 2])
def test_sharded_prototypes_and_dp_step_over_rccl(world):
    if torch.cuda.device_count() < world:
        pytest.skip(f"needs {world} GPUs, this box has {torch.cuda.device_count()} (RCCL refuses two ranks on one device)")
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 33500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, ret), nprocs=world, join=True)
    assert ret["world"] == world and ret["backend"] == "nccl"
    assert ret["ranks_identical_protos"] and ret["equal_single_gpu"], "sharded prototypes differ between ranks / from the one-GPU kernel"
    assert ret["ranks_identical_bank"] and ret["ranks_identical_adapter"], "data-parallel step left the ranks with different parameters"
    assert ret["side_stream_exchange"], "PrototypeExchange (all-gather on the side stream) != the in-line exchange"
    s = sum(range(1, world + 1))
    assert ret["counts"] == [[s, 2 * world]] and ret["n"] == sum(10 + r for r in range(world))
    l_single, l_dist = ret["losses"]
    for a, b in zip(l_single, l_dist):                      # fp32 summation order of the gradient all-reduce only
        assert abs(a - b) <= 1e-4 * max(1.0, abs(a)), (l_single, l_dist)
    if world == 1:
        assert l_single == l_dist and ret["bank_diff"] == 0.0          # one rank through RCCL == no RCCL, bit for bit
