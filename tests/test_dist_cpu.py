"""N>1 path on CPU: world_size-2 (and 3) `gloo` processes run the sharded prototype reduction and the
counter all-reduce of proto_clip_amd.dist with the ORACLE's restatement injected for the per-rank
arithmetic (the HIP kernels need a GPU; what is under test here is the sharding, the single all-gather
payload layout, the rank-order combine and that every rank ends with identical prototypes)."""
import os
import sys

import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

from conftest import REPO


def _worker(rank, world, port, N, K, D, ret):
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import proto_oracle as po
        from proto_clip_amd import synth
        from proto_clip_amd.dist import allreduce_counts, shard_bounds, sharded_prototypes
        split = synth.make_split(N, K, D, 8, 8, seed=4, sigma=3.0)
        rows = split.visual_memory_keys.t().contiguous()
        rows = (rows.float() * 1.3).half()                      # un-normalised, like a learned bank
        labels = torch.arange(N).repeat_interleave(K).int()
        lo, hi = shard_bounds(N * K, rank, world)
        fin = lambda s, c, fp32_out=False: po.proto_finalize(s, c, fp32=fp32_out)
        proto = sharded_prototypes(rows[lo:hi], labels[lo:hi], N, partial_fn=po.partial_sums, finalize_fn=fin)
        single = po.proto_build(rows, N, K)
        # identical on every rank, and identical to the unsharded reduction
        gathered = [torch.empty_like(proto) for _ in range(world)]
        dist.all_gather(gathered, proto)
        same = all(torch.equal(g, gathered[0]) for g in gathered)
        correct = torch.tensor([[rank + 1, 2], [3, 4 * (rank + 1)]], dtype=torch.int32)
        tot, n = allreduce_counts(correct, 10 + rank)
        if rank == 0:
            ret["same"] = same
            ret["equal_single"] = torch.equal(proto, single)
            ret["max_diff"] = (proto.float() - single.float()).abs().max().item()
            ret["counts"] = tot.tolist()
            ret["n"] = n
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world,N,K", [(2, 10, 16), (3, 7, 5), (2, 5, 1)])
def test_sharded_prototypes_gloo(world, N, K):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 29500 + (os.getpid() % 2000) + world
    mp.spawn(_worker, args=(world, port, N, K, 64, ret), nprocs=world, join=True)
    assert ret["same"], "ranks disagree on the prototypes"
    assert ret["equal_single"], f"sharded != single-process reduction (max diff {ret['max_diff']})"
    s = sum(range(1, world + 1))
    assert ret["counts"] == [[s, 2 * world], [3 * world, 4 * s]]
    assert ret["n"] == sum(10 + r for r in range(world))
