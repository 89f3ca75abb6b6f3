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


def _train_worker(rank, world, port, ret):
    """Data-parallel training step (proto_clip_amd/train.py::step_features): every rank differentiates ITS slab of the
    episode's queries with the loss normalised by the TOTAL query count; one flat all-reduce (dist.allreduce_sum_) sums the
    fp32 gradients wrt the two prototype matrices and the adapter parameters.  The per-rank arithmetic is the oracle's
    autograd (the HIP kernels need a GPU); under test: the decomposition, the slab bounds and the single-message layout."""
    sys.path.insert(0, REPO)
    sys.path.insert(0, os.path.join(REPO, "tests"))
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        import numpy as np
        import torch.nn.functional as F
        from golden.spec import train_inputs
        from oracle import train_oracle as to
        from proto_clip_amd.dist import allreduce_sum_, shard_bounds
        from proto_clip_amd.train import sample_epoch
        split, cfg = train_inputs("T_fc")
        g = np.load(os.path.join(REPO, "tests", "golden", "train_T_fc.npz"))
        N, K = int(g["meta"][0]), int(g["meta"][1])
        _, q_idx, q_lab = next(iter(sample_epoch(N, K, np.random.RandomState(1))))       # same episode on every rank
        keys_rows = split.visual_memory_keys.t().contiguous()
        ad = {str(n): torch.from_numpy(g["init__" + str(n)]).clone().requires_grad_() for n in g["names"] if str(n).startswith("fc.")}

        def grads(idx, lab, q_total):
            for p in ad.values():
                p.grad = None
            z_img = F.normalize(keys_rows.view(N, K, -1).float().mean(1), dim=-1).requires_grad_()
            z_txt = F.normalize(split.textual_memory_bank.t().float(), dim=-1).requires_grad_()
            if len(idx):
                zq = to.adapter_fc(keys_rows[torch.as_tensor(idx)], ad).float()
                zq = zq / zq.norm(dim=-1, keepdim=True)
                p = to.P(zq, z_img, z_txt, cfg["alpha"], cfg["beta"])
                loss = -torch.log(p[torch.arange(len(idx)), torch.as_tensor(lab)]).sum() / q_total
                loss.backward()
            zero = lambda t: torch.zeros_like(t, dtype=torch.float32)
            out = [zero(z_img) if z_img.grad is None else z_img.grad.clone(), zero(z_txt) if z_txt.grad is None else z_txt.grad.clone()]
            out += [zero(v) if v.grad is None else v.grad.float().clone() for v in ad.values()]
            return out

        lo, hi = shard_bounds(len(q_idx), rank, world)
        mine = allreduce_sum_(grads(q_idx[lo:hi], q_lab[lo:hi], len(q_idx)))
        full = grads(q_idx, q_lab, len(q_idx))
        if rank == 0:
            ret["rel"] = [((a - b).norm() / b.norm().clamp_min(1e-20)).item() for a, b in zip(mine, full)]
            ret["slabs"] = [shard_bounds(len(q_idx), r, world) for r in range(world)]
            ret["n"] = len(q_idx)
    finally:
        dist.destroy_process_group()


@pytest.mark.parametrize("world", [2, 3])
def test_data_parallel_training_gradients_gloo(world):
    mgr = mp.Manager()
    ret = mgr.dict()
    port = 31500 + (os.getpid() % 2000) + world
    mp.spawn(_train_worker, args=(world, port, ret), nprocs=world, join=True)
    slabs = ret["slabs"]
    assert slabs[0][0] == 0 and slabs[-1][1] == ret["n"] and all(a[1] == b[0] for a, b in zip(slabs, slabs[1:]))
    # fp32 prototype gradients: summation order only; fp16 adapter gradients: each rank's autograd rounds its partial to fp16
    assert ret["rel"][0] < 1e-5 and ret["rel"][1] < 1e-5, ret["rel"]
    assert max(ret["rel"][2:]) < 2e-2, ret["rel"]


def _step_worker(rank, world, port, ret):
    """bench.py's step function (proto_clip_amd.dist.hot_path_step: side-stream prototype exchange + encoder + adapter + classify)
    on a tiny tower with the ORACLE's stages injected: every rank classifies ITS queries against prototypes reduced from ITS slab
    of the support bank; the predictions must equal the single-process run over the whole bank."""
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    if world > 1:
        dist.init_process_group("gloo", rank=rank, world_size=world)
    try:
        from oracle import clip_oracle, proto_oracle as po
        from proto_clip_amd import synth
        from proto_clip_amd.clip.model import random_state_dict
        from proto_clip_amd.dist import PrototypeExchange, hot_path_step, shard_bounds
        kw = dict(embed_dim=64, image_resolution=32, vision_layers=2, vision_width=128, vision_patch_size=8, context_length=77,
                  vocab_size=512, transformer_width=64, transformer_heads=1, transformer_layers=1)
        sd = random_state_dict(seed=11, **kw)
        N, K, D, Q = 9, 4, 64, 24
        split = synth.make_split(N, K, D, 8, 8, seed=2)
        rows = split.visual_memory_keys.t().contiguous()
        labels = torch.arange(N).repeat_interleave(K).int()
        zt = po.l2norm_rows(split.textual_memory_bank.t().contiguous())
        imgs = synth.make_images(Q, 32, seed=5, n_class=N)
        torch.manual_seed(0)
        ad_sd = {"conv1.weight": torch.randn(16, 1, 1, 1).half(), "conv2.weight": (torch.randn(16, 16, 3, 3) / 12).half(),
                 "conv3.weight": (torch.randn(1, 16, 1, 1) / 4).half()}
        for i, c in ((1, 16), (2, 16), (3, 1)):
            ad_sd[f"bn{i}.weight"], ad_sd[f"bn{i}.bias"] = torch.ones(c, 8, 8).half(), torch.zeros(c, 8, 8).half()

        class OraclePath:
            encode = staticmethod(lambda x: clip_oracle.encode_image(sd, x, half=True))
            l2norm = staticmethod(po.l2norm_rows)
            adapt = staticmethod(lambda f: po.l2norm_rows(po.adapter_conv(f, ad_sd, "conv-3x")))
            classify = staticmethod(lambda a, zi, zt_, al, be: po.P(a, zi, zt_, al, be).max(1)[1])

        fin = lambda s, c, fp32_out=False: po.proto_finalize(s, c, fp32=fp32_out)
        lo, hi = shard_bounds(N * K, rank, world)
        qlo, qhi = shard_bounds(Q, rank, world)
        am = hot_path_step(OraclePath, PrototypeExchange(), rows[lo:hi], labels[lo:hi], N, imgs[qlo:qhi], zt, 0.5, 12.0,
                           partial_fn=po.partial_sums, finalize_fn=fin)
        ret[rank] = am.tolist()
    finally:
        if world > 1:
            dist.destroy_process_group()


def test_bench_step_function_two_ranks_match_one():
    mgr = mp.Manager()
    one, two = mgr.dict(), mgr.dict()
    _step_worker(0, 1, 0, one)
    port = 33500 + (os.getpid() % 2000)
    mp.spawn(_step_worker, args=(2, port, two), nprocs=2, join=True)
    assert len(one[0]) == 24 and two[0] + two[1] == one[0]


def _bench_main_worker(rank, world, port, outdir):
    """bench.py's OWN run(): the rank / world environment of torch.distributed.run, build_state's support-set sharding, warm-up, barrier +
    max-over-ranks timing, the self-check collective and the single JSON line — on CPU over gloo with the oracle's stages behind
    proto_clip_amd.dist.hot_path_step (the first real `--gpus 8` run must not die on host logic)."""
    import argparse
    import io
    sys.path.insert(0, REPO)
    os.environ.update(MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port), RANK=str(rank), LOCAL_RANK=str(rank), WORLD_SIZE=str(world))
    import bench
    from oracle import clip_oracle, proto_oracle as po
    from proto_clip_amd import synth
    from proto_clip_amd.clip.model import random_state_dict
    from proto_clip_amd.dist import PrototypeExchange, hot_path_step, shard_bounds
    kw = dict(embed_dim=64, image_resolution=32, vision_layers=2, vision_width=128, vision_patch_size=8, context_length=77,
              vocab_size=512, transformer_width=64, transformer_heads=1, transformer_layers=1)
    N, K, D, B = 9, 4, 64, 12
    calls = {"build": 0, "step": 0, "check": 0}

    def build_state(device, rank_, world_):
        calls["build"] += 1
        assert device.type == "cpu" and (rank_, world_) == (rank, world)
        split = synth.make_split(N, K, D, 8, 8, seed=2)
        rows = split.visual_memory_keys.t().contiguous()
        labels = torch.arange(N).repeat_interleave(K).int()
        lo, hi = shard_bounds(N * K, rank_, world_)
        torch.manual_seed(0)
        ad = {"conv1.weight": torch.randn(16, 1, 1, 1).half(), "conv2.weight": (torch.randn(16, 16, 3, 3) / 12).half(),
              "conv3.weight": (torch.randn(1, 16, 1, 1) / 4).half()}
        for i, c in ((1, 16), (2, 16), (3, 1)):
            ad[f"bn{i}.weight"], ad[f"bn{i}.bias"] = torch.ones(c, 8, 8).half(), torch.zeros(c, 8, 8).half()
        return dict(sd=random_state_dict(seed=11, **kw), ad=ad, bank=rows[lo:hi], bank_labels=labels[lo:hi],
                    text=po.l2norm_rows(split.textual_memory_bank.t().contiguous()), images=synth.make_images(B, 32, seed=5 + rank_, n_class=N))

    def step(st):
        calls["step"] += 1

        class OraclePath:
            encode = staticmethod(lambda x: clip_oracle.encode_image(st["sd"], x, half=True))
            l2norm = staticmethod(po.l2norm_rows)
            adapt = staticmethod(lambda f: po.l2norm_rows(po.adapter_conv(f, st["ad"], "conv-3x")))
            classify = staticmethod(lambda a, zi, zt_, al, be: po.P(a, zi, zt_, al, be).max(1)[1])

        fin = lambda s, c, fp32_out=False: po.proto_finalize(s, c, fp32=fp32_out)
        return hot_path_step(OraclePath, PrototypeExchange(), st["bank"], st["bank_labels"], N, st["images"], st["text"], 0.5, 12.0,
                             partial_fn=po.partial_sums, finalize_fn=fin)

    def self_check(st):
        calls["check"] += 1
        idx = torch.arange(0, B, 3)
        full, sub = step(st), step(dict(st, images=st["images"][idx].contiguous()))
        return "ok" if torch.equal(full[idx], sub) else "MISMATCH"

    class CpuHooks:
        backend, instrument, batch = "gloo", False, B
        device = staticmethod(lambda local: torch.device("cpu"))
        sync = staticmethod(lambda: None)
    CpuHooks.build_state, CpuHooks.step, CpuHooks.self_check = staticmethod(build_state), staticmethod(step), staticmethod(self_check)

    buf = io.StringIO()
    args = argparse.Namespace(gpus=world, steps=3, warmup=1, no_cpu_baseline=True, no_extra=True, cpu_worker=None)
    bench.run(args, CpuHooks, out=buf)
    assert calls["build"] == 1 and calls["check"] == 1 and calls["step"] == 1 + 3 + 2       # warm-up + timed + the self-check's two
    with open(os.path.join(outdir, f"out{rank}.txt"), "w") as f:
        f.write(buf.getvalue())


def test_bench_main_plumbing_two_ranks(tmp_path):
    import json
    port = 35500 + (os.getpid() % 2000)
    mp.spawn(_bench_main_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    out0, out1 = (tmp_path / "out0.txt").read_text(), (tmp_path / "out1.txt").read_text()
    assert out1 == "" and out0.count("\n") == 1                           # ONE line, from rank 0 only
    line = json.loads(out0)
    assert line["n_gpus"] == 2 and line["rccl_world_size"] == 2 and line["steps"] == 3 and line["warmup"] == 1
    assert line["scaling"] == "weak" and line["self_check"] == "ok" and line["higher_is_better"] is True
    assert line["config"]["batch_per_gpu"] == 12 and line["config"]["global_batch"] == 24
    # value = the units ALL ranks processed / the slowest rank's time
    assert abs(line["value"] - 3 * 12 * 2 / (line["ms_per_step"] * 3 / 1e3)) < 1e-6 * line["value"]
    assert "cpu_baseline" not in line and "extra" not in line
