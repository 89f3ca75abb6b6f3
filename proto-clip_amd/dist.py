"""Multi-GPU sharding of the hot path: one process per GPU, `torch.distributed` backend
"nccl" (= RCCL over xGMI on ROCm).

The path shards naturally: encoder forwards and query classification are independent per image
(weights and prototypes replicated); the only exchange is the per-class mean of the support set.
Each rank reduces its slab of the class-sorted support rows to fp32 per-class partial sums [N, D] +
counts [N] (no atomics, deterministic), ONE all-gather moves 2.05 MB/rank (ImageNet, D=512) — a
latency-bound message for which a single direct all-gather over the fully connected xGMI mesh beats
any ring schedule — and every rank combines the W slabs in rank order, so all ranks hold bit-identical
prototypes.  Accuracy counters are all-reduced as int32/int64.

The reference has no distributed code; arithmetic restated here is main.py:399-402.
`partial_fn` / `finalize_fn` default to the HIP kernels; the gloo CPU tests inject the oracle's
restatement to exercise the communication logic without a GPU."""
import torch
import torch.distributed as dist


def world_size(group=None) -> int:
    return dist.get_world_size(group) if dist.is_available() and dist.is_initialized() else 1


def rank(group=None) -> int:
    return dist.get_rank(group) if dist.is_available() and dist.is_initialized() else 0


def allreduce_sum_(tensors, group=None):
    """In-place sum over ranks of a list of same-dtype tensors with ONE all-reduce of their concatenation (the training
    step's gradients are a few MB: latency-bound on xGMI, so one message beats one per tensor).  Returns `tensors`."""
    if not (dist.is_available() and dist.is_initialized()) or not tensors:
        return tensors                                      # (a 1-rank group still goes through RCCL: exercised on one GPU)
    flat = torch.cat([t.reshape(-1) for t in tensors])
    dist.all_reduce(flat, op=dist.ReduceOp.SUM, group=group)
    off = 0
    for t in tensors:
        n = t.numel()
        t.copy_(flat[off:off + n].view(t.shape))
        off += n
    return tensors


def shard_bounds(n_items: int, rank: int, world: int):
    """Contiguous slab [lo, hi) of rank `rank` (sizes differ by at most one)."""
    base, rem = divmod(n_items, world)
    lo = rank * base + min(rank, rem)
    return lo, lo + base + (1 if rank < rem else 0)


def sharded_prototypes(mem_shard, labels_shard, N: int, per_shot_norm: bool = True, fp32_out: bool = False,
                       group=None, partial_fn=None, finalize_fn=None):
    """mem_shard [R_r, D] fp16 = this rank's slab of the support rows with non-decreasing labels.
    Returns the full prototype matrix [N, D], identical on every rank."""
    if partial_fn is None or finalize_fn is None:
        from . import ops
        partial_fn = partial_fn or ops.partial_sums
        finalize_fn = finalize_fn or ops.proto_finalize
    sums, counts = partial_fn(mem_shard, labels_shard, N, per_shot_norm)
    if not dist.is_initialized():
        return finalize_fn(sums[None], counts[None], fp32_out=fp32_out)
    world = dist.get_world_size(group)
    D = sums.shape[1]
    # one message per rank: [N*D sums | N counts (bit-cast to fp32)] so a single all-gather suffices
    payload = torch.cat([sums.reshape(-1), counts.view(torch.float32).reshape(-1)])
    gathered = torch.empty(world * payload.numel(), dtype=torch.float32, device=payload.device)
    dist.all_gather_into_tensor(gathered, payload, group=group)
    gathered = gathered.view(world, -1)
    all_sums = gathered[:, : N * D].reshape(world, N, D).contiguous()
    all_counts = gathered[:, N * D:].contiguous().view(torch.int32).reshape(world, N)
    return finalize_fn(all_sums, all_counts, fp32_out=fp32_out)


class PrototypeExchange:
    """The prototype reduction + all-gather of a step on a SIDE stream, overlapped with the encoder forward (the only coupling
    between ranks is 2.05 MB per rank, needed only by the classification at the END of the step):

        ex.launch(bank_shard, labels_shard, N)      # step start: partial sums -> all-gather -> rank-ordered combine, side stream
        ... encode_image, normalise, adapter on the current stream ...
        zi = ex.result()                            # the current stream waits for the side stream here

    so that at N > 1 the step costs what the encoder costs.  CPU tensors (the gloo tests) run inline."""

    def __init__(self):
        self.stream = None
        self.out = None

    def launch(self, mem_shard, labels_shard, N, **kw):
        if not mem_shard.is_cuda:
            self.out = sharded_prototypes(mem_shard, labels_shard, N, **kw)
            return self
        if self.stream is None:
            self.stream = torch.cuda.Stream(mem_shard.device)
        self.stream.wait_stream(torch.cuda.current_stream(mem_shard.device))      # the bank may have been written on the main stream
        # the operands were allocated on the caller's stream and are READ on the side stream: tell the caching allocator, or a caller that
        # passes temporaries (bank[lo:hi].to(device) per step) and drops them before result() could get the memory handed back to the main
        # stream while partial_sums still reads it
        mem_shard.record_stream(self.stream)
        labels_shard.record_stream(self.stream)
        with torch.cuda.stream(self.stream):
            self.out = sharded_prototypes(mem_shard, labels_shard, N, **kw)
        return self

    def result(self):
        out, self.out = self.out, None
        if self.stream is not None and out is not None and out.is_cuda:
            cur = torch.cuda.current_stream(out.device)
            cur.wait_stream(self.stream)
            out.record_stream(cur)                                                # allocated on the side stream, consumed on this one
        return out


class HipPath:
    """The stages of one hot-path step on the HIP kernels (`hot_path_step`'s default implementation)."""

    def __init__(self, model, adapter):
        self.model, self.adapter = model, adapter

    def encode(self, images):
        return self.model.encode_image(images)                                    # clip/model.py:338

    def l2norm(self, f):
        from . import ops
        return ops.l2norm_rows(f, out=f)                                          # utils.py:352

    def adapt(self, f):
        return self.adapter(f, l2norm_out=True)                                   # model.py:49-78 + main.py:408-409

    def classify(self, a, zi, zt, alpha, beta):
        from . import ops
        return ops.classify(a, zi, zt, alpha, beta, want_p=False, want_argmax=True)[1]   # utils.py:225-244 + main.py:190


def hot_path_step(path, exchange, bank_shard, labels_shard, N, images, text_proto, alpha, beta, **proto_kw):
    """One step of the sharded hot path on THIS rank (bench.py's step): prototypes from the rank-sharded support bank
    (exchanged on the side stream), this rank's query images through encoder -> normalise -> adapter, classification against both
    banks.  Returns the top-1 class of this rank's queries.  `path` supplies the stage implementations (HipPath on the GPU; the
    gloo test injects the oracle's)."""
    exchange.launch(bank_shard, labels_shard, N, **proto_kw)
    f = path.l2norm(path.encode(images))
    a = path.adapt(f)
    zi = exchange.result()
    return path.classify(a, zi, text_proto, alpha, beta)


def allreduce_counts(correct: torch.Tensor, total: int, group=None):
    """Sum integer correct-counts (e.g. the [na, nb] sweep grid) and sample totals over ranks."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return correct, total
    buf = torch.cat([correct.reshape(-1).to(torch.int64), torch.tensor([total], dtype=torch.int64, device=correct.device)])
    dist.all_reduce(buf, op=dist.ReduceOp.SUM, group=group)
    return buf[:-1].reshape(correct.shape), int(buf[-1].item())
