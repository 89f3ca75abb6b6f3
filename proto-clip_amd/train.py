"""Episodic training of the memory banks and the adapter (reference main.py:216-381, utils.py:80-109).

The reference builds the episode graph with eager tensors and lets autograd + torch.optim.AdamW do the rest.  The
graph is static, so here it is written out: one forward and one backward sweep of libpclip kernels per episode
(csrc/pclip_train.hip, pclip_adapter.hip), no tape, fp16 parameters / gradients / AdamW moments as in the reference
(`nn.Embedding(...).to(clip_model.dtype)`, `Adapter(dtype=torch.half)`), fp32 wherever the reference says `.float()`.

    zs = v.view(N, K, D); zs /= |zs|; z_img = mean_K(zs).float(); z_img /= |z_img|          main.py:260-264
    zq = adapter(keys[query_index]).float(); zq /= |zq|                                     main.py:266-275
    z_txt = (t / |t|).float()                                                               main.py:272-279
    p = P(zq, z_img, z_txt, alpha, beta); loss = NLL(log p) [+ InfoNCE(z_img, z_txt) + InfoNCE(z_txt, z_img) ...]

torch is used for memory, indexing of the constant key rows and host bookkeeping only."""
import math

import numpy as np
import torch

from . import dist as pdist
from . import ops
from ._lib import PclipError
from .model import Adapter, Adapter_FC

INFO_NCE_TEMPERATURE = 0.1      # info-nce-pytorch default (the package is not pinned by the reference)


# ---------------------------------------------------------------- episode sampler (host) ---------------------
def sample_epoch(N: int, K: int, rng=np.random):
    """The episodes of one epoch exactly as main.py:228-258 draws them from numpy's global generator: yields
    (class_index, query_index, zq_labels) with the same sequence of np.random calls."""
    class_upper = int(N * 0.4)
    class_lower = max(int(N * 0.2), 1)
    class_indexes = rng.permutation(N)
    start = 0
    while start < N - 1:
        num_class = rng.randint(class_lower, class_upper)
        class_index = sorted(class_indexes[start:min(start + num_class, N - 1)])
        query_index, zq_labels = [], []
        for cls in class_index:
            item_indexes = rng.permutation(K)
            n = rng.randint(1, K) if K > 1 else K
            query = sorted(item_indexes[n:]) if K > 1 else sorted(item_indexes[:n])
            query_index.extend(int(cls) * K + int(q) for q in query)
            zq_labels.extend([int(cls)] * len(query))
        yield class_index, query_index, zq_labels
        start += len(class_index)


def cosine_lr(base_lr: float, epoch: int, t_max: int, eta_min: float = 0.0) -> float:
    """Learning rate after `epoch` calls of CosineAnnealingLR.step() (closed form; main.py:136-137, 312)."""
    return eta_min + (base_lr - eta_min) * (1.0 + math.cos(math.pi * epoch / t_max)) / 2.0


# ---------------------------------------------------------------- adapters with saved activations -------------
def _fc_forward(ad: Adapter_FC, x):
    """Adapter_FC forward with the activations the backward needs.  The stages are launched one by one (the fused
    pclip_adapter_fc_f16 keeps them in scratch) with the UNSPLIT GEMM — `ops.low_latency()` must not change which kernel
    produced the saved h1 / h2 — and the output is formed from exactly those tensors: Linear -> LN -> Linear -> LN, then
    r16(r16(0.2 * h) + r16(0.8 * x)) (model.py:91-95), the blend riding on the second LayerNorm."""
    fc = ad.fc
    with ops.low_latency(False):
        h1 = ops.gemm(x, fc[0].weight)                                         # Linear, no bias (model.py:85)
        a1 = ops.layernorm(h1, fc[1].weight.float(), fc[1].bias.float())
        h2 = ops.gemm(a1, fc[2].weight)
    out = ops.layernorm_blend(h2, fc[3].weight, fc[3].bias, x, ratio=0.2)
    return out, (x, h1, a1, h2)


def _fc_backward(ad: Adapter_FC, saved, g):
    """g [Q, D] fp16 = dL/d(adapter output).  The input rows are constants (main.py:266): parameter gradients only."""
    fc = ad.fc
    x, h1, a1, h2 = saved
    dh2, dg2, db2 = ops.layernorm_backward(h2, fc[3].weight, g, dy_scale=0.2)   # ratio * x (model.py:93-94)
    dw2 = ops.gemm_f32(dh2, a1, trans_a=True)                                   # [D, H]
    da1 = ops.cast_f16(ops.gemm_f32(dh2, fc[2].weight))                         # [Q, H], fp16 like autograd's
    dh1, dg1, db1 = ops.layernorm_backward(h1, fc[1].weight, da1)
    dw1 = ops.gemm_f32(dh1, x, trans_a=True)                                    # [H, D]
    return {fc[0].weight: dw1, fc[1].weight: dg1, fc[1].bias: db1, fc[2].weight: dw2, fc[3].weight: dg2, fc[3].bias: db2}


def _conv_forward(ad: Adapter, x):
    out = ops.adapter_conv(x, ad.c_type == "conv-3x", ad.conv1.weight, ad.bn1.weight, ad.bn1.bias, ad.conv2.weight,
                           ad.bn2.weight, ad.bn2.bias, ad.conv3.weight, ad.bn3.weight, ad.bn3.bias)
    return out, (x,)


def _conv_backward(ad: Adapter, saved, g):
    (x,) = saved
    grads = ops.adapter_conv_backward(x, g, ad.c_type == "conv-3x", ad.conv1.weight, ad.bn1.weight, ad.bn1.bias,
                                      ad.conv2.weight, ad.bn2.weight, ad.bn2.bias, ad.conv3.weight, ad.bn3.weight, ad.bn3.bias)
    names = ("conv1.weight", "bn1.weight", "bn1.bias", "conv2.weight", "bn2.weight", "bn2.bias", "conv3.weight",
             "bn3.weight", "bn3.bias")
    params = dict(ad.named_parameters())
    return {params[n]: grads[n] for n in names if grads.get(n) is not None}


# ---------------------------------------------------------------- the trainer ----------------------------------
class ProtoClipTrainer:
    """Owns the learnable banks (fp16 `[N*K, D]`, `[N, D]`), the adapter and the AdamW state; `step()` runs one episode."""

    def __init__(self, cfg, visual_memory_keys, textual_memory_bank, adapter, alpha, beta):
        D, NK = visual_memory_keys.shape
        self.K = int(cfg["shots"])
        self.N = NK // self.K
        self.D = D
        from .main import check_shape_envelope
        check_shape_envelope(self.N, self.K, D, "fc" if isinstance(adapter, Adapter_FC) else "conv-3x", training=True, eval_path=False)
        self.alpha, self.beta = float(alpha), float(beta)
        self.losses = list(cfg.get("losses", []))
        self.keys_rows = ops.transpose(visual_memory_keys)                      # constant query source (main.py:266)
        self.visual = self.keys_rows.clone()                                    # nn.Embedding weight (main.py:110-112)
        self.textual = ops.transpose(textual_memory_bank).clone()               # main.py:119-121
        self.adapter = adapter
        self.train_vis_mem_only = bool(cfg.get("train_vis_mem_only", False))
        params = list(adapter.parameters()) + [self.visual]                     # main.py:123-128
        if not self.train_vis_mem_only:
            params = [self.visual, self.textual] + list(adapter.parameters())
        self.params = params
        self.base_lr = float(cfg["lr"])
        self.lr = self.base_lr
        self.t_max = int(cfg["train_epoch"]) * NK                               # main.py:136-137
        self.state = {id(p): (torch.zeros_like(p.data if hasattr(p, "data") else p), torch.zeros_like(p.data if hasattr(p, "data") else p), [0])
                      for p in params}
        self.epoch = 0
        self._param_index = {id(p): i for i, p in enumerate(adapter.parameters())}

    # -- forward pieces shared with evaluation ------------------------------------------------------------
    def adapter_forward(self, x):
        if isinstance(self.adapter, Adapter_FC):
            return _fc_forward(self.adapter, x)
        return _conv_forward(self.adapter, x)

    def adapter_backward(self, saved, g):
        if isinstance(self.adapter, Adapter_FC):
            return _fc_backward(self.adapter, saved, g)
        return _conv_backward(self.adapter, saved, g)

    def _info_nce(self, a, b, ga, gb):
        """loss of InfoNCE()(a, b) (utils.py:72-77); accumulates its gradients into ga / gb (fp32, like a / b)."""
        n = a.shape[0]
        an = ops.l2norm_rows_f32(a)
        bn = an if b is a else ops.l2norm_rows_f32(b)
        S = ops.gemm_f32(an, bn, trans_b=True, alpha=1.0 / INFO_NCE_TEMPERATURE)            # logits [n, n]
        loss_rows, dS = ops.softmax_ce_rows(S, 1.0 / n)                                      # mean reduction
        scale = 1.0 / INFO_NCE_TEMPERATURE
        dan = ops.gemm_f32(dS, bn, alpha=scale)                                              # dS @ bn
        if b is a:
            ops.gemm_f32(dS, an, trans_a=True, alpha=scale, out=dan, beta=1.0)               # + dS^T @ an
            ops.l2norm_rows_backward_f32_(ga, a, dan)
        else:
            dbn = ops.gemm_f32(dS, an, trans_a=True, alpha=scale)
            ops.l2norm_rows_backward_f32_(ga, a, dan)
            ops.l2norm_rows_backward_f32_(gb, b, dbn)
        return ops.colsum_f32(loss_rows.view(n, 1), scale=1.0 / n)                           # [1] device scalar

    def step(self, query_index, zq_labels):
        """One episode of main.py (queries = rows `query_index` of the constant key bank, main.py:265-266): forward,
        backward, AdamW.  Returns the 7-tuple of utils.compute_loss_and_matches (device scalars; entries the configured
        `losses` do not produce are None).  Under torch.distributed every rank draws the same episode (same numpy seed) and
        takes its contiguous slab of the queries."""
        dev = self.visual.device
        q_total = len(query_index)
        if pdist.world_size() > 1:
            lo, hi = pdist.shard_bounds(q_total, pdist.rank(), pdist.world_size())
            query_index, zq_labels = query_index[lo:hi], zq_labels[lo:hi]
        qi = torch.as_tensor(query_index, device=dev, dtype=torch.long)
        return self.step_features(self.keys_rows[qi], torch.as_tensor(zq_labels, device=dev, dtype=torch.long), q_total=q_total)

    def step_features(self, xq, labels, q_total=None):
        """One step on explicit query features xq [Q, D] fp16 with labels [Q] — the Proto-CLIP-F-Q^T variant (main.qt.py:198-
        209) feeds `clip_model.encode_image(images)` of the training loader here.

        Data parallel: xq / labels are THIS rank's queries, q_total the number over all ranks (all-reduced when
        None).  The query-dependent gradients — fp32 gradients wrt the two prototype matrices and the adapter parameters,
        plus the loss / match sums — are summed over ranks in ONE all-reduce (RCCL) before the replicated tail (alignment
        losses, prototype-chain backward, fp16 rounding, AdamW), so every rank applies the identical update."""
        N, K = self.N, self.K
        dev = self.visual.device
        xq = xq.to(dev).contiguous()
        labels = labels.to(dev)
        Q = xq.shape[0]
        if q_total is None:
            q_total = Q if not pdist.dist.is_initialized() else int(pdist.allreduce_sum_([torch.tensor([float(Q)], device=dev)])[0].item())
        # ---- forward ----
        z_img = ops.proto_build(self.visual, N, K, per_shot_norm=True, fp32_out=True)        # 260-264
        z_txt = ops.cast_f32(ops.l2norm_rows(self.textual))                                  # 272-279
        g_img = torch.zeros(N, self.D, dtype=torch.float32, device=dev)
        g_txt = torch.zeros(N, self.D, dtype=torch.float32, device=dev)
        sums = torch.zeros(2, dtype=torch.float32, device=dev)                               # [sum_q nll / q_total, matches]
        use_l1 = len(self.losses) == 0 or "L1" in self.losses                                # utils.py:90
        ad_grads = {}
        if Q > 0:
            a, saved = self.adapter_forward(xq)                                              # 267
            zq = ops.proto_build(a, Q, 1, per_shot_norm=False, fp32_out=True)                # .float(); / norm (267, 274)
            d2i, d2t, _ = ops.sqdist_f32(zq, z_img, z_txt)                                   # P (utils.py:225-244)
            gi, gt, rs, nll, _, am = ops.nll_grad(d2i, d2t, labels, N, self.alpha, self.beta, q_total=q_total)
            sums[1] = (am.long() == labels).float().sum()                                    # utils.py:84-85
            if use_l1:
                sums[0:1] = ops.colsum_f32(nll.view(Q, 1), scale=1.0 / q_total)
                gi_v, gt_v = gi[:, :N], gt[:, :N]                                            # views of the padded rows
                # cdist backward: dq = sum_c 2 G[q,c] (q - z_c), dz_c = sum_q 2 G[q,c] (z_c - q)
                gq = ops.gemm_f32(gi_v, z_img, alpha=-2.0)
                ops.gemm_f32(gt_v, z_txt, alpha=-2.0, out=gq, beta=1.0)
                ops.addscaled_rows_(gq, zq, rs, 2.0)
                ops.gemm_f32(gi_v, zq, trans_a=True, alpha=-2.0, out=g_img, beta=1.0)
                ops.gemm_f32(gt_v, zq, trans_a=True, alpha=-2.0, out=g_txt, beta=1.0)
                ops.addscaled_rows_(g_img, z_img, ops.colsum_f32(gi, cols=N), 2.0)
                ops.addscaled_rows_(g_txt, z_txt, ops.colsum_f32(gt, cols=N), 2.0)
                da = ops.proto_backward(a, gq, Q, 1, per_shot_norm=False, final_norm=True)   # fp16, dL/d adapter(x)
                ad_grads = self.adapter_backward(saved, da)                                  # fp32, parameter-shaped
        if pdist.dist.is_initialized():
            if use_l1:                                                                       # a rank without queries still contributes zeros
                for p in self.adapter_params_with_grad():
                    ad_grads.setdefault(p, torch.zeros(p.shape, dtype=torch.float32, device=dev))
            order = sorted(ad_grads, key=lambda p: self._param_index[id(p)])
            pdist.allreduce_sum_([sums, g_img, g_txt] + [ad_grads[p] for p in order])
        matches = sums[1]
        total = torch.zeros(1, dtype=torch.float32, device=dev)
        l1 = l2 = l3 = l4i = l4t = None
        if use_l1:
            l1 = sums[0:1].clone()
            total += l1
        if "L2" in self.losses:
            l2 = self._info_nce(z_img, z_txt, g_img, g_txt)
            total += l2
        if "L3" in self.losses:
            l3 = self._info_nce(z_txt, z_img, g_txt, g_img)
            total += l3
        if "L4" in self.losses:
            l4i = self._info_nce(z_img, z_img, g_img, g_img)
            l4t = self._info_nce(z_txt, z_txt, g_txt, g_txt)
            total += l4i
            total += l4t
        # ---- replicated tail of the backward ----
        grads = {id(p): g for p, g in ad_grads.items()}
        grads[id(self.visual)] = ops.proto_backward(self.visual, g_img, N, K, per_shot_norm=True, final_norm=True)
        if not self.train_vis_mem_only:
            grads[id(self.textual)] = ops.proto_backward(self.textual, g_txt, N, 1, per_shot_norm=True, final_norm=False)
        # ---- AdamW (eps 1e-4, weight decay 0.05; main.py:134-135) ----
        for p in self.params:
            g = grads.get(id(p))
            if g is None:
                continue                                                                      # no gradient: torch skips it
            data = p.data if isinstance(p, torch.nn.Parameter) else p
            g16 = g if g.dtype == torch.float16 else ops.cast_f16(g.contiguous())
            m, v, cnt = self.state[id(p)]
            cnt[0] += 1
            ops.adamw_(data.view(-1), g16.reshape(-1), m.view(-1), v.view(-1), self.lr, cnt[0])
        self.last_grads = grads
        return matches, total, l1, l2, l3, l4i, l4t

    def adapter_params_with_grad(self):
        """Adapter parameters the loss reaches (conv-2x never uses conv2 / bn2: SURVEY fact 7)."""
        named = dict(self.adapter.named_parameters())
        skip = ("conv2.weight", "bn2.weight", "bn2.bias") if getattr(self.adapter, "c_type", None) == "conv-2x" else ()
        return [p for n, p in named.items() if n not in skip]

    def end_epoch(self):
        """scheduler.step() (main.py:312)."""
        self.epoch += 1
        self.lr = cosine_lr(self.base_lr, self.epoch, self.t_max)
        return self.lr

    def train_epoch(self, rng=np.random, clip_model=None, train_loader_F=None):
        """One epoch: the episodes of main.py:228-310, or — with a loader — one step per batch of freshly encoded training
        images (main.qt.py:198-250).  Returns (train accuracy, mean loss, learning rate after scheduler.step())."""
        correct, seen, losses = 0.0, 0, []

        def book(result, n):
            nonlocal correct, seen
            correct += float(result[0].item())
            seen += n
            losses.append(float(result[1].item()))

        if train_loader_F is not None:
            for images, target in train_loader_F:
                with torch.no_grad():
                    feats = clip_model.encode_image(images.cuda())                           # main.qt.py:199-201
                book(self.step_features(feats, target), len(target))
        else:
            for _, query_index, zq_labels in sample_epoch(self.N, self.K, rng):
                book(self.step(query_index, zq_labels), len(zq_labels))
        lr = self.end_epoch()
        return correct / max(seen, 1), sum(losses) / max(len(losses), 1), lr
