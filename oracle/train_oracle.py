"""TEST INFRASTRUCTURE — CPU restatement of the reference's episodic training step (main.py:216-381,
utils.py:72-109) with torch autograd on CPU fp16/fp32 tensors, exactly the mechanism the reference uses
(eager tensors + autograd + torch.optim.AdamW).  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
may import this package; the product path (proto_clip_amd/train.py) runs explicit HIP kernels instead.

Pinned by tests/golden/train_*.npz: the reference's own `run_proto_clip` training loop executed under the Appendix-B
shim (tests/golden/make_golden.py) — per-episode losses, first-step gradients and the trained banks.
`info_nce` is absent from the image: InfoNCE below restates the published defaults of info-nce-pytorch
(temperature 0.1, mean reduction, both sides L2-normalised, cross entropy against the diagonal) — the alignment
losses L2/L3/L4 are therefore pinned only against that restatement."""
import math

import numpy as np
import torch
import torch.nn.functional as F


def sample_epoch(N, K, rng=np.random):
    """Episodes of one epoch (main.py:228-258): same draws, same order."""
    upper, lower = int(N * 0.4), max(int(N * 0.2), 1)
    perm = rng.permutation(N)
    start = 0
    while start < N - 1:
        n_cls = rng.randint(lower, upper)
        classes = sorted(perm[start:min(start + n_cls, N - 1)])
        q_idx, q_lab = [], []
        for c in classes:
            items = rng.permutation(K)
            n = rng.randint(1, K) if K > 1 else K
            query = sorted(items[n:]) if K > 1 else sorted(items[:n])
            q_idx.extend(int(c) * K + int(i) for i in query)
            q_lab.extend([int(c)] * len(query))
        yield classes, q_idx, q_lab
        start += len(classes)


def info_nce(a, b, temperature=0.1):
    a, b = F.normalize(a, dim=-1), F.normalize(b, dim=-1)
    return F.cross_entropy(a @ b.t() / temperature, torch.arange(len(a)))


def P(zq, zi, zt, alpha, beta):
    """utils.py:225-244 on autograd tensors."""
    zq, zi, zt = zq.float(), zi.float(), zt.float()
    di = torch.cdist(zq, zi) ** 2
    dt = torch.cdist(zq, zt) ** 2
    return alpha * F.softmax(-beta * di, dim=1) + (1 - alpha) * F.softmax(-beta * dt, dim=1)


def adapter_conv(x, p, c_type):
    """model.py:49-78 with fp16 parameters p (dict of tensors)."""
    B, D = x.shape
    s = int(math.ceil(math.sqrt(D)))
    x = F.pad(x, (0, s * s - D)).view(-1, 1, s, s)
    out = F.layer_norm(F.conv2d(x, p["conv1.weight"]), [16, s, s], p["bn1.weight"], p["bn1.bias"])
    if c_type == "conv-3x":
        out = F.layer_norm(F.conv2d(out, p["conv2.weight"], padding=1), [16, s, s], p["bn2.weight"], p["bn2.bias"])
    out = F.layer_norm(F.conv2d(out, p["conv3.weight"]), [1, s, s], p["bn3.weight"], p["bn3.bias"])
    out = out + x
    return out.view(-1, 1, s * s)[:, :, :D].reshape(-1, D)


def adapter_fc(x, p):
    """model.py:81-95."""
    h = F.layer_norm(F.linear(x, p["fc.0.weight"]), [p["fc.0.weight"].shape[0]], p["fc.1.weight"], p["fc.1.bias"])
    h = F.layer_norm(F.linear(h, p["fc.2.weight"]), [x.shape[1]], p["fc.3.weight"], p["fc.3.bias"])
    return 0.2 * h + (1 - 0.2) * x


def episode_loss(visual, textual, adapter_params, kind, keys_rows, q_idx, q_lab, N, K, alpha, beta, losses):
    """main.py:260-285 + utils.py:80-109: returns (matches, loss, dict of the individual terms)."""
    D = visual.shape[1]
    zs = visual.view(-1, K, D)
    zs = zs / zs.norm(dim=-1, keepdim=True)
    z_img = zs.mean(dim=1).float()
    z_img = z_img / z_img.norm(dim=-1, keepdim=True)
    zq = keys_rows[torch.as_tensor(q_idx)]
    zq = (adapter_fc(zq, adapter_params) if kind == "fc" else adapter_conv(zq, adapter_params, kind)).float()
    lab = torch.as_tensor(q_lab)
    zq = zq / zq.norm(dim=-1, keepdim=True)
    z_txt = (textual / textual.norm(dim=-1, keepdim=True)).float()
    p = P(zq, z_img, z_txt, alpha, beta)
    matches = (p.max(dim=1)[1] == lab).float().sum()
    terms, loss = {}, 0
    if len(losses) == 0 or "L1" in losses:
        terms["L1"] = F.nll_loss(torch.log(p), lab)
        loss = loss + terms["L1"]
    if "L2" in losses:
        terms["L2"] = info_nce(z_img, z_txt)
        loss = loss + terms["L2"]
    if "L3" in losses:
        terms["L3"] = info_nce(z_txt, z_img)
        loss = loss + terms["L3"]
    if "L4" in losses:
        terms["L4i"], terms["L4t"] = info_nce(z_img, z_img), info_nce(z_txt, z_txt)
        loss = loss + terms["L4i"] + terms["L4t"]
    return matches, loss, terms, p


class Trainer:
    """State of main.py:107-137: fp16 banks + adapter parameters, AdamW(eps=1e-4, wd=0.05), cosine schedule."""

    def __init__(self, cfg, keys, text_bank, adapter_sd, alpha, beta):
        D, NK = keys.shape
        self.K, self.N = cfg["shots"], NK // cfg["shots"]
        self.kind = cfg["adapter"]
        self.alpha, self.beta, self.losses = alpha, beta, list(cfg.get("losses", []))
        self.keys_rows = keys.t().contiguous()
        self.visual = torch.nn.Parameter(keys.t().clone())
        self.textual = torch.nn.Parameter(text_bank.t().clone())
        self.adapter = {k: torch.nn.Parameter(v.clone()) for k, v in adapter_sd.items()}
        ad = list(self.adapter.values())
        params = ad + [self.visual] if cfg.get("train_vis_mem_only", False) else [self.visual, self.textual] + ad
        self.opt = torch.optim.AdamW(params, lr=cfg["lr"], eps=1e-4, weight_decay=0.05)
        self.sched = torch.optim.lr_scheduler.CosineAnnealingLR(self.opt, cfg["train_epoch"] * NK)

    def step(self, q_idx, q_lab):
        matches, loss, terms, _ = episode_loss(self.visual, self.textual, self.adapter, self.kind, self.keys_rows, q_idx,
                                               q_lab, self.N, self.K, self.alpha, self.beta, self.losses)
        self.opt.zero_grad()
        loss.backward()
        grads = {"visual": self.visual.grad, "textual": self.textual.grad, **{k: v.grad for k, v in self.adapter.items()}}
        grads = {k: (None if g is None else g.clone()) for k, g in grads.items()}
        self.opt.step()
        return matches.item(), loss.item(), {k: v.item() for k, v in terms.items()}, grads

    def train_epoch(self, rng=np.random):
        correct, seen, ls = 0.0, 0, []
        for _, q_idx, q_lab in sample_epoch(self.N, self.K, rng):
            m, l, _, _ = self.step(q_idx, q_lab)
            correct += m
            seen += len(q_lab)
            ls.append(l)
        self.sched.step()
        return correct / max(seen, 1), sum(ls) / max(len(ls), 1), self.sched.get_last_lr()[0]
