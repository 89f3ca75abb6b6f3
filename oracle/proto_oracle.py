"""ORACLE — test infrastructure only.  CPU restatement (torch-CPU, fp32 arithmetic with explicit fp16
rounding points) of the reference's algorithm for the Proto-CLIP hot path.  Only tests/,
__graft_entry__.smoke() and bench.py's cpu_baseline leg may import this; the product path
(proto-clip_amd/) never does.

Parity pin: checked against outputs of the reference itself, imported in the build container with
the SURVEY Appendix-B shim (tests/golden/make_golden.py) — the committed fixtures in tests/golden/*.npz
hold those outputs and tests/test_oracle_golden.py compares this file against them on every run.

Every function cites the reference lines it restates (paths relative to the reference tree).
`r16(x)` = round fp32 -> fp16 -> fp32, the rounding every fp16 tensor op of the reference performs."""
import math

import numpy as np
import torch
import torch.nn.functional as F


def r16(x: torch.Tensor) -> torch.Tensor:
    return x.half().float()


# ---- normalisation / prototypes -------------------------------------------------------------------

def l2norm_rows(x16: torch.Tensor) -> torch.Tensor:
    """x / x.norm(dim=-1, keepdim=True) on fp16 tensors (utils.py:352; main.py:182-185, 404-409):
    the norm is accumulated in fp32 and rounded to fp16, the quotient is rounded to fp16."""
    x = x16.float()
    n = r16(x.pow(2).sum(-1, keepdim=True).sqrt())
    return (x / n).half()


def proto_build(mem16: torch.Tensor, N: int, K: int, per_shot_norm: bool = True, fp32: bool = False):
    """main.py:399-402 (eval), 260-264 (train: fp32=True), 173-176 (zero-shot init: per_shot_norm=False).
    zs = mem.view(N,K,D); zs /= ||zs||; z = zs.mean(1); z /= ||z||."""
    D = mem16.shape[1]
    zs = mem16.view(N, K, D)
    if per_shot_norm:
        zs = l2norm_rows(zs.reshape(N * K, D)).view(N, K, D)
    z = r16(zs.float().sum(1) / K)                      # fp16 mean: fp32 accumulate, one rounding
    if fp32:
        return z / z.pow(2).sum(-1, keepdim=True).sqrt()          # .float() then fp32 normalise (262-264)
    return l2norm_rows(z.half())


def bank_reduce(feats16: torch.Tensor, perm: torch.Tensor = None) -> torch.Tensor:
    """utils.py:318-326: mean over augment epochs (fp16), row normalise, sort columns by label.
    Returns rows [R, D] (the reference then stores the transpose [D, R])."""
    A = feats16.shape[0]
    m = r16(feats16.float().sum(0) / A).half()
    keys = l2norm_rows(m)
    return keys if perm is None else keys[perm]


def partial_sums(mem16, labels, N, per_shot_norm=True):
    """Shard-local part of the class mean: fp32 sums of (normalised) rows per class + counts."""
    D = mem16.shape[1]
    rows = l2norm_rows(mem16).float() if per_shot_norm else mem16.float()
    sums = torch.zeros(N, D)
    sums.index_add_(0, labels.long(), rows)
    counts = torch.bincount(labels.long(), minlength=N).int()
    return sums, counts


def proto_finalize(sums, counts, fp32=False):
    """Sum slabs in rank order, divide by the class count, then finish as proto_build."""
    s = sums[0].clone()
    for w in range(1, sums.shape[0]):
        s = s + sums[w]
    c = counts.sum(0).clamp_min(1).float().unsqueeze(1)
    z = r16(s / c)
    if fp32:
        return z / z.pow(2).sum(-1, keepdim=True).sqrt()
    return l2norm_rows(z.half())


# ---- classification ----------------------------------------------------------------------------------

def sqdist(q: torch.Tensor, z: torch.Tensor) -> torch.Tensor:
    """torch.cdist(q.float(), z.float(), p=2).pow(2) (utils.py:230-233).  cdist's matmul path computes
    sqrt(clamp(||q||^2 + ||z||^2 - 2 q.z, 0)); the reference squares it again."""
    q, z = q.float(), z.float()
    v = q.pow(2).sum(-1, keepdim=True) + z.pow(2).sum(-1).unsqueeze(0) - 2.0 * (q @ z.t())
    return v.clamp_min(0).sqrt().pow(2)


def softmax_neg(d2: torch.Tensor, beta: float) -> torch.Tensor:
    """F.softmax(beta * (-d2), dim=1) (utils.py:236, 239), max-subtracted, fp32."""
    x = torch.tensor(beta, dtype=torch.float32) * (-d2)
    e = (x - x.max(dim=1, keepdim=True).values).exp()
    return e / e.sum(dim=1, keepdim=True)


def P_from_dists(d2i, d2t, alpha: float, beta: float) -> torch.Tensor:
    """p = alpha * p_i + (1 - alpha) * p_t (utils.py:242); (1 - alpha) is formed in double, cast to fp32."""
    a = torch.tensor(float(alpha), dtype=torch.float32)
    oma = torch.tensor(1 - float(alpha), dtype=torch.float32)
    return a * softmax_neg(d2i, beta) + oma * softmax_neg(d2t, beta)


def P(zq, z_img_proto, z_text_proto, alpha, beta) -> torch.Tensor:
    """utils.py:225-244."""
    return P_from_dists(sqdist(zq, z_img_proto), sqdist(zq, z_text_proto), alpha, beta)


def hp_grid():
    """main.py:142-146."""
    return np.arange(0, 1 + 0.1, 0.1).round(1), np.concatenate((np.arange(0.1, 1, 0.1), np.arange(1, 21, 1.0)))


def grid_accuracy(feat, labels, zi, zt, alpha_list=None, beta_list=None) -> np.ndarray:
    """The (alpha, beta) loop of main.py:187-199 / 419-430 for one split: rows (alpha, beta, acc)."""
    if alpha_list is None:
        alpha_list, beta_list = hp_grid()
    d2i, d2t = sqdist(feat, zi), sqdist(feat, zt)
    rows = []
    for alpha in alpha_list:
        for beta in beta_list:
            p = P_from_dists(d2i, d2t, alpha, beta)
            acc = (p.max(1)[1] == labels).float().mean().item()
            rows.append([alpha, beta, acc])
    return np.array(rows)


# ---- adapters -----------------------------------------------------------------------------------------

def _ln(x32: torch.Tensor, w, b, eps=1e-5) -> torch.Tensor:
    """nn.LayerNorm over ALL trailing dims of x32[0] with fp32 statistics (biased variance); fp16 output."""
    dims = tuple(range(1, x32.dim()))
    mean = x32.mean(dim=dims, keepdim=True)
    var = (x32 - mean).pow(2).mean(dim=dims, keepdim=True)
    return r16((x32 - mean) / torch.sqrt(var + eps) * w.float() + b.float())


def adapter_fc(x16, sd, ratio=0.2) -> torch.Tensor:
    """Adapter_FC.forward (model.py:91-95) with fp16 parameters `sd` (keys fc.0.weight ... fc.3.bias)."""
    x = x16.float()
    h = r16(x @ sd["fc.0.weight"].float().t())
    h = _ln(h, sd["fc.1.weight"], sd["fc.1.bias"])
    h = r16(h @ sd["fc.2.weight"].float().t())
    h = _ln(h, sd["fc.3.weight"], sd["fc.3.bias"])
    r, omr = torch.tensor(ratio, dtype=torch.float32), torch.tensor(1 - ratio, dtype=torch.float32)
    return r16(r16(r * h) + r16(omr * x)).half()


def adapter_conv(x16, sd, c_type: str) -> torch.Tensor:
    """Adapter.forward (model.py:49-78): pad to s*s, conv1 1x1, LN, [conv2 3x3, LN], conv3 1x1, LN, +identity."""
    B, D = x16.shape
    s = int(math.ceil(math.sqrt(D)))
    x = F.pad(x16.float(), (0, s * s - D)).view(B, 1, s, s)
    out = r16(F.conv2d(x, sd["conv1.weight"].float()))
    out = _ln(out, sd["bn1.weight"], sd["bn1.bias"])
    if c_type == "conv-3x":
        out = r16(F.conv2d(out, sd["conv2.weight"].float(), padding=1))
        out = _ln(out, sd["bn2.weight"], sd["bn2.bias"])
    out = r16(F.conv2d(out, sd["conv3.weight"].float()))
    out = _ln(out, sd["bn3.weight"], sd["bn3.bias"])
    out = r16(out + x)
    return out.view(B, s * s)[:, :D].half()


# ---- whole test pass (main.py:383-455) -------------------------------------------------------------

def run_test_pass(cfg, keys, values, val_f, val_y, test_f, test_y, emb_v, emb_t, adapter_sd):
    """Restatement of the reference's test block given saved banks + adapter state dict."""
    D, NK = keys.shape
    K = cfg["shots"]
    N = NK // K
    zi = proto_build(emb_v, N, K, True)
    zt = l2norm_rows(emb_t)
    ad = (lambda x: adapter_fc(x, adapter_sd)) if cfg["adapter"] == "fc" else (lambda x: adapter_conv(x, adapter_sd, cfg["adapter"]))
    test_a = l2norm_rows(ad(test_f))
    train_a = l2norm_rows(ad(keys.t().contiguous()))
    val_a = ad(val_f)                                    # not normalised (main.py:415)
    train_y = values.argmax(1)
    out = dict(val=grid_accuracy(val_a, val_y, zi, zt), test=grid_accuracy(test_a, test_y, zi, zt),
               train=grid_accuracy(train_a, train_y, zi, zt))
    p = P(test_a, zi, zt, cfg["alpha"], cfg["beta"])
    out["fixed_acc"] = (p.max(1)[1] == test_y).float().mean().item()
    return out


def run_zero_shot(cfg, keys, values, val_f, val_y, test_f, test_y, text_bank):
    """main.py:172-199."""
    D, NK = keys.shape
    K = cfg["shots"]
    N = NK // K
    rows = keys.t().contiguous()
    zi = proto_build(rows, N, K, per_shot_norm=False)
    zt = l2norm_rows(text_bank.t().contiguous())
    train_y = values.argmax(1)
    return dict(val=grid_accuracy(l2norm_rows(val_f), val_y, zi, zt), test=grid_accuracy(l2norm_rows(test_f), test_y, zi, zt),
                train=grid_accuracy(l2norm_rows(rows), train_y, zi, zt))
