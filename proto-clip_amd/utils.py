"""Hot-path primitives with the reference's names and signatures (reference utils.py:54-69, 225-361),
executed by the gfx950 kernels.  Cache-file names and tensor layouts are the reference's (SURVEY §5,
fact 9) so existing caches stay loadable."""
import os
import pickle

import numpy as np
import torch
import torch.nn.functional as F

from . import autograd as pag
from . import ops
from ._lib import PclipError


def get_seed():
    return 1


def dir_exists(path):
    return os.path.exists(path)


def save(obj, filepath, msg):
    print(f"Saving {msg} to {filepath}")
    with open(filepath, "wb") as handle:
        pickle.dump(obj, handle, protocol=pickle.HIGHEST_PROTOCOL)


def load(filepath, msg):
    print(f"Loading {msg} from {filepath}")
    with open(filepath, "rb") as handle:
        return pickle.load(handle)


def beautify(string):
    return string.strip().replace("/", "_").replace("-", "_")


def get_model_dir_root(cfg):
    return f"{cfg['cache_dir']}/models/{beautify(cfg['backbone'])}/K-{cfg['shots']}"


def _all_f16(*ts):
    for t in ts:
        if t.dtype not in (torch.float16, torch.float32):
            raise PclipError(f"P(): unsupported operand dtype {t.dtype} (fp16 or fp32 expected)")
    return all(t.dtype == torch.float16 for t in ts)


def P(zq_imgs_flat, z_img_proto, z_text_proto, alpha, beta):
    """p = alpha * softmax(-beta * d2(q, z_img)) + (1 - alpha) * softmax(-beta * d2(q, z_txt)), fp32 [Q, N]
    (reference utils.py:225-244).  The reference casts every operand with .float(): fp16 operands (the eval path,
    main.py:399-409) go to the fp16-input MFMA kernel, which is lossless for them; if any operand is fp32 (the
    training path, main.py:262-281) everything is promoted and the exact-fp32 MFMA kernel is used."""
    if pag.wants_grad(zq_imgs_flat, z_img_proto, z_text_proto):        # the training loop: p carries an autograd node (main.py:281-309)
        _all_f16(zq_imgs_flat, z_img_proto, z_text_proto)
        return pag.PFn.apply(zq_imgs_flat, z_img_proto, z_text_proto, alpha, beta)
    if _all_f16(zq_imgs_flat, z_img_proto, z_text_proto):
        p, _, _, _ = ops.classify(zq_imgs_flat, z_img_proto, z_text_proto, alpha, beta, want_p=True, want_argmax=False)
        return p
    d2i, d2t, _ = ops.sqdist_f32(zq_imgs_flat, z_img_proto, z_text_proto)
    p, _, _, _ = ops.fuse_probs(d2i, d2t, z_img_proto.shape[0], alpha, beta, want_p=True)
    return p


def _as_f16_rows(t: torch.Tensor, what: str) -> torch.Tensor:
    if t.dtype == torch.float16:
        return t
    raise PclipError(f"{what} is {t.dtype}: the fused argmax / top-k entry points take the fp16 tensors of the eval path")


def P_argmax(zq_imgs_flat, z_img_proto, z_text_proto, alpha, beta):
    """`P(...).max(1)[1]` without materialising p (main.py:190, 341-345); int64 [Q] like torch's max."""
    _, am, _, _ = ops.classify(_as_f16_rows(zq_imgs_flat, "zq"), _as_f16_rows(z_img_proto, "z_img_proto"),
                               _as_f16_rows(z_text_proto, "z_text_proto"), alpha, beta, want_p=False, want_argmax=True)
    return am.long()


def P_topk(zq_imgs_flat, z_img_proto, z_text_proto, alpha, beta, k):
    """`P(...).topk(k)` fused (toolkit proto_clip_classifier.py:146-147) -> (values [Q,k], indices [Q,k])."""
    _, _, tp, ti = ops.classify(_as_f16_rows(zq_imgs_flat, "zq"), _as_f16_rows(z_img_proto, "z_img_proto"),
                                _as_f16_rows(z_text_proto, "z_text_proto"), alpha, beta, want_p=False,
                                want_argmax=False, topk=k)
    return tp, ti.long()


def image_prototypes(memory_rows: torch.Tensor, N: int, K: int, per_shot_norm: bool = True, fp32: bool = False):
    """The prototype block the reference inlines seven times (main.py:399-402, 260-264, 173-176, ...):
    view [N,K,D] -> (normalise) -> mean over K -> normalise.  memory_rows is [N*K, D] fp16."""
    return ops.proto_build(memory_rows, N, K, per_shot_norm=per_shot_norm, fp32_out=fp32)


def text_prototypes(text_rows: torch.Tensor):
    """zs_text / zs_text.norm(dim=-1, keepdim=True) in fp16 (main.py:404-405)."""
    return ops.l2norm_rows(text_rows)


def cls_acc(output, target, topk=1):
    pred = output.topk(topk, 1, True, True)[1].t()
    correct = pred.eq(target.view(1, -1).expand_as(pred))
    acc = float(correct[:topk].reshape(-1).float().sum(0, keepdim=True).cpu().numpy())
    return 100 * acc / target.shape[0]


def get_target_inds(info):
    """Episode ground-truth labels [n_class, k_query, 1] int64 on the GPU from info = (n_class, k_support, k_query)
    (reference utils.py:112-122; no caller in the reference — kept for API completeness)."""
    n_class, _, k_query = info
    return torch.arange(n_class, device="cuda").view(n_class, 1, 1).expand(n_class, k_query, 1).long()


def clip_classifier(classnames, template, clip_model, tokenize=None):
    """Textual memory bank [D, N] fp16 (reference utils.py:256-273).  All N*T prompts go through the
    text tower in large batches (the reference issues N calls of T prompts) and the per-class
    normalise -> mean over templates -> normalise is one prototype-reduction kernel."""
    if tokenize is None:
        from .clip import tokenize
    texts = [t.format(c.replace("_", " ")) for c in classnames for t in template]
    tokens = tokenize(texts).cuda()
    emb = clip_model.encode_text(tokens)                               # [N*T, D] fp16
    N, T = len(classnames), len(template)
    weights = ops.proto_build(emb, N, T, per_shot_norm=True)           # utils.py:267-270
    return classnames, ops.transpose(weights)                          # stack(dim=1) -> [D, N]


def get_textual_memory_bank(cfg, classnames, template, clip_model, tokenize=None):
    msg = "text_memory_bank"
    model_dir_root = get_model_dir_root(cfg)
    os.makedirs(model_dir_root, exist_ok=True)
    path = os.path.join(model_dir_root, f"text_mb_{beautify(cfg['backbone'])}_K_{cfg['shots']}.pkl")
    if dir_exists(path):
        return classnames, load(path, msg)
    text_prompts, textual_memory_bank = clip_classifier(classnames, template, clip_model, tokenize)
    save(textual_memory_bank, path, msg)
    return text_prompts, textual_memory_bank


def build_cache_model(cfg, clip_model, train_loader_cache):
    """Visual memory bank (reference utils.py:284-332): keys [D, N*K] fp16 (columns sorted by label),
    values one-hot [N*K, N] int64; same cache files."""
    model_dir_root = get_model_dir_root(cfg) + "/aug"
    os.makedirs(model_dir_root, exist_ok=True)
    key_path = f"{model_dir_root}/visual_mb_keys_aug_{cfg['augment_epoch']}_{cfg['shots']}_shots.pt"
    value_path = f"{model_dir_root}/visual_mb_values_aug_{cfg['augment_epoch']}_{cfg['shots']}_shots.pt"
    if dir_exists(key_path) and dir_exists(value_path):
        return torch.load(key_path), torch.load(value_path)

    feats, labels = [], []
    with torch.no_grad():
        for augment_idx in range(cfg["augment_epoch"]):
            print("Augment Epoch: {:} / {:}".format(augment_idx, cfg["augment_epoch"]))
            epoch = []
            for images, target in train_loader_cache:
                epoch.append(clip_model.encode_image(images.cuda()))
                if augment_idx == 0:
                    labels.append(target.cuda())
            feats.append(torch.cat(epoch, dim=0))
    feats = torch.stack(feats, dim=0)                                  # [A, N*K, D] fp16
    cache_values = torch.cat(labels, dim=0)
    index = torch.argsort(cache_values, stable=True)                   # utils.py:324 (order within a class is free)
    keys_rows = ops.bank_reduce(feats, perm=index)                     # mean_A -> normalise -> gather (318-326)
    cache_keys = ops.transpose(keys_rows)                              # [D, N*K] (320)
    cache_values = F.one_hot(cache_values[index])
    torch.save(cache_keys, key_path)
    torch.save(cache_values, value_path)
    return cache_keys, cache_values


def pre_load_features(cfg, split, clip_model, loader):
    """Query features [Q, D] fp16 (unit norm in fp16) + labels (reference utils.py:335-361)."""
    root_dir_prefix = f"{get_model_dir_root(cfg)}/{split}"
    feature_path, label_path = f"{root_dir_prefix}_features.pt", f"{root_dir_prefix}_labels.pt"
    if dir_exists(feature_path) and dir_exists(label_path):
        print(f"Loading cached features and labels from {root_dir_prefix}")
        return torch.load(feature_path), torch.load(label_path)
    print(f"Creating cached (features, labels) and saving to {root_dir_prefix}")
    features, labels = [], []
    with torch.no_grad():
        for images, target in loader:
            f = clip_model.encode_image(images.cuda())
            features.append(ops.l2norm_rows(f, out=f))                 # utils.py:352, in place
            labels.append(target.cuda())
    features, labels = torch.cat(features), torch.cat(labels)
    torch.save(features, feature_path)
    torch.save(labels, label_path)
    return features, labels


def accuracy_from_counts(correct, Q: int) -> np.ndarray:
    """`(p.max(1)[1] == labels).float().mean().item()` from integer correct-counts: the reference's
    mean is an fp32 sum of 0/1 divided by Q in fp32 (main.py:190-191)."""
    c = np.asarray(correct, dtype=np.float32)
    return (c / np.float32(Q)).astype(np.float64)


def InfoNCELoss(A, B):
    """InfoNCE()(A, B) of utils.py:72-77 with the info-nce-pytorch defaults (temperature 0.1, mean reduction, both sides
    L2-normalised, cross entropy against the diagonal); A, B fp32 [n, D].  A device scalar; when an operand requires grad it
    carries an autograd node (proto_clip_amd.autograd.InfoNceFn)."""
    if pag.wants_grad(A, B):
        return pag.InfoNceFn.apply(A, B)
    an, bn = ops.l2norm_rows_f32(A.float().contiguous()), ops.l2norm_rows_f32(B.float().contiguous())
    S = ops.gemm_f32(an, bn, trans_b=True, alpha=10.0)
    rows, _ = ops.softmax_ce_rows(S, 1.0 / A.shape[0])
    return ops.colsum_f32(rows.view(-1, 1), scale=1.0 / A.shape[0])[0]


def compute_loss_and_matches(p, target_inds, z_img_proto, z_text_proto, cfg):
    """Loss and accuracy of one episode (reference utils.py:80-109): the same 7-tuple.  When p (or a prototype matrix) carries an
    autograd graph — the reference's own loop, main.py:281-309 — the loss does too (NllMeanFn / InfoNceFn) and
    `train_loss.backward()` runs the explicit backward kernels; otherwise forward values only.  ProtoClipTrainer.step remains the
    fast path (NLL fused with P, p never materialised).  As in the reference, `neg_log_loss` (index 2) is returned as None."""
    require = p.dtype == torch.float32 and p.is_cuda
    if not require:
        raise PclipError("compute_loss_and_matches: p must be the fp32 CUDA tensor returned by P()")
    nll, _, y_hat = ops.nll_rows(p.detach(), target_inds)
    matches = (y_hat.long() == target_inds).float().sum()
    loss = torch.zeros((), dtype=torch.float32, device=p.device)
    img2txt = txt2img = img_inter = txt_inter = None
    losses = cfg["losses"]
    if len(losses) == 0 or "L1" in losses:
        if pag.wants_grad(p):
            loss = loss + pag.NllMeanFn.apply(p, target_inds)
        else:
            loss = loss + ops.colsum_f32(nll.view(-1, 1), scale=1.0 / p.shape[0])[0]
    if "L2" in losses:
        img2txt = InfoNCELoss(z_img_proto, z_text_proto)
        loss = loss + img2txt
    if "L3" in losses:
        txt2img = InfoNCELoss(z_text_proto, z_img_proto)
        loss = loss + txt2img
    if "L4" in losses:
        img_inter, txt_inter = InfoNCELoss(z_img_proto, z_img_proto), InfoNCELoss(z_text_proto, z_text_proto)
        loss = loss + img_inter + txt_inter
    return matches, loss, None, img2txt, txt2img, img_inter, txt_inter
