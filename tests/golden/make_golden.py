#!/usr/bin/env python3
"""Generates the committed golden fixtures by RUNNING THE REFERENCE ITSELF (imported read-only from
/root/reference under the SURVEY Appendix-B shim) on seeded synthetic inputs.  Runs only in the build
container; nothing of the reference (source, bytecode, pickled modules) is written to this repo —
fixtures hold input seeds/shapes and the reference's OUTPUT tensors only.

    python tests/golden/make_golden.py            # all fixtures
    python tests/golden/make_golden.py C2 tiny    # a subset

Inputs are regenerated in the tests from proto_clip_amd.synth / clip.model.random_state_dict with the
same seeds, so they are not stored (except the small random adapter state dicts).
"""
import io
import os
import sys
import tempfile
import types
import contextlib

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
REPO = os.path.dirname(os.path.dirname(HERE))
REF = "/root/reference"
sys.path.insert(0, REPO)

from proto_clip_amd import synth                                   # noqa: E402
from proto_clip_amd.clip.model import random_state_dict            # noqa: E402
sys.path.insert(0, HERE)
from spec import (E2E, E2E_CASE, E2E_VARIANTS, ENCODERS, ENCODERS_FULL, FEWSHOT, RESNETS, TRAIN, TRAIN_QT, e2e_arch, e2e_images, e2e_jitter, e2e_state_dict, e2e_variant_images, fewshot_inputs,   # noqa: E402
                  randomize_adapter_, train_inputs)


# ---------------------------------------------------------------- Appendix-B shim -----------------
def install_shim():
    def mod(name, **attrs):
        m = types.ModuleType(name)
        m.__dict__.update(attrs)
        sys.modules[name] = m
        return m

    ident = lambda *a, **k: (lambda x: x)

    class _Interp:
        BICUBIC = 3

    tv = mod("torchvision")
    tf = mod("torchvision.transforms", Compose=ident, Resize=ident, CenterCrop=ident, ToTensor=ident, Normalize=ident,
             RandomResizedCrop=ident, RandomHorizontalFlip=ident, InterpolationMode=_Interp)
    tff = mod("torchvision.transforms.functional", to_tensor=lambda x: torch.zeros(3, 4, 4))
    tv.transforms, tf.functional = tf, tff
    tv.datasets = mod("torchvision.datasets")
    mod("ftfy", fix_text=lambda s: s)
    mod("gdown")

    class SummaryWriter:
        def __init__(self, *a, **k): pass
        def add_scalar(self, *a, **k): pass
        def add_image(self, *a, **k): pass
        def close(self): pass

    mod("torch.utils.tensorboard", SummaryWriter=SummaryWriter)

    class InfoNCE(torch.nn.Module):       # PyPI info-nce-pytorch defaults (absent here; training only)
        def forward(self, q, k):
            q, k = torch.nn.functional.normalize(q, dim=-1), torch.nn.functional.normalize(k, dim=-1)
            return torch.nn.functional.cross_entropy(q @ k.t() / 0.1, torch.arange(len(q)))

    mod("info_nce", InfoNCE=InfoNCE)
    torch.Tensor.cuda = lambda self, *a, **k: self
    torch.nn.Module.cuda = lambda self, *a, **k: self
    torch.cuda.manual_seed_all = lambda *a, **k: None
    import matplotlib
    matplotlib.use("Agg")
    sys.path.insert(0, REF)


def import_reference():
    install_shim()
    scratch = tempfile.mkdtemp(prefix="pclip_golden_")
    os.chdir(scratch)                       # the reference writes ./caches and ./plots
    import main as ref_main                 # noqa
    import utils as ref_utils               # noqa
    import model as ref_model               # noqa
    import clip as ref_clip                 # noqa
    import clip.model as ref_clip_model     # noqa
    ref_main.plot_tsne = lambda *a, **k: None       # visualisation, off the hot path
    return ref_main, ref_utils, ref_model, ref_clip, ref_clip_model, scratch


def savez(name, **arrays):
    path = os.path.join(HERE, name + ".npz")
    np.savez_compressed(path, **{k: (v.numpy() if isinstance(v, torch.Tensor) else np.asarray(v)) for k, v in arrays.items()})
    print(f"wrote {path} ({os.path.getsize(path) / 1024:.0f} KiB)")


# ---------------------------------------------------------------- few-shot configs ------------------
def adapter_state(ref_model, kind, D, seed):
    torch.manual_seed(seed)
    ad = ref_model.Adapter_FC(D, dtype=torch.half) if kind == "fc" else ref_model.Adapter(D, c_type=kind, dtype=torch.half)
    return randomize_adapter_(ad, seed)


def make_fewshot(name, ref_main, ref_utils, ref_model, scratch):
    N, K, D, Qv, Qt, alpha, beta, kind, unnorm, sigma = FEWSHOT[name]
    split, emb_v, emb_t, cfg = fewshot_inputs(name)
    cfg.update(cache_dir=os.path.join(scratch, "caches", name), logs_dir_path="logs")
    ad = adapter_state(ref_model, kind, D, seed=7)
    model_dir = f"{ref_utils.get_model_dir_root(cfg)}/alpha-beta/{alpha}-{beta}"
    os.makedirs(model_dir, exist_ok=True)
    prefix = f"{model_dir}/best_lr_{cfg['lr']}_aug_{cfg['augment_epoch']}_epochs_{cfg['train_epoch']}"
    torch.save(torch.nn.Parameter(emb_v), prefix + "_v.pt")
    torch.save(torch.nn.Parameter(emb_t), prefix + "_t.pt")
    torch.save(ad.state_dict(), prefix + "_a.pt")

    calls = []
    real_P = ref_utils.P

    n_grid = 319 * 3

    def spy_P(zq, zi, zt, a, b):
        p = real_P(zq, zi, zt, a, b)
        keep_p = len(calls) >= 2 * n_grid               # only the two single calls after the grids keep p
        calls.append((zq, zi, zt, float(a), float(b), p if keep_p else None, p.max(1)[1].to(torch.int16)))
        return p

    ref_main.P = spy_P
    clip_stub = types.SimpleNamespace(dtype=torch.float16)
    buf = io.StringIO()
    with contextlib.redirect_stdout(buf):
        ref_main.run_proto_clip(cfg, split.visual_memory_keys, split.visual_memory_values, split.val_features,
                                split.val_labels, split.test_features, split.test_labels, split.textual_memory_bank,
                                clip_stub, [str(i) for i in range(N)])
    ref_main.P = real_P
    log = buf.getvalue()
    assert len(calls) == 2 * n_grid + 2, len(calls)
    zs_calls, test_calls = calls[:n_grid], calls[n_grid:2 * n_grid]
    fixed_call, hp_call = calls[2 * n_grid], calls[2 * n_grid + 1]

    def grid(cs, off):
        # rebuild accuracy the way the reference does (main.py:190-199) from the recorded p tensors
        return cs[off::3]

    def acc_rows(cs, labels):
        return np.array([[a, b, (am.long() == labels).float().mean().item()] for (_, _, _, a, b, _, am) in cs])

    train_labels = split.visual_memory_values.argmax(1)
    sel = torch.from_numpy(synth.randint(16, Qt, 1, 900))
    zq_fixed, zi_test, zt_test, a_f, b_f, p_fixed, _ = fixed_call
    adapted_val = test_calls[0][0]          # adapter(val) un-normalised (main.py:415)
    savez(
        "fewshot_" + name,
        meta=np.array([N, K, D, Qv, Qt]), alpha=alpha, beta=beta,
        adapter_keys=np.array(list(ad.state_dict().keys())),
        **{"adapter__" + k: v for k, v in ad.state_dict().items()},
        zs_proto_img=zs_calls[0][1], zs_proto_txt=zs_calls[0][2],
        zs_val=acc_rows(grid(zs_calls, 0), split.val_labels), zs_test=acc_rows(grid(zs_calls, 1), split.test_labels),
        zs_train=acc_rows(grid(zs_calls, 2), train_labels),
        test_proto_img=zi_test, test_proto_txt=zt_test,
        test_val=acc_rows(grid(test_calls, 0), split.val_labels), test_test=acc_rows(grid(test_calls, 1), split.test_labels),
        test_train=acc_rows(grid(test_calls, 2), train_labels),
        adapted_test_norm=zq_fixed[:64], adapted_val_raw=adapted_val[:64],
        fixed_argmax=p_fixed.max(1)[1].to(torch.int16), fixed_acc=(p_fixed.max(1)[1] == split.test_labels).float().mean().item(),
        p_rows_idx=sel, p_rows=p_fixed[sel],
        hp_alpha=hp_call[3], hp_beta=hp_call[4],
        hp_acc=(hp_call[5].max(1)[1] == split.test_labels).float().mean().item(),
    )
    assert "Fixed-alp-beta" in log


# ---------------------------------------------------------------- training runs ----------------------
def make_train(name, ref_main, ref_utils, scratch):
    """The reference's own training loop (main.py:216-381) on a small seeded split.  Recorded: the adapter as the
    reference initialised it, every episode's labels / loss terms, the gradients and updated parameters of the first
    three optimizer steps, the parameters after the last step and the per-epoch validation accuracy."""
    N, K, D, Qv, Qt, alpha, beta, kind, sigma, vis_only, losses, epochs, lr = TRAIN[name]
    split, cfg = train_inputs(name)
    cfg.update(cache_dir=os.path.join(scratch, "caches", name), logs_dir_path="logs")
    episodes, steps, holder = [], [], {}
    real_clm = ref_main.compute_loss_and_matches

    def spy_clm(p, target, zi, zt, c):
        out = real_clm(p, target, zi, zt, c)
        terms = [float("nan") if t is None else float(t) for t in out[2:5]]
        # out[2] (neg_log_loss) is never assigned by the reference; recompute L1 the way utils.py:90-93 does
        l1 = float(torch.nn.NLLLoss()(torch.log(p), target)) if (len(c["losses"]) == 0 or "L1" in c["losses"]) else float("nan")
        episodes.append((target.clone(), float(out[0]), float(out[1]), l1, terms[1], terms[2]))
        return out

    real_step = torch.optim.AdamW.step

    def spy_step(self, *a, **k):
        params = [p for g in self.param_groups for p in g["params"]]
        holder["opt"] = self
        if len(steps) < 3:
            before = [p.detach().clone() for p in params]
            grads = [None if p.grad is None else p.grad.detach().clone() for p in params]
            r = real_step(self, *a, **k)
            steps.append((before, grads, [p.detach().clone() for p in params]))
            return r
        r = real_step(self, *a, **k)
        holder["last"] = [p.detach().clone() for p in params]       # the test block later reloads the BEST adapter (main.py:391)
        return r

    ref_main.compute_loss_and_matches = spy_clm
    torch.optim.AdamW.step = spy_step
    clip_stub = types.SimpleNamespace(dtype=torch.float16)
    buf = io.StringIO()
    torch.manual_seed(1)
    np.random.seed(1)
    try:
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
            ref_main.run_proto_clip(cfg, split.visual_memory_keys, split.visual_memory_values, split.val_features,
                                    split.val_labels, split.test_features, split.test_labels, split.textual_memory_bank,
                                    clip_stub, [str(i) for i in range(N)])
    finally:
        ref_main.compute_loss_and_matches = real_clm
        torch.optim.AdamW.step = real_step
    log = buf.getvalue()
    import re
    val_acc = [float(x) for x in re.findall(r"val accuracy: ([0-9.]+)%", log)]
    fixed = float(re.search(r"Fixed-alp-beta: Proto-CLIP's test accuracy: ([0-9.]+)%", log).group(1))
    params = [p for g in holder["opt"].param_groups for p in g["params"]]
    # parameter order of main.py:123-128
    if kind == "fc":
        ad_names = ["fc.0.weight", "fc.1.weight", "fc.1.bias", "fc.2.weight", "fc.3.weight", "fc.3.bias"]
    else:
        ad_names = ["conv1.weight", "bn1.weight", "bn1.bias", "conv2.weight", "bn2.weight", "bn2.bias", "conv3.weight",
                    "bn3.weight", "bn3.bias"]
    names = ad_names + ["visual"] if vis_only else ["visual", "textual"] + ad_names
    assert len(names) == len(params), (len(names), len(params))
    arrays = dict(meta=np.array([N, K, D, Qv, Qt]), names=np.array(names), n_episodes=len(episodes),
                  ep_sizes=np.array([len(e[0]) for e in episodes]), ep_labels=torch.cat([e[0] for e in episodes]).to(torch.int16),
                  ep_matches=np.array([e[1] for e in episodes]), ep_loss=np.array([e[2] for e in episodes]),
                  ep_l1=np.array([e[3] for e in episodes]), ep_l2=np.array([e[4] for e in episodes]),
                  ep_l3=np.array([e[5] for e in episodes]), val_acc=np.array(val_acc), fixed_acc=fixed)
    for si, (before, grads, after) in enumerate(steps):
        for n, b, g, a in zip(names, before, grads, after):
            if si == 0:
                arrays[f"init__{n}"] = b
            if g is not None:
                arrays[f"grad{si}__{n}"] = g
            arrays[f"after{si}__{n}"] = a
    for n, p in zip(names, holder["last"]):
        arrays[f"final__{n}"] = p
    savez("train_" + name, **arrays)


def make_train_qt(name, ref_utils, ref_model, ref_clip_model, scratch):
    """The reference's Proto-CLIP-F-Q^T loop (main.qt.py:75-330) itself: queries = clip_model.encode_image(batch) of a training image loader (main.qt.py:198-201),
    both memory banks and the adapter learnable, prototypes over every class.  Towers: the small ViT of the image -> logits fixtures with a 256-wide embedding (the fc adapter's kernels take multiples of 256; fp16 weights); banks from the
    reference's own builders on seeded images; the loader = the support images in fixed batches.  Recorded: the banks / features the run starts from, the adapter as
    initialised, and for the first three optimizer steps the encoded query features, labels, loss terms, gradients and updated parameters."""
    import importlib.util
    import builtins
    from datasets.imagenet import imagenet_classes, imagenet_templates
    spec_ = importlib.util.spec_from_file_location("ref_main_qt", os.path.join(REF, "main.qt.py"))
    qt = importlib.util.module_from_spec(spec_)
    spec_.loader.exec_module(qt)
    qt.plot_tsne = lambda *a, **k: None
    c = TRAIN_QT[name]
    N, K = c["N"], c["K"]
    case = dict(E2E_CASE, N=N, K=K, seed=c["seed"])
    sd = random_state_dict(seed=c["sd_seed"], **dict(E2E, embed_dim=c["embed_dim"]))
    with contextlib.redirect_stdout(io.StringIO()):
        m = ref_clip_model.build_model({k: v.clone() for k, v in sd.items()})
    (sup_x, sup_y), (val_x, val_y), (test_x, test_y) = e2e_images(case)
    classnames = [imagenet_classes[i] for i in (0, 1, 2, 21, 15, 43, 7, 99)][:N]
    cfg = dict(shots=K, backbone="ViT-B/16", dataset="synthetic_" + name, only_test=False, lr=c["lr"], augment_epoch=1, train_epoch=c["epochs"], alpha=c["alpha"],
               beta=c["beta"], adapter=c["adapter"], train_vis_mem_only=False, losses=c["losses"], cache_dir=os.path.join(scratch, "caches", name), logs_dir_path="logs")
    with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
        keys, values = ref_utils.build_cache_model(cfg, m, [(sup_x, sup_y)])
        val_f, val_l = ref_utils.pre_load_features(cfg, "val", m, [(val_x, val_y)])
        test_f, test_l = ref_utils.pre_load_features(cfg, "test", m, [(test_x, test_y)])
        _, text_bank = ref_utils.clip_classifier(classnames, imagenet_templates[:2], m)
    keys, val_f, test_f, text_bank = keys.half(), val_f.half(), test_f.half(), text_bank.half()
    bs = c["batch"]
    loader = [(sup_x[i:i + bs], sup_y[i:i + bs]) for i in range(0, len(sup_y), bs)]
    feats, episodes, steps, holder = [], [], [], {}
    real_enc = m.encode_image

    def spy_enc(images):
        f = real_enc(images)
        feats.append(f.detach().clone())
        return f

    m.encode_image = spy_enc
    real_clm = qt.compute_loss_and_matches

    def spy_clm(p, target, zi, zt, cc):
        out = real_clm(p, target, zi, zt, cc)
        l1 = float(torch.nn.NLLLoss()(torch.log(p), target))
        episodes.append((target.clone(), float(out[0]), float(out[1]), l1, float(out[3]), float(out[4])))
        return out

    real_step = torch.optim.AdamW.step

    def spy_step(self, *a, **k):
        params = [p for g in self.param_groups for p in g["params"]]
        holder["opt"] = self
        if len(steps) < 3:
            before = [p.detach().clone() for p in params]
            grads = [None if p.grad is None else p.grad.detach().clone() for p in params]
            r = real_step(self, *a, **k)
            steps.append((before, grads, [p.detach().clone() for p in params]))
            return r
        return real_step(self, *a, **k)

    qt.compute_loss_and_matches = spy_clm
    torch.optim.AdamW.step = spy_step
    real_input = builtins.input
    builtins.input = lambda *a, **k: ""
    buf = io.StringIO()
    torch.manual_seed(1)
    np.random.seed(1)
    try:
        with contextlib.redirect_stdout(buf), contextlib.redirect_stderr(io.StringIO()):
            qt.run_proto_clip(cfg, keys, values, val_f, val_l, test_f, test_l, text_bank, m, classnames, loader)
    finally:
        torch.optim.AdamW.step = real_step
        builtins.input = real_input
    import re
    log = buf.getvalue()
    val_acc = [float(x) for x in re.findall(r"val accuracy: ([0-9.]+)%", log)]
    ad_names = ["fc.0.weight", "fc.1.weight", "fc.1.bias", "fc.2.weight", "fc.3.weight", "fc.3.bias"] if c["adapter"] == "fc" else \
        ["conv1.weight", "bn1.weight", "bn1.bias", "conv2.weight", "bn2.weight", "bn2.bias", "conv3.weight", "bn3.weight", "bn3.bias"]
    names = ["visual", "textual"] + ad_names                     # parameter order of main.qt.py:98-99
    params = [p for g in holder["opt"].param_groups for p in g["params"]]
    assert len(names) == len(params), (len(names), len(params))
    arrays = dict(names=np.array(names), keys=keys, values=values.to(torch.int16), text_bank=text_bank, val_acc=np.array(val_acc), n_steps=len(episodes),
                  ep_matches=np.array([e[1] for e in episodes]), ep_loss=np.array([e[2] for e in episodes]), ep_l1=np.array([e[3] for e in episodes]),
                  ep_l2=np.array([e[4] for e in episodes]), ep_l3=np.array([e[5] for e in episodes]))
    for si, (before, grads, after) in enumerate(steps):
        arrays[f"zq{si}"] = feats[si]
        arrays[f"labels{si}"] = episodes[si][0].to(torch.int16)
        for n, b, g, a in zip(names, before, grads, after):
            if si == 0:
                arrays[f"init__{n}"] = b
            if g is not None:
                arrays[f"grad{si}__{n}"] = g
            arrays[f"after{si}__{n}"] = a
    print("%s: %d steps, val acc per epoch %s, loss %.4f -> %.4f" % (name, len(episodes), val_acc, episodes[0][2], episodes[-1][2]))
    savez("train_" + name, **arrays)


# ---------------------------------------------------------------- shipped checkpoints ---------------
def make_shipped(ref_model):
    """The two adapter checkpoints the reference ships on seeded inputs: real trained weights."""
    for tag, sub, kind, D in (("imagenetF", "imagenet-F", "conv-2x", 1024), ("fewsol198F", "fewsol-198-F", "fc", 768)):
        sd = torch.load(os.path.join(REF, "pretrained_ckpt", sub, "query_adapter.pt"), map_location="cpu")
        ad = ref_model.Adapter_FC(D, dtype=torch.half) if kind == "fc" else ref_model.Adapter(D, c_type=kind, dtype=torch.half)
        ad.load_state_dict(sd)
        x = synth.make_split(8, 8, D, 8, 8, seed=3).visual_memory_keys.t().contiguous()     # 64 unit rows
        with torch.no_grad():
            y = ad(x)
        tb = torch.load(os.path.join(REF, "pretrained_ckpt", sub, "memory_bank_t.pt"), map_location="cpu").detach()
        zt = tb / tb.norm(dim=-1, keepdim=True)                                             # main.py:404-405 on the real bank
        savez("shipped_" + tag, y=y, text_rows_head=tb[:32], text_proto_head=zt[:32], n_text=np.array(tb.shape))


# ---------------------------------------------------------------- encoders --------------------------
def synth_tokens(n, vocab, seed):
    """SOT, random body, EOT (= highest id, as the reference's argmax gather requires), zero padding."""
    t = torch.zeros(n, 77, dtype=torch.long)
    lens = synth.randint(n, 20, seed, 300) + 3
    body = synth.randint(n * 77, vocab - 2, seed, 301).reshape(n, 77)
    for i in range(n):
        t[i, 0] = vocab - 2
        t[i, 1:lens[i]] = torch.from_numpy(body[i, 1:lens[i]])
        t[i, lens[i]] = vocab - 1
    return t


def make_encoder(tag, kw, ref_clip_model, ref_utils):
    sd = random_state_dict(seed=11, **kw)
    with contextlib.redirect_stdout(io.StringIO()):
        m16 = ref_clip_model.build_model({k: v.clone() for k, v in sd.items()})      # fp16 weights (GPU-path precision)
        m32 = ref_clip_model.build_model({k: v.clone() for k, v in sd.items()}).float()   # clip.load(device='cpu')
    res = kw["image_resolution"]
    imgs = synth.make_images(24, res, seed=5, n_class=6)
    toks = synth_tokens(12, kw["vocab_size"], seed=5)
    with torch.no_grad():
        f16, f32 = m16.encode_image(imgs), m32.encode_image(imgs)
        t16, t32 = m16.encode_text(toks), m32.encode_text(toks)
        # hot-path callers on top of the fp16 model (utils.py:284-361), list-of-batches loaders
        labels = torch.from_numpy(synth.randint(24, 6, 5, 51))
        loader = [(imgs[:10], labels[:10]), (imgs[10:], labels[10:])]
        cfg = dict(cache_dir=os.path.join(os.getcwd(), "enc_" + tag), backbone="tiny", shots=4, augment_epoch=2)
        with contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            keys, values = ref_utils.build_cache_model(cfg, m16, loader)
            feats, flabels = ref_utils.pre_load_features(cfg, "val", m16, loader)
    savez("encoder_" + tag, img_f16=f16, img_f32=f32, txt_f16=t16, txt_f32=t32, tokens=toks, cache_keys=keys,
          cache_values=values.to(torch.int16), cache_labels=labels, pre_features=feats, pre_labels=flabels)


def make_encoder_full(tag, ref_clip_model):
    """The reference's OWN towers at a full-size architecture (spec.ENCODERS_FULL: the hyper-parameters build_model infers from OpenAI's
    ViT-B/16 checkpoint, clip/model.py:397-434): encode_image (clip/model.py:221-238, 338-339) on 8 images and encode_text
    (clip/model.py:341-354) on 8 prompts, with fp16 weights (its GPU-path precision) and as the fp32 model clip.load(device='cpu')
    yields.  Outputs only (8 x 512 x 4 tensors); seeded random-init weights (no checkpoints in the build environment)."""
    from proto_clip_amd.clip.model import BACKBONES
    kw = BACKBONES[ENCODERS_FULL[tag]["backbone"]]
    n_img, n_txt = ENCODERS_FULL[tag]["n_img"], ENCODERS_FULL[tag]["n_txt"]
    sd = random_state_dict(seed=ENCODERS_FULL[tag]["sd_seed"], **kw)
    with contextlib.redirect_stdout(io.StringIO()):
        m16 = ref_clip_model.build_model({k: v.clone() for k, v in sd.items()})
        m32 = ref_clip_model.build_model({k: v.clone() for k, v in sd.items()}).float()
    imgs = synth.make_images(n_img, kw["image_resolution"], seed=5, n_class=6)
    toks = synth_tokens(n_txt, kw["vocab_size"], seed=5)
    with torch.no_grad():
        f32, t32 = m32.encode_image(imgs), m32.encode_text(toks)
        f16, t16 = m16.encode_image(imgs), m16.encode_text(toks)
    savez("encoder_" + tag, img_f16=f16, img_f32=f32, txt_f16=t16, txt_f32=t32, tokens=toks)


def make_resnet(tag, kw, ref_clip_model):
    """ModifiedResNet tower of the reference (fp16-weight and fp32 variants) on seeded weights / images."""
    sd = random_state_dict(seed=13, **kw)
    with contextlib.redirect_stdout(io.StringIO()):
        m16 = ref_clip_model.build_model({k: v.clone() for k, v in sd.items()})
        m32 = ref_clip_model.build_model({k: v.clone() for k, v in sd.items()}).float()
    imgs = synth.make_images(6, kw["image_resolution"], seed=5, n_class=6)
    with torch.no_grad():
        f16, f32 = m16.encode_image(imgs), m32.encode_image(imgs)
    savez("encoder_" + tag, img_f16=f16, img_f32=f32)


def make_tokenizer(ref_clip):
    """clip.tokenize of the reference (clip/clip.py:194-230) on: the original six prompts, every ImageNet template x 50 class
    names (the prompts utils.clip_classifier builds, utils.py:262-263), and punctuation / unicode / html-entity /
    whitespace cases.  ftfy is absent from the image (shim: identity), so the unicode cases avoid mojibake."""
    from datasets.imagenet import imagenet_classes, imagenet_templates
    prompts = ["a photo of a dog.", "a centered satellite photo of annual crop land.", "itap of a forest.",
               "a bad photo of the tench, tinca tinca.", "A Photo Of The Large golden_retriever!!", "art of the 3-d   printer's nozzle"]
    names = imagenet_classes[:30] + imagenet_classes[400:410] + imagenet_classes[-10:]
    prompts += [t.format(c.replace("_", " ")) for c in names for t in imagenet_templates]
    prompts += ["caf\u00e9 cr\u00e8me br\u00fbl\u00e9e", "na\u00efve \u00fcber-stra\u00dfe", "\u65e5\u672c\u8a9e\u306e\u5199\u771f", "emoji \U0001f600 test", "tom &amp; jerry's &lt;cat&gt;", "  spaces\t and\n newlines  ",
                "they're we've i'm he'll she'd it's don't", "price: $1,234.56 (approx.) #42 @home 100%", "MiXeD CaSe WORDS and 12345 67890", "x", "",
                "a photo of a " + "very " * 30 + "long prompt"]
    ids = ref_clip.tokenize(prompts)
    too_long = "word " * 100
    savez("tokenizer", prompts=np.array(prompts), ids=ids.to(torch.int32), truncated=ref_clip.tokenize(too_long, truncate=True).to(torch.int32))


# ---------------------------------------------------------------- image -> logits chain ------------------
def make_e2e(name, ref_main, ref_utils, ref_model, ref_clip_model, scratch):
    """Fixture `name` of spec.E2E_VARIANTS (seeded weights / images / adapter; the `trained` ones carry trained-CLIP-like LayerNorm
    parameters and residual-stream outlier channels, spec.trained_like_).  Images through the reference's whole hot path (utils.py:256-361 bank builders on the reference's CLIP towers, then
    main.py:383-441 via run_proto_clip with a spy on P): support images -> build_cache_model, prompts -> clip_classifier,
    query images -> pre_load_features -> adapter -> normalise -> P.  Run twice: fp16-weight towers (the reference's GPU
    precision) and fp32 towers with the features cast to fp16 (the reference CPU path) — their disagreement is the
    yard-stick the GPU test's tolerance is stated against."""
    from datasets.imagenet import imagenet_classes, imagenet_templates
    var = E2E_VARIANTS[name]
    c = var["case"]
    N, K = c["N"], c["K"]
    classnames = [imagenet_classes[i] for i in (0, 1, 2, 21, 15, 43)][:N]
    templates = imagenet_templates[:c["n_templates"]]
    sd = e2e_state_dict(name)
    (sup_x, sup_y), (val_x, val_y), (test_x, test_y) = e2e_variant_images(name)
    ad = adapter_state(ref_model, c["adapter"], e2e_arch(name)["embed_dim"], seed=var["adapter_seed"])
    out = {}
    clean = (sup_x, val_x, test_x)
    for tag in ("f16", "f32", "f16_jitter"):
        with contextlib.redirect_stdout(io.StringIO()):
            m = ref_clip_model.build_model({k: v.clone() for k, v in sd.items()})
        if tag == "f32":
            m = m.float()
        sup_x, val_x, test_x = clean
        if tag == "f16_jitter":      # the fp16 chain again on images with 2 % of the pixels moved by one fp16 ulp: the comparator's self-noise
            sup_x, val_x, test_x = (e2e_jitter(x, c["seed"] + i) for i, x in enumerate(clean))
        cfg = dict(shots=K, backbone="ViT-B/16", dataset="synthetic_" + name + "_" + tag, only_test=True, lr=0.0001, augment_epoch=c["augment_epoch"],
                   train_epoch=1, alpha=c["alpha"], beta=c["beta"], adapter=c["adapter"], train_vis_mem_only=True, losses=["L1"],
                   cache_dir=os.path.join(scratch, "caches", name + "_" + tag), logs_dir_path="logs")
        with torch.no_grad(), contextlib.redirect_stdout(io.StringIO()), contextlib.redirect_stderr(io.StringIO()):
            keys, values = ref_utils.build_cache_model(cfg, m, [(sup_x[:10], sup_y[:10]), (sup_x[10:], sup_y[10:])])
            val_f, val_l = ref_utils.pre_load_features(cfg, "val", m, [(val_x, val_y)])
            test_f, test_l = ref_utils.pre_load_features(cfg, "test", m, [(test_x[:20], test_y[:20]), (test_x[20:], test_y[20:])])
            _, text_bank = ref_utils.clip_classifier(classnames, templates, m)
        keys, val_f, test_f, text_bank = keys.half(), val_f.half(), test_f.half(), text_bank.half()
        model_dir = f"{ref_utils.get_model_dir_root(cfg)}/alpha-beta/{c['alpha']}-{c['beta']}"
        os.makedirs(model_dir, exist_ok=True)
        prefix = f"{model_dir}/best_lr_{cfg['lr']}_aug_{cfg['augment_epoch']}_epochs_{cfg['train_epoch']}"
        torch.save(torch.nn.Parameter(keys.t().contiguous()), prefix + "_v.pt")        # the banks a fresh training run starts from
        torch.save(torch.nn.Parameter(text_bank.t().contiguous()), prefix + "_t.pt")
        torch.save(ad.state_dict(), prefix + "_a.pt")
        calls, real_P = [], ref_utils.P

        def spy_P(zq, zi, zt, a, b):
            p = real_P(zq, zi, zt, a, b)
            calls.append((zq, zi, zt, float(a), float(b), p))
            return p

        ref_main.P = spy_P
        with contextlib.redirect_stdout(io.StringIO()):
            ref_main.run_proto_clip(cfg, keys, values, val_f, val_l, test_f, test_l, text_bank, types.SimpleNamespace(dtype=torch.float16),
                                    classnames)
        ref_main.P = real_P
        n_grid = 319 * 3
        assert len(calls) == 2 * n_grid + 2, len(calls)
        zq, zi, zt, a, b, p = calls[2 * n_grid]                      # the fixed-(alpha, beta) test call, main.py:433
        assert (a, b) == (c["alpha"], c["beta"])
        if tag == "f16_jitter":
            out.update({"p_f16_jitter": p, "test_features_f16_jitter": test_f})
            continue
        out.update({f"keys_{tag}": keys, f"test_features_{tag}": test_f, f"text_bank_{tag}": text_bank, f"adapted_{tag}": zq,
                    f"proto_img_{tag}": zi, f"proto_txt_{tag}": zt, f"p_{tag}": p, f"argmax_{tag}": p.max(1)[1].to(torch.int16),
                    f"acc_{tag}": (p.max(1)[1] == test_l).float().mean().item()})
        if tag == "f16":
            out["values"] = values.to(torch.int16)
    print("%s: acc f16 %.3f, f32 %.3f; max|p16 - p32| %.3e; max|p16 - p16(jittered input)| %.3e; argmax agree %d/%d" % (
        name, out["acc_f16"], out["acc_f32"], (out["p_f16"] - out["p_f32"]).abs().max().item(),
        (out["p_f16"] - out["p_f16_jitter"]).abs().max().item(), int((out["argmax_f16"] == out["argmax_f32"]).sum()), len(test_y)))
    savez(name, classnames=np.array(classnames), templates=np.array(templates), adapter_keys=np.array(list(ad.state_dict().keys())),
          **{"adapter__" + k: v for k, v in ad.state_dict().items()}, **out)


def main():
    want = set(sys.argv[1:])
    ref_main, ref_utils, ref_model, ref_clip, ref_clip_model, scratch = import_reference()
    todo = lambda k: not want or k in want
    for name in FEWSHOT:
        if todo(name):
            make_fewshot(name, ref_main, ref_utils, ref_model, scratch)
    for name in TRAIN:
        if todo(name):
            make_train(name, ref_main, ref_utils, scratch)
    for name in TRAIN_QT:
        if todo(name):
            make_train_qt(name, ref_utils, ref_model, ref_clip_model, scratch)
    if todo("shipped"):
        make_shipped(ref_model)
    for tag, kw in ENCODERS.items():
        if todo(tag):
            make_encoder(tag, kw, ref_clip_model, ref_utils)
    for tag in ENCODERS_FULL:
        if todo(tag):
            make_encoder_full(tag, ref_clip_model)
    for tag, kw in RESNETS.items():
        if todo(tag):
            make_resnet(tag, kw, ref_clip_model)
    if todo("tokenizer"):
        make_tokenizer(ref_clip)
    for name in E2E_VARIANTS:
        if todo(name) or "e2e" in want:
            make_e2e(name, ref_main, ref_utils, ref_model, ref_clip_model, scratch)


if __name__ == "__main__":
    main()
