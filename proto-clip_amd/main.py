"""Experiment driver with the reference's CLI and `run_proto_clip` signature (reference main.py:24-102,
105-465, 474-548): memory-bank construction, zero-shot (alpha, beta) search, the episodic training loop
(main.py:216-381 -> `train_proto_clip` / proto_clip_amd.train.ProtoClipTrainer; main.qt.py through main_qt.run_proto_clip)
and the test pass over saved banks / adapters.  `only_test: True` configs skip training as in the reference.

Differences by design: the 957 `P` calls + `.item()` syncs of each grid search (main.py:187-199, 419-430)
become three distance GEMMs + three sweep kernels; results (the three [319, 3] arrays, their pickle
files, the selected (alpha, beta)) are the reference's.  One CLI difference (also listed in INTEGRATION.md):
`populate_cfg_using_args` copies --only_test and --train_vis_memory_only into cfg when they are given; the reference parses
them and then ignores them (its yaml decides)."""
import argparse
import os
import random

import numpy as np
import torch
import yaml

from . import ops
from .model import Adapter, Adapter_FC
from .utils import (accuracy_from_counts, beautify, build_cache_model, get_model_dir_root, get_seed,
                    get_textual_memory_bank, load, pre_load_features, save)


def get_arguments(argv=None):
    parser = argparse.ArgumentParser()
    parser.add_argument("--logs", dest="logs_dir_path", help="log directory path", required=False)
    parser.add_argument("--config", dest="config", help="settings of Proto-CLIP in yaml format", required=True)
    parser.add_argument("--alpha", dest="alpha", help="alpha", type=float, required=False)
    parser.add_argument("--beta", dest="beta", help="beta", type=float, required=False)
    parser.add_argument("--adapter", dest="adapter", help="adapter to use: ['conv-3x', 'conv-2x', 'fc']", type=str,
                        required=False)
    parser.add_argument("--train_vis_memory_only", dest="train_vis_mem_only", help="train visual memory only",
                        action="store_true")
    parser.add_argument("--only_test", dest="only_test", help="flag to perorm only testing", action="store_true")
    parser.add_argument("--shots", dest="shots", help="shots in few-shot setups", type=int, required=False)
    parser.add_argument("--losses", nargs="+", dest="losses", help="List of loss aliases: {'L1', 'L2', 'L3'}",
                        required=False)
    parser.add_argument("--backbone", dest="backbone", help="backbones: [ViT-B/16, ViT-B/32, ViT-L/14]", type=str,
                        required=False)
    parser.add_argument("--dataset", dest="dataset", help="dataset alias", required=False)
    return parser.parse_args(argv)


def populate_cfg_using_args(cfg, args):
    """Truthy CLI values overlay the YAML (main.py:52-71); note `--alpha 0` is a no-op there too."""
    for key in ("logs_dir_path", "alpha", "beta", "adapter", "shots", "losses", "backbone", "dataset"):
        v = getattr(args, key, None)
        if v:
            cfg[key] = v
    # the reference declares these store_true flags but never copies them into cfg (main.py:36-39, 52-71);
    # honouring them is what the README's command lines intend
    if getattr(args, "train_vis_mem_only", False):
        cfg["train_vis_mem_only"] = True
    if getattr(args, "only_test", False):
        cfg["only_test"] = True
    return cfg


def hp_grid(rounded: bool = True):
    """alpha in {0, .1, ..., 1} (rounded in main.py:142-144, left as np.arange yields them in main.qt.py:110-111),
    beta in {.1...0.9} U {1...20} (main.py:145-146)."""
    alpha_list = np.arange(0, 1 + 0.1, 0.1)
    if rounded:
        alpha_list = alpha_list.round(1)
    beta_list = np.concatenate((np.arange(0.1, 1, 0.1), np.arange(1, 21, 1.0)))
    return alpha_list, beta_list


def grid_accuracy(features, labels, z_img_proto, z_text_proto, alpha_list, beta_list):
    """[na*nb, 3] float64 rows (alpha, beta, acc) in the reference's alpha-major order (main.py:187-199)."""
    d2i, d2t, _ = ops.sqdist(features, z_img_proto, z_text_proto)
    correct = ops.hp_sweep(d2i, d2t, z_img_proto.shape[0], labels, alpha_list, beta_list)
    acc = accuracy_from_counts(correct.cpu().numpy(), features.shape[0])        # single host sync per split
    aa, bb = np.meshgrid(alpha_list, beta_list, indexing="ij")
    return np.stack([aa.ravel(), bb.ravel(), acc.ravel()], axis=1)


def select_hp(val_acc_list):
    """First maximum of validation accuracy, alpha-major order (utils.py:197-203)."""
    idx = int(val_acc_list[:, 2].argmax())
    return val_acc_list[idx, 0], val_acc_list[idx, 1], val_acc_list[idx, 2], idx


def fixed_accuracy(features, labels, z_img_proto, z_text_proto, alpha, beta):
    _, am, _, _ = ops.classify(features, z_img_proto, z_text_proto, alpha, beta, want_p=False, want_argmax=True)
    correct = (am.long() == labels.to(am.device)).sum().item()
    return float(accuracy_from_counts(correct, features.shape[0]))


def make_adapter(cfg, ndim):
    if "conv" in cfg["adapter"]:
        return Adapter(ndim, c_type=cfg["adapter"], dtype=torch.half).cuda()
    if cfg["adapter"] == "fc":
        return Adapter_FC(ndim, dtype=torch.half).cuda()
    raise ValueError(f"unknown adapter {cfg['adapter']!r}")


def train_proto_clip(cfg, visual_memory_keys, textual_memory_bank, adapter, val_features, val_labels, alpha, beta,
                     clip_model=None, train_loader_F=None, subdir="alpha-beta"):
    """The training loop of reference main.py:216-381: episodes of proto_clip_amd.train, per-epoch validation with the
    inference kernels, best-on-validation checkpoints under the reference's file names."""
    from .train import ProtoClipTrainer
    K = cfg["shots"]
    trainer = ProtoClipTrainer(cfg, visual_memory_keys, textual_memory_bank, adapter, alpha, beta)
    N = trainer.N
    model_dir = f"{get_model_dir_root(cfg)}/{subdir}/{alpha}-{beta}"
    model_prefix = f"best_lr_{cfg['lr']}_aug_{cfg['augment_epoch']}_epochs_{cfg['train_epoch']}"
    os.makedirs(model_dir, exist_ok=True)
    pv, pt, pa = (os.path.join(model_dir, f"{model_prefix}_{s}.pt") for s in ("v", "t", "a"))
    best_acc, best_epoch, history = 0.0, 0, []
    for epoch in range(cfg["train_epoch"]):
        print("Train Epoch: {:} / {:}".format(epoch, cfg["train_epoch"]))
        train_acc, train_loss, lr = trainer.train_epoch(clip_model=clip_model, train_loader_F=train_loader_F)   # main.py:228-312
        print("LR: {:.6f}, Acc: {:.4f}%, Loss: {:.4f}".format(lr, train_acc * 100, train_loss))
        with torch.no_grad():                                                                # main.py:318-345
            z_img_proto = ops.proto_build(trainer.visual, N, K, per_shot_norm=True)
            z_text_proto = ops.l2norm_rows(trainer.textual)
            val_f = adapter(val_features, l2norm_out=True)
            _, am, tp, _ = ops.classify(val_f, z_img_proto, z_text_proto, alpha, beta, want_p=False, want_argmax=True, topk=1)
            val_acc = float(accuracy_from_counts((am.long() == val_labels.to(am.device)).sum().item(), val_f.shape[0]))
            val_loss = float(-torch.log(tp[:, 0]).mean().item())
        print("**** Proto-CLIP's val accuracy: {:.2f}% | loss: {:.2f}***\n".format(val_acc * 100, val_loss))
        if val_acc >= best_acc:                                                              # main.py:359-364
            best_acc, best_epoch = val_acc, epoch
            torch.save(torch.nn.Parameter(trainer.visual.clone()), pv)
            torch.save(torch.nn.Parameter(trainer.textual.clone()), pt)
            torch.save(adapter.state_dict(), pa)
        history.append(dict(epoch=epoch, lr=lr, train_acc=train_acc, train_loss=train_loss, val_acc=val_acc, val_loss=val_loss))
    print(f"Best model: best_val_acc = {best_acc * 100: .2f}, best_val_epoch = {best_epoch}")
    return dict(history=history, best_acc=best_acc, best_epoch=best_epoch, trainer=trainer)


# per-dataset (search_scale, search_step) constants the reference writes into cfg before a run (main.py:74-103; Tip-Adapter
# legacy, read by nothing on this path — kept so that a cfg dict leaves run_proto_clip with the same keys)
_SEARCH = {"caltech101": ([12, 5], [200, 20]), "dtd": ([13, 13], [200, 20]), "eurosat": ([12, 10], [200, 20]),
           "fgvc": ([30, 30], [200, 20]), "food101": ([10, 10], [200, 20]), "imagenet": ([7, 3], [200, 20]),
           "oxford_flowers": ([50, 50], [200, 20]), "oxford_pets": ([7, 3], [200, 20]), "stanford_cars": ([20, 10], [200, 20]),
           "sun397": ([12, 10], [200, 20]), "ucf101": ([7, 3], [200, 20]), "fewsol": ([13, 13], [200, 20])}


def search_scale_step(cfg):
    cfg["search_scale"], cfg["search_step"] = _SEARCH.get(cfg.get("dataset"), (None, None))
    return cfg


def check_shape_envelope(N, K, D, adapter, training, eval_path=True):
    """The kernels' hard limits, checked up front with one clear message instead of a PclipError from deep inside a step
    (README 'Supported shapes'; the reference itself has none of these limits — every dataset and backbone it ships fits)."""
    problems = []
    if N > 4096:
        problems.append(f"{N} classes: the softmax / fusion / (alpha, beta)-sweep kernels hold a class row in registers (N <= 4096)")
    if eval_path and (D % 64 or D > 4096):
        problems.append(f"feature dim {D}: the fp16 distance GEMM of the evaluation path needs a multiple of 64, <= 4096")
    if adapter in ("conv-2x", "conv-3x") and D > 1024:
        problems.append(f"conv adapter at D = {D}: its [16, s, s] stack lives in LDS (D <= 1024)")
    if adapter == "fc" and (D % 256):
        problems.append(f"fc adapter at D = {D}: D and D / 4 must be multiples of 64")
    if training and K > 32:
        problems.append(f"{K} shots: the prototype backward keeps a class's shots in LDS (K <= 32)")
    if training and D > 2048:
        problems.append(f"training at D = {D}: the LayerNorm / prototype backward kernels take D <= 2048")
    if problems:
        from ._lib import PclipError
        raise PclipError("outside the supported shape envelope: " + "; ".join(problems))


def run_proto_clip(cfg, visual_memory_keys, visual_memory_values, val_features, val_labels, test_features,
                   test_labels, textual_memory_bank, clip_model, text_prompts, train_loader_F=None, variant="main"):
    """Reference main.py:105-465.  Returns a dict of everything it computed."""
    ndim, NxK = visual_memory_keys.shape
    K = cfg["shots"]
    N = NxK // K
    check_shape_envelope(N, K, ndim, cfg.get("adapter", "fc"), training=not cfg.get("only_test", False))
    cfg = search_scale_step(cfg)                                        # main.py:111
    qt = variant == "qt"                                               # main.qt.py: queries from the image loader
    subdir = "best-alpha-beta" if qt else "alpha-beta"                 # main.qt.py:292, 327
    alpha_list, beta_list = hp_grid(rounded=not qt)
    model_dir_root = get_model_dir_root(cfg)
    os.makedirs(model_dir_root, exist_ok=True)
    tag = f"{beautify(cfg['backbone'])}_K_{cfg['shots']}"
    paths = {s: os.path.join(model_dir_root, f"zero_shot_hp_search_{s}_{tag}.pkl") for s in ("val", "test", "train")}
    train_labels = torch.argmax(visual_memory_values, dim=1)
    keys_rows = ops.transpose(visual_memory_keys)                      # [N*K, D]
    text_rows = ops.transpose(textual_memory_bank)                     # [N, D]
    out = {}

    with torch.no_grad():
        if all(os.path.exists(p) for p in paths.values()):
            val_acc_list, test_acc_list, train_acc_list = (load(paths[s], f"hp based on {s} set") for s in ("val", "test", "train"))
        else:
            # zero-shot-init prototypes: bank rows are already unit norm (main.py:173-178)
            z_img_proto = ops.proto_build(keys_rows, N, K, per_shot_norm=False)
            z_text_proto = ops.l2norm_rows(text_rows)
            train_f = ops.l2norm_rows(keys_rows)                       # 179-180
            val_f = ops.l2norm_rows(val_features)                      # 182-185
            test_f = ops.l2norm_rows(test_features)
            val_acc_list = grid_accuracy(val_f, val_labels, z_img_proto, z_text_proto, alpha_list, beta_list)
            test_acc_list = grid_accuracy(test_f, test_labels, z_img_proto, z_text_proto, alpha_list, beta_list)
            train_acc_list = grid_accuracy(train_f, train_labels, z_img_proto, z_text_proto, alpha_list, beta_list)
            for s, arr in (("val", val_acc_list), ("test", test_acc_list), ("train", train_acc_list)):
                save(arr, paths[s], f"hp based on {s} set")
        a, b, acc, idx = select_hp(val_acc_list)
        print(f"alpha: {a: .3f}, beta:{b: .3f} | Max val-acc: {acc * 100: .3f} | "
              f"Max test-acc-using-val-alpha-beta: {test_acc_list[idx, 2] * 100: .3f}")
        out["zero_shot"] = dict(val=val_acc_list, test=test_acc_list, train=train_acc_list, best_alpha=a, best_beta=b)

    best_alpha, best_beta = cfg["alpha"], cfg["beta"]                  # main.py:213-214
    # the reference constructs nn.Embedding(NxK, ndim) (whose default init draws NxK*ndim normals) BEFORE the adapter
    # (main.py:110-117): draw them too, so that the adapter initialises identically under the same torch seed
    torch.nn.Embedding(num_embeddings=NxK, embedding_dim=ndim)
    adapter = make_adapter(cfg, ndim)
    if not cfg.get("only_test", False):
        if qt and train_loader_F is None:
            raise ValueError("the main.qt.py variant trains on images: pass train_loader_F")
        out["train"] = train_proto_clip(cfg, visual_memory_keys, textual_memory_bank, adapter, val_features, val_labels,
                                        best_alpha, best_beta, clip_model=clip_model, train_loader_F=train_loader_F if qt else None,
                                        subdir=subdir)

    with torch.no_grad():
        print("Testing...")
        model_dir = f"{model_dir_root}/{subdir}/{best_alpha}-{best_beta}"
        model_prefix = f"best_lr_{cfg['lr']}_aug_{cfg['augment_epoch']}_epochs_{cfg['train_epoch']}"
        pv, pt, pa = (os.path.join(model_dir, f"{model_prefix}_{s}.pt") for s in ("v", "t", "a"))
        try:
            embeddings_v = torch.load(pv).cuda()
            embeddings_t = torch.load(pt).cuda()
            adapter.load_state_dict(torch.load(pa))
        except Exception:
            raise FileNotFoundError(f"File does not exist: {pv} and {pt}")
        z_img_proto = ops.proto_build(embeddings_v.detach(), N, K, per_shot_norm=True)        # 399-402
        z_text_proto = ops.l2norm_rows(embeddings_t.detach())                                 # 404-405
        test_f = adapter(test_features, l2norm_out=True)                                      # 407-409
        train_f = adapter(keys_rows, l2norm_out=True)                                         # 411-413
        val_f = adapter(val_features)                                  # adapted, NOT normalised (main.py:415)
        val_acc_list = grid_accuracy(val_f, val_labels, z_img_proto, z_text_proto, alpha_list, beta_list)
        test_acc_list = grid_accuracy(test_f, test_labels, z_img_proto, z_text_proto, alpha_list, beta_list)
        train_acc_list = grid_accuracy(train_f, train_labels, z_img_proto, z_text_proto, alpha_list, beta_list)
        fixed = fixed_accuracy(test_f, test_labels, z_img_proto, z_text_proto, best_alpha, best_beta)   # 436-438
        print("**** Fixed-alp-beta: Proto-CLIP's test accuracy: {:.2f}% ****\n".format(fixed * 100))
        print("fixed_best_alpha", best_alpha, "fixed_best_beta", best_beta)
        a, b, _, _ = select_hp(val_acc_list)
        searched = fixed_accuracy(test_f, test_labels, z_img_proto, z_text_proto, a, b)                 # 448-450
        print("**** HP-search: Proto-CLIP's test accuracy: {:.2f}% ****\n".format(searched * 100))
        print("hp_search_best_alpha", a, "hp_search_best_beta", b)
        out["test"] = dict(val=val_acc_list, test=test_acc_list, train=train_acc_list, fixed_acc=fixed,
                           hp_alpha=a, hp_beta=b, hp_acc=searched)
    return out


def main(argv=None, dataset=None, clip_model=None):
    """CLI entry (reference main.py:474-548).  Dataset readers (JPEG/PIL host work, reference datasets/*)
    are out of scope: pass `dataset` = object with train_loader / val_loader / test_loader / classnames /
    template, or use proto_clip_amd.synth for seeded synthetic splits."""
    args = get_arguments(argv)
    assert os.path.exists(args.config)
    cfg = yaml.load(open(args.config, "r"), Loader=yaml.Loader)
    if args.dataset is None:
        raise SystemExit("Please provide alias of dataset")
    cfg = populate_cfg_using_args(cfg, args)
    cfg["cache_dir"] = os.path.join("./caches", cfg["dataset"])
    os.makedirs(cfg["cache_dir"], exist_ok=True)
    print("\nRunning configs.")
    print(cfg, "\n")
    if clip_model is None:
        from . import clip
        clip_model, _ = clip.load(cfg["backbone"])
    clip_model.eval()
    seed = get_seed()
    random.seed(seed)
    np.random.seed(seed)
    torch.manual_seed(seed)
    if dataset is None:
        raise SystemExit("dataset readers are not part of this build: supply loaders (see main() docstring)")
    keys, values = build_cache_model(cfg, clip_model, dataset.train_loader)
    text_prompts, text_bank = get_textual_memory_bank(cfg, dataset.classnames, dataset.template, clip_model)
    val_f, val_y = pre_load_features(cfg, "val", clip_model, dataset.val_loader)
    test_f, test_y = pre_load_features(cfg, "test", clip_model, dataset.test_loader)
    return run_proto_clip(cfg, keys, values, val_f, val_y, test_f, test_y, text_bank, clip_model, text_prompts)


if __name__ == "__main__":
    main()
