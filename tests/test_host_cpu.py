"""CPU-side checks (no GPU, no compute calls): the C-ABI library loads and exports every symbol
include/pclip.h declares, argument validation fails loudly before any launch, host logic (config overlay,
grid definition, sharding arithmetic, synthetic generator, tokenizer) behaves like the reference's."""
import ctypes
import os
import re
import types

import numpy as np
import pytest
import torch

from conftest import REPO, golden
from proto_clip_amd import _lib, synth


def header_symbols():
    txt = open(os.path.join(REPO, "include", "pclip.h")).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(pclip_[a-z0-9_]+)\s*\(", txt)))


def test_library_exports_every_declared_symbol():
    lib = _lib.load()
    names = header_symbols()
    assert len(names) >= 25
    for n in names:
        assert hasattr(lib, n), f"{n} declared in include/pclip.h but not exported by libpclip.so"
    assert sorted(_lib.EXPORTED_SYMBOLS) == names          # the ctypes table binds exactly the header
    assert lib.pclip_abi_version() == 1


def test_argument_validation_is_loud_and_precedes_any_launch():
    lib = _lib.load()
    buf = ctypes.c_void_p(0x1000)      # never dereferenced: validation rejects first
    assert lib.pclip_l2norm_rows_f16(buf, buf, 4, 510, None, None) == -1
    assert b"multiple of 8" in lib.pclip_last_error()
    assert lib.pclip_sqdist_f16(buf, buf, None, 4, 3, 72, None, None, None, buf, None, 64, None, 0, None) == -1
    assert b"multiple of 64" in lib.pclip_last_error()
    assert lib.pclip_proto_build_f16(buf, 3, 0, 512, 1, buf, None, None, None) == -1
    assert lib.pclip_fuse_probs(buf, buf, 4, 5000, 5056, 0.5, 0.5, 1.0, buf, None, None, None, 0, None) == -1
    assert lib.pclip_attention_f16(buf, buf, 1, 300, 2, 64, 0, None) == -1
    assert lib.pclip_attention_f16(buf, buf, 1, 50, 2, 32, 0, None) == -1
    assert lib.pclip_gemm_f16(buf, 100, buf, 100, buf, 8, 4, 8, 100, None, 0, None, None) == -1
    assert lib.pclip_classify_f16(buf, buf, buf, 4, 3, 64, None, None, None, 0.5, 0.5, 1.0, None, buf, None, None, 0,
                                  buf, 16, None) == -3       # workspace too small
    assert lib.pclip_workspace_bytes(2, 50000, 1000, 512) >= 2 * 50000 * 1000 * 4
    assert lib.pclip_attention_config(3, 0) == -1 and lib.pclip_attention_config(-1, 0) == 0


def test_cpu_tensors_are_refused():
    from proto_clip_amd import PclipError, ops
    from proto_clip_amd.utils import P
    x = torch.zeros(4, 512, dtype=torch.float16)
    with pytest.raises(PclipError):
        ops.l2norm_rows(x)
    with pytest.raises(PclipError):
        P(x, x, x, 0.5, 1.0)


def test_missing_library_fails_loudly(monkeypatch):
    monkeypatch.setattr(_lib, "_lib", None)
    monkeypatch.setattr(_lib, "LIB_PATH", "/nonexistent/libpclip.so")
    with pytest.raises(_lib.PclipError, match="no CPU/eager fallback"):
        _lib.load()


def test_hp_grid_and_selection_rule():
    from proto_clip_amd.main import hp_grid, select_hp
    a, b = hp_grid()
    assert len(a) == 11 and len(b) == 29 and a[3] == 0.3 and b[8] == pytest.approx(0.9) and b[9] == 1.0 and b[-1] == 20.0
    acc = np.zeros((319, 3))
    acc[:, 2] = 0.5
    acc[[40, 100], 2] = 0.9          # two equal maxima: the first (alpha-major order) wins, utils.py:197-203
    assert select_hp(acc)[3] == 40


def test_cfg_overlay_follows_reference_truthiness():
    from proto_clip_amd.main import get_arguments, populate_cfg_using_args
    cfg = dict(alpha=0.5, beta=12, adapter="conv-2x", shots=16, backbone="RN50", dataset="imagenet")
    args = get_arguments(["--config", "x.yml", "--alpha", "0", "--beta", "3", "--adapter", "fc", "--backbone", "ViT-B/16",
                          "--dataset", "eurosat", "--train_vis_memory_only", "--only_test"])
    out = populate_cfg_using_args(dict(cfg), args)
    assert out["alpha"] == 0.5            # `--alpha 0` is falsy and ignored, as in main.py:56-57
    assert out["beta"] == 3 and out["adapter"] == "fc" and out["backbone"] == "ViT-B/16" and out["dataset"] == "eurosat"
    assert out["train_vis_mem_only"] is True and out["only_test"] is True


def test_accuracy_from_counts_is_fp32_mean():
    from proto_clip_amd.utils import accuracy_from_counts
    for c, q in [(766, 999), (1, 3), (49999, 50000), (0, 7)]:
        ref = (torch.arange(q) < c).float().mean().item()
        assert float(accuracy_from_counts(c, q)) == ref


def test_shard_bounds_partition():
    from proto_clip_amd.dist import shard_bounds
    for n, w in [(16000, 8), (3168, 8), (10, 3), (5, 8), (0, 4)]:
        b = [shard_bounds(n, r, w) for r in range(w)]
        assert b[0][0] == 0 and b[-1][1] == n
        assert all(b[i][1] == b[i + 1][0] for i in range(w - 1))
        sizes = [hi - lo for lo, hi in b]
        assert max(sizes) - min(sizes) <= 1


def test_synth_is_deterministic_and_laid_out_like_the_reference():
    s1, s2 = synth.make_split(10, 4, 64, 30, 40), synth.make_split(10, 4, 64, 30, 40)
    assert torch.equal(s1.visual_memory_keys, s2.visual_memory_keys) and torch.equal(s1.test_features, s2.test_features)
    assert s1.visual_memory_keys.shape == (64, 40) and s1.visual_memory_keys.dtype == torch.float16    # [D, N*K]
    assert s1.visual_memory_values.shape == (40, 10) and s1.visual_memory_values.dtype == torch.int64
    assert torch.equal(s1.visual_memory_values.argmax(1), torch.arange(10).repeat_interleave(4))       # sorted by class
    assert s1.textual_memory_bank.shape == (64, 10)
    n = s1.val_features.float().norm(dim=-1)
    assert (n - 1).abs().max() < 2e-3
    # known answers of the portable PRNG: published first output of splitmix64(seed=0) is 0xE220A8397B1DCDAF
    assert int(synth._splitmix64(np.arange(2, dtype=np.uint64), 0)[1]) == 0xE220A8397B1DCDAF
    assert synth.uniform(3, 1).tolist() == [0.8833108082136427, 0.43152799704851, 0.0264337715925978]
    np.testing.assert_allclose(synth.normal((4,), 1, 0), [0.40576161808654904, 1.244090962132839, 0.288987556045254,
                                                          0.364806048344448], rtol=1e-14)


def test_tokenizer_matches_reference_ids(monkeypatch):
    """clip.tokenize against ids produced by the reference's tokenizer (tests/golden/tokenizer.npz: the six probe prompts, all
    7 ImageNet templates x 50 class names, punctuation / unicode / html entities / whitespace / contractions / digits, the
    empty string, a 77-token-overflow).  The merge table ships with the package, so this runs everywhere."""
    from proto_clip_amd.clip import simple_tokenizer
    from proto_clip_amd.clip.clip import tokenize
    monkeypatch.delenv("PCLIP_BPE_VOCAB", raising=False)
    monkeypatch.setattr(simple_tokenizer, "_default", None)
    g = golden("tokenizer")
    prompts = [str(p) for p in g["prompts"]]
    assert len(prompts) >= 360
    ids = tokenize(prompts)
    assert ids.shape == (len(prompts), 77) and ids.dtype == torch.int64
    ref = torch.from_numpy(g["ids"]).long()
    bad = [prompts[i] for i in range(len(prompts)) if not torch.equal(ids[i], ref[i])]
    assert not bad, bad[:5]
    assert ids[0, :9].tolist() == [49406, 320, 1125, 539, 320, 1929, 269, 49407, 0]      # SURVEY §8c probe
    assert torch.equal(tokenize("a photo of a dog."), ids[:1])                           # str input == one-element list
    with pytest.raises(RuntimeError):
        tokenize("word " * 100)
    assert torch.equal(tokenize("word " * 100, truncate=True), torch.from_numpy(g["truncated"]).long())
    tok = simple_tokenizer._default
    assert tok.decode(tok.encode("a photo of a dog.")).strip() == "a photo of a dog ."


def test_tokenizer_missing_merge_table_is_a_loud_error(monkeypatch, tmp_path):
    from proto_clip_amd.clip import simple_tokenizer
    from proto_clip_amd.clip.clip import tokenize
    monkeypatch.setattr(simple_tokenizer, "_default", None)
    monkeypatch.setenv("PCLIP_BPE_VOCAB", str(tmp_path / "nope.txt.gz"))
    with pytest.raises(FileNotFoundError, match="PCLIP_BPE_VOCAB"):
        tokenize("a photo of a dog.")
    monkeypatch.delenv("PCLIP_BPE_VOCAB")
    assert os.path.isfile(simple_tokenizer.default_bpe())                                # the shipped copy
    assert tokenize("a photo of a dog.")[0, 0].item() == 49406


def test_clip_load_checkpoint_formats(tmp_path):
    """clip.load's file handling (reference clip/clip.py:92-139): a plain state-dict file, a TorchScript archive (what OpenAI
    ships: the reference calls torch.jit.load(...).state_dict() and rebuilds, clip/clip.py:126-139), a name resolved inside
    download_root, and the error paths.  Building the model needs no GPU until `.to(device)`, so device='cpu' parameters are
    checked here; the GPU test (tests/test_gpu_encoder.py::test_clip_load_runs_on_gpu) pushes images through the result."""
    from proto_clip_amd.clip import clip as pclip
    from proto_clip_amd.clip.model import random_state_dict
    from conftest import TINY
    sd = random_state_dict(seed=3, **TINY)
    path_sd = tmp_path / "tiny_sd.pt"
    torch.save(sd, path_sd)

    class Holder(torch.nn.Module):                # a scriptable module whose state_dict() has OpenAI's key names
        def __init__(self, sd_):
            super().__init__()
            for k, v in sd_.items():
                mod, parts = self, k.split(".")
                for part in parts[:-1]:
                    if not hasattr(mod, part):
                        mod.add_module(part, torch.nn.Module())
                    mod = getattr(mod, part)
                mod.register_buffer(parts[-1], v.clone())

        def forward(self, x):
            return x

    # plus the three metadata entries OpenAI's archives carry (clip/model.py:427-429 deletes them)
    meta = dict(sd, input_resolution=torch.tensor(TINY["image_resolution"]), context_length=torch.tensor(77), vocab_size=torch.tensor(TINY["vocab_size"]))
    path_jit = tmp_path / "tiny_jit.pt"
    torch.jit.save(torch.jit.script(Holder(meta)), str(path_jit))
    got = pclip._state_dict_from_file(str(path_jit))
    assert set(got) == set(meta) and all(torch.equal(got[k], meta[k]) for k in meta)
    got = pclip._state_dict_from_file(str(path_sd))
    assert list(got) == list(sd)
    for path in (path_sd, path_jit):
        model, preprocess = pclip.load(str(path), device="cpu")
        msd = model.state_dict()
        assert set(msd) == set(sd)
        for k, v in sd.items():                   # convert_weights: fp16 for Linear / conv / projections, fp32 elsewhere
            want = v.half() if msd[k].dtype == torch.float16 else v.float()
            assert torch.equal(msd[k], want.to(msd[k].dtype)), k
        assert msd["visual.conv1.weight"].dtype == torch.float16 and msd["visual.ln_pre.weight"].dtype == torch.float32
        assert model.visual.input_resolution == TINY["image_resolution"] and preprocess.n_px == TINY["image_resolution"]
    # a model NAME is looked up in download_root under the reference's file names (clip/clip.py:30-38, 118-121); no download
    root = tmp_path / "cache"
    root.mkdir()
    torch.save(sd, root / "ViT-B-16.pt")
    model, _ = pclip.load("ViT-B/16", device="cpu", download_root=str(root))
    assert set(model.state_dict()) == set(sd)
    with pytest.raises(RuntimeError, match="not found and downloading is disabled"):
        pclip.load("RN50", device="cpu", download_root=str(root))
    with pytest.raises(RuntimeError, match="available models"):
        pclip.load("ViT-Z/99", device="cpu")
    with pytest.raises(Exception, match="jit"):
        pclip.load(str(path_sd), device="cpu", jit=True)
    assert pclip.available_models() == ["RN50", "RN101", "ViT-B/32", "ViT-B/16", "ViT-L/14"]


def test_model_surface_and_state_dict_names():
    """Adapters keep the reference's key names/shapes; CLIP keeps OpenAI's."""
    from proto_clip_amd.clip.model import build_model, random_state_dict
    from proto_clip_amd.model import Adapter, Adapter_FC
    from conftest import TINY
    a = Adapter(1024, "conv-2x", dtype=torch.half)
    sd = a.state_dict()
    assert list(sd) == ["conv1.weight", "bn1.weight", "bn1.bias", "conv2.weight", "bn2.weight", "bn2.bias", "conv3.weight",
                        "bn3.weight", "bn3.bias"]
    assert sd["conv1.weight"].shape == (16, 1, 1, 1) and sd["bn1.weight"].shape == (16, 32, 32) and sd["conv2.weight"].shape == (16, 16, 3, 3)
    assert sd["conv3.weight"].shape == (1, 16, 1, 1) and sd["bn3.bias"].shape == (1, 32, 32) and sd["bn1.weight"].dtype == torch.float16
    f = Adapter_FC(768, dtype=torch.half).state_dict()
    assert {k: tuple(v.shape) for k, v in f.items()} == {"fc.0.weight": (192, 768), "fc.1.weight": (192,), "fc.1.bias": (192,),
                                                         "fc.2.weight": (768, 192), "fc.3.weight": (768,), "fc.3.bias": (768,)}
    assert Adapter(512, "conv-3x").bn1.weight.shape == (16, 23, 23) and Adapter(768, "conv-3x").bn3.weight.shape == (1, 28, 28)
    ck = "/root/reference/pretrained_ckpt/imagenet-F/query_adapter.pt"
    if os.path.exists(ck):                       # the shipped checkpoints load unchanged
        a.load_state_dict(torch.load(ck, map_location="cpu"))
        Adapter_FC(768, dtype=torch.half).load_state_dict(torch.load(ck.replace("imagenet-F", "fewsol-198-F"), map_location="cpu"))
    sd = random_state_dict(seed=11, **TINY)
    m = build_model({k: v.clone() for k, v in sd.items()})
    assert set(m.state_dict()) == set(sd)
    assert m.dtype == torch.float16 and m.visual.input_resolution == 32
    assert m.state_dict()["visual.ln_pre.weight"].dtype == torch.float32          # convert_weights semantics
    assert m.state_dict()["transformer.resblocks.0.attn.in_proj_weight"].dtype == torch.float16
    with pytest.raises(Exception):
        m.encode_image(torch.zeros(1, 3, 32, 32))                                   # CPU tensor: loud, no fallback


# ---------------------------------------------------------------- training host logic -------------------------
def test_episode_sampler_reproduces_the_reference_run():
    """proto_clip_amd.train.sample_epoch against the episodes the reference drew (tests/golden/train_*.npz)."""
    from conftest import golden
    from golden.spec import TRAIN
    from proto_clip_amd.train import sample_epoch
    for name in TRAIN:
        g = golden("train_" + name)
        N, K = int(g["meta"][0]), int(g["meta"][1])
        rng = np.random.RandomState(1)
        labels, sizes = [], []
        for _ in range(TRAIN[name][11]):
            for classes, q_idx, q_lab in sample_epoch(N, K, rng):
                assert all(i // K == l for i, l in zip(q_idx, q_lab)) and sorted(set(q_lab)) == [int(c) for c in classes if int(c) in q_lab]
                labels.extend(q_lab)
                sizes.append(len(q_lab))
        assert sizes == list(g["ep_sizes"]) and labels == list(g["ep_labels"].astype(int))


def test_cosine_lr_matches_torch_scheduler():
    from proto_clip_amd.train import cosine_lr
    p = torch.nn.Parameter(torch.zeros(1))
    opt = torch.optim.AdamW([p], lr=0.003)
    sched = torch.optim.lr_scheduler.CosineAnnealingLR(opt, 40)
    for epoch in range(1, 45):
        opt.step()
        sched.step()
        assert abs(sched.get_last_lr()[0] - cosine_lr(0.003, epoch, 40)) <= 1e-12 + 1e-9 * 0.003


def test_preprocess_host_rules_match_the_oracle():
    """Resize(int) output size, CenterCrop offsets and tap counts computed by the host wrapper (proto_clip_amd/preprocess.py)
    against the oracle's restatement of torchvision / Pillow."""
    from oracle import preprocess_oracle as pp
    from proto_clip_amd import preprocess as dp
    for h, w, n in [(375, 500, 224), (500, 375, 224), (224, 224, 224), (227, 225, 224), (61, 60, 32), (1000, 37, 64)]:
        assert dp.resize_output_size(h, w, n) == pp.resize_output_size(h, w, n)
    for i, o in [(500, 224), (224, 224), (100, 224), (375, 298), (37, 64), (1000, 64)]:
        ks = dp._ksize(i, o)
        assert ks == (1 if i == o else pp.precompute_coeffs(i, o)[0])
    t = dp.RandomTrainTransform(size=224)
    torch.manual_seed(0)
    for h, w in [(375, 500), (64, 64), (31, 257)]:
        top, left, ch, cw = t.get_params(h, w)
        assert 0 <= top and 0 <= left and top + ch <= h and left + cw <= w and ch > 0 and cw > 0


def test_shape_envelope_is_checked_up_front():
    """The kernels' hard limits surface as ONE clear error before any launch; every shape the reference ships passes."""
    from proto_clip_amd._lib import PclipError
    from proto_clip_amd.main import check_shape_envelope
    for N, K, D, ad in ((1000, 16, 512, "conv-3x"), (100, 1, 1024, "conv-3x"), (10, 16, 512, "fc"), (198, 16, 768, "fc"), (37, 4, 512, "conv-2x")):
        check_shape_envelope(N, K, D, ad, training=True)
    for args, what in (((5000, 16, 512, "fc", False), "classes"), ((10, 64, 512, "fc", True), "shots"), ((10, 4, 2048, "conv-3x", False), "conv adapter"),
                       ((10, 4, 576, "fc", False), "fc adapter"), ((10, 4, 500, "conv-2x", False), "feature dim")):
        with pytest.raises(PclipError, match=what):
            check_shape_envelope(*args)
    check_shape_envelope(10, 64, 512, "fc", training=False)          # the shot limit only binds the training step
