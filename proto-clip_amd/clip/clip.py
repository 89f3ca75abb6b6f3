"""`clip.load` / `clip.tokenize` with the reference's signatures (clip/clip.py:92-139, 194-230).

There is no network in the build environment, so `load` never downloads: it accepts a model name whose
checkpoint already sits in `download_root` (default ~/.cache/clip, the reference's cache directory,
clip/clip.py:118) or a path to a checkpoint, and builds the gfx950 model from its state dict.
`name="random:<backbone>"` builds seeded random-init weights of that architecture (benchmarks/tests)."""
import os
from typing import List, Union

import torch

from .._lib import PclipError
from .model import BACKBONES, build_model, random_state_dict

_FILES = {"RN50": "RN50.pt", "RN101": "RN101.pt", "ViT-B/32": "ViT-B-32.pt", "ViT-B/16": "ViT-B-16.pt", "ViT-L/14": "ViT-L-14.pt"}


def available_models() -> List[str]:
    return list(_FILES)


def _state_dict_from_file(path):
    try:
        jit = torch.jit.load(path, map_location="cpu").eval()     # OpenAI ships TorchScript archives
        return jit.state_dict()
    except RuntimeError:
        return torch.load(path, map_location="cpu")


def load(name: str, device: Union[str, torch.device] = "cuda", jit: bool = False, download_root: str = None):
    """-> (model, preprocess).  The model lives on `device` (must be a ROCm GPU: there is no CPU path)."""
    if jit:
        raise PclipError("jit=True is not supported: the gfx950 towers are not TorchScript modules")
    if name.startswith("random:"):
        backbone = name.split(":", 1)[1]
        if backbone not in BACKBONES:
            raise RuntimeError(f"Model {backbone} not found; available models = {available_models()}")
        sd = random_state_dict(seed=1, **BACKBONES[backbone])
    elif name in _FILES:
        path = os.path.join(download_root or os.path.expanduser("~/.cache/clip"), _FILES[name])
        if not os.path.isfile(path):
            raise RuntimeError(f"checkpoint {path} not found and downloading is disabled (no network); "
                               f"place the OpenAI checkpoint there or pass a path")
        sd = _state_dict_from_file(path)
    elif os.path.isfile(name):
        sd = _state_dict_from_file(name)
    else:
        raise RuntimeError(f"Model {name} not found; available models = {available_models()}")
    model = build_model(sd).to(device)
    return model, _transform(model.visual.input_resolution)


def _transform(n_px):
    """Resize(bicubic) -> CenterCrop -> RGB -> ToTensor -> Normalize (clip/clip.py:77-84) on the GPU: the returned callable takes
    a PIL image or a uint8 HWC array and returns the normalised `[3, n_px, n_px]` fp32 tensor (on the device — the callers'
    `.cuda()` becomes a no-op); `.batch(list)` processes many images in three launches.  Bit-identical to PIL + torchvision
    (proto_clip_amd/preprocess.py)."""
    from ..preprocess import ClipPreprocess

    class _Preprocess(ClipPreprocess):
        def batch(self, images):
            import numpy as np
            return super().batch([np.asarray(im.convert("RGB")) if hasattr(im, "convert") else im for im in images])

    return _Preprocess(n_px)


def tokenize(texts: Union[str, List[str]], context_length: int = 77, truncate: bool = False) -> torch.LongTensor:
    """[n, context_length] int64, SOT=49406 / EOT=49407, zero padded (clip/clip.py:194-230)."""
    from .simple_tokenizer import default_tokenizer
    tok = default_tokenizer()
    if isinstance(texts, str):
        texts = [texts]
    sot, eot = tok.encoder["<|startoftext|>"], tok.encoder["<|endoftext|>"]
    result = torch.zeros(len(texts), context_length, dtype=torch.long)
    for i, text in enumerate(texts):
        tokens = [sot] + tok.encode(text) + [eot]
        if len(tokens) > context_length:
            if not truncate:
                raise RuntimeError(f"Input {texts[i]} is too long for context length {context_length}")
            tokens = tokens[:context_length]
            tokens[-1] = eot
        result[i, :len(tokens)] = torch.tensor(tokens)
    return result
