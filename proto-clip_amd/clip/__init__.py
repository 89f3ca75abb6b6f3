"""Drop-in for the reference's vendored `clip` package surface used on the hot path
(clip/clip.py:92 `load`, :194 `tokenize`; clip/model.py `build_model`)."""
from .clip import available_models, load, tokenize
from .model import BACKBONES, CLIP, build_model, random_state_dict

__all__ = ["available_models", "load", "tokenize", "build_model", "CLIP", "BACKBONES", "random_state_dict"]
