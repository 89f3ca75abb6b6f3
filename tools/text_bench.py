#!/usr/bin/env python3
"""Textual memory bank at ImageNet size: 1000 classes x 7 templates through tokenizer + text tower + prototype
reduction; host tokenisation and device time separately.  Random-init ViT-B/16 text tower (12 x 512, 8 heads, 77 tokens)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import clip as pclip, ops
from proto_clip_amd.clip.model import BACKBONES, build_model, random_state_dict
from proto_clip_amd.utils import clip_classifier

model = build_model(random_state_dict(seed=1, **BACKBONES["ViT-B/16"])).cuda()
classes = [f"class number {i} thing" for i in range(1000)]
templates = ["itap of a {}.", "a bad photo of the {}.", "a origami {}.", "a photo of the large {}.", "a {} in a video game.", "art of the {}.",
             "a photo of the small {}."]                                              # datasets/imagenet.py:193-199
texts = [t.format(c) for c in classes for t in templates]
# synthetic token ids (the BPE merge table is not shipped with this repository): SOT, 5 - 12 word pieces, EOT = highest id
g = torch.Generator().manual_seed(1)
V = BACKBONES["ViT-B/16"]["vocab_size"]
toks = torch.zeros(len(texts), 77, dtype=torch.long)
for i in range(len(texts)):
    n = int(torch.randint(5, 13, (1,), generator=g))
    toks[i, 0] = V - 2
    toks[i, 1:1 + n] = torch.randint(1, V - 2, (n,), generator=g)
    toks[i, 1 + n] = V - 1
toks = toks.cuda()
N, T = len(classes), len(templates)
with torch.no_grad():
    model.encode_text(toks[:64]); torch.cuda.synchronize()
    t0 = time.perf_counter(); emb = model.encode_text(toks); torch.cuda.synchronize(); t_enc = time.perf_counter() - t0
    t0 = time.perf_counter(); w = ops.transpose(ops.proto_build(emb, N, T, per_shot_norm=True)); torch.cuda.synchronize(); t_red = time.perf_counter() - t0
print(f"{len(texts)} prompts: encode_text {t_enc * 1e3:.1f} ms = {len(texts) / t_enc:.0f} prompts/s "
      f"({len(texts) * 5.96e9 / t_enc / 1e12:.0f} TFLOP/s-equivalent), per-class normalise / mean / normalise + transpose {t_red * 1e6:.0f} us")
