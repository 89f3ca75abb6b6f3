"""CLIP towers on the gfx950 kernels (reference clip/model.py:155-434).

`build_model(state_dict)` accepts an OpenAI-format CLIP state dict (same key names the reference's
`build_model` consumes, clip/model.py:397-434) and returns an object with the reference's surface:
`encode_image`, `encode_text`, `dtype`, `eval()`, `state_dict()`, `visual.input_resolution`.
Parameters live in an nn.Module tree with the reference's names, so `load_state_dict` of real OpenAI
weights works unchanged; the forward passes are sequences of libpclip launches (MFMA linears with
fused bias/QuickGELU/residual epilogues, fp32-statistics LayerNorm, whole-sequence attention).

Precision follows `convert_weights` (clip/model.py:373-394): Linear/conv/projection weights fp16,
LayerNorm and embedding parameters fp32, activations fp16 with fp32 accumulation.
Only the transformer towers (ViT-B/32, ViT-B/16, ViT-L/14 and every text tower) are built; the
ModifiedResNet tower (RN50/RN101, clip/model.py:95-152) is not — BASELINE pins its only config (C1) to
the CPU path."""
import os
from collections import OrderedDict

import numpy as np
import torch
from torch import nn

from .. import ops
from .._lib import PclipError


class _Box(nn.Module):
    """Parameter container; attribute names mirror the reference's module tree."""


def _linear(out_f, in_f, bias=True):
    b = _Box()
    b.weight = nn.Parameter(torch.empty(out_f, in_f), requires_grad=False)
    if bias:
        b.bias = nn.Parameter(torch.empty(out_f), requires_grad=False)
    return b


def _ln(width):
    b = _Box()
    b.weight = nn.Parameter(torch.ones(width), requires_grad=False)
    b.bias = nn.Parameter(torch.zeros(width), requires_grad=False)
    return b


def _resblock(width):
    blk = _Box()
    blk.attn = _Box()
    blk.attn.in_proj_weight = nn.Parameter(torch.empty(3 * width, width), requires_grad=False)
    blk.attn.in_proj_bias = nn.Parameter(torch.zeros(3 * width), requires_grad=False)
    blk.attn.out_proj = _linear(width, width)
    blk.ln_1 = _ln(width)
    blk.mlp = _Box()
    blk.mlp.c_fc = _linear(4 * width, width)
    blk.mlp.c_proj = _linear(width, 4 * width)
    blk.ln_2 = _ln(width)
    return blk


def _transformer(width, layers):
    t = _Box()
    t.width, t.layers = width, layers
    t.resblocks = nn.Sequential(*[_resblock(width) for _ in range(layers)])
    return t


def _run_blocks(x, blocks, B, L, heads, causal):
    """ResidualAttentionBlock.forward (clip/model.py:187-190) per layer on x [B*L, W] fp16.
    Each residual add is fused into the LayerNorm that reads its result, so the stream of a block is
        h = LN1(x [+ d])   qkv = in_proj(h)   a = attention(qkv)   d = out_proj(a)
        h = LN2(x += d)    f = QuickGELU(c_fc(h))                  d = c_proj(f)
    Returns (x, d): the block stack's output is x + d, left for the caller's final LayerNorm to fuse."""
    d = None
    for blk in blocks:
        if d is None:
            h = ops.layernorm(x, blk.ln_1.weight, blk.ln_1.bias)
        else:
            h = ops.add_layernorm(x, d, blk.ln_1.weight, blk.ln_1.bias)
        qkv = ops.gemm(h, blk.attn.in_proj_weight, blk.attn.in_proj_bias)
        a = ops.attention(qkv, B, L, heads, causal=causal)
        d = ops.gemm(a, blk.attn.out_proj.weight, blk.attn.out_proj.bias)
        h = ops.add_layernorm(x, d, blk.ln_2.weight, blk.ln_2.bias)
        f = ops.gemm(h, blk.mlp.c_fc.weight, blk.mlp.c_fc.bias, act=1)
        d = ops.gemm(f, blk.mlp.c_proj.weight, blk.mlp.c_proj.bias)
    return x, d


class _Cached:
    """Derived (cast / transposed / padded) copies of parameters, refreshed when the source changes."""

    def __init__(self):
        self._c = {}

    def get(self, key, src, fn):
        tag = (src.data_ptr(), src._version, src.dtype, src.device)
        hit = self._c.get(key)
        if hit is None or hit[0] != tag:
            hit = (tag, fn(src.detach()))
            self._c[key] = hit
        return hit[1]


class VisionTransformer(nn.Module):
    """clip/model.py:204-238."""

    def __init__(self, input_resolution, patch_size, width, layers, heads, output_dim):
        super().__init__()
        self.input_resolution, self.patch_size = input_resolution, patch_size
        self.width, self.heads, self.output_dim = width, heads, output_dim
        self.conv1 = _Box()
        self.conv1.weight = nn.Parameter(torch.empty(width, 3, patch_size, patch_size), requires_grad=False)
        self.class_embedding = nn.Parameter(torch.empty(width), requires_grad=False)
        self.positional_embedding = nn.Parameter(torch.empty((input_resolution // patch_size) ** 2 + 1, width),
                                                 requires_grad=False)
        self.ln_pre = _ln(width)
        self.transformer = _transformer(width, layers)
        self.ln_post = _ln(width)
        self.proj = nn.Parameter(torch.empty(width, output_dim), requires_grad=False)
        self._cache = _Cached()
        # images per pass: bounds activation memory (c_fc output = chunk*L*4W*2 B) and keeps tile counts
        # a large multiple of the 512 resident GEMM workgroups (tail quantisation); tuned on MI355X
        self.chunk = int(os.environ.get("PCLIP_VIT_CHUNK", "256"))

    def forward(self, x: torch.Tensor):
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != self.input_resolution or x.shape[3] != self.input_resolution:
            raise PclipError(f"expected images [B,3,{self.input_resolution},{self.input_resolution}], got {tuple(x.shape)}")
        outs = [self._forward_chunk(x[i:i + self.chunk]) for i in range(0, x.shape[0], self.chunk)]
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)

    def _forward_chunk(self, img):
        B, P, W = img.shape[0], self.patch_size, self.width
        G = self.input_resolution // P
        L = G * G + 1
        if img.dtype == torch.float32:
            img = ops.cast_f16(img)                      # image.type(self.dtype), clip/model.py:339
        elif img.dtype != torch.float16:
            raise PclipError(f"unsupported image dtype {img.dtype}")
        img = img.contiguous()
        kp = 3 * P * P
        kpad = (kp + 63) // 64 * 64

        def pad_w(w):
            w2 = w.reshape(W, kp)
            if kpad != kp:
                w2 = torch.cat([w2, w2.new_zeros(W, kpad - kp)], dim=1)
            return w2.contiguous()

        wconv = self._cache.get("conv1", self.conv1.weight, pad_w)
        cls16 = self._cache.get("cls", self.class_embedding, lambda t: t.half())
        pos16 = self._cache.get("pos", self.positional_embedding, lambda t: t.half().contiguous())
        projT = self._cache.get("projT", self.proj, lambda t: t.t().contiguous())
        cols = ops.im2col_patches(img, P)                                   # conv1 as GEMM (clip/model.py:222)
        patch = ops.gemm(cols, wconv)
        x = ops.vit_assemble_tokens(patch, cls16, pos16, B, G * G, W)       # clip/model.py:225-226
        x = ops.layernorm(x, self.ln_pre.weight, self.ln_pre.bias)          # 227
        x, d = _run_blocks(x, self.transformer.resblocks, B, L, self.heads, causal=False)   # 229-231
        if d is None:
            cls = ops.layernorm(x, self.ln_post.weight, self.ln_post.bias, rows=B, ld=L * W)
        else:                                                               # ln_post((x + d)[:, 0, :]), 233
            cls = ops.add_layernorm(x, d, self.ln_post.weight, self.ln_post.bias, update_x=False, rows=B, ld=L * W)
        return ops.gemm(cls, projT)                                         # x @ proj, 235-236


class CLIP(nn.Module):
    """clip/model.py:241-370 (transformer towers only)."""

    def __init__(self, embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size, context_length,
                 vocab_size, transformer_width, transformer_heads, transformer_layers):
        super().__init__()
        if isinstance(vision_layers, (tuple, list)):
            raise PclipError("ModifiedResNet towers (RN50/RN101) are not built in this round; use a ViT backbone")
        self.context_length = context_length
        self.visual = VisionTransformer(image_resolution, vision_patch_size, vision_width, vision_layers,
                                        vision_width // 64, embed_dim)
        self.transformer = _transformer(transformer_width, transformer_layers)
        self.transformer_heads = transformer_heads
        self.vocab_size = vocab_size
        self.token_embedding = _Box()
        self.token_embedding.weight = nn.Parameter(torch.empty(vocab_size, transformer_width), requires_grad=False)
        self.positional_embedding = nn.Parameter(torch.empty(context_length, transformer_width), requires_grad=False)
        self.ln_final = _ln(transformer_width)
        self.text_projection = nn.Parameter(torch.empty(transformer_width, embed_dim), requires_grad=False)
        self.logit_scale = nn.Parameter(torch.ones([]) * np.log(1 / 0.07), requires_grad=False)
        self._cache = _Cached()
        self.text_chunk = 1024

    @property
    def dtype(self):
        return self.visual.conv1.weight.dtype

    def encode_image(self, image):
        with torch.no_grad():
            return self.visual(image)

    def encode_text(self, text):
        """text [n, context_length] int64 -> [n, embed_dim] fp16 (clip/model.py:341-354).  Unlike the
        reference's per-class calls (utils.py:264-266, n = #templates) any n is efficient here."""
        with torch.no_grad():
            outs = [self._encode_text_chunk(text[i:i + self.text_chunk]) for i in range(0, text.shape[0], self.text_chunk)]
            return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)

    def _encode_text_chunk(self, text):
        B, L = text.shape
        W = self.transformer.width
        emb16 = self._cache.get("tok", self.token_embedding.weight, lambda t: t.half().contiguous())
        pos16 = self._cache.get("pos", self.positional_embedding, lambda t: t.half().contiguous())
        projT = self._cache.get("tprojT", self.text_projection, lambda t: t.t().contiguous())
        x = ops.text_embed(text, emb16, pos16)                                   # 342-344
        x, d = _run_blocks(x, self.transformer.resblocks, B, L, self.transformer_heads, causal=True)   # 345-347
        if d is None:
            x = ops.layernorm(x, self.ln_final.weight, self.ln_final.bias)       # 348
        else:
            x = ops.add_layernorm(x, d, self.ln_final.weight, self.ln_final.bias, update_x=False)
        eot = ops.gather_eot(x, text, B, L, W)                                   # x[arange, text.argmax(-1)]
        return ops.gemm(eot, projT)                                              # @ text_projection, 352

    def forward(self, image, text):
        raise NotImplementedError("contrastive forward (clip/model.py:356-370) is not on the Proto-CLIP hot path")


def convert_weights(model: nn.Module):
    """fp16 for Linear/conv/attention/projection parameters, fp32 elsewhere (clip/model.py:373-394)."""
    half_suffixes = ("in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias", "c_fc.weight", "c_fc.bias",
                     "c_proj.weight", "c_proj.bias", "conv1.weight", "visual.proj", "text_projection")
    for name, p in model.named_parameters():
        p.data = p.data.half() if name.endswith(half_suffixes) else p.data.float()


def build_model(state_dict: dict):
    """Same shape inference as the reference's build_model (clip/model.py:397-434)."""
    if "visual.proj" not in state_dict:
        raise PclipError("ModifiedResNet checkpoints (RN50/RN101) are not supported by this build")
    vision_width = state_dict["visual.conv1.weight"].shape[0]
    vision_layers = len([k for k in state_dict if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    vision_patch_size = state_dict["visual.conv1.weight"].shape[-1]
    grid_size = round((state_dict["visual.positional_embedding"].shape[0] - 1) ** 0.5)
    image_resolution = vision_patch_size * grid_size
    embed_dim = state_dict["text_projection"].shape[1]
    context_length = state_dict["positional_embedding"].shape[0]
    vocab_size = state_dict["token_embedding.weight"].shape[0]
    transformer_width = state_dict["ln_final.weight"].shape[0]
    transformer_heads = transformer_width // 64
    transformer_layers = len(set(k.split(".")[2] for k in state_dict if k.startswith("transformer.resblocks")))
    model = CLIP(embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size, context_length,
                 vocab_size, transformer_width, transformer_heads, transformer_layers)
    sd = OrderedDict((k, v) for k, v in state_dict.items() if k not in ("input_resolution", "context_length", "vocab_size"))
    convert_weights(model)
    model.load_state_dict(sd)
    convert_weights(model)
    return model.eval()


# Backbone hyper-parameters OpenAI's checkpoints resolve to.
BACKBONES = {
    "ViT-B/32": dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=32,
                     context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8, transformer_layers=12),
    "ViT-B/16": dict(embed_dim=512, image_resolution=224, vision_layers=12, vision_width=768, vision_patch_size=16,
                     context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8, transformer_layers=12),
    "ViT-L/14": dict(embed_dim=768, image_resolution=224, vision_layers=24, vision_width=1024, vision_patch_size=14,
                     context_length=77, vocab_size=49408, transformer_width=768, transformer_heads=12, transformer_layers=12),
}


def random_state_dict(seed: int = 1, **kw) -> "OrderedDict[str, torch.Tensor]":
    """Seeded random-init weights with the statistics of CLIP.initialize_parameters
    (clip/model.py:297-324) — there are no pretrained checkpoints in the build environment."""
    g = torch.Generator().manual_seed(seed)
    W, Lv, P = kw["vision_width"], kw["vision_layers"], kw["vision_patch_size"]
    Wt, Lt, E = kw["transformer_width"], kw["transformer_layers"], kw["embed_dim"]
    G = kw["image_resolution"] // P
    rn = lambda *shape, std=1.0: torch.randn(*shape, generator=g) * std
    sd = OrderedDict()
    sd["visual.class_embedding"] = rn(W, std=W ** -0.5)
    sd["visual.positional_embedding"] = rn(G * G + 1, W, std=W ** -0.5)
    sd["visual.proj"] = rn(W, E, std=W ** -0.5)
    sd["visual.conv1.weight"] = rn(W, 3, P, P, std=(3 * P * P) ** -0.5)
    for nm in ("ln_pre", "ln_post"):
        sd[f"visual.{nm}.weight"] = 1 + rn(W, std=0.02)
        sd[f"visual.{nm}.bias"] = rn(W, std=0.02)

    def blocks(prefix, width, layers):
        proj_std, attn_std, fc_std = (width ** -0.5) * ((2 * layers) ** -0.5), width ** -0.5, (2 * width) ** -0.5
        for i in range(layers):
            p = f"{prefix}resblocks.{i}."
            sd[p + "attn.in_proj_weight"] = rn(3 * width, width, std=attn_std)
            sd[p + "attn.in_proj_bias"] = rn(3 * width, std=0.02)
            sd[p + "attn.out_proj.weight"] = rn(width, width, std=proj_std)
            sd[p + "attn.out_proj.bias"] = rn(width, std=0.02)
            sd[p + "ln_1.weight"] = 1 + rn(width, std=0.02)
            sd[p + "ln_1.bias"] = rn(width, std=0.02)
            sd[p + "mlp.c_fc.weight"] = rn(4 * width, width, std=fc_std)
            sd[p + "mlp.c_fc.bias"] = rn(4 * width, std=0.02)
            sd[p + "mlp.c_proj.weight"] = rn(width, 4 * width, std=proj_std)
            sd[p + "mlp.c_proj.bias"] = rn(width, std=0.02)
            sd[p + "ln_2.weight"] = 1 + rn(width, std=0.02)
            sd[p + "ln_2.bias"] = rn(width, std=0.02)

    blocks("visual.transformer.", W, Lv)
    blocks("transformer.", Wt, Lt)
    sd["token_embedding.weight"] = rn(kw["vocab_size"], Wt, std=0.02)
    sd["positional_embedding"] = rn(kw["context_length"], Wt, std=0.01)
    sd["ln_final.weight"] = 1 + rn(Wt, std=0.02)
    sd["ln_final.bias"] = rn(Wt, std=0.02)
    sd["text_projection"] = rn(Wt, E, std=Wt ** -0.5)
    sd["logit_scale"] = torch.ones([]) * np.log(1 / 0.07)
    return sd
