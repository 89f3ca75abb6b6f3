"""CLIP towers on the gfx950 kernels (reference clip/model.py:155-434).

`build_model(state_dict)` accepts an OpenAI-format CLIP state dict (same key names the reference's
`build_model` consumes, clip/model.py:397-434) and returns an object with the reference's surface:
`encode_image`, `encode_text`, `dtype`, `eval()`, `state_dict()`, `visual.input_resolution`.
Parameters live in an nn.Module tree with the reference's names, so `load_state_dict` of real OpenAI
weights works unchanged; the forward passes are sequences of libpclip launches (MFMA linears with
fused bias/QuickGELU/residual epilogues, fp32-statistics LayerNorm, whole-sequence attention).

Precision follows `convert_weights` (clip/model.py:373-394): Linear/conv/projection weights fp16,
LayerNorm and embedding parameters fp32, activations fp16 with fp32 accumulation.
Both vision towers are built: VisionTransformer (ViT-B/32, ViT-B/16, ViT-L/14) and ModifiedResNet (RN50 /
RN101: NHWC activations, 1x1 convs as GEMMs and 3x3 convs as implicit GEMMs with the eval BatchNorm (+ReLU) in their
epilogue; im2col + GEMM only for the strided 3-channel first convolution of the stem)."""
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


# (Rounds 2 - 5 carried two opt-in forms of the LayerNorms — folded into the consuming linear (PCLIP_LN_FOLD=1: an independent rounding, +3.5 % in round 3, -0.9 % against
# the four-wave GEMM of round 5) and computed inside the residual GEMM's launch (PCLIP_RES_LN_FUSE=1: same bits, -1.2 %).  Both lost to the plain LayerNorm pass and
# were removed in round 6: profiles/r03_e2e_fold_study.json, r04_ab_res_ln.txt, r06_bench_v0.json.)


def _run_blocks(x, blocks, B, L, heads, causal, select=None, first_token=False, h0=None):
    """ResidualAttentionBlock.forward (clip/model.py:187-190) per layer on x [B*L, W] fp16 (updated IN PLACE).
    Both residual adds ride in the epilogue of the GEMM that produces the addend (pclip_gemm_f16 with `residual`: the residual
    rows are read in the coalesced store pass, r16(x + r16(acc + bias)) — the reference's two roundings); each LayerNorm is a
    read-x / write-h pass (the reference's rounding point h = r16(LN(x))) followed by the ordinary linear:
        h = ln_1(x)   qkv = in_proj(h)   a = attention(qkv)   x += out_proj(a)
        h = ln_2(x)   f = QuickGELU(c_fc(h))                  x += c_proj(f)
    In low-latency mode (ops.low_latency: split-K linears of a serving request) the addend is produced by the split-K kernel
    and the add stays in the LayerNorm pass (pclip_add_layernorm_f16).
    Returns (x, d): the stack's output is x (+ d when d is not None — low-latency mode leaves the last add to the caller's
    final LayerNorm), [B, W] rows picked by `select` when given.

    `select(t)` picks, from a [B*L, W] tensor, the B rows the caller reads after the stack (the class token of the vision
    tower, clip/model.py:233; the EOT token of the text tower, :350).  The reference pushes every token through the last
    block and then discards all but that row; here the last block's out_proj, LN2 and MLP run on those B rows only — the
    same arithmetic for the rows that matter (a GEMM row does not depend on the other rows), 9/12 of one layer's linear
    FLOPs saved (6 % of a 12-layer tower).  `first_token` (vision tower: the
    selected row is token 0 of every sequence) additionally projects the last block's QUERIES for those B rows only and runs
    its attention for that one query per image (keys / values still come from every token).
    `h0`: the first block's ln_1 output when the caller's stem produced it."""
    n = len(blocks)
    d = None

    def norm_of(x_, ln):
        return None if ln is None else ops.layernorm(x_, ln.weight, ln.bias)

    def linear(h, w, bias, act=0, rows=None):
        """act(h w^T + bias); `rows` = a row range of (w, bias) (the q / kv thirds of in_proj)."""
        sl = slice(None) if rows is None else rows
        return ops.gemm(h, w[sl], bias[sl], act=act)

    def add_linear(x_, a_, lin, ln):
        """x_ += lin(a_) ; returns (x_, ln(x_))."""
        if ops.splitk_active(a_.shape[0]):
            dd = ops.gemm(a_, lin.weight, lin.bias)
            return x_, ops.add_layernorm(x_, dd, ln.weight, ln.bias)
        ops.gemm(a_, lin.weight, lin.bias, residual=x_, out=x_)
        return x_, norm_of(x_, ln)

    h = None
    if n > 0:
        h = h0 if h0 is not None else norm_of(x, blocks[0].ln_1)
    for i, blk in enumerate(blocks):
        last = i == n - 1
        w, bias = blk.attn.in_proj_weight, blk.attn.in_proj_bias
        if select is not None and last and first_token and not causal:
            W = x.shape[1]
            kv = linear(h, w, bias, rows=slice(W, 3 * W))                           # keys | values of every token
            q = linear(select(h), w, bias, rows=slice(0, W))                        # queries of the class tokens
            a, x = ops.attention_first_queries(q, kv, B, L, 1, heads), select(x)
        else:
            qkv = linear(h, w, bias)
            a = ops.attention(qkv, B, L, heads, causal=causal)
            if select is not None and last:
                a, x = select(a), select(x)              # x holds the residual stream entering this block's out_proj add
        x, h = add_linear(x, a, blk.attn.out_proj, blk.ln_2)
        f = linear(h, blk.mlp.c_fc.weight, blk.mlp.c_fc.bias, act=1)
        if last:
            if ops.splitk_active(f.shape[0]):
                d = ops.gemm(f, blk.mlp.c_proj.weight, blk.mlp.c_proj.bias)      # the caller's final LayerNorm adds it
            else:
                ops.gemm(f, blk.mlp.c_proj.weight, blk.mlp.c_proj.bias, residual=x, out=x)
        else:
            x, h = add_linear(x, f, blk.mlp.c_proj, blocks[i + 1].ln_1)
    if select is not None and n == 0:
        x = select(x)
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
        # images per pass: bounds activation memory (c_fc output = chunk*L*4W*2 B = 1.2 GB for ViT-B/16 at 1024);
        # larger passes waste less of the GEMMs' last round of persistent tiles (measured on MI355X: 1024 > 512 > 256)
        self.chunk = int(os.environ.get("PCLIP_VIT_CHUNK", "1024"))

    def forward(self, x: torch.Tensor):
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != self.input_resolution or x.shape[3] != self.input_resolution:
            raise PclipError(f"expected images [B,3,{self.input_resolution},{self.input_resolution}], got {tuple(x.shape)}")
        outs = [self._forward_chunk(x[i:i + self.chunk]) for i in range(0, x.shape[0], self.chunk)]
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)

    def _forward_chunk(self, img):
        B, P, W = img.shape[0], self.patch_size, self.width
        G = self.input_resolution // P
        L = G * G + 1
        if img.dtype not in (torch.float32, torch.float16):   # fp32: image.type(self.dtype) (clip/model.py:339) rides on the im2col gather
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
        blocks = self.transformer.resblocks
        h0 = None
        if len(blocks) > 0:                                                 # tokens + ln_pre + the first block's ln_1 in one pass
            x, h0 = ops.vit_embed_ln(patch, cls16, pos16, B, G * G, W, self.ln_pre.weight, self.ln_pre.bias, blocks[0].ln_1.weight,
                                     blocks[0].ln_1.bias)                   # clip/model.py:225-227, 188
        else:
            x = ops.vit_assemble_tokens(patch, cls16, pos16, B, G * G, W)   # clip/model.py:225-226
            x = ops.layernorm(x, self.ln_pre.weight, self.ln_pre.bias)      # 227
        pick_cls = lambda t: t.view(B, L, W)[:, 0, :].contiguous()          # x[:, 0, :], 233 (taken before the last block's tail)
        x, d = _run_blocks(x, blocks, B, L, self.heads, causal=False, select=pick_cls, first_token=True, h0=h0)   # 229-231
        if d is None:                                                       # ln_post(x[:, 0, :]), 233
            cls = ops.layernorm(x, self.ln_post.weight, self.ln_post.bias)
        else:
            cls = ops.add_layernorm(x, d, self.ln_post.weight, self.ln_post.bias, update_x=False)
        return ops.gemm(cls, projT)                                         # x @ proj, 235-236


def _bn(ch):
    b = _Box()
    b.weight = nn.Parameter(torch.ones(ch), requires_grad=False)
    b.bias = nn.Parameter(torch.zeros(ch), requires_grad=False)
    b.register_buffer("running_mean", torch.zeros(ch))
    b.register_buffer("running_var", torch.ones(ch))
    b.register_buffer("num_batches_tracked", torch.tensor(0, dtype=torch.long))
    b.eps = 1e-5
    return b


def _conv(cout, cin, k):
    b = _Box()
    b.weight = nn.Parameter(torch.empty(cout, cin, k, k), requires_grad=False)
    return b


class _Bottleneck(_Box):
    """Parameter tree of clip/model.py:10-38 (conv1 1x1, conv2 3x3, avgpool(stride), conv3 1x1, optional
    downsample = avgpool(stride) -> 1x1 conv -> bn, whose Sequential keys are '0' and '1')."""

    def __init__(self, inplanes, planes, stride):
        super().__init__()
        self.conv1, self.bn1 = _conv(planes, inplanes, 1), _bn(planes)
        self.conv2, self.bn2 = _conv(planes, planes, 3), _bn(planes)
        self.conv3, self.bn3 = _conv(planes * 4, planes, 1), _bn(planes * 4)
        self.stride = stride
        self.downsample = None
        if stride > 1 or inplanes != planes * 4:
            self.downsample = nn.ModuleDict({"0": _conv(planes * 4, inplanes, 1), "1": _bn(planes * 4)})


class ModifiedResNet(nn.Module):
    """clip/model.py:95-152 on the gfx950 kernels.  Activations are NHWC fp16 rows [B*H*W, C]: 1x1 convolutions are
    plain MFMA GEMMs, 3x3 convolutions are an im2col gather + the same GEMM, BatchNorm(eval)+ReLU(+residual) is one
    streaming pass, the attention pool reuses the transformer attention kernel (head dim 64)."""

    def __init__(self, layers, output_dim, heads, input_resolution=224, width=64):
        super().__init__()
        self.output_dim, self.input_resolution, self.heads = output_dim, input_resolution, heads
        self.conv1, self.bn1 = _conv(width // 2, 3, 3), _bn(width // 2)
        self.conv2, self.bn2 = _conv(width // 2, width // 2, 3), _bn(width // 2)
        self.conv3, self.bn3 = _conv(width, width // 2, 3), _bn(width)
        inpl = width
        for li, (planes, nblk, stride) in enumerate(zip((width, width * 2, width * 4, width * 8), layers, (1, 2, 2, 2)), 1):
            blocks = [_Bottleneck(inpl, planes, stride)]
            inpl = planes * 4
            blocks += [_Bottleneck(inpl, planes, 1) for _ in range(1, nblk)]
            setattr(self, f"layer{li}", nn.Sequential(*blocks))
        embed = width * 32
        if embed // heads != 64:
            raise PclipError(f"attention pool head dim {embed // heads} unsupported (the attention kernel needs 64)")
        self.attnpool = _Box()
        self.attnpool.positional_embedding = nn.Parameter(torch.empty((input_resolution // 32) ** 2 + 1, embed), requires_grad=False)
        for nm in ("k_proj", "q_proj", "v_proj"):
            setattr(self.attnpool, nm, _linear(embed, embed))
        self.attnpool.c_proj = _linear(output_dim, embed)
        self._cache = _Cached()
        # images per pass: small passes are launch-bound and leave the tail tiles of every convolution idle — 1024 images in passes of 128 / 256 / 512 / 1024:
        # RN50 33.4 / 37.0 / 39.6 / 41.6 k img/s, RN101 22.5 / 25.7 / 27.0 / 28.5 k (profiles/r06_rn_chunk_probe.txt); the widest activation of a 1024-image pass
        # (layer1's 56 x 56 x 256 fp16) is 1.6 GB.  Same bits for every pass size (tests/test_gpu_encoder.py).
        # (Widths whose narrow convolutions go through a materialised im2col matrix — RN50x4 / x16: 40 / 48 stem channels — keep passes of 256: the matrix of a
        # 1024-image pass would be 16 - 34 GB.)
        self.chunk = int(os.environ.get("PCLIP_RN_CHUNK", "1024" if width // 2 in (32, 64) else "256"))

    # -- helpers ---------------------------------------------------------------------------------------------------
    def _bn_affine(self, key, bn):
        def fold(_):
            inv = torch.rsqrt(bn.running_var.float() + bn.eps)
            scale = bn.weight.float() * inv
            return torch.stack([scale, bn.bias.float() - bn.running_mean.float() * scale]).contiguous()
        tag_src = bn.weight            # refreshed when the affine weight tensor is replaced; running stats are frozen in eval
        ss = self._cache.get(("bn", key), tag_src, fold)
        return ss[0], ss[1]

    def _w1x1(self, key, conv):
        return self._cache.get(("w1", key), conv.weight, lambda w: w.reshape(w.shape[0], w.shape[1]).contiguous())

    def _w3x3(self, key, conv):
        def prep(w):                   # [Cout, Cin, 3, 3] -> [Cout, (ky, kx, Cin)] padded to the GEMM's K-tile
            co, ci = w.shape[0], w.shape[1]
            w2 = w.permute(0, 2, 3, 1).reshape(co, 9 * ci)
            kpad = (9 * ci + 63) // 64 * 64
            if kpad != 9 * ci:
                w2 = torch.cat([w2, w2.new_zeros(co, kpad - 9 * ci)], dim=1)
            return w2.contiguous()
        return self._cache.get(("w3", key), conv.weight, prep)

    def _conv3_bn_relu(self, key, x, strides, B, H, W, C, conv, bn, stride=1):
        sc, sh = self._bn_affine(key, bn)
        if stride == 1 and (C % 64 == 0 or C in (8, 16, 32)) and (conv.weight.shape[0] % 64 == 0 or conv.weight.shape[0] == 32) \
                and strides == (H * W * C, W * C, C, 1):
            return ops.conv3x3_bn(x, self._w3x3(key, conv), sc, sh, B, H, W, C, relu=True)   # implicit GEMM: no im2col buffer
        cols = ops.im2col3x3(x, strides, B, H, W, C, stride)
        return ops.gemm_bn(cols, self._w3x3(key, conv), sc, sh, relu=True)                   # relu(bn(conv3x3(x))) in one launch

    def _bottleneck(self, key, blk, x, B, H, W, Cin):
        planes = blk.conv1.weight.shape[0]
        sc, sh = self._bn_affine(key + ".1", blk.bn1)
        out = ops.gemm_bn(x, self._w1x1(key + ".1", blk.conv1), sc, sh, relu=True)           # relu(bn1(conv1(x)))
        out = self._conv3_bn_relu(key + ".2", out, (H * W * planes, W * planes, planes, 1), B, H, W, planes, blk.conv2, blk.bn2)
        Ho, Wo = H, W
        if blk.stride > 1:
            out = ops.avgpool_nhwc(out, B, H, W, planes, blk.stride)                         # anti-aliased stride
            Ho, Wo = H // blk.stride, W // blk.stride
        identity = x
        if blk.downsample is not None:
            idn = ops.avgpool_nhwc(x, B, H, W, Cin, blk.stride) if blk.stride > 1 else x
            dsc, dsh = self._bn_affine(key + ".d", blk.downsample["1"])
            identity = ops.gemm_bn(idn, self._w1x1(key + ".d", blk.downsample["0"]), dsc, dsh, relu=False)
        sc, sh = self._bn_affine(key + ".3", blk.bn3)
        out = ops.gemm_bn_res_relu(out, self._w1x1(key + ".3", blk.conv3), sc, sh, identity)  # relu(bn3(conv3(out)) + identity), one launch
        return out, Ho, Wo, planes * 4

    def forward(self, x):
        R = self.input_resolution
        if x.dim() != 4 or x.shape[1] != 3 or x.shape[2] != R or x.shape[3] != R:
            raise PclipError(f"expected images [B,3,{R},{R}], got {tuple(x.shape)}")
        outs = [self._forward_chunk(x[i:i + self.chunk]) for i in range(0, x.shape[0], self.chunk)]
        return outs[0] if len(outs) == 1 else torch.cat(outs, dim=0)

    def _forward_chunk(self, img):
        B, R = img.shape[0], self.input_resolution
        w2 = self.conv1.weight.shape[0]
        if ops.stem_conv_applies(R, w2) and img.dtype in (torch.float32, torch.float16):
            # stem conv1 + bn1 + relu straight from the NCHW images (fp32 rounded to fp16 on the way): no cast pass, no im2col matrix
            sc, sh = self._bn_affine("s1", self.bn1)
            x = ops.stem_conv_bn(img, self._w3x3("s1", self.conv1), sc, sh, relu=True)
        else:
            if img.dtype == torch.float32:
                img = ops.cast_f16(img)
            img = img.contiguous()
            # stem (clip/model.py:138-142): the NCHW image is read through its strides by the first im2col
            x = self._conv3_bn_relu("s1", img, (3 * R * R, R, 1, R * R), B, R, R, 3, self.conv1, self.bn1, stride=2)
        H = W = (R - 1) // 2 + 1
        x = self._conv3_bn_relu("s2", x, (H * W * w2, W * w2, w2, 1), B, H, W, w2, self.conv2, self.bn2)
        C = self.conv3.weight.shape[0]
        if ops.conv3x3_pool_applies(H, W, w2, C):           # conv3 + bn3 + relu + avgpool(2) in one launch (the same bits)
            sc, sh = self._bn_affine("s3", self.bn3)
            x = ops.conv3x3_bn_pool(x, self._w3x3("s3", self.conv3), sc, sh, B, H, W, w2)
        else:
            x = self._conv3_bn_relu("s3", x, (H * W * w2, W * w2, w2, 1), B, H, W, w2, self.conv3, self.bn3)
            x = ops.avgpool_nhwc(x, B, H, W, C, 2)
        H, W = H // 2, W // 2
        for li in (1, 2, 3, 4):
            for bi, blk in enumerate(getattr(self, f"layer{li}")):
                x, H, W, C = self._bottleneck(f"l{li}.{bi}", blk, x, B, H, W, C)
        # attention pool (clip/model.py:67-92): q = k = v = tokens, separate projections, output of token 0
        ap = self.attnpool
        pos16 = self._cache.get("appos", ap.positional_embedding, lambda t: t.half().contiguous())
        wqkv = self._cache.get("apw", ap.q_proj.weight, lambda _: torch.cat([ap.q_proj.weight, ap.k_proj.weight, ap.v_proj.weight]).detach().contiguous())
        bqkv = self._cache.get("apb", ap.q_proj.bias, lambda _: torch.cat([ap.q_proj.bias, ap.k_proj.bias, ap.v_proj.bias]).detach().contiguous())
        L = H * W + 1
        tok = ops.attnpool_tokens(x, pos16, B, H * W, C)
        # only the pooled (mean) token's output is returned (clip/model.py:91 `x[0]`): project queries for token 0 only and
        # run a one-query attention per (image, head); keys / values still come from all HW + 1 tokens
        kv = ops.gemm(tok, wqkv[C:], bqkv[C:])
        q0 = ops.gemm(tok.view(B, L * C)[:, :C].contiguous(), wqkv[:C], bqkv[:C])
        a0 = ops.attention_first_queries(q0, kv, B, L, 1, self.heads)
        return ops.gemm(a0, ap.c_proj.weight, ap.c_proj.bias)


class CLIP(nn.Module):
    """clip/model.py:241-370 (transformer towers only)."""

    def __init__(self, embed_dim, image_resolution, vision_layers, vision_width, vision_patch_size, context_length,
                 vocab_size, transformer_width, transformer_heads, transformer_layers):
        super().__init__()
        self.context_length = context_length
        if isinstance(vision_layers, (tuple, list)):
            self.visual = ModifiedResNet(vision_layers, embed_dim, vision_width * 32 // 64, image_resolution, vision_width)
        else:
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
        pick_eot = lambda t: ops.gather_eot(t, text, B, L, W)                    # x[arange, text.argmax(-1)], 350
        x, d = _run_blocks(x, self.transformer.resblocks, B, L, self.transformer_heads, causal=True, select=pick_eot)   # 345-347
        if d is None:
            eot = ops.layernorm(x, self.ln_final.weight, self.ln_final.bias)     # 348 (row-wise: commutes with the gather)
        else:
            eot = ops.add_layernorm(x, d, self.ln_final.weight, self.ln_final.bias, update_x=False)
        return ops.gemm(eot, projT)                                              # @ text_projection, 352

    def forward(self, image, text):
        raise NotImplementedError("contrastive forward (clip/model.py:356-370) is not on the Proto-CLIP hot path")


def convert_weights(model: nn.Module):
    """fp16 for Linear/conv/attention/projection parameters, fp32 elsewhere (clip/model.py:373-394)."""
    half_suffixes = ("in_proj_weight", "in_proj_bias", "out_proj.weight", "out_proj.bias", "c_fc.weight", "c_fc.bias",
                     "c_proj.weight", "c_proj.bias", "visual.proj", "text_projection", "q_proj.weight", "q_proj.bias",
                     "k_proj.weight", "k_proj.bias", "v_proj.weight", "v_proj.bias")
    for name, p in model.named_parameters():
        is_conv = p.dim() == 4                               # every nn.Conv2d weight (ViT patch conv, all ResNet convs)
        p.data = p.data.half() if (is_conv or name.endswith(half_suffixes)) else p.data.float()


def build_model(state_dict: dict):
    """Same shape inference as the reference's build_model (clip/model.py:397-434)."""
    if "visual.proj" in state_dict:
        vision_width = state_dict["visual.conv1.weight"].shape[0]
        vision_layers = len([k for k in state_dict if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
        vision_patch_size = state_dict["visual.conv1.weight"].shape[-1]
        grid_size = round((state_dict["visual.positional_embedding"].shape[0] - 1) ** 0.5)
        image_resolution = vision_patch_size * grid_size
    else:
        vision_layers = tuple(len(set(k.split(".")[2] for k in state_dict if k.startswith(f"visual.layer{b}"))) for b in (1, 2, 3, 4))
        vision_width = state_dict["visual.layer1.0.conv1.weight"].shape[0]
        output_width = round((state_dict["visual.attnpool.positional_embedding"].shape[0] - 1) ** 0.5)
        vision_patch_size = None
        assert output_width ** 2 + 1 == state_dict["visual.attnpool.positional_embedding"].shape[0]
        image_resolution = output_width * 32
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
    "RN50": dict(embed_dim=1024, image_resolution=224, vision_layers=(3, 4, 6, 3), vision_width=64, vision_patch_size=None,
                 context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8, transformer_layers=12),
    "RN101": dict(embed_dim=512, image_resolution=224, vision_layers=(3, 4, 23, 3), vision_width=64, vision_patch_size=None,
                  context_length=77, vocab_size=49408, transformer_width=512, transformer_heads=8, transformer_layers=12),
    "ViT-L/14": dict(embed_dim=768, image_resolution=224, vision_layers=24, vision_width=1024, vision_patch_size=14,
                     context_length=77, vocab_size=49408, transformer_width=768, transformer_heads=12, transformer_layers=12),
}


def random_state_dict(seed: int = 1, **kw) -> "OrderedDict[str, torch.Tensor]":
    """Seeded random-init weights with the statistics of CLIP.initialize_parameters
    (clip/model.py:297-324) — there are no pretrained checkpoints in the build environment."""
    g = torch.Generator().manual_seed(seed)
    Wt, Lt, E = kw["transformer_width"], kw["transformer_layers"], kw["embed_dim"]
    rn = lambda *shape, std=1.0: torch.randn(*shape, generator=g) * std
    sd = OrderedDict()
    if isinstance(kw["vision_layers"], (tuple, list)):
        w = kw["vision_width"]

        def conv(name, co, ci, k):
            sd[name + ".weight"] = rn(co, ci, k, k, std=(ci * k * k) ** -0.5)

        def bn(name, ch, gamma=1.0):
            sd[name + ".weight"] = gamma * (1 + rn(ch, std=0.05))
            sd[name + ".bias"] = rn(ch, std=0.05)
            sd[name + ".running_mean"] = rn(ch, std=0.1)
            sd[name + ".running_var"] = 1 + 0.2 * torch.rand(ch, generator=g)
            sd[name + ".num_batches_tracked"] = torch.tensor(0, dtype=torch.long)

        conv("visual.conv1", w // 2, 3, 3); bn("visual.bn1", w // 2)
        conv("visual.conv2", w // 2, w // 2, 3); bn("visual.bn2", w // 2)
        conv("visual.conv3", w, w // 2, 3); bn("visual.bn3", w)
        inpl = w
        for li, (planes, nblk, stride) in enumerate(zip((w, 2 * w, 4 * w, 8 * w), kw["vision_layers"], (1, 2, 2, 2)), 1):
            for bi in range(nblk):
                p = f"visual.layer{li}.{bi}."
                st = stride if bi == 0 else 1
                conv(p + "conv1", planes, inpl, 1); bn(p + "bn1", planes)
                conv(p + "conv2", planes, planes, 3); bn(p + "bn2", planes)
                conv(p + "conv3", planes * 4, planes, 1); bn(p + "bn3", planes * 4, gamma=0.5)
                if st > 1 or inpl != planes * 4:
                    conv(p + "downsample.0", planes * 4, inpl, 1); bn(p + "downsample.1", planes * 4)
                inpl = planes * 4
        emb = w * 32
        sd["visual.attnpool.positional_embedding"] = rn((kw["image_resolution"] // 32) ** 2 + 1, emb, std=emb ** -0.5)
        for nm in ("k_proj", "q_proj", "v_proj"):
            sd[f"visual.attnpool.{nm}.weight"] = rn(emb, emb, std=emb ** -0.5)
            sd[f"visual.attnpool.{nm}.bias"] = rn(emb, std=0.02)
        sd["visual.attnpool.c_proj.weight"] = rn(E, emb, std=emb ** -0.5)
        sd["visual.attnpool.c_proj.bias"] = rn(E, std=0.02)
    else:
        W, Lv, P = kw["vision_width"], kw["vision_layers"], kw["vision_patch_size"]
        G = kw["image_resolution"] // P
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

    if not isinstance(kw["vision_layers"], (tuple, list)):
        blocks("visual.transformer.", kw["vision_width"], kw["vision_layers"])
    blocks("transformer.", Wt, Lt)
    sd["token_embedding.weight"] = rn(kw["vocab_size"], Wt, std=0.02)
    sd["positional_embedding"] = rn(kw["context_length"], Wt, std=0.01)
    sd["ln_final.weight"] = 1 + rn(Wt, std=0.02)
    sd["ln_final.bias"] = rn(Wt, std=0.02)
    sd["text_projection"] = rn(Wt, E, std=Wt ** -0.5)
    sd["logit_scale"] = torch.ones([]) * np.log(1 / 0.07)
    return sd
