"""ORACLE — test infrastructure only (see proto_oracle.py header).  CPU restatement of the CLIP
transformer towers (reference clip/model.py:155-238, 338-354) as explicit fp32 matrix arithmetic on an
OpenAI-format state dict.  `half=True` applies the fp16 rounding points of the reference's GPU
precision (convert_weights, clip/model.py:373-394: fp16 Linear/conv/projection weights and fp16
activations, fp32 LayerNorm); `half=False` is the fp32 model `clip.load(device="cpu")` yields.

Parity pin: tests/golden/encoder_*.npz hold encode_image / encode_text outputs of the reference's own
modules (fp16-weight and fp32 variants) on the same seeded weights and inputs; tests compare both.
Attention internals of the reference (torch's fused SDPA) round differently from any restatement, so
this oracle is pinned to those fixtures by tolerance, not bit-exactly."""
import torch
import torch.nn.functional as F


def _r(x, half):
    return x.half().float() if half else x


def _w(sd, key, half, force=False):
    """Parameter as fp32 values.  Tensors that convert_weights halves (Linear / conv / attention /
    projection weights and biases) are fp16-rounded in BOTH modes: build_model converts the model to fp16
    before load_state_dict (clip/model.py:432-433), so even the fp32 CPU model (`.float()` afterwards,
    clip/clip.py:137-138) carries fp16-representable weights — as OpenAI's fp16 checkpoints do anyway."""
    t = sd[key].float()
    return t.half().float() if force else t


def layer_norm(x, w, b, eps=1e-5):
    """clip/model.py:155-161: computed in fp32 whatever the activation dtype."""
    m = x.mean(-1, keepdim=True)
    v = (x - m).pow(2).mean(-1, keepdim=True)
    return (x - m) / torch.sqrt(v + eps) * w + b


def _linear(x, sd, wkey, bkey, half):
    y = x @ _w(sd, wkey, half, True).t()
    if bkey is not None:
        y = y + _w(sd, bkey, half, True)
    return _r(y, half)


def _blocks(x, sd, prefix, layers, heads, mask, half):
    """x [B, L, W]; ResidualAttentionBlock.forward (clip/model.py:187-190) for every layer."""
    B, L, W = x.shape
    dh = W // heads
    for i in range(layers):
        p = f"{prefix}resblocks.{i}."
        h = _r(layer_norm(x, sd[p + "ln_1.weight"].float(), sd[p + "ln_1.bias"].float()), half)
        qkv = _linear(h, sd, p + "attn.in_proj_weight", p + "attn.in_proj_bias", half)
        q, k, v = (t.view(B, L, heads, dh).transpose(1, 2) for t in qkv.split(W, dim=-1))
        s = (q @ k.transpose(-1, -2)) * (dh ** -0.5)
        if mask is not None:
            s = s + mask
        a = _r(_r(torch.softmax(s, dim=-1), half) @ v, half)
        a = a.transpose(1, 2).reshape(B, L, W)
        x = _r(x + _linear(a, sd, p + "attn.out_proj.weight", p + "attn.out_proj.bias", half), half)
        h = _r(layer_norm(x, sd[p + "ln_2.weight"].float(), sd[p + "ln_2.bias"].float()), half)
        f = _linear(h, sd, p + "mlp.c_fc.weight", p + "mlp.c_fc.bias", half)
        f = _r(f * _r(torch.sigmoid(_r(1.702 * f, half)), half), half)          # QuickGELU, clip/model.py:164-166
        x = _r(x + _linear(f, sd, p + "mlp.c_proj.weight", p + "mlp.c_proj.bias", half), half)
    return x


def encode_image(sd, images, half=True):
    """VisionTransformer.forward (clip/model.py:221-238) behind CLIP.encode_image (338-339)."""
    w = _w(sd, "visual.conv1.weight", half, True)
    W, _, P, _ = w.shape
    B = images.shape[0]
    x = _r(images.float(), half)
    cols = F.unfold(x, kernel_size=P, stride=P).transpose(1, 2)               # [B, G*G, 3*P*P]
    x = _r(cols @ w.reshape(W, -1).t(), half)                                  # conv1, 222-224
    cls = _r(sd["visual.class_embedding"].float(), half).expand(B, 1, W)
    x = torch.cat([cls, x], dim=1)                                             # 225
    x = _r(x + _r(sd["visual.positional_embedding"].float(), half), half)      # 226
    x = _r(layer_norm(x, sd["visual.ln_pre.weight"].float(), sd["visual.ln_pre.bias"].float()), half)
    layers = len([k for k in sd if k.startswith("visual.") and k.endswith(".attn.in_proj_weight")])
    x = _blocks(x, sd, "visual.transformer.", layers, W // 64, None, half)
    x = _r(layer_norm(x[:, 0, :], sd["visual.ln_post.weight"].float(), sd["visual.ln_post.bias"].float()), half)
    out = _r(x @ _w(sd, "visual.proj", half, True), half)                      # 235-236
    return out.half() if half else out


def encode_text(sd, text, half=True):
    """CLIP.encode_text (clip/model.py:341-354)."""
    Wt = sd["ln_final.weight"].shape[0]
    L = text.shape[1]
    x = _r(sd["token_embedding.weight"].float()[text], half)
    x = _r(x + _r(sd["positional_embedding"].float(), half), half)
    mask = torch.full((L, L), float("-inf")).triu_(1)                           # 326-332
    layers = len(set(k.split(".")[2] for k in sd if k.startswith("transformer.resblocks")))
    x = _blocks(x, sd, "transformer.", layers, Wt // 64, mask, half)
    x = _r(layer_norm(x, sd["ln_final.weight"].float(), sd["ln_final.bias"].float()), half)
    eot = x[torch.arange(x.shape[0]), text.argmax(dim=-1)]
    out = _r(eot @ _w(sd, "text_projection", half, True), half)
    return out.half() if half else out


# ---- ModifiedResNet tower (clip/model.py:10-152) ----------------------------------------------------------

def _bn_eval(x, sd, p, half):
    """nn.BatchNorm2d in eval mode with fp32 statistics on a (possibly fp16-rounded) activation."""
    w, b = sd[p + ".weight"].float(), sd[p + ".bias"].float()
    m, v = sd[p + ".running_mean"].float(), sd[p + ".running_var"].float()
    y = (x - m[None, :, None, None]) / torch.sqrt(v[None, :, None, None] + 1e-5) * w[None, :, None, None] + b[None, :, None, None]
    return _r(y, half)


def _conv(x, sd, key, half, stride=1, padding=0):
    return _r(F.conv2d(x, _w(sd, key + ".weight", half, True), stride=stride, padding=padding), half)


def _avgpool(x, k, half):
    return _r(F.avg_pool2d(x, k), half) if k > 1 else x


def _bottleneck(x, sd, p, stride, half):
    """Bottleneck.forward (clip/model.py:40-53)."""
    out = torch.relu(_bn_eval(_conv(x, sd, p + "conv1", half), sd, p + "bn1", half))
    out = torch.relu(_bn_eval(_conv(out, sd, p + "conv2", half, padding=1), sd, p + "bn2", half))
    out = _avgpool(out, stride, half)
    out = _bn_eval(_conv(out, sd, p + "conv3", half), sd, p + "bn3", half)
    identity = x
    if p + "downsample.0.weight" in sd:
        identity = _bn_eval(_conv(_avgpool(x, stride, half), sd, p + "downsample.0", half), sd, p + "downsample.1", half)
    return torch.relu(_r(out + identity, half))


def encode_image_resnet(sd, images, half=True):
    """ModifiedResNet.forward + AttentionPool2d.forward (clip/model.py:137-152, 67-92)."""
    x = _r(images.float(), half)
    x = torch.relu(_bn_eval(_conv(x, sd, "visual.conv1", half, stride=2, padding=1), sd, "visual.bn1", half))
    x = torch.relu(_bn_eval(_conv(x, sd, "visual.conv2", half, padding=1), sd, "visual.bn2", half))
    x = torch.relu(_bn_eval(_conv(x, sd, "visual.conv3", half, padding=1), sd, "visual.bn3", half))
    x = _avgpool(x, 2, half)
    for li, stride in ((1, 1), (2, 2), (3, 2), (4, 2)):
        nblk = len(set(k.split(".")[2] for k in sd if k.startswith(f"visual.layer{li}.")))
        for bi in range(nblk):
            x = _bottleneck(x, sd, f"visual.layer{li}.{bi}.", stride if bi == 0 else 1, half)
    B, C, H, W = x.shape
    t = x.reshape(B, C, H * W).permute(0, 2, 1)                                    # [B, HW, C]
    t = torch.cat([_r(t.mean(dim=1, keepdim=True), half), t], dim=1)
    t = _r(t + _r(sd["visual.attnpool.positional_embedding"].float(), half), half)
    heads = C // 64
    q = _linear(t, sd, "visual.attnpool.q_proj.weight", "visual.attnpool.q_proj.bias", half)
    k = _linear(t, sd, "visual.attnpool.k_proj.weight", "visual.attnpool.k_proj.bias", half)
    v = _linear(t, sd, "visual.attnpool.v_proj.weight", "visual.attnpool.v_proj.bias", half)
    L = t.shape[1]
    q, k, v = (z.view(B, L, heads, 64).transpose(1, 2) for z in (q, k, v))
    a = _r(_r(torch.softmax(q @ k.transpose(-1, -2) * 0.125, dim=-1), half) @ v, half)
    a = a.transpose(1, 2).reshape(B, L, C)[:, 0, :]
    out = _linear(a, sd, "visual.attnpool.c_proj.weight", "visual.attnpool.c_proj.bias", half)
    return out.half() if half else out
