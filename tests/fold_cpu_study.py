"""Study (not a collected test; CPU, runs in the build container): what does folding the LayerNorms into the linears
(clip/model.py LN_FOLD, pclip_gemm_ln_f16) cost against the REFERENCE's own image -> logits chain, fixture by fixture?
The oracle's towers are run three ways on every spec.E2E_VARIANTS fixture and compared with the reference's p (fp16-weight towers
and fp32 towers, tests/golden/<variant>.npz):
  unfolded  — the oracle as it is (the reference's rounding points: h = r16(LN(x)), then the linear);
  folded    — an fp32 emulation of the kernels' folded arithmetic: Wf = r16(gamma * W), colsum of the rounded Wf, one-pass
              statistics var = E[x^2] - mu^2, y = r16(rstd * (x Wf^T - mu colsum) + (beta W^T + b));
  variants  — folded with two-pass statistics / with mean-shifted one-pass statistics, to see which ingredient matters.
    python tests/fold_cpu_study.py            (prints one line per fixture and mode; summary in profiles/r03_fold_cpu_study.json)"""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from golden.spec import E2E, E2E_VARIANTS, e2e_images, e2e_state_dict  # noqa: E402
from oracle import clip_oracle as co, proto_oracle as po  # noqa: E402
from proto_clip_amd.clip import clip as pclip  # noqa: E402

r16 = lambda t: t.half().float()
MODE = {"fold": False, "stats": "onepass"}


def folded_linear(x, gamma, beta, W, b):
    Wf = r16(gamma * W)
    cs = Wf.sum(-1)
    bf = (beta * W).sum(-1) + (b if b is not None else 0.0)
    D = x.shape[-1]
    if MODE["stats"] == "twopass":
        mu = x.mean(-1, keepdim=True)
        var = (x - mu).pow(2).mean(-1, keepdim=True)
    elif MODE["stats"] == "shifted":
        c = x[..., :1]
        s, q = (x - c).sum(-1, keepdim=True), (x - c).pow(2).sum(-1, keepdim=True)
        m = s / D
        mu, var = c + m, (q / D - m * m).clamp_min(0.0)
    else:
        s, q = x.sum(-1, keepdim=True), (x * x).sum(-1, keepdim=True)
        mu = s / D
        var = (q / D - mu * mu).clamp_min(0.0)
    rstd = 1.0 / torch.sqrt(var + 1e-5)
    return r16(rstd * (x @ Wf.t() - mu * cs) + bf)


def blocks(x, sd, prefix, layers, heads, mask):
    B, L, W = x.shape
    dh = W // heads
    for i in range(layers):
        p = f"{prefix}resblocks.{i}."
        g1, b1 = sd[p + "ln_1.weight"].float(), sd[p + "ln_1.bias"].float()
        qkv = folded_linear(x, g1, b1, r16(sd[p + "attn.in_proj_weight"].float()), r16(sd[p + "attn.in_proj_bias"].float()))
        q, k, v = (t.view(B, L, heads, dh).transpose(1, 2) for t in qkv.split(W, dim=-1))
        s = (q @ k.transpose(-1, -2)) * (dh ** -0.5)
        if mask is not None:
            s = s + mask
        a = r16(r16(torch.softmax(s, dim=-1)) @ v).transpose(1, 2).reshape(B, L, W)
        x = r16(x + co._linear(a, sd, p + "attn.out_proj.weight", p + "attn.out_proj.bias", True))
        g2, b2 = sd[p + "ln_2.weight"].float(), sd[p + "ln_2.bias"].float()
        f = folded_linear(x, g2, b2, r16(sd[p + "mlp.c_fc.weight"].float()), r16(sd[p + "mlp.c_fc.bias"].float()))
        f = r16(f * r16(torch.sigmoid(r16(1.702 * f))))
        x = r16(x + co._linear(f, sd, p + "mlp.c_proj.weight", p + "mlp.c_proj.bias", True))
    return x


def chain(sd, sup_x, sup_y, test_x, tok, ad_sd, c, half, fold, n_templates):
    N, K = c["N"], c["K"]
    real = co._blocks
    if fold:
        co._blocks = lambda x, sd_, prefix, layers, heads, mask, half_: blocks(x, sd_, prefix, layers, heads, mask)
    try:
        order = torch.from_numpy(np.argsort(np.asarray(sup_y), kind="stable"))
        # build_cache_model with augment_epoch passes over the same images: the mean over identical epochs is the identity in fp16
        keys = po.l2norm_rows(co.encode_image(sd, sup_x, half=half).half())[order]
        zi = po.proto_build(keys, N, K)
        zt = po.proto_build(co.encode_text(sd, tok, half=half).half(), N, n_templates)
        tf = po.l2norm_rows(co.encode_image(sd, test_x, half=half).half())
    finally:
        co._blocks = real
    zq = po.l2norm_rows(po.adapter_conv(tf, ad_sd, c["adapter"]))
    return po.P(zq, zi, zt, c["alpha"], c["beta"]), tf


def main():
    out = {}
    for name, var in E2E_VARIANTS.items():
        path = os.path.join(HERE, "golden", name + ".npz")
        if not os.path.exists(path):
            continue
        g = np.load(path)
        c = var["case"]
        sd = e2e_state_dict(name)
        (sup_x, sup_y), _, (test_x, _) = e2e_images(c)
        classnames, templates = [str(x) for x in g["classnames"]], [str(x) for x in g["templates"]]
        tok = pclip.tokenize([t.format(cn.replace("_", " ")) for cn in classnames for t in templates])
        ad_sd = {str(k): torch.from_numpy(g["adapter__" + str(k)]) for k in g["adapter_keys"]}
        p16, p32 = torch.from_numpy(g["p_f16"]), torch.from_numpy(g["p_f32"])
        f16 = torch.from_numpy(g["test_features_f16"]).float()
        gap = (p16 - p32).abs().max().item()
        res = {"reference_gap_f16_vs_f32": gap, "tol": max(2 * gap, 1e-3)}
        for tag, fold, stats in (("unfolded", False, None), ("folded_onepass", True, "onepass"), ("folded_twopass", True, "twopass"),
                                 ("folded_shifted", True, "shifted")):
            MODE["stats"] = stats
            p, tf = chain(sd, sup_x, sup_y, test_x, tok, ad_sd, c, True, fold, len(templates))
            res[tag] = {"vs_ref16": (p - p16).abs().max().item(), "vs_ref32": (p - p32).abs().max().item(),
                        "features_rel_vs_ref16": ((tf.float() - f16).norm(dim=-1) / f16.norm(dim=-1)).max().item()}
        out[name] = res
        print(name, json.dumps(res), flush=True)
    os.makedirs(os.path.join(os.path.dirname(HERE), "profiles"), exist_ok=True)
    json.dump(out, open(os.path.join(os.path.dirname(HERE), "profiles", "r03_fold_cpu_study.json"), "w"), indent=1)


if __name__ == "__main__":
    main()
