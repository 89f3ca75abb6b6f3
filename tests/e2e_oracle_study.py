"""Study (CPU; `python tests/e2e_oracle_study.py`): how far is the ORACLE — the per-operation restatement of the reference with its
fp16 rounding points, pinned to the reference's fp32 towers at 5e-6 — from the reference's own fp16 chain on the seven image -> logits
fixtures?  The reference ran its fp16 towers on the CPU (torch's CPU half kernels: their own accumulation order and intermediate
rounding); any restatement, the HIP path included, lands one draw of fp16 noise away from them.  Prints, per fixture, max|p - p_ref16|
and max|p - p_ref32| of the oracle's fp16 and fp32 chains next to the fixture's own fp16 <-> fp32 gap -> profiles/r03_e2e_oracle_study.json;
tests/test_oracle_golden.py::test_e2e_chain_oracle asserts the fp32 side."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from golden.spec import E2E_VARIANTS, e2e_images, e2e_state_dict, e2e_variant_images  # noqa: E402
from oracle import clip_oracle as co, proto_oracle as po  # noqa: E402


def oracle_chain(name, half):
    """utils.py:284-332 (one distinct augment epoch: the mean over identical epochs is the identity), 335-361, 256-273, then
    main.py:399-409 and utils.P — on the oracle's towers."""
    from proto_clip_amd.clip import clip as pclip
    g = np.load(os.path.join(HERE, "golden", name + ".npz"), allow_pickle=False)
    c = E2E_VARIANTS[name]["case"]
    N, K = c["N"], c["K"]
    sd = e2e_state_dict(name)
    (sup_x, sup_y), _, (test_x, _) = e2e_variant_images(name)
    classnames, templates = [str(x) for x in g["classnames"]], [str(x) for x in g["templates"]]
    tok = pclip.tokenize([t.format(cn.replace("_", " ")) for cn in classnames for t in templates])
    ad_sd = {str(k): torch.from_numpy(g["adapter__" + str(k)]) for k in g["adapter_keys"]}
    enc = co.encode_image_resnet if E2E_VARIANTS[name].get("arch") in ("rn", "rn50") else co.encode_image
    order = torch.from_numpy(np.argsort(np.asarray(sup_y), kind="stable"))
    T = len(templates)
    if half:        # fp16 towers: every normalisation / mean is fp16 arithmetic on fp16 tensors (the reference's GPU precision)
        keys = po.l2norm_rows(enc(sd, sup_x, half=True).half())[order]
        tf = po.l2norm_rows(enc(sd, test_x, half=True).half())
        txt = co.encode_text(sd, tok, half=True).half()
    else:           # fp32 towers (the reference CPU path): normalised in fp32, cast to fp16 where the reference casts (`.half()` of the banks)
        n32 = lambda x: x / x.norm(dim=-1, keepdim=True)
        keys = n32(n32(enc(sd, sup_x, half=False).float()))[order].half()          # utils.py:305-311: per-epoch normalise, mean, normalise
        tf = n32(enc(sd, test_x, half=False).float()).half()                       # utils.py:349-351
        e = n32(co.encode_text(sd, tok, half=False).float()).view(N, T, -1).mean(dim=1)       # utils.py:266-270
        txt = None
        text_bank = n32(e).half()                                                              # [N, D] = clip_weights.t()
    zi = po.proto_build(keys, N, K)
    zt = po.proto_build(txt, N, T) if half else po.l2norm_rows(text_bank)
    zq = po.l2norm_rows(po.adapter_fc(tf, ad_sd) if c["adapter"] == "fc" else po.adapter_conv(tf, ad_sd, c["adapter"]))
    p = po.P(zq, zi, zt, c["alpha"], c["beta"])
    return g, dict(test_features=tf, proto_img=zi, proto_txt=zt, adapted=zq, p=p)


def rel(a, b):
    a, b = a.float(), b.float()
    return ((a - b).norm(dim=-1) / b.norm(dim=-1)).max().item()


def run(names=None):
    out = {}
    for name in names or E2E_VARIANTS:
        row = {}
        for half in (True, False):
            g, st = oracle_chain(name, half)
            tag = "oracle16" if half else "oracle32"
            p16, p32 = torch.from_numpy(g["p_f16"]), torch.from_numpy(g["p_f32"])
            row[tag + "_vs_ref16"] = (st["p"] - p16).abs().max().item()
            row[tag + "_vs_ref32"] = (st["p"] - p32).abs().max().item()
            ref = "f16" if half else "f32"
            row[tag + "_stages_vs_ref" + ref[1:]] = {k: rel(st[k], torch.from_numpy(g[k + "_" + ref])) for k in ("test_features", "proto_img", "proto_txt", "adapted")}
        row["gap"] = (p16 - p32).abs().max().item()
        row["tol"] = max(2 * row["gap"], 1e-3)
        out[name] = row
        print(name, json.dumps(row), flush=True)
    return out


if __name__ == "__main__":
    res = run()
    # the committed yard-stick of the GPU gate (tests/test_gpu_e2e.py::test_distribution_against_oracle_chain): the ORACLE's distances to the
    # reference's chains, fixture by fixture; tests/test_oracle_golden.py::test_e2e_chain_oracle re-derives a subset on every CPU run
    json.dump(res, open(os.path.join(HERE, "golden", "e2e_oracle_chain.json"), "w"), indent=1, sort_keys=True)
    os.makedirs(os.path.join(os.path.dirname(HERE), "profiles"), exist_ok=True)
    json.dump(res, open(os.path.join(os.path.dirname(HERE), "profiles", "r04_e2e_oracle_study.json"), "w"), indent=1)
