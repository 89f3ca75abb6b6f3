"""The image -> logits chain of tests/test_gpu_e2e.py as a function (shared with the study `python tests/e2e_chain.py`, which
prints every fixture's numbers with the LayerNorm fold on and off: profiles/r03_e2e_fold_study.json)."""
import json
import os
import sys
import tempfile

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
if os.path.dirname(HERE) not in sys.path:
    sys.path.insert(0, os.path.dirname(HERE))
if HERE not in sys.path:
    sys.path.insert(0, HERE)
from golden.spec import E2E, E2E_VARIANTS, e2e_arch, e2e_images, e2e_state_dict, e2e_variant_images  # noqa: E402

STAGES = ("test_features", "text_bank", "adapted", "proto_img", "proto_txt")


def rel_err(a, b):
    a, b = a.float().cpu(), b.float().cpu()
    return ((a - b).norm(dim=-1) / b.norm(dim=-1)).max().item()


def run_variant(name, tmp_dir=None):
    """GPU chain of fixture `name` (utils.py:256-361 bank builders on the HIP towers, then main.py:399-409, 433-435) and its
    distances to the reference's chain.  Returns a dict of tensors and numbers."""
    from proto_clip_amd import ops
    from proto_clip_amd.clip.model import build_model
    from proto_clip_amd.model import Adapter, Adapter_FC
    from proto_clip_amd.utils import build_cache_model, clip_classifier, pre_load_features
    g = np.load(os.path.join(HERE, "golden", name + ".npz"), allow_pickle=False)
    c = E2E_VARIANTS[name]["case"]
    N, K, D = c["N"], c["K"], e2e_arch(name)["embed_dim"]
    model = build_model(e2e_state_dict(name)).cuda()
    (sup_x, sup_y), _, (test_x, test_y) = e2e_variant_images(name)
    tmp_dir = tmp_dir or tempfile.mkdtemp(prefix="pclip_e2e_")
    cfg = dict(cache_dir=str(tmp_dir), backbone="ViT-B/16", shots=K, augment_epoch=c["augment_epoch"], dataset="synthetic_" + name)
    classnames, templates = [str(x) for x in g["classnames"]], [str(x) for x in g["templates"]]
    ad_sd = {str(k): torch.from_numpy(g["adapter__" + str(k)]) for k in g["adapter_keys"]}
    with torch.no_grad():
        keys, values = build_cache_model(cfg, model, [(sup_x[:10], sup_y[:10]), (sup_x[10:], sup_y[10:])])      # utils.py:284-332
        test_f, test_l = pre_load_features(cfg, "test", model, [(test_x[:20], test_y[:20]), (test_x[20:], test_y[20:])])   # 335-361
        _, text_bank = clip_classifier(classnames, templates, model)                                          # 256-273, real tokenizer
        adapter = Adapter_FC(D, dtype=torch.half) if c["adapter"] == "fc" else Adapter(D, c["adapter"], dtype=torch.half)
        adapter.load_state_dict(ad_sd)
        adapter = adapter.cuda()
        zi = ops.proto_build(ops.transpose(keys), N, K)                                  # main.py:399-402
        zt = ops.l2norm_rows(ops.transpose(text_bank))                                   # 404-405
        zq = adapter(test_f, l2norm_out=True)                                            # 407-409
        p, am, _, _ = ops.classify(zq, zi, zt, c["alpha"], c["beta"], want_p=True, want_argmax=True)   # utils.py:225-244, main.py:433-435
    p, am = p.cpu(), am.cpu().long()
    p16, p32 = torch.from_numpy(g["p_f16"]), torch.from_numpy(g["p_f32"])
    gap = (p16 - p32).abs().max().item()
    # per-VECTOR relative errors: the textual bank is [D, N] (utils.py:272) — its vectors are the columns (round 2 took the rows: 6
    # numbers per "vector", some of them ~0, which is where its 7e-3 came from)
    vec = lambda k, t: t.t() if k == "text_bank" else t
    stage = {k: rel_err(vec(k, a), vec(k, torch.from_numpy(g[k + "_f16"]))) for k, a in
             (("test_features", test_f), ("text_bank", text_bank), ("adapted", zq), ("proto_img", zi), ("proto_txt", zt))}
    # the reference's own fp16 <-> fp32 disagreement per stage: the yard-stick of the stage bounds
    stage_gap = {k: rel_err(vec(k, torch.from_numpy(g[k + "_f32"])), vec(k, torch.from_numpy(g[k + "_f16"]))) for k in STAGES}
    jitter = (torch.from_numpy(g["p_f16_jitter"]) - p16).abs().max().item() if "p_f16_jitter" in g.files else None
    return dict(g=g, c=c, p=p, am=am, p16=p16, p32=p32, gap=gap, jitter=jitter, d16=(p - p16).abs().max().item(), d32=(p - p32).abs().max().item(),
                stage=stage, stage_gap=stage_gap, test_l=test_l.cpu(), test_y=test_y, sup_y=sup_y, values=values.cpu(), zq=zq, zi=zi, zt=zt)


def main():
    from proto_clip_amd.clip import model as M
    out = {}
    for name in E2E_VARIANTS:
        if not os.path.exists(os.path.join(HERE, "golden", name + ".npz")):
            continue
        row = {}
        r = run_variant(name)
        srt = r["p16"].sort(dim=1).values
        margin = srt[:, -1] - srt[:, -2]
        tol = max(2 * r["gap"], 1e-3)
        decided = margin > 2 * tol
        ref_am = torch.from_numpy(r["g"]["argmax_f16"]).long()
        row["gpu"] = dict(d16=r["d16"], d32=r["d32"], stage=r["stage"], top1_flips_decided=int((r["am"][decided] != ref_am[decided]).sum()))
        row.update(gap=r["gap"], jitter=r["jitter"], tol=tol, stage_gap=r["stage_gap"], decided=int(decided.sum()))
        out[name] = row
        print(name, json.dumps(row), flush=True)
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump(out, open("gpurun_out/e2e_chain_study.json", "w"), indent=1)


if __name__ == "__main__":
    main()
