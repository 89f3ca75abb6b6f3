"""Study (imported by tests/test_gpu_e2e.py; also `python tests/chain_parity_study.py [n_seeds]` on a GPU box): how far is the GPU's
image -> logits chain from the reference CPU path over several seeded weight sets / image sets of the e2e case
(tests/golden/spec.py::E2E)?  It is compared with the ORACLE's fp32 towers (pinned to the reference's fp32 model at 5e-6, tests/test_oracle_golden.py)
with the features cast to fp16 — the reference CPU path of SURVEY 8d — and with the oracle's fp16 towers.  Prints one line per
seed and the maxima; the summary is kept in profiles/."""
import json
import os
import sys

import numpy as np
import torch

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, os.path.dirname(HERE))
sys.path.insert(0, HERE)
from golden.spec import E2E, E2E_CASE, e2e_images  # noqa: E402
from oracle import clip_oracle as co, proto_oracle as po  # noqa: E402
from proto_clip_amd import ops  # noqa: E402
from proto_clip_amd.clip import clip as pclip, model as M  # noqa: E402
from proto_clip_amd.clip.model import build_model, random_state_dict  # noqa: E402
from proto_clip_amd.model import Adapter  # noqa: E402

CLASSES = ["tench", "goldfish", "great white shark", "kite", "robin", "bullfrog"]
TEMPLATES = ["a bad photo of a {}.", "a photo of many {}.", "a sculpture of a {}."]


def oracle_chain(sd, sup_x, sup_y, test_x, tok, ad_sd, c, half):
    N, K, T = c["N"], c["K"], len(TEMPLATES)
    f16 = lambda t: t.half()
    order = torch.from_numpy(np.argsort(np.asarray(sup_y), kind="stable"))
    keys = po.l2norm_rows(f16(co.encode_image(sd, sup_x, half=half)))[order]          # one augment epoch: mean over epochs = identity
    zi = po.proto_build(keys, N, K)
    zt = po.proto_build(f16(co.encode_text(sd, tok, half=half)), N, T)
    tf = po.l2norm_rows(f16(co.encode_image(sd, test_x, half=half)))
    zq = po.l2norm_rows(po.adapter_conv(tf, ad_sd, c["adapter"]))
    return po.P(zq, zi, zt, c["alpha"], c["beta"])


def gpu_chain(model, sup_x, sup_y, test_x, tok, adapter, c):
    N, K, T = c["N"], c["K"], len(TEMPLATES)
    order = torch.from_numpy(np.argsort(np.asarray(sup_y), kind="stable")).cuda()
    with torch.no_grad():
        keys = ops.l2norm_rows(model.encode_image(sup_x.cuda()))[order].contiguous()
        zi = ops.proto_build(keys, N, K)
        zt = ops.proto_build(model.encode_text(tok.cuda()), N, T, per_shot_norm=True)
        tf = ops.l2norm_rows(model.encode_image(test_x.cuda()))
        zq = adapter(tf, l2norm_out=True)
        p, am, _, _ = ops.classify(zq, zi, zt, c["alpha"], c["beta"], want_p=True, want_argmax=True)
    return p.cpu()


def run(n):
    tok = pclip.tokenize([t.format(cn) for cn in CLASSES for t in TEMPLATES])
    rows = []
    for s in range(n):
        case = dict(E2E_CASE, seed=E2E_CASE["seed"] + 7 * s)
        sd = random_state_dict(seed=17 + s, **E2E)
        (sup_x, sup_y), _, (test_x, test_y) = e2e_images(case)
        torch.manual_seed(9 + s)
        adapter = Adapter(E2E["embed_dim"], case["adapter"], dtype=torch.half)
        ad_sd = {k: v.clone() for k, v in adapter.state_dict().items()}
        adapter = adapter.cuda()
        p32 = oracle_chain(sd, sup_x, sup_y, test_x, tok, ad_sd, case, half=False)
        p16 = oracle_chain(sd, sup_x, sup_y, test_x, tok, ad_sd, case, half=True)
        model = build_model({k: v.clone() for k, v in sd.items()}).cuda()
        res = {"seed": s, "oracle16_vs_32": (p16 - p32).abs().max().item()}
        p = gpu_chain(model, sup_x, sup_y, test_x, tok, adapter, case)
        res["gpu_vs_32"] = (p - p32).abs().max().item()
        res["gpu_vs_16"] = (p - p16).abs().max().item()
        srt = p32.sort(dim=1).values
        decided = (srt[:, -1] - srt[:, -2]) > 2e-3
        res["gpu_top1_flips_decided"] = int((p.max(1)[1][decided] != p32.max(1)[1][decided]).sum())
        rows.append(res)
        print(json.dumps(res), flush=True)
    keys = [k for k in rows[0] if k != "seed"]
    summary = {k: {"max": max(r[k] for r in rows), "mean": sum(r[k] for r in rows) / len(rows)} for k in keys}
    return rows, summary


def main():
    rows, summary = run(int(sys.argv[1]) if len(sys.argv) > 1 else 6)
    print("SUMMARY " + json.dumps(summary))
    os.makedirs("gpurun_out", exist_ok=True)
    json.dump({"rows": rows, "summary": summary}, open("gpurun_out/chain_parity_study.json", "w"), indent=1)


if __name__ == "__main__":
    main()
