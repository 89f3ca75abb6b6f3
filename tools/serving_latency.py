#!/usr/bin/env python3
"""Small-batch request latency of the serving entry (toolkit consumer): eager launches vs hipGraph
replay, random-init weights of the named backbone, synthetic banks (FewSOL-198 shapes for ViT-L/14)."""
import argparse, os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd import synth
from proto_clip_amd.clip.model import BACKBONES, build_model, random_state_dict
from proto_clip_amd.model import Adapter_FC
from proto_clip_amd.serving import ProtoClipClassifier


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--backbone", default="ViT-B/16")
    args = ap.parse_args()
    kw = BACKBONES[args.backbone]
    model = build_model(random_state_dict(seed=1, **kw)).cuda()
    D, N, K = kw["embed_dim"], 198, 16
    split = synth.make_split(N, K, D, 8, 8, seed=1, sigma=3.0)
    ev = (split.visual_memory_keys.t().float() * 1.2).half().contiguous().cuda()
    et = (split.textual_memory_bank.t().float() * 1.4).half().contiguous().cuda()
    adapter = Adapter_FC(D, dtype=torch.half).cuda()
    clf = ProtoClipClassifier(model, ev, et, adapter, shots=K, alpha=0.2, beta=12.0, top_k=5)
    for bs in (1, 4, 8, 32):
        imgs = synth.make_images(bs, kw["image_resolution"], seed=3, n_class=N).cuda()

        def run(n):
            torch.cuda.synchronize()
            t0 = time.perf_counter()
            for _ in range(n):
                tp, ti = clf.classify(imgs)
            torch.cuda.synchronize()
            return (time.perf_counter() - t0) / n * 1e3

        run(3)
        eager = run(20)
        clf.capture(bs)
        run(3)
        graph = run(20)
        print(f"{args.backbone} batch {bs:3d}: eager {eager:7.3f} ms   hipGraph replay {graph:7.3f} ms   ({eager / graph:4.2f}x)", flush=True)


if __name__ == "__main__":
    main()
