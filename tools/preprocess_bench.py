#!/usr/bin/env python3
"""Throughput of the device pre-processing: ImageNet-like 500x375 uint8 images -> [B,3,224,224], images already
resident on the GPU (kernel rate) and including the host -> device copy of the decoded bytes; PIL + numpy on one host core beside it."""
import os, sys, time
import numpy as np, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from proto_clip_amd.preprocess import ClipPreprocess, RandomTrainTransform

B = 256
rng = np.random.RandomState(0)
host = [rng.randint(0, 256, (375, 500, 3), dtype=np.uint8) for _ in range(B)]
dev = [torch.from_numpy(h).cuda() for h in host]
for name, tf in (("clip eval transform -> fp32", ClipPreprocess(224)), ("clip eval transform -> fp16", ClipPreprocess(224, out_dtype=torch.float16)),
                 ("random train transform -> fp32", RandomTrainTransform(224))):
    tf.batch(dev); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        out = tf.batch(dev)
    torch.cuda.synchronize()
    t_dev = (time.perf_counter() - t0) / 10
    t0 = time.perf_counter()
    for _ in range(3):
        out = tf.batch(host)
    torch.cuda.synchronize()
    t_h2d = (time.perf_counter() - t0) / 3
    byts = B * (375 * 500 * 3 + 2 * 375 * 224 * 3 + 3 * 224 * 224 * out.element_size())
    print(f"{name:32s}: resident {B / t_dev:9.0f} img/s ({byts / t_dev / 1e9:6.1f} GB/s algorithmic), from host arrays {B / t_h2d:8.0f} img/s", flush=True)
try:
    from PIL import Image
    from oracle import preprocess_oracle as po
    t0 = time.perf_counter()
    for h in host[:32]:
        im = Image.fromarray(h)
        oh, ow = po.resize_output_size(375, 500, 224)
        r = np.asarray(im.resize((ow, oh), Image.BICUBIC))
        top, left = po.center_crop_offsets(oh, ow, 224)
        po.to_tensor_normalize(r[top:top + 224, left:left + 224])
    print(f"PIL + numpy on one host core             : {32 / (time.perf_counter() - t0):9.0f} img/s")
except Exception as e:
    print("PIL baseline skipped:", e)
