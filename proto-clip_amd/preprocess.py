"""Image pre-processing on the GPU with the reference's transform semantics.

`ClipPreprocess(n_px)` = clip/clip.py:77-84 `_transform` (Resize(n_px, BICUBIC) -> CenterCrop(n_px) -> ToTensor -> Normalize);
`RandomTrainTransform(size)` = datasets/imagenet.py:8-23 (RandomResizedCrop(size, scale=(0.5, 1), BICUBIC) ->
RandomHorizontalFlip(0.5) -> ToTensor -> Normalize).  Inputs are DECODED images: uint8 HWC RGB arrays / tensors (JPEG decoding
stays on the host — PIL in the reference).  The pixel arithmetic is Pillow's 8-bit bicubic resample restated in
csrc/pclip_preprocess.hip, bit-identical to `Image.resize`; whole batches go through three kernel launches.  The random crop
parameters are drawn from torch's RNG in torchvision's order (restated: torchvision is not in the image)."""
import math

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, stream

CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _ksize(in_size: int, out_size: int) -> int:
    if in_size == out_size:
        return 1
    return int(math.ceil(2.0 * max(float(in_size) / out_size, 1.0))) * 2 + 1


def _as_device_u8(img) -> torch.Tensor:
    if not isinstance(img, torch.Tensor):
        img = torch.from_numpy(np.ascontiguousarray(np.asarray(img)))
    if img.dtype != torch.uint8 or img.dim() != 3 or img.shape[2] != 3:
        raise _lib.PclipError(f"expected a uint8 HWC RGB image, got {tuple(img.shape)} {img.dtype}")
    return img.cuda().contiguous()


def preprocess_batch(images, boxes, resized, windows, flips, n_px: int, mean=CLIP_MEAN, std=CLIP_STD, out_dtype=torch.float32):
    """images: list of uint8 HWC RGB; boxes[i] = (top, left, h, w) resized to resized[i] = (rs_h, rs_w); windows[i] = (top, left) of
    the n_px x n_px window kept; flips[i] bool.  Returns [B, 3, n_px, n_px] on the GPU."""
    imgs = [_as_device_u8(im) for im in images]
    B = len(imgs)
    out = torch.empty(B, 3, n_px, n_px, dtype=out_dtype, device="cuda")
    if B == 0:
        return out
    desc = np.zeros((B, 16), dtype=np.int32)
    coef_off, tmp_bytes = 0, 0
    for i, im in enumerate(imgs):
        (bt, bl, bh, bw), (rh, rw), (wt, wl) = boxes[i], resized[i], windows[i]
        if not (0 <= bt and 0 <= bl and bh > 0 and bw > 0 and bt + bh <= im.shape[0] and bl + bw <= im.shape[1]):
            raise _lib.PclipError(f"image {i}: box {boxes[i]} outside {tuple(im.shape[:2])}")
        if not (0 <= wt and 0 <= wl and wt + n_px <= rh and wl + n_px <= rw):
            raise _lib.PclipError(f"image {i}: window {windows[i]} + {n_px} outside the resized box {resized[i]}")
        ks_h, ks_v = _ksize(bw, rw), _ksize(bh, rh)
        desc[i] = (im.shape[0], im.shape[1], bt, bl, bh, bw, rh, rw, wt, wl, int(bool(flips[i])), coef_off, 0, ks_h, ks_v, 0)
        coef_off += n_px * (2 + ks_h) + n_px * (2 + ks_v)
    tmp_base = (coef_off * 4 + 255) // 256 * 256
    for i in range(B):
        desc[i, 12] = tmp_base + tmp_bytes
        tmp_bytes += (int(desc[i, 4]) * n_px * 3 + 255) // 256 * 256
    ws = torch.empty(tmp_base + tmp_bytes, dtype=torch.uint8, device="cuda")
    srcs = torch.tensor([im.data_ptr() for im in imgs], dtype=torch.int64, device="cuda")
    d = torch.from_numpy(desc).cuda()
    check(_lib.load().pclip_preprocess_u8(ptr(srcs), ptr(d), B, n_px, int(desc[:, 4].max()), *[float(np.float32(v)) for v in mean],
                                          *[float(np.float32(v)) for v in std], ptr(out), int(out_dtype == torch.float16), ptr(ws),
                                          stream()), "pclip_preprocess_u8")
    return out


def resize_output_size(h: int, w: int, size: int):
    """torchvision Resize(int): shorter side -> size, the other int(size * long / short).  Returns (out_h, out_w)."""
    short, long = (w, h) if w <= h else (h, w)
    new_long = int(size * long / short)
    return (new_long, size) if w <= h else (size, new_long)


class ClipPreprocess:
    """`preprocess` of clip.load (clip/clip.py:77-84).  Call with one image -> [3, n, n], or `.batch(list)` -> [B, 3, n, n]."""

    def __init__(self, n_px: int, out_dtype=torch.float32):
        self.n_px, self.out_dtype = n_px, out_dtype

    def batch(self, images):
        n = self.n_px
        boxes, resized, windows = [], [], []
        for im in images:
            h, w = int(im.shape[0]), int(im.shape[1])
            rh, rw = resize_output_size(h, w, n)
            boxes.append((0, 0, h, w))
            resized.append((rh, rw))
            windows.append((int(round((rh - n) / 2.0)), int(round((rw - n) / 2.0))))        # CenterCrop
        return preprocess_batch(images, boxes, resized, windows, [False] * len(images), n, out_dtype=self.out_dtype)

    def __call__(self, image):
        return self.batch([image])[0]


class RandomTrainTransform:
    """datasets/imagenet.py:8-23: RandomResizedCrop(size, scale=(0.5, 1), ratio=(3/4, 4/3), BICUBIC) + RandomHorizontalFlip(0.5)."""

    def __init__(self, size: int = 224, scale=(0.5, 1.0), ratio=(3.0 / 4.0, 4.0 / 3.0), p_flip: float = 0.5, out_dtype=torch.float32):
        self.size, self.scale, self.ratio, self.p_flip, self.out_dtype = size, scale, ratio, p_flip, out_dtype

    def get_params(self, height: int, width: int):
        """torchvision RandomResizedCrop.get_params: up to 10 draws from torch's global RNG, then the central fallback."""
        area = height * width
        log_ratio = (math.log(self.ratio[0]), math.log(self.ratio[1]))
        for _ in range(10):
            target_area = area * torch.empty(1).uniform_(self.scale[0], self.scale[1]).item()
            aspect_ratio = math.exp(torch.empty(1).uniform_(log_ratio[0], log_ratio[1]).item())
            w = int(round(math.sqrt(target_area * aspect_ratio)))
            h = int(round(math.sqrt(target_area / aspect_ratio)))
            if 0 < w <= width and 0 < h <= height:
                i = torch.randint(0, height - h + 1, size=(1,)).item()
                j = torch.randint(0, width - w + 1, size=(1,)).item()
                return i, j, h, w
        in_ratio = float(width) / float(height)
        if in_ratio < min(self.ratio):
            w, h = width, int(round(width / min(self.ratio)))
        elif in_ratio > max(self.ratio):
            h, w = height, int(round(height * max(self.ratio)))
        else:
            w, h = width, height
        return (height - h) // 2, (width - w) // 2, h, w

    def batch(self, images):
        n = self.size
        boxes, flips = [], []
        for im in images:                       # per image: crop draws, then the flip draw — the order Compose applies them
            boxes.append(self.get_params(int(im.shape[0]), int(im.shape[1])))
            flips.append(bool(torch.rand(1).item() < self.p_flip))
        return preprocess_batch(images, boxes, [(n, n)] * len(images), [(0, 0)] * len(images), flips, n, out_dtype=self.out_dtype)

    def __call__(self, image):
        return self.batch([image])[0]
