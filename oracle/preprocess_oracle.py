"""TEST INFRASTRUCTURE — CPU restatement of the image pre-processing the reference applies before `encode_image`
(clip/clip.py:77-84 `_transform`: Resize(n_px, BICUBIC) -> CenterCrop(n_px) -> ToTensor -> Normalize; datasets/imagenet.py:8-23
`get_random_train_tfm`: RandomResizedCrop(224, scale=(0.5, 1), BICUBIC) -> RandomHorizontalFlip -> ToTensor -> Normalize).

The arithmetic lives in two third-party dependencies of the reference that are not under /root/reference:
  * Pillow (`Image.resize(size, BICUBIC)`, present in this image: 12.2.0) — restated here from its published algorithm
    (libImaging/Resample.c: `precompute_coeffs` in double precision, `normalize_coeffs_8bpc` to 22-bit fixed point, a
    horizontal then a vertical 8-bit pass, each rounding to uint8) and PINNED against Pillow itself in
    tests/test_preprocess_cpu.py on seeded images, bit for bit;
  * torchvision.transforms (absent from the image): output-size rule of Resize(int), CenterCrop offsets, ToTensor (/255 in
    fp32) and Normalize ((x - mean) / std in fp32) restated from the published implementation; RandomResizedCrop.get_params /
    RandomHorizontalFlip draw order restated likewise — **parity of the random draws is unpinned**, the pixel arithmetic is
    pinned through Pillow + IEEE fp32.
Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this package."""
import math

import numpy as np

PRECISION_BITS = 32 - 8 - 2
CLIP_MEAN = (0.48145466, 0.4578275, 0.40821073)
CLIP_STD = (0.26862954, 0.26130258, 0.27577711)


def _bicubic(x: float) -> float:
    a = -0.5
    if x < 0.0:
        x = -x
    if x < 1.0:
        return ((a + 2.0) * x - (a + 3.0)) * x * x + 1
    if x < 2.0:
        return (((x - 5) * x + 8) * x - 4) * a
    return 0.0


def precompute_coeffs(in_size: int, out_size: int):
    """Resample.c precompute_coeffs + normalize_coeffs_8bpc for the box [0, in_size): (ksize, bounds [out,2], kk [out,ksize] int32)."""
    scale = float(in_size) / out_size
    filterscale = max(scale, 1.0)
    support = 2.0 * filterscale
    ksize = int(math.ceil(support)) * 2 + 1
    bounds = np.zeros((out_size, 2), dtype=np.int32)
    kk = np.zeros((out_size, ksize), dtype=np.int32)
    ss = 1.0 / filterscale
    for xx in range(out_size):
        center = 0.0 + (xx + 0.5) * scale
        xmin = int(center - support + 0.5)
        if xmin < 0:
            xmin = 0
        xmax = int(center + support + 0.5)
        if xmax > in_size:
            xmax = in_size
        xmax -= xmin
        w = [_bicubic((x + xmin - center + 0.5) * ss) for x in range(xmax)]
        ww = 0.0
        for v in w:
            ww += v
        for x in range(xmax):
            k = w[x] / ww if ww != 0.0 else w[x]
            kk[xx, x] = int(-0.5 + k * (1 << PRECISION_BITS)) if k < 0 else int(0.5 + k * (1 << PRECISION_BITS))
        bounds[xx] = (xmin, xmax)
    return ksize, bounds, kk


def _resample_axis(img: np.ndarray, out_size: int, axis: int) -> np.ndarray:
    """One 8-bit pass along `axis` of an HWC uint8 image."""
    in_size = img.shape[axis]
    _, bounds, kk = precompute_coeffs(in_size, out_size)
    src = np.moveaxis(img, axis, 0).astype(np.int64)
    out = np.empty((out_size,) + src.shape[1:], dtype=np.uint8)
    for xx in range(out_size):
        xmin, xmax = bounds[xx]
        acc = np.full(src.shape[1:], 1 << (PRECISION_BITS - 1), dtype=np.int64)
        for x in range(xmax):
            acc += src[xmin + x] * int(kk[xx, x])
        out[xx] = np.clip(acc >> PRECISION_BITS, 0, 255).astype(np.uint8)
    return np.moveaxis(out, 0, axis)


def resize_bicubic(img: np.ndarray, out_h: int, out_w: int) -> np.ndarray:
    """PIL `Image.fromarray(img).resize((out_w, out_h), Image.BICUBIC)` for an HWC uint8 RGB array: horizontal pass first
    (skipped when the width is unchanged), then vertical (skipped when the height is unchanged)."""
    if out_w != img.shape[1]:
        img = _resample_axis(img, out_w, 1)
    if out_h != img.shape[0]:
        img = _resample_axis(img, out_h, 0)
    return img


def resize_output_size(h: int, w: int, size: int):
    """torchvision Resize(int): the shorter side becomes `size`, the other int(size * long / short)."""
    short, long = (w, h) if w <= h else (h, w)
    new_short, new_long = size, int(size * long / short)
    return (new_long, new_short) if w <= h else (new_short, new_long)      # (out_h, out_w)


def center_crop_offsets(h: int, w: int, n: int):
    return int(round((h - n) / 2.0)), int(round((w - n) / 2.0))


def to_tensor_normalize(img: np.ndarray, mean=CLIP_MEAN, std=CLIP_STD) -> np.ndarray:
    """ToTensor + Normalize: CHW fp32, (x / 255 - mean) / std with every step in IEEE fp32."""
    x = img.astype(np.float32).transpose(2, 0, 1) / np.float32(255)
    m = np.asarray(mean, dtype=np.float32)[:, None, None]
    s = np.asarray(std, dtype=np.float32)[:, None, None]
    return ((x - m).astype(np.float32) / s).astype(np.float32)


def clip_transform(img: np.ndarray, n_px: int) -> np.ndarray:
    """clip/clip.py:77-84 on an HWC uint8 RGB array -> [3, n_px, n_px] fp32."""
    oh, ow = resize_output_size(img.shape[0], img.shape[1], n_px)
    r = resize_bicubic(img, oh, ow)
    top, left = center_crop_offsets(oh, ow, n_px)
    return to_tensor_normalize(r[top:top + n_px, left:left + n_px])


def resized_crop_flip(img: np.ndarray, top: int, left: int, h: int, w: int, size: int, flip: bool) -> np.ndarray:
    """datasets/imagenet.py:8-23 given the drawn parameters: crop -> resize to size x size -> optional hflip -> tensor."""
    r = resize_bicubic(img[top:top + h, left:left + w], size, size)
    if flip:
        r = r[:, ::-1]
    return to_tensor_normalize(np.ascontiguousarray(r))
