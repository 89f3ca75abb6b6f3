"""Pins oracle/preprocess_oracle.py (restatement of Pillow's bicubic resample, the arithmetic behind the reference's
`Resize(n_px, BICUBIC)` / `RandomResizedCrop`, clip/clip.py:77-84, datasets/imagenet.py:8-23) against Pillow itself, bit for
bit, on seeded images: up- and down-scaling, odd sizes, identity axes, crops."""
import numpy as np
import pytest

from oracle import preprocess_oracle as pp

PIL = pytest.importorskip("PIL.Image")


def _img(h, w, seed):
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, size=(h // 7 + 2, w // 7 + 2, 3)).astype(np.float64)      # blocky structure + noise
    up = np.kron(base, np.ones((7, 7, 1)))[:h, :w]
    return np.clip(up + rng.normal(0, 20, size=(h, w, 3)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("h,w,oh,ow", [(37, 53, 17, 24), (64, 48, 149, 112), (120, 200, 67, 112), (33, 33, 33, 20), (40, 30, 25, 30),
                                       (500, 375, 298, 224), (9, 300, 224, 224), (231, 17, 5, 64)])
def test_resize_matches_pillow(h, w, oh, ow):
    img = _img(h, w, h * 1000 + w)
    ref = np.asarray(PIL.fromarray(img).resize((ow, oh), PIL.BICUBIC))
    assert np.array_equal(pp.resize_bicubic(img, oh, ow), ref)


@pytest.mark.parametrize("h,w,n", [(50, 80, 32), (375, 500, 224), (224, 224, 224), (300, 225, 64), (61, 60, 32)])
def test_clip_transform_matches_pillow_pipeline(h, w, n):
    """Resize(n) -> CenterCrop(n) -> ToTensor -> Normalize with Pillow doing the resize and the torchvision rules restated."""
    img = _img(h, w, h + w)
    oh, ow = pp.resize_output_size(h, w, n)
    assert min(oh, ow) == n
    r = np.asarray(PIL.fromarray(img).resize((ow, oh), PIL.BICUBIC))
    top, left = pp.center_crop_offsets(oh, ow, n)
    ref = pp.to_tensor_normalize(r[top:top + n, left:left + n])
    out = pp.clip_transform(img, n)
    assert out.shape == (3, n, n) and out.dtype == np.float32
    assert np.array_equal(out, ref)


def test_resized_crop_flip_matches_pillow():
    img = _img(90, 130, 5)
    for (top, left, h, w, flip) in [(3, 10, 70, 99, False), (0, 0, 90, 130, True), (20, 40, 33, 57, True)]:
        r = np.asarray(PIL.fromarray(img).crop((left, top, left + w, top + h)).resize((48, 48), PIL.BICUBIC))
        if flip:
            r = r[:, ::-1]
        assert np.array_equal(pp.resized_crop_flip(img, top, left, h, w, 48, flip), pp.to_tensor_normalize(np.ascontiguousarray(r)))


def test_torchvision_size_rules():
    assert pp.resize_output_size(375, 500, 224) == (224, 298)
    assert pp.resize_output_size(500, 375, 224) == (298, 224)
    assert pp.resize_output_size(100, 100, 64) == (64, 64)
    assert pp.center_crop_offsets(224, 298, 224) == (0, 37)
    assert pp.center_crop_offsets(227, 225, 224) == (2, 0)          # round-half-even of 1.5 and 0.5
