"""Device pre-processing (csrc/pclip_preprocess.hip, proto_clip_amd/preprocess.py) against the oracle restatement of
Pillow + torchvision semantics (oracle/preprocess_oracle.py, pinned bit for bit to Pillow in tests/test_preprocess_cpu.py):
integer / byte work and IEEE fp32 -> the bar is bit-exact."""
import numpy as np
import pytest
import torch

from oracle import preprocess_oracle as pp

pytestmark = pytest.mark.gpu


def _img(h, w, seed):
    rng = np.random.RandomState(seed)
    base = rng.randint(0, 256, size=(h // 7 + 2, w // 7 + 2, 3)).astype(np.float64)
    up = np.kron(base, np.ones((7, 7, 1)))[:h, :w]
    return np.clip(up + rng.normal(0, 20, size=(h, w, 3)), 0, 255).astype(np.uint8)


@pytest.mark.parametrize("n", [32, 224])
def test_clip_preprocess_batch_bit_exact(n):
    from proto_clip_amd.preprocess import ClipPreprocess
    sizes = [(50, 80), (375, 500), (500, 375), (n, n), (300, 225), (61, 60), (n, 3 * n), (2 * n + 1, n), (40, 33)]
    imgs = [_img(h, w, h * 7 + w) for h, w in sizes]
    out = ClipPreprocess(n).batch(imgs).cpu().numpy()
    for i, im in enumerate(imgs):
        assert np.array_equal(out[i], pp.clip_transform(im, n)), sizes[i]
    half = ClipPreprocess(n, out_dtype=torch.float16).batch(imgs[:3]).cpu()
    assert torch.equal(half, torch.from_numpy(out[:3]).half())              # fused cast == encode_image's image.type(dtype)
    one = ClipPreprocess(n)(imgs[1]).cpu().numpy()
    assert np.array_equal(one, out[1])


def test_random_train_transform_bit_exact_and_rng_order():
    from proto_clip_amd.preprocess import RandomTrainTransform
    imgs = [_img(90, 130, 5), _img(200, 150, 6), _img(64, 64, 7), _img(31, 257, 8)]
    tfm = RandomTrainTransform(size=48)
    torch.manual_seed(3)
    out = tfm.batch(imgs).cpu().numpy()
    torch.manual_seed(3)                                                    # replay the draws on the host
    saw_flip = set()
    for i, im in enumerate(imgs):
        top, left, h, w = tfm.get_params(im.shape[0], im.shape[1])
        flip = bool(torch.rand(1).item() < 0.5)
        saw_flip.add(flip)
        assert 0 < h <= im.shape[0] and 0 < w <= im.shape[1]
        assert np.array_equal(out[i], pp.resized_crop_flip(im, top, left, h, w, 48, flip)), i
    assert len(saw_flip) == 2 or len(imgs) < 3


def test_preprocess_feeds_the_encoder():
    """uint8 images -> device preprocess (fp16 out) -> encode_image == the same pixels through the fp32 entry."""
    from conftest import ENCODERS
    from proto_clip_amd.clip.model import build_model, random_state_dict
    from proto_clip_amd.preprocess import ClipPreprocess
    kw = ENCODERS["tiny"]
    model = build_model(random_state_dict(seed=11, **kw)).cuda()
    n = kw["image_resolution"]
    imgs = [_img(40 + 3 * i, 57 - 2 * i, i) for i in range(5)]
    x32 = ClipPreprocess(n).batch(imgs)
    with torch.no_grad():
        f32 = model.encode_image(x32)
        f16 = model.encode_image(ClipPreprocess(n, out_dtype=torch.float16).batch(imgs))
    assert torch.equal(f32, f16)


def test_preprocess_empty_batch_and_bad_input():
    from proto_clip_amd._lib import PclipError
    from proto_clip_amd.preprocess import ClipPreprocess, preprocess_batch
    assert ClipPreprocess(32).batch([]).shape == (0, 3, 32, 32)
    with pytest.raises(PclipError):
        ClipPreprocess(32).batch([np.zeros((40, 40), dtype=np.uint8)])            # not HWC RGB
    with pytest.raises(PclipError):
        preprocess_batch([_img(40, 40, 1)], [(0, 0, 50, 40)], [(32, 32)], [(0, 0)], [False], 32)     # box outside the image
