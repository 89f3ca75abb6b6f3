"""Seeded synthetic few-shot splits.

No datasets or pretrained CLIP weights exist in the build environment, so every parity / bench input
is generated here from a *portable* counter-based PRNG (splitmix64 -> Box-Muller in float64), which
gives bit-identical inputs in the build container and on the GPU box; only expected outputs are
committed as fixtures.  Seed 1 echoes the reference's ``utils.get_seed`` (utils.py:22-26).

Feature model: class centres c_n ~ N(0, I_D); a support/query feature is normalize(c_n + sigma*eps)
stored fp16 exactly as the reference stores them (fp16 normalise, Appendix A); the textual bank is
normalize(c_n + 0.5*eps).  Support rows are sorted by class (utils.py:324-326).
"""
import numpy as np
import torch

_MASK = np.uint64(0xFFFFFFFFFFFFFFFF)


def _splitmix64(idx: np.ndarray, seed: int) -> np.ndarray:
    """splitmix64 output for counters ``idx`` (uint64) under ``seed``; pure uint64 wraparound."""
    with np.errstate(over="ignore"):
        z = (idx.astype(np.uint64) + np.uint64(seed & 0xFFFFFFFFFFFFFFFF)) * np.uint64(0x9E3779B97F4A7C15)
        z = (z ^ (z >> np.uint64(30))) * np.uint64(0xBF58476D1CE4E5B9)
        z = (z ^ (z >> np.uint64(27))) * np.uint64(0x94D049BB133111EB)
        z = z ^ (z >> np.uint64(31))
    return z


def uniform(n: int, seed: int, stream: int = 0) -> np.ndarray:
    """n float64 uniforms in (0,1), reproducible across machines."""
    idx = np.arange(n, dtype=np.uint64) + (np.uint64(stream) << np.uint64(40))
    bits = _splitmix64(idx, seed) >> np.uint64(11)          # 53 random bits
    return (bits.astype(np.float64) + 0.5) * (1.0 / 9007199254740992.0)


def normal(shape, seed: int, stream: int = 0) -> np.ndarray:
    """Standard normals (float64) via Box-Muller on two uniform streams."""
    n = int(np.prod(shape))
    m = (n + 1) // 2
    u1 = uniform(m, seed, 2 * stream)
    u2 = uniform(m, seed, 2 * stream + 1)
    r = np.sqrt(-2.0 * np.log(u1))
    z = np.concatenate([r * np.cos(2.0 * np.pi * u2), r * np.sin(2.0 * np.pi * u2)])[:n]
    return z.reshape(shape)


def randint(n: int, high: int, seed: int, stream: int = 0) -> np.ndarray:
    return (uniform(n, seed, stream) * high).astype(np.int64) % high


def _l2n_f16(x32: torch.Tensor) -> torch.Tensor:
    """fp32 -> fp16 cast then the reference's fp16 normalise (utils.py:352): r16(f / r16(||f||))."""
    h = x32.half()
    n = h.float().pow(2).sum(-1, keepdim=True).sqrt().half()
    return (h.float() / n.float()).half()


class FewShotSplit:
    """Container mirroring what main.py:529-544 hands to run_proto_clip."""

    def __init__(self, keys, values, text_bank, val_f, val_y, test_f, test_y, N, K, D):
        self.visual_memory_keys = keys          # [D, N*K] fp16, columns sorted by class
        self.visual_memory_values = values      # [N*K, N] int64 one-hot
        self.textual_memory_bank = text_bank    # [D, N] fp16
        self.val_features, self.val_labels = val_f, val_y
        self.test_features, self.test_labels = test_f, test_y
        self.N, self.K, self.D = N, K, D


def make_split(N: int, K: int, D: int, Q_val: int, Q_test: int, seed: int = 1, sigma: float = 0.8,
               sigma_text: float = 0.5, unnormalized_text: bool = False) -> FewShotSplit:
    """Seeded synthetic few-shot split with the reference's tensor layouts."""
    centres = normal((N, D), seed, 0)
    sup = centres[:, None, :] + sigma * normal((N, K, D), seed, 1)
    keys_rows = _l2n_f16(torch.from_numpy(sup.reshape(N * K, D)).float())      # [N*K, D]
    txt = centres + sigma_text * normal((N, D), seed, 2)
    txt_t = torch.from_numpy(txt).float()
    if unnormalized_text:      # learned banks are not unit norm (pretrained_ckpt/fewsol-198-F)
        text_rows = (txt_t / txt_t.norm(dim=-1, keepdim=True) * 1.45).half()
    else:
        text_rows = _l2n_f16(txt_t)

    def queries(Q, stream):
        y = randint(Q, N, seed, 100 + stream)
        f = centres[y] + sigma * normal((Q, D), seed, 3 + stream)
        return _l2n_f16(torch.from_numpy(f).float()), torch.from_numpy(y)

    val_f, val_y = queries(Q_val, 0)
    test_f, test_y = queries(Q_test, 1)
    labels = torch.arange(N).repeat_interleave(K)
    values = torch.nn.functional.one_hot(labels, N)
    return FewShotSplit(keys_rows.t().contiguous(), values, text_rows.t().contiguous(),
                        val_f, val_y, test_f, test_y, N, K, D)


def make_images(B: int, res: int, seed: int = 1, stream: int = 50, n_class: int = 0, labels=None) -> torch.Tensor:
    """Pre-processed image batch [B,3,res,res] fp32: per-class low-frequency pattern + noise
    (SURVEY §8d 'Full-path configs').  Values are in the range CLIP's Normalize produces.  `labels` (int array [B])
    fixes the class of every image; otherwise classes are drawn from the seeded stream (`image_labels`)."""
    noise = normal((B, 3, res, res), seed, stream).astype(np.float32)
    if n_class > 0 or labels is not None:
        y = np.asarray(labels, dtype=np.int64) if labels is not None else randint(B, n_class, seed, stream + 1)
        yy, xx = np.meshgrid(np.arange(res), np.arange(res), indexing="ij")
        fx = (y % 7 + 1)[:, None, None, None]
        fy = (y // 7 % 7 + 1)[:, None, None, None]
        pat = np.sin(2 * np.pi * fx * xx[None, None] / res) * np.cos(2 * np.pi * fy * yy[None, None] / res)
        noise = (0.7 * pat + 0.5 * noise).astype(np.float32)
    return torch.from_numpy(noise)


# Config shapes: name -> (D, N, K, Q_val, Q_test, alpha, beta, adapter)
CONFIG_SHAPES = {
    "C1_caltech101_rn50": (1024, 100, 1, 1649, 2465, 0.8, 9.0, "conv-3x"),
    "C2_eurosat_vitb32": (512, 10, 16, 5400, 8100, 1.0, 0.7, "fc"),
    "C3_imagenet_vitb16": (512, 1000, 16, 50000, 50000, 0.5, 12.0, "conv-3x"),
    "C5_fewsol198_vitl14": (768, 198, 16, 666, 32, 0.2, 12.0, "fc"),
}
