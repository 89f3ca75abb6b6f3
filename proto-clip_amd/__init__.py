"""proto_clip_amd — MI355X (gfx950) implementation of Proto-CLIP's data-parallel hot path: CLIP encoder
forwards that fill the memory banks, the per-class prototype reduction, query adapters, and the
dual-bank distance-softmax classification, behind the reference's Python surface (main.py / model.py /
utils.py / clip).  Arithmetic runs in libpclip.so (hand-written HIP, C ABI in include/pclip.h);
importing this package needs no GPU, calling into it does."""
__version__ = "0.1.0"

from . import _lib  # noqa: F401  (ctypes binding; the shared library is loaded on first use)
from ._lib import PclipError  # noqa: F401
