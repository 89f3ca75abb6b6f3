"""Tensor-level wrappers over the C ABI (include/pclip.h).  PyTorch is used only to own device memory
and streams; every arithmetic step below is a libpclip kernel.  All functions require CUDA (ROCm)
tensors and raise PclipError otherwise — there is deliberately no CPU path."""
import math
import os

import numpy as np
import torch

from . import _lib
from ._lib import check, ptr, require_cuda, stream

_ws_cache = {}


def _workspace(nbytes: int, device) -> torch.Tensor:
    """Scratch for one call (the C ABI never allocates).  Outside graph capture: a grow-only buffer per (device, stream) — calls
    on one stream are ordered, so they can share it.  During a hipGraph capture nothing is cached: the buffer comes from the
    capture's private memory pool and must belong to THAT graph only (a cached pointer would be baked into later captures whose
    pool does not own it); torch's caching allocator reuses the block between the calls of one capture."""
    if torch.cuda.is_current_stream_capturing():
        return torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
    key = (device, torch.cuda.current_stream(device).cuda_stream)
    buf = _ws_cache.get(key)
    if buf is None or buf.numel() < nbytes:
        buf = torch.empty(max(nbytes, 1 << 20), dtype=torch.uint8, device=device)
        _ws_cache[key] = buf
    return buf


def evict_workspace(stream_obj):
    """Forget the scratch buffer cached for `stream_obj` (a torch.cuda.Stream that will not be used again)."""
    for key in [k for k in _ws_cache if k[1] == stream_obj.cuda_stream]:
        del _ws_cache[key]


def release_workspaces():
    """Drop the cached scratch buffers (e.g. after a warm-up on a side stream that will not be used again)."""
    _ws_cache.clear()


def _f16c(t: torch.Tensor) -> torch.Tensor:
    if t.dtype != torch.float16:
        raise _lib.PclipError(f"expected a float16 tensor, got {t.dtype}")
    return t if t.is_contiguous() else t.contiguous()


def l2norm_rows(x: torch.Tensor, out: torch.Tensor = None, want_sq: bool = False):
    """r16(x / r16(||x||)) per row — utils.py:352, main.py:182-185, 408-409."""
    require_cuda(x)
    x = _f16c(x)
    R, D = x.shape
    y = torch.empty_like(x) if out is None else out
    sq = torch.empty(R, dtype=torch.float32, device=x.device) if want_sq else None
    check(_lib.load().pclip_l2norm_rows_f16(ptr(x), ptr(y), R, D, ptr(sq), stream()), "pclip_l2norm_rows_f16")
    return (y, sq) if want_sq else y


def row_sqnorm(x: torch.Tensor) -> torch.Tensor:
    require_cuda(x)
    x = _f16c(x)
    sq = torch.empty(x.shape[0], dtype=torch.float32, device=x.device)
    check(_lib.load().pclip_row_sqnorm_f16(ptr(x), x.shape[0], x.shape[1], ptr(sq), stream()), "pclip_row_sqnorm_f16")
    return sq


def transpose(x: torch.Tensor) -> torch.Tensor:
    """Materialised fp16 transpose (bank layout [D, N*K] <-> [N*K, D], utils.py:320)."""
    require_cuda(x)
    x = _f16c(x)
    R, C = x.shape
    y = torch.empty(C, R, dtype=torch.float16, device=x.device)
    check(_lib.load().pclip_transpose_f16(ptr(x), R, C, ptr(y), stream()), "pclip_transpose_f16")
    return y


def proto_build(mem: torch.Tensor, N: int, K: int, per_shot_norm: bool = True, fp32_out: bool = False,
                want_sq: bool = False):
    """Prototype reduction (main.py:399-402 eval / 260-264 train / 173-176 zero-shot init).
    mem [N*K, D] fp16 -> proto [N, D] (fp16, or fp32 un-rounded for the train path)."""
    require_cuda(mem)
    mem = _f16c(mem)
    if mem.shape[0] != N * K:
        raise _lib.PclipError(f"memory bank has {mem.shape[0]} rows, expected N*K={N * K}")
    D = mem.shape[1]
    p16 = None if fp32_out else torch.empty(N, D, dtype=torch.float16, device=mem.device)
    p32 = torch.empty(N, D, dtype=torch.float32, device=mem.device) if fp32_out else None
    sq = torch.empty(N, dtype=torch.float32, device=mem.device) if want_sq else None
    check(_lib.load().pclip_proto_build_f16(ptr(mem), N, K, D, int(per_shot_norm), ptr(p16), ptr(p32), ptr(sq),
                                            stream()), "pclip_proto_build_f16")
    out = p32 if fp32_out else p16
    return (out, sq) if want_sq else out


def bank_reduce(feats: torch.Tensor, perm: torch.Tensor = None) -> torch.Tensor:
    """feats [A, R, D] fp16 -> normalise(r16(mean_A)) rows, optionally gathered by perm (utils.py:318-326)."""
    require_cuda(feats, perm)
    feats = _f16c(feats)
    A, R, D = feats.shape
    if perm is not None:
        perm = perm.to(torch.int32).contiguous()
    keys = torch.empty(R, D, dtype=torch.float16, device=feats.device)
    check(_lib.load().pclip_bank_reduce_f16(ptr(feats), A, R, D, ptr(perm), ptr(keys), stream()),
          "pclip_bank_reduce_f16")
    return keys


def partial_sums(mem: torch.Tensor, labels: torch.Tensor, N: int, per_shot_norm: bool = True):
    """This rank's fp32 per-class sums [N, D] + int32 counts [N] (labels must be non-decreasing)."""
    require_cuda(mem, labels)
    mem = _f16c(mem)
    labels = labels.to(torch.int32).contiguous()
    R, D = mem.shape
    sums = torch.empty(N, D, dtype=torch.float32, device=mem.device)
    counts = torch.empty(N, dtype=torch.int32, device=mem.device)
    check(_lib.load().pclip_partial_sums_f16(ptr(mem), ptr(labels), R, N, D, int(per_shot_norm), ptr(sums),
                                             ptr(counts), stream()), "pclip_partial_sums_f16")
    return sums, counts


def proto_finalize(sums: torch.Tensor, counts: torch.Tensor, fp32_out: bool = False, want_sq: bool = False):
    """sums [W, N, D] fp32, counts [W, N] int32 (rank-major) -> prototypes as proto_build."""
    require_cuda(sums, counts)
    if sums.dim() == 2:
        sums, counts = sums[None], counts[None]
    sums, counts = sums.contiguous(), counts.contiguous()
    W, N, D = sums.shape
    p16 = None if fp32_out else torch.empty(N, D, dtype=torch.float16, device=sums.device)
    p32 = torch.empty(N, D, dtype=torch.float32, device=sums.device) if fp32_out else None
    sq = torch.empty(N, dtype=torch.float32, device=sums.device) if want_sq else None
    check(_lib.load().pclip_proto_finalize(ptr(sums), ptr(counts), W, N, D, ptr(p16), ptr(p32), ptr(sq), stream()),
          "pclip_proto_finalize")
    out = p32 if fp32_out else p16
    return (out, sq) if want_sq else out


def padded_ld(N: int) -> int:
    return (N + 63) // 64 * 64


def sqdist(q: torch.Tensor, zi: torch.Tensor, zt: torch.Tensor = None, q_sq=None, zi_sq=None, zt_sq=None):
    """Squared distances of fp16 queries to both prototype banks: fp32 [Q, ldd] each (cdist(...)**2 of
    utils.py:230-233).  Returns (d2i, d2t, ldd); columns >= N are unspecified padding."""
    require_cuda(q, zi, zt)
    q, zi = _f16c(q), _f16c(zi)
    zt = None if zt is None else _f16c(zt)
    Q, D = q.shape
    N = zi.shape[0]
    ldd = padded_ld(N)
    d2i = torch.empty(Q, ldd, dtype=torch.float32, device=q.device)
    d2t = torch.empty(Q, ldd, dtype=torch.float32, device=q.device) if zt is not None else None
    nws = _lib.workspace_bytes(_lib.OP_SQDIST, Q, N, D)
    ws = _workspace(nws, q.device)
    check(_lib.load().pclip_sqdist_f16(ptr(q), ptr(zi), ptr(zt), Q, N, D, ptr(q_sq), ptr(zi_sq), ptr(zt_sq),
                                       ptr(d2i), ptr(d2t), ldd, ptr(ws), ws.numel(), stream()), "pclip_sqdist_f16")
    return d2i, d2t, ldd


def sqdist_f32(q: torch.Tensor, zi: torch.Tensor, zt: torch.Tensor = None):
    """fp32-operand squared distances (training path: fp32 prototypes / adapted queries, main.py:262-281)."""
    require_cuda(q, zi, zt)
    q, zi = q.float().contiguous(), zi.float().contiguous()
    zt = None if zt is None else zt.float().contiguous()
    Q, D = q.shape
    N = zi.shape[0]
    ldd = padded_ld(N)
    d2i = torch.empty(Q, ldd, dtype=torch.float32, device=q.device)
    d2t = torch.empty(Q, ldd, dtype=torch.float32, device=q.device) if zt is not None else None
    check(_lib.load().pclip_sqdist_f32(ptr(q), ptr(zi), ptr(zt), Q, N, D, ptr(d2i), ptr(d2t), ldd, stream()),
          "pclip_sqdist_f32")
    return d2i, d2t, ldd


def fuse_probs(d2i, d2t, N: int, alpha: float, beta: float, want_p=True, want_argmax=False, topk: int = 0):
    """alpha*softmax(-beta*d2i) + (1-alpha)*softmax(-beta*d2t) (utils.py:236-242) + argmax / top-k."""
    require_cuda(d2i, d2t)
    Q, ldd = d2i.shape
    dev = d2i.device
    p = torch.empty(Q, N, dtype=torch.float32, device=dev) if want_p else None
    am = torch.empty(Q, dtype=torch.int32, device=dev) if want_argmax else None
    tp = torch.empty(Q, topk, dtype=torch.float32, device=dev) if topk else None
    ti = torch.empty(Q, topk, dtype=torch.int32, device=dev) if topk else None
    # the reference forms (1 - alpha) in Python double precision and lets torch cast it to fp32
    a32, oma32 = float(np.float32(alpha)), float(np.float32(1 - float(alpha)))
    check(_lib.load().pclip_fuse_probs(ptr(d2i), ptr(d2t), Q, N, ldd, a32, oma32, float(np.float32(beta)), ptr(p),
                                       ptr(am), ptr(tp), ptr(ti), topk, stream()), "pclip_fuse_probs")
    return p, am, tp, ti


def classify(q, zi, zt, alpha: float, beta: float, want_p=False, want_argmax=True, topk: int = 0,
             q_sq=None, zi_sq=None, zt_sq=None):
    """One-call classification (sqdist + fuse) with distances kept in scratch."""
    require_cuda(q, zi, zt)
    q, zi, zt = _f16c(q), _f16c(zi), _f16c(zt)
    Q, D = q.shape
    N = zi.shape[0]
    dev = q.device
    p = torch.empty(Q, N, dtype=torch.float32, device=dev) if want_p else None
    am = torch.empty(Q, dtype=torch.int32, device=dev) if want_argmax else None
    tp = torch.empty(Q, topk, dtype=torch.float32, device=dev) if topk else None
    ti = torch.empty(Q, topk, dtype=torch.int32, device=dev) if topk else None
    ws = _workspace(_lib.workspace_bytes(_lib.OP_CLASSIFY, Q, N, D), dev)
    a32, oma32 = float(np.float32(alpha)), float(np.float32(1 - float(alpha)))
    check(_lib.load().pclip_classify_f16(ptr(q), ptr(zi), ptr(zt), Q, N, D, ptr(q_sq), ptr(zi_sq), ptr(zt_sq), a32,
                                         oma32, float(np.float32(beta)), ptr(p), ptr(am), ptr(tp), ptr(ti), topk,
                                         ptr(ws), ws.numel(), stream()), "pclip_classify_f16")
    return p, am, tp, ti


def proto_classify(mem, N: int, K: int, q, zt, alpha: float, beta: float, per_shot_norm: bool = True, want_p=False, want_argmax=True, topk: int = 0):
    """main.py:399-405 + utils.py:225-244 + main.py:190: prototypes from the memory bank and the classification of q against them.  Returns
    (z_img_proto [N, D] fp16, p, argmax, topk_p, topk_i): `proto_build` followed by `classify` (4.0 + 6.4 us at EuroSAT's size).  (A one-launch form — builder and
    consumer workgroups in one grid — was built in round 5 and measured SLOWER, 12.8 - 16.8 us: the cross-XCD hand-over of the prototype rows costs more than the
    launch it saves; removed in round 6, profiles/r05_c2_phases.txt.)"""
    require_cuda(mem, q, zt)
    mem, q, zt = _f16c(mem), _f16c(q), _f16c(zt)
    if mem.shape[0] != N * K:
        raise _lib.PclipError(f"memory bank has {mem.shape[0]} rows, expected N*K={N * K}")
    zi = proto_build(mem, N, K, per_shot_norm)
    return (zi,) + tuple(classify(q, zi, zt, alpha, beta, want_p=want_p, want_argmax=want_argmax, topk=topk))


CLASSIFY_ROUTES = ("two stages", "one launch, small N", "one launch, mid N", "fused row panels")


def classify_route(Q: int, N: int, D: int, alpha: float, beta: float, want_p=False, want_argmax=True, topk: int = 0, has_zt: bool = True) -> str:
    """The kernels `classify` takes for a call of this shape under the current settings (pclip_classify_route).  The routes differ in fp32 summation order only —
    at a near-tie of p (top-2 margin < ~1e-6) the argmax may depend on the route, i.e. on the batch size; `classify_two_stage()` pins one arithmetic."""
    a32, oma32 = float(np.float32(alpha)), float(np.float32(1 - float(alpha)))
    r = _lib.load().pclip_classify_route(Q, N, D, a32, oma32, float(np.float32(beta)), int(has_zt), int(bool(want_p)), int(bool(want_argmax)), topk,
                                         _lib.workspace_bytes(_lib.OP_CLASSIFY, Q, N, D))
    return CLASSIFY_ROUTES[r]


class classify_mid:
    """`with ops.classify_mid(mode):` — routing of the one-launch mid-N kernel (32 < N <= 256): 0 off (two stages), 1 by size (default), 2 every shape it can run."""
    def __init__(self, mode: int):
        self.mode = mode

    def __enter__(self):
        self.before = _lib.load().pclip_classify_mid_config(self.mode)
        return self

    def __exit__(self, *exc):
        _lib.load().pclip_classify_mid_config(self.before if self.before >= 0 else 1)
        return False


class classify_two_stage:
    """`with ops.classify_two_stage():` — classification through pclip_sqdist_f16 + pclip_fuse_probs instead of the fused row-panel kernel / the one-launch mid-N
    kernel (their reference)."""
    def __enter__(self):
        self.before = _lib.load().pclip_classify_panel_config(0)
        self.before_mid = _lib.load().pclip_classify_mid_config(0)
        return self

    def __exit__(self, *exc):
        _lib.load().pclip_classify_panel_config(self.before if self.before >= 0 else 1)
        _lib.load().pclip_classify_mid_config(self.before_mid if self.before_mid >= 0 else 1)
        return False


class classify_fused:
    """`with ops.classify_fused():` — the fused row-panel kernel for EVERY argmax-only call it can run (by default only calls with enough panels to fill the chip;
    the one-launch mid-N kernel, which would take N <= 256 first, is switched off inside)."""
    def __enter__(self):
        self.before = _lib.load().pclip_classify_panel_config(2)
        self.before_mid = _lib.load().pclip_classify_mid_config(0)
        return self

    def __exit__(self, *exc):
        _lib.load().pclip_classify_panel_config(self.before if self.before >= 0 else 1)
        _lib.load().pclip_classify_mid_config(self.before_mid if self.before_mid >= 0 else 1)
        return False


class classify_panel_passes:
    """`with ops.classify_panel_passes(mode):` — 0 one pass + candidates with proof (default), 1 always two passes, 2 candidates computed, second pass forced (tests)."""
    def __init__(self, mode: int):
        self.mode = mode

    def __enter__(self):
        self.prev = _lib.load().pclip_classify_panel_passes(self.mode)
        return self

    def __exit__(self, *exc):
        _lib.load().pclip_classify_panel_passes(self.prev if self.prev >= 0 else 0)
        return False


def classify_panel_stats(reset: bool = False, tiles: bool = False):
    """(panels classified by the fused kernel, panels that needed a second pass[, class tiles those second passes walked]) since the last reset; synchronises."""
    import ctypes
    out = (ctypes.c_int * 3)()
    check(_lib.load().pclip_classify_panel_stats(ctypes.cast(out, ctypes.c_void_p), int(reset)), "pclip_classify_panel_stats")
    return (int(out[0]), int(out[1]), int(out[2])) if tiles else (int(out[0]), int(out[1]))


def classify_panel_distances(q, zi, zt, exact: bool = False):
    """Test hook: the distances the fused row-panel kernel forms for its first tile, ([256, 128] visual, [256, 128] textual) fp32; exact: with the sqrt round trip."""
    require_cuda(q, zi, zt)
    Q, D = q.shape
    N = zi.shape[0]
    dump = torch.empty(2, 256, 128, dtype=torch.float32, device=q.device)
    ws = _workspace(_lib.workspace_bytes(_lib.OP_CLASSIFY, max(Q, 4096), N, D), q.device)
    check(_lib.load().pclip_classify_panel_dump_f16(ptr(q), ptr(zi), ptr(zt), Q, N, D, ptr(dump), int(exact), ptr(ws), ws.numel(), stream()), "pclip_classify_panel_dump_f16")
    return dump[0], dump[1]


def hp_sweep(d2i, d2t, N: int, labels, alphas, betas) -> torch.Tensor:
    """Correct-counts int32 [na, nb] for every (alpha, beta) pair (main.py:187-199 / 419-430)."""
    require_cuda(d2i, d2t, labels)
    Q, ldd = d2i.shape
    dev = d2i.device
    a = np.asarray(alphas, dtype=np.float64)
    a32 = torch.tensor(a.astype(np.float32), device=dev)
    oma32 = torch.tensor((1 - a).astype(np.float32), device=dev)
    b32 = torch.tensor(np.asarray(betas, dtype=np.float64).astype(np.float32), device=dev)
    labels = labels.to(torch.int32).contiguous()
    correct = torch.zeros(len(a), len(b32), dtype=torch.int32, device=dev)
    check(_lib.load().pclip_hp_sweep(ptr(d2i), ptr(d2t), ptr(labels), Q, N, ldd, ptr(a32), ptr(oma32), len(a),
                                     ptr(b32), len(b32), ptr(correct), stream()), "pclip_hp_sweep")
    return correct


def adapter_fc(x, w1, g1, b1, w2, g2, b2, ratio: float = 0.2, l2norm_out: bool = False, want_sq: bool = False):
    """Adapter_FC.forward (model.py:91-95), optionally fused with the following row normalise."""
    require_cuda(x, w1, w2)
    x = _f16c(x)
    B, D = x.shape
    H = w1.shape[0]
    y = torch.empty_like(x)
    sq = torch.empty(B, dtype=torch.float32, device=x.device) if want_sq else None
    ws = _workspace(_lib.workspace_bytes(_lib.OP_ADAPTER_FC, B, H, D), x.device)
    r32, omr32 = float(np.float32(ratio)), float(np.float32(1 - ratio))
    check(_lib.load().pclip_adapter_fc_f16(ptr(x), B, D, H, ptr(_f16c(w1)), ptr(_f16c(g1)), ptr(_f16c(b1)),
                                           ptr(_f16c(w2)), ptr(_f16c(g2)), ptr(_f16c(b2)), r32, omr32,
                                           int(l2norm_out), ptr(y), ptr(sq), ptr(ws), ws.numel(), stream()),
          "pclip_adapter_fc_f16")
    return (y, sq) if want_sq else y


def layernorm_blend(h, gamma, beta, x, ratio: float = 0.2, l2norm_out: bool = False, eps: float = 1e-5):
    """r16(r16(ratio * LN(h)) + r16((1 - ratio) * x)) — the last stage of Adapter_FC.forward (model.py:92-95) on its own."""
    require_cuda(h, gamma, beta, x)
    h, x = _f16c(h), _f16c(x)
    R, D = h.shape
    y = torch.empty_like(h)
    r32, omr32 = float(np.float32(ratio)), float(np.float32(1 - ratio))
    check(_lib.load().pclip_layernorm_blend_f16(ptr(h), ptr(_f16c(gamma)), ptr(_f16c(beta)), eps, ptr(x), r32, omr32, int(l2norm_out),
                                                ptr(y), None, R, D, stream()), "pclip_layernorm_blend_f16")
    return y


def adapter_conv(x, three_x: bool, conv1, ln1w, ln1b, conv2, ln2w, ln2b, conv3, ln3w, ln3b,
                 l2norm_out: bool = False, want_sq: bool = False):
    """Adapter.forward (model.py:49-78): conv-2x / conv-3x feature adapter, whole pipeline in one kernel."""
    require_cuda(x, conv1)
    x = _f16c(x)
    B, D = x.shape
    y = torch.empty_like(x)
    sq = torch.empty(B, dtype=torch.float32, device=x.device) if want_sq else None
    f = lambda t: None if t is None else ptr(_f16c(t))
    check(_lib.load().pclip_adapter_conv_f16(ptr(x), B, D, int(three_x), f(conv1), f(ln1w), f(ln1b), f(conv2),
                                             f(ln2w), f(ln2b), f(conv3), f(ln3w), f(ln3b), int(l2norm_out), ptr(y),
                                             ptr(sq), stream()), "pclip_adapter_conv_f16")
    return (y, sq) if want_sq else y


# ---- encoder building blocks -------------------------------------------------------------------

def gemm(a, w, bias=None, act: int = 0, residual=None, out=None):
    """out[M,N] = epilogue(a[M,K] @ w[N,K]^T) — nn.Linear with fused bias/QuickGELU/residual."""
    require_cuda(a, w)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float16, device=a.device)
    lib = _lib.load()
    if residual is None and M <= _SPLITK_MAX_M and _SPLITK:
        need = lib.pclip_gemm_splitk_workspace(M, N, K)
        if need and a.stride(0) % 8 == 0 and w.stride(0) % 8 == 0 and out.stride(0) % 8 == 0 and \
                not ((a.data_ptr() | w.data_ptr() | out.data_ptr()) & 15) and (bias is None or not (bias.data_ptr() & 15)):
            ws = _workspace(need, a.device)
            check(lib.pclip_gemm_splitk_f16(ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(out), out.stride(0), M, N, K, ptr(bias), act,
                                            ptr(ws), ws.numel(), stream()), "pclip_gemm_splitk_f16")
            return out
    check(lib.pclip_gemm_f16(ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(out), out.stride(0), M, N, K,
                             ptr(bias), act, ptr(residual), stream()), "pclip_gemm_f16")
    return out


def gemm4w(a, w, bias=None, act: int = 0, residual=None, out=None, var: int = 0):
    """ops.gemm on the four-wave asm-loop kernel explicitly (csrc/pclip_gemm4w.hip; ops.gemm routes its 256 x 256 tiles there by itself).  var: 0 product loop,
    1 race-stress build, 6 the no-epilogue ablation (stores nothing), 8 the stamped diagnostic build (pclip_gemm4w_stamp_buffer).  Raises PclipError for shapes outside the kernel (N % 256, K % 64, K >= 192)."""
    require_cuda(a, w)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float16, device=a.device)
    check(_lib.load().pclip_gemm4w_var_f16(ptr(a), a.stride(0), ptr(w), w.stride(0), ptr(out), out.stride(0), M, N, K, ptr(bias), act, ptr(residual), var,
                                           stream()), "pclip_gemm4w_f16")
    return out


class gemm_eight_wave:
    """`with ops.gemm_eight_wave():` — ops.gemm's 256 x 256 tiles on the eight-wave kernel (the four-wave kernel's bit-identity reference)."""
    def __enter__(self):
        self.before = _lib.load().pclip_gemm4w_config(0)
        return self

    def __exit__(self, *exc):
        _lib.load().pclip_gemm4w_config(1 if self.before != 0 else 0)
        return False


# split-K for small M (serving requests).  Opt-in (`with ops.low_latency():`, or PCLIP_GEMM_SPLITK=1): a split call sums K in
# slices, so its last fp16 bit can differ from the unsplit kernel's — the batch paths keep one arithmetic for every batch size,
# the serving entry takes the latency.
_SPLITK = os.environ.get("PCLIP_GEMM_SPLITK", "0") == "1"
_SPLITK_MAX_M = 4096


def splitk_active(M: int) -> bool:
    """True when ops.gemm would consider the split-K kernel for an M-row call (low-latency mode and a small M)."""
    return _SPLITK and M <= _SPLITK_MAX_M


class low_latency:
    """Context manager: inside it, ops.gemm uses pclip_gemm_splitk_f16 for the shapes pclip_gemm_splitk_workspace accepts."""

    def __init__(self, enabled: bool = True):
        self.enabled = enabled

    def __enter__(self):
        global _SPLITK
        self.prev, _SPLITK = _SPLITK, self.enabled
        return self

    def __exit__(self, *exc):
        global _SPLITK
        _SPLITK = self.prev
        return False


def gemm_bn(a, w, scale, shift, relu: bool = True, out=None):
    """relu?(bn(a @ w.T)) with eval-mode BatchNorm folded to fp32 per-column scale / shift — conv + bn (+ relu) of the ResNet
    tower in one launch (clip/model.py:43-52)."""
    require_cuda(a, w, scale, shift)
    a, w = _f16c(a), _f16c(w)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float16, device=a.device)
    check(_lib.load().pclip_gemm_bn_f16(ptr(a), K, ptr(w), w.shape[1], ptr(out), N, M, N, K, ptr(scale), ptr(shift), int(relu),
                                        stream()), "pclip_gemm_bn_f16")
    return out


def gemm_bn_res_relu(a, w, scale, shift, residual, out=None):
    """relu(bn(a @ w.T) + residual): conv3 + bn3 + identity add + ReLU of a bottleneck in one launch where the fused epilogue
    applies (N % 64 == 0, aligned operands), otherwise GEMM followed by bn_act — the same values either way."""
    require_cuda(a, w, scale, shift, residual)
    a, w, residual = _f16c(a), _f16c(w), _f16c(residual)
    M, K = a.shape
    N = w.shape[0]
    if out is None:
        out = torch.empty(M, N, dtype=torch.float16, device=a.device)
    if N % 64 == 0 and K % 64 == 0 and not ((a.data_ptr() | w.data_ptr() | out.data_ptr() | residual.data_ptr() | scale.data_ptr() | shift.data_ptr()) & 15):
        check(_lib.load().pclip_gemm_bn_res_f16(ptr(a), K, ptr(w), w.shape[1], ptr(out), N, M, N, K, ptr(scale), ptr(shift), ptr(residual),
                                                stream()), "pclip_gemm_bn_res_f16")
        return out
    gemm(a, w, out=out)
    return bn_act(out, scale, shift, residual=residual, relu=True, out=out)


_zero_line = {}


def conv3x3_bn(x, w, scale, shift, B: int, H: int, W: int, Cin: int, relu: bool = True):
    """relu?(bn(conv3x3(x))) (stride 1, pad 1) on NHWC rows x [B*H*W, Cin] with w [Cout, 9*Cin] in (ky, kx, Cin) order:
    implicit GEMM, no im2col buffer.  Cin a multiple of 64 — or 8 / 16 / 32 with the rows of w zero-padded to a multiple of 64 —
    and Cout a multiple of 64."""
    require_cuda(x, w, scale, shift)
    x, w = _f16c(x), _f16c(w)
    Cout = w.shape[0]
    if w.shape[1] != (9 * Cin + 63) // 64 * 64 or x.numel() != B * H * W * Cin:
        raise _lib.PclipError("conv3x3_bn: shape mismatch")
    z = _zero_line.get(x.device)
    if z is None:
        z = _zero_line[x.device] = torch.zeros(64, dtype=torch.float16, device=x.device)
    y = torch.empty(B * H * W, Cout, dtype=torch.float16, device=x.device)
    check(_lib.load().pclip_conv3x3_bn_f16(ptr(x), ptr(w), ptr(z), B, H, W, Cin, Cout, ptr(scale), ptr(shift), int(relu), ptr(y),
                                           stream()), "pclip_conv3x3_bn_f16")
    return y


class conv_strip:
    """`with ops.conv_strip(on):` — routing of the narrow 3x3 convolutions (Cin, Cout in {32, 64} at 56 x 56 / 112 x 112) to csrc/pclip_conv_strip.hip (default on)
    or to the implicit-GEMM kernel (off: its reference in the tests)."""
    def __init__(self, on: bool):
        self.mode = int(bool(on))

    def __enter__(self):
        self.before = _lib.load().pclip_conv3x3_strip_config(self.mode)
        return self

    def __exit__(self, *exc):
        _lib.load().pclip_conv3x3_strip_config(self.before)
        return False


def conv_strip_applies(B: int, H: int, W: int, Cin: int, Cout: int) -> bool:
    return bool(_lib.load().pclip_conv3x3_strip_applies(B, H, W, Cin, Cout))


def conv3x3_pool_applies(H: int, W: int, Cin: int, Cout: int) -> bool:
    return bool(_lib.load().pclip_conv3x3_pool_applies(H, W, Cin, Cout))


def conv3x3_bn_pool(x, w, scale, shift, B: int, H: int, W: int, Cin: int):
    """avgpool2(relu(bn(conv3x3(x)))) in one launch: [B * (H / 2) * (W / 2), Cout] (the stem's tail, clip/model.py:104-105, 142-143)."""
    require_cuda(x, w, scale, shift)
    x, w = _f16c(x), _f16c(w)
    Cout = w.shape[0]
    if w.shape[1] != (9 * Cin + 63) // 64 * 64 or x.numel() != B * H * W * Cin:
        raise _lib.PclipError("conv3x3_bn_pool: shape mismatch")
    y = torch.empty(B * (H // 2) * (W // 2), Cout, dtype=torch.float16, device=x.device)
    check(_lib.load().pclip_conv3x3_bn_pool_f16(ptr(x), ptr(w), B, H, W, Cin, Cout, ptr(scale), ptr(shift), ptr(y), stream()), "pclip_conv3x3_bn_pool_f16")
    return y


def stem_conv_applies(R: int, Cout: int) -> bool:
    return bool(_lib.load().pclip_stem_conv_applies(R, Cout))


def stem_conv_bn(img, w, scale, shift, relu: bool = True):
    """relu?(bn(conv3x3 stride 2 pad 1)) of NCHW images [B, 3, R, R] (fp32 or fp16) -> NHWC rows [B * Ho * Ho, Cout] fp16, w [Cout, 64] in im2col column order
    (clip/model.py:100-102, 138): no im2col matrix, no separate cast."""
    require_cuda(img, w, scale, shift)
    if img.dim() != 4 or img.shape[1] != 3 or img.shape[2] != img.shape[3] or img.dtype not in (torch.float32, torch.float16) or w.shape[1] != 64:
        raise _lib.PclipError("stem_conv_bn: images [B, 3, R, R] fp32 / fp16 and w [Cout, 64] expected")
    img, w = img.contiguous(), _f16c(w)
    B, R, Cout = img.shape[0], img.shape[2], w.shape[0]
    Ho = (R - 1) // 2 + 1
    y = torch.empty(B * Ho * Ho, Cout, dtype=torch.float16, device=img.device)
    check(_lib.load().pclip_stem_conv_bn_f16(ptr(img), int(img.dtype == torch.float32), B, R, ptr(w), Cout, ptr(scale), ptr(shift), int(relu), ptr(y), stream()),
          "pclip_stem_conv_bn_f16")
    return y


def layernorm(x, gamma, beta, eps: float = 1e-5, out=None, rows: int = None, ld: int = None):
    """fp16 in/out LayerNorm with fp32 statistics and fp32 affine (clip/model.py:155-161)."""
    require_cuda(x, gamma, beta)
    D = x.shape[-1]
    R = x.numel() // D if rows is None else rows
    ld = D if ld is None else ld
    if out is None:
        out = torch.empty(R, D, dtype=torch.float16, device=x.device)
    check(_lib.load().pclip_layernorm_f16(ptr(x), ld, ptr(gamma), ptr(beta), eps, ptr(out), R, D, stream()),
          "pclip_layernorm_f16")
    return out


def add_layernorm(x, delta, gamma, beta, eps: float = 1e-5, out=None, update_x: bool = True, rows: int = None,
                  ld: int = None):
    """xs = r16(x + delta) (written back into x when update_x) and returns r16(LayerNorm(xs)): the residual add of
    clip/model.py:188-189 fused into the LayerNorm that consumes it."""
    require_cuda(x, delta, gamma, beta)
    D = x.shape[-1]
    R = x.numel() // D if rows is None else rows
    ld = D if ld is None else ld
    if out is None:
        out = torch.empty(R, D, dtype=torch.float16, device=x.device)
    check(_lib.load().pclip_add_layernorm_f16(ptr(x), ptr(delta), ld, ptr(x) if update_x else None, ptr(gamma), ptr(beta),
                                              eps, ptr(out), R, D, stream()), "pclip_add_layernorm_f16")
    return out


def attention(qkv, B: int, L: int, H: int, causal: bool = False, out=None):
    require_cuda(qkv)
    W = H * 64
    if out is None:
        out = torch.empty(B * L, W, dtype=torch.float16, device=qkv.device)
    check(_lib.load().pclip_attention_f16(ptr(qkv), ptr(out), B, L, H, 64, int(causal), stream()),
          "pclip_attention_f16")
    return out


def attention_first_queries(q, kv, B: int, L: int, Lq: int, H: int):
    """Attention output of the first Lq tokens only: q [B*Lq, W] (projected queries of those tokens), kv [B*L, 2W] (keys |
    values of every token).  Same arithmetic per query row as `attention`."""
    require_cuda(q, kv)
    W = H * 64
    out = torch.empty(B * Lq, W, dtype=torch.float16, device=q.device)
    check(_lib.load().pclip_attention_q_f16(ptr(q), W, Lq * W, ptr(kv), 2 * W, 0, W, ptr(out), B, L, Lq, H, 64, 0, stream()),
          "pclip_attention_q_f16")
    return out


def cast_f16(x: torch.Tensor) -> torch.Tensor:
    require_cuda(x)
    x = x.contiguous()
    y = torch.empty(x.shape, dtype=torch.float16, device=x.device)
    check(_lib.load().pclip_cast_f32_f16(ptr(x), ptr(y), x.numel(), stream()), "pclip_cast_f32_f16")
    return y


def im2col_patches(img: torch.Tensor, P: int) -> torch.Tensor:
    require_cuda(img)
    B, C, R, _ = img.shape
    G = R // P
    ld = (3 * P * P + 63) // 64 * 64        # K padded to the GEMM's K-tile (ViT-L/14: 588 -> 640)
    cols = torch.empty(B * G * G, ld, dtype=torch.float16, device=img.device)
    if img.dtype == torch.float32:             # fused with the cast to fp16 (clip/model.py:339)
        check(_lib.load().pclip_im2col_patches_f32(ptr(img), B, R, P, ptr(cols), ld, stream()), "pclip_im2col_patches_f32")
        return cols
    check(_lib.load().pclip_im2col_patches_f16(ptr(img), B, R, P, ptr(cols), ld, stream()),
          "pclip_im2col_patches_f16")
    return cols


def vit_assemble_tokens(patch_emb, class_emb, pos_emb, B: int, G2: int, W: int) -> torch.Tensor:
    tokens = torch.empty(B * (G2 + 1), W, dtype=torch.float16, device=patch_emb.device)
    check(_lib.load().pclip_vit_assemble_tokens_f16(ptr(patch_emb), ptr(class_emb), ptr(pos_emb), B, G2, W,
                                                    ptr(tokens), stream()), "pclip_vit_assemble_tokens_f16")
    return tokens


def vit_embed_ln(patch_emb, class_emb, pos_emb, B: int, G2: int, W: int, g_pre, b_pre, g_1, b_1, eps: float = 1e-5):
    """(x0, h) = (ln_pre(tokens), ln_1(x0)) with tokens = [class ; patches] + pos, one pass (clip/model.py:225-227, 188)."""
    R = B * (G2 + 1)
    x0 = torch.empty(R, W, dtype=torch.float16, device=patch_emb.device)
    f = lambda t: t if t is None or t.dtype == torch.float32 else t.float()
    g_pre, b_pre, g_1, b_1 = f(g_pre), f(b_pre), f(g_1), f(b_1)
    h = torch.empty_like(x0)
    check(_lib.load().pclip_vit_embed_ln_f16(ptr(patch_emb), ptr(class_emb), ptr(pos_emb), B, G2, W, ptr(g_pre), ptr(b_pre), ptr(g_1),
                                             ptr(b_1), eps, ptr(x0), ptr(h), stream()), "pclip_vit_embed_ln_f16")
    return x0, h


def text_embed(tokens, tok_emb, pos_emb) -> torch.Tensor:
    require_cuda(tokens, tok_emb)
    B, L = tokens.shape
    vocab, W = tok_emb.shape
    tokens = tokens.to(torch.int64).contiguous()
    x = torch.empty(B * L, W, dtype=torch.float16, device=tok_emb.device)
    check(_lib.load().pclip_text_embed_f16(ptr(tokens), ptr(tok_emb), ptr(pos_emb), B, L, W, vocab, ptr(x),
                                           stream()), "pclip_text_embed_f16")
    return x


def gather_eot(x, tokens, B: int, L: int, W: int) -> torch.Tensor:
    tokens = tokens.to(torch.int64).contiguous()
    out = torch.empty(B, W, dtype=torch.float16, device=x.device)
    check(_lib.load().pclip_gather_eot_f16(ptr(x), ptr(tokens), B, L, W, ptr(out), stream()), "pclip_gather_eot_f16")
    return out


# ---- ModifiedResNet building blocks (activations NHWC fp16) ---------------------------------------------

def im2col3x3(x, strides, B: int, H: int, W: int, C: int, stride: int = 1) -> torch.Tensor:
    """[B*Ho*Wo, round_up(9*C, 64)] fp16 rows for a 3x3 / pad 1 convolution; `strides` = element strides
    (batch, y, x, channel) of the input buffer."""
    require_cuda(x)
    Ho, Wo = (H - 1) // stride + 1, (W - 1) // stride + 1
    ld = (9 * C + 63) // 64 * 64
    cols = torch.empty(B * Ho * Wo, ld, dtype=torch.float16, device=x.device)
    sb, sh, sw, sc = strides
    check(_lib.load().pclip_im2col3x3_f16(ptr(x), sb, sh, sw, sc, B, H, W, C, stride, ptr(cols), ld, stream()),
          "pclip_im2col3x3_f16")
    return cols


def bn_act(x, scale, shift, residual=None, relu: bool = True, out=None) -> torch.Tensor:
    require_cuda(x, scale, shift, residual)
    rows, C = x.shape
    if out is None:
        out = torch.empty_like(x)
    check(_lib.load().pclip_bn_act_f16(ptr(x), ptr(scale), ptr(shift), ptr(residual), int(relu), ptr(out), rows, C, stream()),
          "pclip_bn_act_f16")
    return out


def avgpool_nhwc(x, B: int, H: int, W: int, C: int, k: int) -> torch.Tensor:
    require_cuda(x)
    y = torch.empty(B * (H // k) * (W // k), C, dtype=torch.float16, device=x.device)
    check(_lib.load().pclip_avgpool_nhwc_f16(ptr(x), B, H, W, C, k, ptr(y), stream()), "pclip_avgpool_nhwc_f16")
    return y


def attnpool_tokens(x, pos16, B: int, HW: int, C: int) -> torch.Tensor:
    require_cuda(x, pos16)
    t = torch.empty(B * (HW + 1), C, dtype=torch.float16, device=x.device)
    check(_lib.load().pclip_attnpool_tokens_f16(ptr(x), ptr(pos16), B, HW, C, ptr(t), stream()), "pclip_attnpool_tokens_f16")
    return t


# ---------------------------------------------------------------- training step (csrc/pclip_train.hip) ----------
def cast_f32(x: torch.Tensor) -> torch.Tensor:
    """fp16 -> fp32 copy (`.float()`)."""
    require_cuda(x)
    x = _f16c(x)
    y = torch.empty(x.shape, dtype=torch.float32, device=x.device)
    check(_lib.load().pclip_cast_f16_f32(ptr(x), ptr(y), x.numel(), stream()), "pclip_cast_f16_f32")
    return y


def gemm_f32(a: torch.Tensor, b: torch.Tensor, trans_a: bool = False, trans_b: bool = False, alpha: float = 1.0,
             out: torch.Tensor = None, beta: float = 0.0) -> torch.Tensor:
    """out = alpha * op(a) @ op(b) + beta * out in fp32 on the fp32 MFMA; a, b are 2-D row-major (unit column stride, any
    row stride) fp16 or fp32."""
    require_cuda(a, b)
    for t in (a, b):
        if t.dtype not in (torch.float16, torch.float32) or t.dim() != 2 or t.stride(1) != 1:
            raise _lib.PclipError("gemm_f32 takes row-major 2-D fp16/fp32 tensors")
    M, K = (a.shape[1], a.shape[0]) if trans_a else a.shape
    Kb, N = (b.shape[1], b.shape[0]) if trans_b else b.shape
    if K != Kb:
        raise _lib.PclipError(f"gemm_f32: inner dimensions differ ({K} vs {Kb})")
    rsa, csa = (1, a.stride(0)) if trans_a else (a.stride(0), 1)
    rsb, csb = (1, b.stride(0)) if trans_b else (b.stride(0), 1)
    if out is None:
        out = torch.empty(M, N, dtype=torch.float32, device=a.device)
        beta = 0.0
    ws = _workspace(32 * M * N * 4, a.device) if K >= 1024 and ((M + 63) // 64) * ((N + 63) // 64) <= 128 else None
    check(_lib.load().pclip_gemm_f32(ptr(a), int(a.dtype == torch.float16), rsa, csa, ptr(b), int(b.dtype == torch.float16),
                                     rsb, csb, ptr(out), out.stride(0), M, N, K, alpha, beta, ptr(ws),
                                     ws.numel() if ws is not None else 0, stream()), "pclip_gemm_f32")
    return out


def colsum_f32(x: torch.Tensor, rows: int = None, cols: int = None, scale: float = 1.0, out: torch.Tensor = None) -> torch.Tensor:
    """scale * x.sum(0) for a 2-D fp32 tensor (leading dimension = x.stride(0)); deterministic.  With `out` the sum is
    ADDED to it."""
    require_cuda(x)
    R = x.shape[0] if rows is None else rows
    C = x.shape[1] if cols is None else cols
    acc = out is not None
    if out is None:
        out = torch.empty(C, dtype=torch.float32, device=x.device)
    ws = _workspace(64 * C * 4, x.device) if R > 128 else None
    check(_lib.load().pclip_colsum_f32(ptr(x), x.stride(0), R, C, scale, ptr(out), int(acc), ptr(ws), ws.numel() if ws is not None else 0,
                                       stream()), "pclip_colsum_f32")
    return out


def addscaled_rows_(c: torch.Tensor, x: torch.Tensor, rowscale: torch.Tensor, s: float) -> torch.Tensor:
    """c[r, :] += s * rowscale[r] * x[r, :] in place (fp32)."""
    require_cuda(c, x, rowscale)
    R, D = c.shape
    check(_lib.load().pclip_addscaled_rows_f32(ptr(c), c.stride(0), ptr(x), x.stride(0), ptr(rowscale), s, R, D, stream()),
          "pclip_addscaled_rows_f32")
    return c


def nll_grad(d2i, d2t, labels, N: int, alpha: float, beta: float, q_total: int = None):
    """From the distance rows of sqdist_f32: gradients (gi, gt) of mean NLL(log P) wrt them, rowsum(gi + gt), the per-query
    -log p[y], max probability and argmax (utils.py:84-93)."""
    require_cuda(d2i, d2t, labels)
    Q, ldd = d2i.shape
    gi, gt = torch.empty_like(d2i), torch.empty_like(d2t)
    rs, nll, pmax = (torch.empty(Q, dtype=torch.float32, device=d2i.device) for _ in range(3))
    am = torch.empty(Q, dtype=torch.int32, device=d2i.device)
    lab = labels.to(torch.int32)
    check(_lib.load().pclip_nll_grad(ptr(d2i), ptr(d2t), ptr(lab), Q, Q if q_total is None else q_total, N, ldd, alpha, 1.0 - alpha, beta, ptr(gi), ptr(gt),
                                     ptr(rs), ptr(nll), ptr(pmax), ptr(am), stream()), "pclip_nll_grad")
    return gi, gt, rs, nll, pmax, am


def fuse_probs_backward(d2i, d2t, dp, N: int, alpha: float, beta: float):
    """Backward of `fuse_probs` for an arbitrary upstream dp [Q, N] fp32: (gi, gt) wrt the two distance rows (padded like them,
    columns >= N unspecified) and rowsum(gi + gt) — the autograd-transparent utils.P."""
    require_cuda(d2i, d2t, dp)
    Q, ldd = d2i.shape
    if dp.dtype != torch.float32 or dp.shape != (Q, N) or dp.stride(1) != 1:
        raise _lib.PclipError("fuse_probs_backward: dp must be a row-major fp32 [Q, N] tensor")
    gi, gt = torch.empty_like(d2i), torch.empty_like(d2t)
    rs = torch.empty(Q, dtype=torch.float32, device=d2i.device)
    a32, oma32 = float(np.float32(alpha)), float(np.float32(1 - float(alpha)))
    check(_lib.load().pclip_fuse_probs_backward(ptr(d2i), ptr(d2t), ptr(dp), dp.stride(0), Q, N, ldd, a32, oma32, float(np.float32(beta)),
                                                ptr(gi), ptr(gt), ptr(rs), stream()), "pclip_fuse_probs_backward")
    return gi, gt, rs


def nll_mean_backward(p: torch.Tensor, labels: torch.Tensor, g: torch.Tensor) -> torch.Tensor:
    """dp of nn.NLLLoss()(torch.log(p), labels) for the upstream gradient g (fp32 device scalar)."""
    require_cuda(p, labels, g)
    Q, N = p.shape
    dp = torch.empty(Q, N, dtype=torch.float32, device=p.device)
    g = g.reshape(1).float().contiguous()
    check(_lib.load().pclip_nll_mean_backward(ptr(p), p.stride(0), ptr(labels.to(torch.int32)), Q, N, ptr(g), ptr(dp), N, stream()),
          "pclip_nll_mean_backward")
    return dp


def nll_rows(p: torch.Tensor, labels: torch.Tensor):
    """(-log p[q, y_q], max_c p[q, c], argmax_c p[q, c]) per row of a materialised fp32 p [Q, N] (utils.py:84-93)."""
    require_cuda(p, labels)
    if p.dtype != torch.float32 or p.dim() != 2 or p.stride(1) != 1:
        raise _lib.PclipError("nll_rows: p must be a row-major fp32 [Q, N] tensor")
    Q, N = p.shape
    nll, pmax = (torch.empty(Q, dtype=torch.float32, device=p.device) for _ in range(2))
    am = torch.empty(Q, dtype=torch.int32, device=p.device)
    lab = labels.to(torch.int32)
    check(_lib.load().pclip_nll_rows(ptr(p), p.stride(0), ptr(lab), Q, N, ptr(nll), ptr(pmax), ptr(am), stream()), "pclip_nll_rows")
    return nll, pmax, am


def softmax_ce_rows(S: torch.Tensor, scale: float):
    """Rows of S as logits against the diagonal: (per-row loss, scale * (softmax - I))."""
    require_cuda(S)
    R, C = S.shape
    dS = torch.empty_like(S)
    loss = torch.empty(R, dtype=torch.float32, device=S.device)
    check(_lib.load().pclip_softmax_ce_rows(ptr(S), S.stride(0), R, C, scale, ptr(dS), dS.stride(0), ptr(loss), stream()),
          "pclip_softmax_ce_rows")
    return loss, dS


def proto_backward(mem: torch.Tensor, g: torch.Tensor, N: int, K: int, per_shot_norm: bool, final_norm: bool) -> torch.Tensor:
    """Gradient wrt the fp16 rows `mem` [N*K, D] of the prototype chain given g [N, D] fp32 wrt its fp32 output."""
    require_cuda(mem, g)
    mem = _f16c(mem)
    D = mem.shape[1]
    if g.dtype != torch.float32 or g.shape != (N, D) or not g.is_contiguous():
        raise _lib.PclipError("proto_backward: g must be a contiguous fp32 [N, D] tensor")
    dmem = torch.empty_like(mem)
    check(_lib.load().pclip_proto_backward_f16(ptr(mem), ptr(g), N, K, D, int(per_shot_norm), int(final_norm), ptr(dmem),
                                               stream()), "pclip_proto_backward_f16")
    return dmem


def layernorm_backward(x, gamma, dy, eps: float = 1e-5, dy_scale: float = 1.0):
    """fp16 LayerNorm backward over the last dimension: (dx fp16, dgamma fp32 [D], dbeta fp32 [D])."""
    require_cuda(x, gamma, dy)
    x, gamma, dy = _f16c(x), _f16c(gamma), _f16c(dy)
    R, D = x.shape
    nblk = max(1, min(256, (R + 3) // 4))
    dx = torch.empty_like(x)
    part = torch.empty(nblk, 2, D, dtype=torch.float32, device=x.device)
    check(_lib.load().pclip_layernorm_backward_f16(ptr(x), D, ptr(gamma), ptr(dy), D, R, D, eps, dy_scale, ptr(dx), D, ptr(part),
                                                   nblk, stream()), "pclip_layernorm_backward_f16")
    sums = colsum_f32(part.view(nblk, 2 * D))
    return dx, sums[:D], sums[D:]


def adamw_(p, g, m, v, lr: float, step: int, beta1: float = 0.9, beta2: float = 0.999, eps: float = 1e-4,
           weight_decay: float = 0.05):
    """In-place torch.optim.AdamW step on an fp16 parameter with fp16 moments (main.py:134-135)."""
    require_cuda(p, g, m, v)
    for t in (p, g, m, v):
        if t.dtype != torch.float16 or not t.is_contiguous():
            raise _lib.PclipError("adamw_: fp16 contiguous tensors expected")
    check(_lib.load().pclip_adamw_f16(ptr(p), ptr(g), ptr(m), ptr(v), p.numel(), lr, beta1, beta2, eps, weight_decay, step,
                                      stream()), "pclip_adamw_f16")
    return p


def l2norm_rows_f32(x: torch.Tensor, eps: float = 1e-12) -> torch.Tensor:
    """F.normalize(x, dim=-1) on fp32 rows."""
    require_cuda(x)
    R, D = x.shape
    y = torch.empty_like(x)
    check(_lib.load().pclip_l2norm_rows_f32(ptr(x), ptr(y), R, D, eps, stream()), "pclip_l2norm_rows_f32")
    return y


def l2norm_rows_backward_f32_(gx: torch.Tensor, x: torch.Tensor, gy: torch.Tensor, eps: float = 1e-12, accumulate: bool = True):
    """gx (+)= backward of F.normalize(x) for upstream gy (all fp32 [R, D])."""
    require_cuda(gx, x, gy)
    R, D = x.shape
    check(_lib.load().pclip_l2norm_rows_backward_f32(ptr(x), ptr(gy), ptr(gx), R, D, eps, int(accumulate), stream()),
          "pclip_l2norm_rows_backward_f32")
    return gx


def adapter_conv_backward(x, g, three_x: bool, conv1, ln1w, ln1b, conv2, ln2w, ln2b, conv3, ln3w, ln3b, chunk: int = 512):
    """Parameter gradients (fp32, shaped like the parameters) of the conv adapter for upstream g = dL/d(output); rows are
    processed `chunk` at a time (per-row contributions live in scratch, then a deterministic column sum)."""
    require_cuda(x, g)
    x, g = _f16c(x), _f16c(g)
    B, D = x.shape
    s = int(math.ceil(math.sqrt(D)))
    s2, n1 = s * s, 16 * s * s
    dev = x.device
    f32 = lambda *shape: torch.zeros(*shape, dtype=torch.float32, device=dev)
    out = {"conv1.weight": f32(16), "conv3.weight": f32(16), "bn1.weight": f32(n1), "bn1.bias": f32(n1), "bn3.weight": f32(s2),
           "bn3.bias": f32(s2)}
    if three_x:
        out.update({"conv2.weight": f32(2304), "bn2.weight": f32(n1), "bn2.bias": f32(n1)})
    lib = _lib.load()
    # rows of partial sums a launch over nb rows writes: the persistent MFMA kernel (conv-3x, D <= 576) one per workgroup — then the whole batch is
    # ONE launch + one column sum per parameter; the per-row kernels one per input row — then the batch is walked in chunks of `chunk` rows
    persistent = B > 0 and lib.pclip_adapter_conv_backward_partials(B, D, int(three_x)) < B
    c = max(B, 1) if persistent else min(chunk, max(B, 1))
    rows_of = lambda nb: lib.pclip_adapter_conv_backward_partials(nb, D, int(three_x))
    rmax = max(rows_of(c), 1)
    sc = lambda n: torch.empty(rmax, n, dtype=torch.float32, device=dev)
    pw1, pw3, pg1, pb1, pg3, pb3 = sc(16), sc(16), sc(n1), sc(n1), sc(s2), sc(s2)
    pw2, pg2, pb2 = (sc(2304), sc(n1), sc(n1)) if three_x else (None, None, None)
    for lo in range(0, B, c):
        nb = min(c, B - lo)
        check(lib.pclip_adapter_conv_backward_f16(
            ptr(x[lo:lo + nb]), ptr(g[lo:lo + nb]), nb, D, int(three_x), ptr(conv1), ptr(ln1w), ptr(ln1b),
            ptr(conv2) if three_x else None, ptr(ln2w) if three_x else None, ptr(ln2b) if three_x else None, ptr(conv3), ptr(ln3w),
            ptr(pw1), ptr(pw2), ptr(pw3), ptr(pg1), ptr(pb1), ptr(pg2), ptr(pb2), ptr(pg3), ptr(pb3), stream()),
            "pclip_adapter_conv_backward_f16")
        for name, part in (("conv1.weight", pw1), ("conv3.weight", pw3), ("bn1.weight", pg1), ("bn1.bias", pb1), ("bn3.weight", pg3),
                           ("bn3.bias", pb3), ("conv2.weight", pw2), ("bn2.weight", pg2), ("bn2.bias", pb2)):
            if part is not None:
                colsum_f32(part, rows=rows_of(nb), out=out[name])
    shapes = {"conv1.weight": (16, 1, 1, 1), "conv3.weight": (1, 16, 1, 1), "conv2.weight": (16, 16, 3, 3), "bn1.weight": (16, s, s),
              "bn1.bias": (16, s, s), "bn2.weight": (16, s, s), "bn2.bias": (16, s, s), "bn3.weight": (1, s, s), "bn3.bias": (1, s, s)}
    return {k: v.view(shapes[k]) for k, v in out.items()}
