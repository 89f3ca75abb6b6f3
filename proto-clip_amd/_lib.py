"""ctypes binding of libpclip.so (include/pclip.h).  There is no fallback: if the library is missing
or a call fails, a loud exception is raised — the product path never routes through a CPU or
PyTorch-eager substitute."""
import ctypes
import os
from ctypes import c_char_p, c_double, c_float, c_int, c_long, c_size_t, c_void_p

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "libpclip.so")

OP_SQDIST, OP_CLASSIFY, OP_ADAPTER_FC = 1, 2, 3


class PclipError(RuntimeError):
    pass


_lib = None

# name -> argtypes (restype is int unless listed in _RESTYPES); mirrors include/pclip.h one to one
_P = c_void_p
_SIGS = {
    "pclip_abi_version": [],
    "pclip_last_error": [],
    "pclip_device_cus": [],
    "pclip_gemm_kernel_launches": [],
    "pclip_l2norm_rows_f16": [_P, _P, c_int, c_int, _P, _P],
    "pclip_row_sqnorm_f16": [_P, c_int, c_int, _P, _P],
    "pclip_proto_build_f16": [_P, c_int, c_int, c_int, c_int, _P, _P, _P, _P],
    "pclip_bank_reduce_f16": [_P, c_int, c_int, c_int, _P, _P, _P],
    "pclip_transpose_f16": [_P, c_int, c_int, _P, _P],
    "pclip_partial_sums_f16": [_P, _P, c_int, c_int, c_int, c_int, _P, _P, _P],
    "pclip_proto_finalize": [_P, _P, c_int, c_int, c_int, _P, _P, _P, _P],
    "pclip_sqdist_f16": [_P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, _P, c_int, _P, c_size_t, _P],
    "pclip_sqdist_f32": [_P, _P, _P, c_int, c_int, c_int, _P, _P, c_int, _P],
    "pclip_fuse_probs": [_P, _P, c_int, c_int, c_int, c_float, c_float, c_float, _P, _P, _P, _P, c_int, _P],
    "pclip_classify_f16": [_P, _P, _P, c_int, c_int, c_int, _P, _P, _P, c_float, c_float, c_float, _P, _P, _P,
                           _P, c_int, _P, c_size_t, _P],
    "pclip_classify_panel_config": [c_int],
    "pclip_classify_panel_passes": [c_int],
    "pclip_classify_panel_stats": [_P, c_int],
    "pclip_classify_panel_dump_f16": [_P, _P, _P, c_int, c_int, c_int, _P, c_int, _P, c_size_t, _P],
    "pclip_gemm_timing": [_P, c_int],
    "pclip_gemm_timing_count": [],
    "pclip_classify_route": [c_int, c_int, c_int, c_float, c_float, c_float, c_int, c_int, c_int, c_int, c_size_t],
    "pclip_classify_mid_config": [c_int],
    "pclip_hp_sweep": [_P, _P, _P, c_int, c_int, c_int, _P, _P, c_int, _P, c_int, _P, _P],
    "pclip_adapter_fc_f16": [_P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, c_float, c_float, c_int, _P, _P, _P,
                             c_size_t, _P],
    "pclip_layernorm_blend_f16": [_P, _P, _P, c_float, _P, c_float, c_float, c_int, _P, _P, c_int, c_int, _P],
    "pclip_adapter_conv_f16": [_P, c_int, c_int, c_int, _P, _P, _P, _P, _P, _P, _P, _P, _P, c_int, _P, _P, _P],
    "pclip_gemm_f16": [_P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, _P],
    "pclip_gemm4w_f16": [_P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, _P],
    "pclip_gemm4w_var_f16": [_P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, c_int, _P],
    "pclip_gemm4w_stamp_buffer": [_P],
    "pclip_gemm4w_config": [c_int],
    "pclip_gemm_splitk_workspace": [c_int, c_int, c_int],
    "pclip_gemm_splitk_f16": [_P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, c_int, _P, c_size_t, _P],
    "pclip_gemm_bn_f16": [_P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, _P, c_int, _P],
    "pclip_gemm_bn_res_f16": [_P, c_int, _P, c_int, _P, c_int, c_int, c_int, c_int, _P, _P, _P, _P],
    "pclip_conv3x3_bn_f16": [_P, _P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, c_int, _P, _P],
    "pclip_conv3x3_strip_applies": [c_int, c_int, c_int, c_int, c_int],
    "pclip_conv3x3_strip_config": [c_int],
    "pclip_conv3x3_pool_applies": [c_int, c_int, c_int, c_int],
    "pclip_conv3x3_bn_pool_f16": [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P, _P, _P],
    "pclip_stem_conv_applies": [c_int, c_int],
    "pclip_stem_conv_bn_f16": [_P, c_int, c_int, c_int, _P, c_int, _P, _P, c_int, _P, _P],
    "pclip_layernorm_f16": [_P, c_int, _P, _P, c_float, _P, c_int, c_int, _P],
    "pclip_add_layernorm_f16": [_P, _P, c_int, _P, _P, _P, c_float, _P, c_int, c_int, _P],
    "pclip_attention_f16": [_P, _P, c_int, c_int, c_int, c_int, c_int, _P],
    "pclip_attention_q_f16": [_P, c_int, c_long, _P, c_int, c_int, c_int, _P, c_int, c_int, c_int, c_int, c_int, c_int, _P],
    "pclip_attention_config": [c_int, c_int],
    "pclip_im2col_patches_f16": [_P, c_int, c_int, c_int, _P, c_int, _P],
    "pclip_im2col_patches_f32": [_P, c_int, c_int, c_int, _P, c_int, _P],
    "pclip_vit_assemble_tokens_f16": [_P, _P, _P, c_int, c_int, c_int, _P, _P],
    "pclip_vit_embed_ln_f16": [_P, _P, _P, c_int, c_int, c_int, _P, _P, _P, _P, c_float, _P, _P, _P],
    "pclip_text_embed_f16": [_P, _P, _P, c_int, c_int, c_int, c_int, _P, _P],
    "pclip_gather_eot_f16": [_P, _P, c_int, c_int, c_int, _P, _P],
    "pclip_im2col3x3_f16": [_P, c_long, c_long, c_long, c_long, c_int, c_int, c_int, c_int, c_int, _P, c_int, _P],
    "pclip_bn_act_f16": [_P, _P, _P, _P, c_int, _P, c_size_t, c_int, _P],
    "pclip_avgpool_nhwc_f16": [_P, c_int, c_int, c_int, c_int, c_int, _P, _P],
    "pclip_attnpool_tokens_f16": [_P, _P, c_int, c_int, c_int, _P, _P],
    "pclip_cast_f32_f16": [_P, _P, c_size_t, _P],
    "pclip_cast_f16_f32": [_P, _P, c_size_t, _P],
    "pclip_gemm_f32": [_P, c_int, c_long, c_long, _P, c_int, c_long, c_long, _P, c_int, c_int, c_int, c_int, c_float, c_float, _P, c_size_t, _P],
    "pclip_colsum_f32": [_P, c_int, c_int, c_int, c_float, _P, c_int, _P, c_size_t, _P],
    "pclip_adapter_conv_backward_f16": [_P, _P, c_int, c_int, c_int] + [_P] * 8 + [_P] * 9 + [_P],
    "pclip_adapter_conv_backward_partials": [c_int, c_int, c_int],
    "pclip_addscaled_rows_f32": [_P, c_int, _P, c_int, _P, c_float, c_int, c_int, _P],
    "pclip_nll_grad": [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, _P, _P, _P, _P, _P, _P, _P],
    "pclip_nll_rows": [_P, c_int, _P, c_int, c_int, _P, _P, _P, _P],
    "pclip_fuse_probs_backward": [_P, _P, _P, c_int, c_int, c_int, c_int, c_float, c_float, c_float, _P, _P, _P, _P],
    "pclip_nll_mean_backward": [_P, c_int, _P, c_int, c_int, _P, _P, c_int, _P],
    "pclip_softmax_ce_rows": [_P, c_int, c_int, c_int, c_float, _P, c_int, _P, _P],
    "pclip_l2norm_rows_f32": [_P, _P, c_int, c_int, c_float, _P],
    "pclip_l2norm_rows_backward_f32": [_P, _P, _P, c_int, c_int, c_float, c_int, _P],
    "pclip_proto_backward_f16": [_P, _P, c_int, c_int, c_int, c_int, c_int, _P, _P],
    "pclip_layernorm_backward_f16": [_P, c_int, _P, _P, c_int, c_int, c_int, c_float, c_float, _P, c_int, _P, c_int, _P],
    "pclip_adamw_f16": [_P, _P, _P, _P, c_size_t, c_double, c_double, c_double, c_double, c_double, c_int, _P],
    "pclip_preprocess_u8": [_P, _P, c_int, c_int, c_int, c_float, c_float, c_float, c_float, c_float, c_float, _P, c_int, _P, _P],
    "pclip_workspace_bytes": [c_int, c_int, c_int, c_int],
}
_RESTYPES = {"pclip_last_error": c_char_p, "pclip_workspace_bytes": c_size_t, "pclip_gemm_splitk_workspace": c_size_t, "pclip_gemm_kernel_launches": c_long}
EXPORTED_SYMBOLS = tuple(_SIGS)


def load():
    """Load libpclip.so once; raise PclipError loudly if it is not built (run __graft_entry__.build())."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise PclipError(
            f"{LIB_PATH} is missing: the HIP extension is not built. Run `python -c 'import __graft_entry__ as g; "
            "g.build()'` (or `make -C proto-clip_amd/csrc`). There is no CPU/eager fallback for this path.")
    lib = ctypes.CDLL(LIB_PATH)
    for name, argtypes in _SIGS.items():
        fn = getattr(lib, name)     # AttributeError if a declared symbol is not exported
        fn.argtypes = argtypes
        fn.restype = _RESTYPES.get(name, c_int)
    if lib.pclip_abi_version() != 1:
        raise PclipError(f"libpclip ABI version {lib.pclip_abi_version()} != 1")
    _lib = lib
    return lib


def check(rc: int, what: str):
    if rc != 0:
        msg = load().pclip_last_error()
        raise PclipError(f"{what} failed (rc={rc}): {msg.decode() if msg else '?'}")


def ptr(t):
    """Device pointer of a tensor (None -> NULL)."""
    return None if t is None else c_void_p(t.data_ptr())


def stream():
    """The current stream of the CURRENT device; `require_cuda` has checked that the operands live there."""
    return c_void_p(torch.cuda.current_stream().cuda_stream)


def require_cuda(*tensors):
    """Operands must be device tensors, all on one GPU, and that GPU must be the current device: the kernels are launched on the
    current device's stream (one process per GPU is the deployment model; a bank on another GPU would otherwise be a silent peer
    access or a fault).  `with torch.cuda.device(x.device):` around the call is the way to drive a second GPU from one process."""
    dev = None
    for t in tensors:
        if t is None:
            continue
        if not t.is_cuda:
            raise PclipError("libpclip operates on device tensors only (got a CPU tensor); there is no CPU fallback")
        if dev is None:
            dev = t.device
        elif t.device != dev:
            raise PclipError(f"operands on different GPUs ({dev} and {t.device}): move them to one device")
    if dev is not None and dev.index != torch.cuda.current_device():
        raise PclipError(f"operands live on {dev} but the current device is cuda:{torch.cuda.current_device()}: call "
                         f"torch.cuda.set_device({dev.index}) or wrap the call in `with torch.cuda.device({dev.index}):`")


def workspace_bytes(op: int, Q: int, N: int, D: int) -> int:
    return int(load().pclip_workspace_bytes(op, Q, N, D))
