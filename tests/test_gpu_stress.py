"""Race-stress build of the kernels with hand-counted waits: proto-clip_amd/libpclip_stress.so = the same sources under -DPCLIP_RACE_STRESS
(csrc/pclip_gemm.h stress_jitter: every counted s_waitcnt vmcnt(N), every LDS-only barrier and every LDS-DMA burst of the persistent kernels first pauses its wave
for 0 / 256 / 1024 cycles, pseudo-randomly per wave and call).  A wait that is a piece too weak or a barrier that does not cover a refill reads stale LDS in some
launch; here every such kernel must reproduce the NORMAL library's bits, repeatedly: the eight-wave GEMM in every tile configuration (staggered refill, ring
kernel, residual / QuickGELU epilogues), sqdist_big's norm strips, the fused row-panel classification, the one-launch mid-N classification (wave-private LDS-DMA
rings, counted per-wave waits), the narrow 3x3 convolution (ping-pong input blocks), attention in
every piece-count class (query-first and looping kernels, causal).  The four-wave asm loop has its own jittered variant (test_gpu_encoder.py)."""
import ctypes
import os

import pytest
import torch

from proto_clip_amd import _lib

pytestmark = pytest.mark.gpu
STRESS = os.path.join(os.path.dirname(_lib.LIB_PATH), "libpclip_stress.so")


@pytest.fixture(scope="module")
def ops():
    from proto_clip_amd import ops as _ops
    _lib.load()
    return _ops


@pytest.fixture(scope="module")
def slib():
    if not os.path.exists(STRESS):
        pytest.fail(f"{STRESS} is missing: `make -C proto-clip_amd/csrc` builds it beside libpclip.so")
    os.environ["PCLIP_GEMM_CFG_LIVE"] = "1"                 # the stress library re-reads PCLIP_GEMM_CFG per call (read at ITS first call)
    lib = ctypes.CDLL(STRESS)
    for name, argtypes in _lib._SIGS.items():
        fn = getattr(lib, name)
        fn.argtypes = argtypes
        fn.restype = _lib._RESTYPES.get(name, ctypes.c_int)
    lib.pclip_gemm4w_config(0)                              # 256 x 256 tiles on the EIGHT-wave kernel: the one whose waits are under test here
    yield lib
    os.environ.pop("PCLIP_GEMM_CFG", None)


def _gemm(lib, a, w, bias, act, res, out):
    M, K = a.shape
    rc = lib.pclip_gemm_f16(_lib.ptr(a), a.stride(0), _lib.ptr(w), w.stride(0), _lib.ptr(out), out.stride(0), M, w.shape[0], K, _lib.ptr(bias), act, _lib.ptr(res), _lib.stream())
    assert rc == 0, lib.pclip_last_error()
    return out


@pytest.mark.parametrize("cfg", ["0", "1", "2", "4", ""])
@pytest.mark.parametrize("M,N,K", [(30000, 768, 768), (9000, 2304, 768), (4000, 768, 3072), (197, 768, 768)])
def test_gemm_under_jitter(ops, slib, cfg, M, N, K):
    g = torch.Generator(device="cuda").manual_seed(M + N + K)
    a = (torch.randn(M, K, device="cuda", generator=g) * 0.5).half()
    w = (torch.randn(N, K, device="cuda", generator=g) * K ** -0.5).half()
    bias = torch.randn(N, device="cuda", generator=g).half()
    res = torch.randn(M, N, device="cuda", generator=g).half()
    if cfg: os.environ["PCLIP_GEMM_CFG"] = cfg
    else: os.environ.pop("PCLIP_GEMM_CFG", None)
    try:
        for act, b, r in ((0, bias, None), (1, bias, None), (0, bias, res)):
            ref = ops.gemm(a, w, b, act, r)
            for _ in range(3):
                out = torch.full((M, N), 7.0, device="cuda", dtype=torch.float16)
                assert torch.equal(_gemm(slib, a, w, b, act, r, out), ref), (cfg, act)
    finally:
        os.environ.pop("PCLIP_GEMM_CFG", None)


def test_sqdist_big_and_fused_classify_under_jitter(ops, slib):
    Q, N, D = 26000, 1000, 512
    nrm = torch.nn.functional.normalize
    g = torch.Generator(device="cuda").manual_seed(9)
    zi = nrm(torch.randn(N, D, device="cuda", generator=g), dim=-1).half()
    zt = nrm(torch.randn(N, D, device="cuda", generator=g), dim=-1).half()
    q = nrm(torch.randn(Q, D, device="cuda", generator=g), dim=-1).half()
    d2i, d2t, _ = ops.sqdist(q, zi, zt)
    ldd = ops.padded_ld(N)
    ws = torch.empty(_lib.workspace_bytes(_lib.OP_CLASSIFY, Q, N, D), dtype=torch.uint8, device="cuda")
    with ops.classify_fused():
        _, am_ref, _, _ = ops.classify(q, zi, zt, 0.5, 12.0, want_p=False, want_argmax=True)
    slib.pclip_classify_panel_config(2)
    for _ in range(3):
        a = torch.full((Q, ldd), float("nan"), device="cuda")
        b = torch.full((Q, ldd), float("nan"), device="cuda")
        rc = slib.pclip_sqdist_f16(_lib.ptr(q), _lib.ptr(zi), _lib.ptr(zt), Q, N, D, None, None, None, _lib.ptr(a), _lib.ptr(b), ldd, _lib.ptr(ws), ws.numel(), _lib.stream())
        assert rc == 0, slib.pclip_last_error()
        assert torch.equal(a[:, :N], d2i[:, :N]) and torch.equal(b[:, :N], d2t[:, :N])
        am = torch.full((Q,), -1, dtype=torch.int32, device="cuda")
        rc = slib.pclip_classify_f16(_lib.ptr(q), _lib.ptr(zi), _lib.ptr(zt), Q, N, D, None, None, None, 0.5, 0.5, 12.0, None, _lib.ptr(am), None, None, 0, _lib.ptr(ws),
                                     ws.numel(), _lib.stream())
        assert rc == 0, slib.pclip_last_error()
        assert torch.equal(am, am_ref)


@pytest.mark.parametrize("Q,N,D", [(2465, 100, 1024), (666, 198, 768), (5000, 64, 512), (9000, 37, 512), (300, 256, 1024), (40, 129, 768)])
def test_classify_mid_under_jitter(ops, slib, Q, N, D):
    """The one-launch mid-N classification streams its bank rows by LDS-DMA into wave-private rings and waits with hand-counted vmcnt (csrc/pclip_classify_mid.hip):
    in the stress build every counted wait and LDS barrier first pauses its wave at random — p and argmax must be the normal library's bits, launch after launch,
    for the eight- and the sixteen-wave form, one and several query groups per workgroup."""
    g = torch.Generator(device="cuda").manual_seed(Q + N + D)
    nrm = torch.nn.functional.normalize
    zi = nrm(torch.randn(N, D, device="cuda", generator=g), dim=-1).half()
    zt = nrm(torch.randn(N, D, device="cuda", generator=g), dim=-1).half()
    q = nrm(torch.randn(Q, D, device="cuda", generator=g) + 2 * zi[torch.arange(Q, device="cuda") % N].float(), dim=-1).half()
    with ops.classify_mid(2):
        p_ref, am_ref, _, _ = ops.classify(q, zi, zt, 0.5, 12.0, want_p=True, want_argmax=True)
    ws = torch.empty(_lib.workspace_bytes(_lib.OP_CLASSIFY, Q, N, D), dtype=torch.uint8, device="cuda")
    slib.pclip_classify_mid_config(2)
    for rep in range(4):
        p = torch.full((Q, N), float("nan"), device="cuda")
        am = torch.full((Q,), -1, dtype=torch.int32, device="cuda")
        rc = slib.pclip_classify_f16(_lib.ptr(q), _lib.ptr(zi), _lib.ptr(zt), Q, N, D, None, None, None, 0.5, 0.5, 12.0, _lib.ptr(p), _lib.ptr(am), None, None, 0,
                                     _lib.ptr(ws), ws.numel(), _lib.stream())
        assert rc == 0, slib.pclip_last_error()
        assert torch.equal(p, p_ref) and torch.equal(am, am_ref), f"launch {rep}"


@pytest.mark.parametrize("B,H,W,Cin,Cout", [(300, 56, 56, 64, 64), (40, 112, 112, 32, 64), (40, 112, 112, 32, 32), (50, 8, 56, 64, 32)])
def test_conv_strip_under_jitter(ops, slib, B, H, W, Cin, Cout):
    """The narrow 3x3 convolution kernel (csrc/pclip_conv_strip.hip) ping-pongs two LDS input blocks filled by LDS-DMA: one vmcnt(0) + LDS barrier per tile covers
    both the landing of this tile's block and the hand-back of the other.  With waves paused at random around them, launch after launch, the normal library's bits."""
    g = torch.Generator(device="cuda").manual_seed(B + H + Cout)
    x = (torch.randn(B * H * W, Cin, device="cuda", generator=g) * 0.7).half()
    w = (torch.randn(Cout, (9 * Cin + 63) // 64 * 64, device="cuda", generator=g) * (9 * Cin) ** -0.5).half()
    sc, sh = 1 + 0.3 * torch.randn(Cout, device="cuda", generator=g), 0.2 * torch.randn(Cout, device="cuda", generator=g)
    with ops.conv_strip(True):
        assert ops.conv_strip_applies(B, H, W, Cin, Cout)
        ref = ops.conv3x3_bn(x, w, sc, sh, B, H, W, Cin, relu=True)
    z = torch.zeros(64, dtype=torch.float16, device="cuda")
    slib.pclip_conv3x3_strip_config(1)
    for rep in range(3):
        y = torch.full((B * H * W, Cout), float("nan"), dtype=torch.float16, device="cuda")
        rc = slib.pclip_conv3x3_bn_f16(_lib.ptr(x), _lib.ptr(w), _lib.ptr(z), B, H, W, Cin, Cout, _lib.ptr(sc), _lib.ptr(sh), 1, _lib.ptr(y), _lib.stream())
        assert rc == 0, slib.pclip_last_error()
        assert torch.equal(y, ref), f"launch {rep}"


@pytest.mark.parametrize("B,L,H,causal", [(64, 197, 12, False), (32, 257, 16, False), (128, 50, 12, False), (256, 77, 8, True), (16, 129, 12, False), (16, 224, 12, True),
                                          (8, 280, 4, False)])
def test_attention_under_jitter(ops, slib, B, L, H, causal):
    g = torch.Generator(device="cuda").manual_seed(B + L)
    qkv = torch.randn(B * L, 3 * H * 64, device="cuda", generator=g).half()
    ref = ops.attention(qkv, B, L, H, causal)
    for _ in range(3):
        out = torch.full((B * L, H * 64), float("nan"), dtype=torch.float16, device="cuda")
        rc = slib.pclip_attention_f16(_lib.ptr(qkv), _lib.ptr(out), B, L, H, 64, int(causal), _lib.stream())
        assert rc == 0, slib.pclip_last_error()
        assert torch.equal(out, ref)


def test_jitter_is_active(ops, slib):
    """The stress library really pauses waves: the same GEMM takes measurably longer through it (otherwise the tests above prove nothing)."""
    a = torch.randn(60000, 768, device="cuda").half()
    w = (torch.randn(768, 768, device="cuda") * 0.03).half()
    out = torch.empty(60000, 768, device="cuda", dtype=torch.float16)

    def t(fn):
        fn(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        return e0.elapsed_time(e1)

    with ops.gemm_eight_wave():
        t_norm = t(lambda: ops.gemm(a, w, None, 0, None, out))
    t_stress = t(lambda: _gemm(slib, a, w, None, 0, None, out))
    print(f"\n[observed] eight-wave GEMM 60000x768x768: normal {t_norm / 10 * 1e3:.1f} us, stress build {t_stress / 10 * 1e3:.1f} us")
    assert t_stress > 1.15 * t_norm
