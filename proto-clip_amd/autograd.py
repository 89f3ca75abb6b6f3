"""torch.autograd.Function wrappers that make the reference's OWN training loop run on the HIP kernels with only the import swap
(reference main.py:260-310, main.qt.py:198-250):

    zs_imgs = visual_embeddings.weight.view(-1, K, ndim); ... z_img_proto = ...            # the caller's eager tensors
    zq_imgs = adapter(zq_imgs).float()                                                      # Adapter / Adapter_FC  -> here
    p = P(zq_imgs, z_img_proto, z_text_proto, alpha, beta)                                  # utils.P               -> here
    ... = compute_loss_and_matches(p, zq_labels, z_img_proto, z_text_proto, cfg)            # NLL / InfoNCE         -> here
    optimizer.zero_grad(); train_loss.backward(retain_graph=True); optimizer.step()         # torch autograd + torch.optim.AdamW

Every forward is the inference kernel of that stage, every backward the explicit backward kernel `train.ProtoClipTrainer` uses
(csrc/pclip_train.hip, pclip_adapter.hip) — the trainer stays the fast path (no tape, fused NLL + P, one flat all-reduce); this
module is the drop-in contract of SURVEY 8(b): "nn.Module semantics".  Gradients are returned in the dtype of the input they
belong to (fp16 parameters get fp16 gradients, as autograd on the reference's fp16 modules delivers them).

Not differentiated: the INPUT of the conv adapter (the reference feeds it constants: rows of the frozen key bank, main.py:265-266,
or encode_image outputs under no_grad, main.qt.py:199-201) — asking for it raises."""
import torch

from . import ops
from ._lib import PclipError

INFO_NCE_TEMPERATURE = 0.1      # info-nce-pytorch default


def _like(g32: torch.Tensor, ref: torch.Tensor) -> torch.Tensor:
    """fp32 gradient -> dtype / shape of the tensor it belongs to."""
    if g32 is None:
        return None
    g = g32.reshape(ref.shape)
    return ops.cast_f16(g.contiguous()) if ref.dtype == torch.float16 else g


class AdapterConvFn(torch.autograd.Function):
    """Adapter.forward (model.py:49-78); backward = pclip_adapter_conv_backward_f16 (forward recomputed in LDS)."""

    @staticmethod
    def forward(ctx, x, three_x, conv1, ln1w, ln1b, conv2, ln2w, ln2b, conv3, ln3w, ln3b):
        ctx.three_x = bool(three_x)
        ctx.save_for_backward(x, conv1, ln1w, ln1b, conv2, ln2w, ln2b, conv3, ln3w, ln3b)
        return ops.adapter_conv(x, three_x, conv1, ln1w, ln1b, conv2, ln2w, ln2b, conv3, ln3w, ln3b)

    @staticmethod
    def backward(ctx, g):
        if ctx.needs_input_grad[0]:
            raise NotImplementedError("the conv adapter's INPUT gradient is not on the Proto-CLIP path (its input rows are constants, "
                                      "main.py:265-266 / main.qt.py:199-201); detach the features")
        x, conv1, ln1w, ln1b, conv2, ln2w, ln2b, conv3, ln3w, ln3b = ctx.saved_tensors
        gr = ops.adapter_conv_backward(x, g.contiguous(), ctx.three_x, conv1, ln1w, ln1b, conv2, ln2w, ln2b, conv3, ln3w, ln3b)
        f = lambda name, ref: _like(gr.get(name), ref)
        # conv-2x never touches conv2 / bn2: their gradient stays None, as in the reference
        return (None, None, f("conv1.weight", conv1), f("bn1.weight", ln1w), f("bn1.bias", ln1b), f("conv2.weight", conv2),
                f("bn2.weight", ln2w), f("bn2.bias", ln2b), f("conv3.weight", conv3), f("bn3.weight", ln3w), f("bn3.bias", ln3b))


class AdapterFcFn(torch.autograd.Function):
    """Adapter_FC.forward (model.py:81-95) stage by stage (the saved activations are exactly the tensors the output was formed
    from) and its backward: two LayerNorm backwards + fp32 MFMA GEMMs for the weight / activation gradients."""

    @staticmethod
    def forward(ctx, x, w1, g1, b1, w2, g2, b2):
        with ops.low_latency(False):
            h1 = ops.gemm(x, w1)
            a1 = ops.layernorm(h1, g1.float(), b1.float())
            h2 = ops.gemm(a1, w2)
        out = ops.layernorm_blend(h2, g2, b2, x, ratio=0.2)
        ctx.save_for_backward(x, h1, a1, h2, w1, g1, w2, g2)
        return out

    @staticmethod
    def backward(ctx, g):
        x, h1, a1, h2, w1, g1, w2, g2 = ctx.saved_tensors
        g = g.contiguous()
        dh2, dg2, db2 = ops.layernorm_backward(h2, g2, g, dy_scale=0.2)             # ratio * LN(h2) (model.py:93-94)
        dw2 = ops.gemm_f32(dh2, a1, trans_a=True)
        da1 = ops.cast_f16(ops.gemm_f32(dh2, w2))
        dh1, dg1, db1 = ops.layernorm_backward(h1, g1, da1)
        dw1 = ops.gemm_f32(dh1, x, trans_a=True)
        dx = None
        if ctx.needs_input_grad[0]:                                                 # 0.8 * g through the blend + the first Linear
            dx32 = ops.gemm_f32(dh1, w1)
            dx32 = dx32 + 0.8 * g.float()
            dx = _like(dx32, x)
        return dx, _like(dw1, w1), _like(dg1, g1), _like(db1, g1), _like(dw2, w2), _like(dg2, g2), _like(db2, g2)


class PFn(torch.autograd.Function):
    """utils.P (utils.py:225-244) on fp32 operands: exact-fp32 MFMA distances + softmax fusion; backward through both softmaxes
    (pclip_fuse_probs_backward) and cdist(...)**2:  dq = sum_c 2 G[q,c] (q - z_c),  dz_c = sum_q 2 G[q,c] (z_c - q)."""

    @staticmethod
    def forward(ctx, zq, zi, zt, alpha, beta):
        q32, i32, t32 = (t.float().contiguous() for t in (zq, zi, zt))
        d2i, d2t, _ = ops.sqdist_f32(q32, i32, t32)
        N = i32.shape[0]
        p, _, _, _ = ops.fuse_probs(d2i, d2t, N, alpha, beta, want_p=True)
        ctx.alpha, ctx.beta, ctx.N = float(alpha), float(beta), N
        ctx.dtypes = (zq.dtype, zi.dtype, zt.dtype)
        ctx.save_for_backward(q32, i32, t32, d2i, d2t)
        return p

    @staticmethod
    def backward(ctx, dp):
        q32, i32, t32, d2i, d2t = ctx.saved_tensors
        N = ctx.N
        gi, gt, rs = ops.fuse_probs_backward(d2i, d2t, dp.float().contiguous(), N, ctx.alpha, ctx.beta)
        gi_v, gt_v = gi[:, :N], gt[:, :N]
        gq = gimg = gtxt = None
        if ctx.needs_input_grad[0]:
            gq = ops.gemm_f32(gi_v, i32, alpha=-2.0)
            ops.gemm_f32(gt_v, t32, alpha=-2.0, out=gq, beta=1.0)
            ops.addscaled_rows_(gq, q32, rs, 2.0)
        if ctx.needs_input_grad[1]:
            gimg = ops.gemm_f32(gi_v, q32, trans_a=True, alpha=-2.0)
            ops.addscaled_rows_(gimg, i32, ops.colsum_f32(gi, cols=N), 2.0)
        if ctx.needs_input_grad[2]:
            gtxt = ops.gemm_f32(gt_v, q32, trans_a=True, alpha=-2.0)
            ops.addscaled_rows_(gtxt, t32, ops.colsum_f32(gt, cols=N), 2.0)
        cast = lambda g, dt: None if g is None else (ops.cast_f16(g) if dt == torch.float16 else g)
        return cast(gq, ctx.dtypes[0]), cast(gimg, ctx.dtypes[1]), cast(gtxt, ctx.dtypes[2]), None, None


class NllMeanFn(torch.autograd.Function):
    """nn.NLLLoss()(torch.log(p), target) (utils.py:90-93): mean over the rows of -log p[q, y_q]."""

    @staticmethod
    def forward(ctx, p, target):
        if p.dtype != torch.float32 or not p.is_contiguous():
            p = p.float().contiguous()
        nll, _, _ = ops.nll_rows(p, target)
        ctx.save_for_backward(p, target)
        return ops.colsum_f32(nll.view(-1, 1), scale=1.0 / p.shape[0])[0]

    @staticmethod
    def backward(ctx, g):
        p, target = ctx.saved_tensors
        return ops.nll_mean_backward(p, target, g), None


class InfoNceFn(torch.autograd.Function):
    """InfoNCE()(A, B) of utils.py:72-77 (info-nce-pytorch defaults: temperature 0.1, both sides normalised, cross entropy against
    the diagonal, mean) with its gradient wrt both operands."""

    @staticmethod
    def forward(ctx, A, B):
        a, b = A.float().contiguous(), B.float().contiguous()
        an, bn = ops.l2norm_rows_f32(a), ops.l2norm_rows_f32(b)
        n = a.shape[0]
        S = ops.gemm_f32(an, bn, trans_b=True, alpha=1.0 / INFO_NCE_TEMPERATURE)
        rows, dS = ops.softmax_ce_rows(S, 1.0 / n)
        ctx.save_for_backward(a, b, an, bn, dS)
        ctx.dtypes = (A.dtype, B.dtype)
        return ops.colsum_f32(rows.view(n, 1), scale=1.0 / n)[0]

    @staticmethod
    def backward(ctx, g):
        a, b, an, bn, dS = ctx.saved_tensors
        scale = 1.0 / INFO_NCE_TEMPERATURE
        ga = gb = None
        if ctx.needs_input_grad[0]:
            dan = ops.gemm_f32(dS, bn, alpha=scale)
            ga = ops.l2norm_rows_backward_f32_(torch.empty_like(a), a, dan, accumulate=False) * g
        if ctx.needs_input_grad[1]:
            dbn = ops.gemm_f32(dS, an, trans_a=True, alpha=scale)
            gb = ops.l2norm_rows_backward_f32_(torch.empty_like(b), b, dbn, accumulate=False) * g
        cast = lambda t, dt: None if t is None else (ops.cast_f16(t.contiguous()) if dt == torch.float16 else t)
        return cast(ga, ctx.dtypes[0]), cast(gb, ctx.dtypes[1])


def wants_grad(*tensors) -> bool:
    return torch.is_grad_enabled() and any(t is not None and t.requires_grad for t in tensors)
