/* pclip.h — C ABI of libpclip.so, the MI355X (gfx950) hot path of Proto-CLIP.
 *
 * The reference (IRVLUTD/Proto-CLIP) is pure Python and has no FFI of its own; each
 * entry point below replaces the eager-PyTorch arithmetic of one reference function, cited as
 * file:line relative to the reference tree.  The host layer (the Python modules under proto-clip_amd/) mirrors the
 * reference's Python signatures and calls these through ctypes (binding shown in INTEGRATION.md).
 *
 * Conventions
 *  - every pointer is a DEVICE pointer owned by the caller (torch `data_ptr()`); fp16 buffers are
 *    `const void*` (IEEE binary16, row-major, innermost dimension contiguous);
 *  - work is enqueued on the hipStream_t passed as `stream` (void*; NULL = default stream); no entry
 *    point synchronises, allocates or frees; scratch comes from the caller (`ws`, sized by
 *    pclip_workspace_bytes) so every call is hipGraph-capturable and thread-safe per stream;
 *  - return value: 0 = ok, <0 = error (PCLIP_E_*), text via pclip_last_error() (thread-local);
 *  - no torch types, no C++ types, no exceptions cross this boundary.
 */
#ifndef PCLIP_H
#define PCLIP_H
#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define PCLIP_ABI_VERSION 1
#define PCLIP_OK 0
#define PCLIP_E_INVALID (-1)   /* bad shape / null pointer / unsupported size */
#define PCLIP_E_LAUNCH (-2)    /* hipLaunch / hipGetLastError failure */
#define PCLIP_E_WORKSPACE (-3) /* workspace too small */

typedef void* pclip_stream_t;

int pclip_abi_version(void);
const char* pclip_last_error(void);
/* number of compute units of the current device (used by host code to size batches) */
int pclip_device_cus(void);
/* Measurement hook of the GEMM kernels (bench.py's roofline): buf = [nslots][2] unsigned 64-bit of device memory, begin words pre-set to ~0 and end words to 0
 * by the caller; every GEMM kernel launch made while the buffer is set takes the next slot and its workgroups fold the device's constant 100 MHz counter
 * (s_memrealtime) into it — slot[0] = first instruction of the launch, slot[1] = its last: the span rocprofv3 reports as the kernel's duration, taken inside the
 * running step without a launch or event between the kernels.  pclip_gemm_timing(NULL, 0) switches it off (the product); pclip_gemm_timing_count() = slots taken. */
int pclip_gemm_timing(void* buf, int nslots);
int pclip_gemm_timing_count(void);
/* cumulative number of GEMM KERNEL launches issued by pclip_gemm_f16 in this process (one call may be split into
 * two launches, see the dispatch in csrc/pclip_encoder.hip); bench.py uses the delta to quote per-launch figures */
long pclip_gemm_kernel_launches(void);

/* ---- memory-bank / prototype reductions -------------------------------------------------- */

/* In-place-capable row L2 normalise: y = r16(x / r16(||x||)).  utils.py:352, main.py:182-185,
 * 408-409.  sq_out (nullable) receives the fp32 squared norm of the *output* row. */
int pclip_l2norm_rows_f16(const void* x, void* y, int R, int D, float* sq_out, pclip_stream_t stream);

/* fp32 squared norms of fp16 rows (the ||.||^2 terms torch.cdist adds, utils.py:230-233). */
int pclip_row_sqnorm_f16(const void* x, int R, int D, float* sq_out, pclip_stream_t stream);

/* Prototype reduction, main.py:399-402 (eval), 260-264 (train), 173-176 (zero-shot init):
 * mem [N*K, D] fp16 (class c owns rows c*K..c*K+K-1) -> proto [N, D].
 *   per_shot_norm=1: zs = r16(m / r16||m||) first (eval/train); 0: skip (zero-shot init).
 *   proto_f16 (nullable): r16(z / r16||z||) with z = r16(mean_K zs)      — eval path
 *   proto_f32 (nullable): fp32 z / ||z||  (no rounding)                     — train path
 *   proto_sq  (nullable): fp32 ||proto_f16||^2 per class. */
int pclip_proto_build_f16(const void* mem, int N, int K, int D, int per_shot_norm, void* proto_f16,
                          float* proto_f32, float* proto_sq, pclip_stream_t stream);

/* Visual-bank reduction, utils.py:318-326: feats [A, R, D] fp16 ->
 * keys[j] = normalise(r16(mean_A feats[:, perm ? perm[j] : j, :])), row-major [R, D] fp16. */
int pclip_bank_reduce_f16(const void* feats, int A, int R, int D, const int32_t* perm, void* keys,
                          pclip_stream_t stream);

/* fp16 matrix transpose x[R,C] -> y[C,R] (bank layout [D, N*K] <-> [N*K, D], utils.py:320). */
int pclip_transpose_f16(const void* x, int R, int C, void* y, pclip_stream_t stream);

/* Multi-GPU shard of the prototype mean: rows of this rank's support slab with
 * NON-DECREASING int32 labels -> fp32 per-class sums of (optionally per-shot normalised) rows and
 * int32 counts.  Classes absent from the slab get zero sums/counts.  Deterministic (no atomics). */
int pclip_partial_sums_f16(const void* mem, const int32_t* labels, int R, int N, int D, int per_shot_norm,
                           float* sums, int32_t* counts, pclip_stream_t stream);

/* Combine W gathered slabs (sums [W,N,D], counts [W,N]) in rank order, then finish as
 * pclip_proto_build_f16 does.  W=1 reproduces the single-GPU result. */
int pclip_proto_finalize(const float* sums, const int32_t* counts, int W, int N, int D, void* proto_f16,
                         float* proto_f32, float* proto_sq, pclip_stream_t stream);

/* ---- classification: utils.py:225-244 `P` ------------------------------------------------ */

/* Squared Euclidean distances against both prototype banks in one pass over the queries:
 * d2x[q, n] = (sqrt(max(||q||^2 + ||z_n||^2 - 2 q.z_n, 0)))^2, fp32, leading dimension ldd >= N.
 * q [Q,D], zi/zt [N,D] fp16; the dot products run on fp16-input / fp32-accumulate MFMA (exact
 * products of the fp16 operands).  q_sq/zi_sq/zt_sq (nullable) are precomputed fp32
 * squared norms; when NULL they are computed into `ws`.  zt may be NULL (single bank). */
int pclip_sqdist_f16(const void* q, const void* zi, const void* zt, int Q, int N, int D,
                     const float* q_sq, const float* zi_sq, const float* zt_sq,
                     float* d2i, float* d2t, int ldd, void* ws, size_t ws_bytes, pclip_stream_t stream);

/* fp32-operand variant for the training path (main.py:262-281: fp32 prototypes and adapted queries): exact
 * fp32 arithmetic on v_mfma_f32_32x32x2_f32; any D; norms computed in the kernel. */
int pclip_sqdist_f32(const float* q, const float* zi, const float* zt, int Q, int N, int D, float* d2i,
                     float* d2t, int ldd, pclip_stream_t stream);

/* p = alpha*softmax(-beta*d2i) + one_minus_alpha*softmax(-beta*d2t) over classes (max-subtracted,
 * fp32).  Outputs (each nullable): p [Q, N] dense; argmax [Q] (lowest index among ties,
 * main.py:190); top-k probabilities/indices [Q, k] sorted descending (toolkit
 * proto_clip_classifier.py:146-147), k <= 16. */
int pclip_fuse_probs(const float* d2i, const float* d2t, int Q, int N, int ldd, float alpha,
                     float one_minus_alpha, float beta, float* p, int32_t* argmax, float* topk_p,
                     int32_t* topk_i, int k, pclip_stream_t stream);

/* Convenience: pclip_sqdist_f16 + pclip_fuse_probs with the distance rows held in `ws`. */
int pclip_classify_f16(const void* q, const void* zi, const void* zt, int Q, int N, int D,
                       const float* q_sq, const float* zi_sq, const float* zt_sq, float alpha,
                       float one_minus_alpha, float beta, float* p, int32_t* argmax, float* topk_p,
                       int32_t* topk_i, int k, void* ws, size_t ws_bytes, pclip_stream_t stream);

/* The route pclip_classify_f16 takes for a call of this shape under the current settings: 0 = two stages (pclip_sqdist_f16 + pclip_fuse_probs), 1 = one launch for
 * small class counts (N <= 16; N <= 32 with top-k), 2 = one launch for 16 < N <= 256, 3 = fused row panels (argmax only, Q N large).  The routes agree with the
 * reference's p to 1e-5 and with each other up to fp32 summation order: a query whose top two classes are closer than ~1e-6 in p may get either of them depending
 * on the route, i.e. on the batch it is classified in.  Callers that need one arithmetic for every batch size force the two stages (pclip_classify_mid_config(0),
 * pclip_classify_panel_config(0), env PCLIP_CLASSIFY_SMALL=0).  ws_bytes: the workspace the call would pass. */
int pclip_classify_route(int Q, int N, int D, float alpha, float one_minus_alpha, float beta, int has_zt, int want_p, int want_argmax, int topk, size_t ws_bytes);

/* The fused row-panel classification walks the class tiles ONCE where it can prove the result (csrc/pclip_classify_panel.hip: per group of 16 classes the nearest
 * class of each bank is kept as a candidate, everybody else is bounded through the group's second smallest distances; a panel whose rows all satisfy
 * max bound < best candidate is finished from the candidates — the very argmax of the second pass — and only the others walk the tiles again).
 * pclip_classify_panel_passes: 0 = that (default; env PCLIP_CLASSIFY_PANEL_PASSES), 1 = always two passes (round 5's first form), 2 = candidates computed but every panel
 * sent through the second pass (tests); returns the previous mode, a negative argument only queries.  pclip_classify_panel_stats: out3[0] = panels classified,
 * out3[1] = panels that needed a second pass, out3[2] = class tiles those second passes walked (a second pass covers only the tiles that hold a bound the proof
 * could not beat), since the last reset (host pointer to three ints; synchronises the device). */
int pclip_classify_panel_passes(int mode);
int pclip_classify_panel_stats(int* out3, int reset);

/* One-launch classification for mid-sized class counts (csrc/pclip_classify_mid.hip; utils.py:225-244 + main.py:190): pclip_classify_f16 takes it by itself for
 * 16 < N <= 256 with both banks, D % 128 == 0, p and / or argmax (no top-k), Q N <= 2e6.  mode 1 = that routing (default; env PCLIP_CLASSIFY_MID), 2 = every shape the kernel can
 * run (tests), 0 = off (two stages), < 0 = query only.  Returns the previous setting (-1 = not decided yet). */
int pclip_classify_mid_config(int mode);

/* Test entry of the fused large-N classification (csrc/pclip_classify_panel.hip; pclip_classify_f16 takes that path by itself for N > 32 when only the argmax is
 * asked for): the distances it forms for its first tile — dump [2][256][128] fp32 = d2 of query rows 0..255 x classes 0..127, visual bank then textual bank —
 * with exact != 0 (torch.cdist's sqrt -> square round trip kept: PCLIP_CLASSIFY_PANEL_EXACT=1) they must be the bits pclip_sqdist_f16 writes (utils.py:230-233),
 * with exact == 0 (the product's arithmetic: max(v, 0), a third faster) within one fp32 ulp of them.  ws as for pclip_classify_f16. */
/* Routing of pclip_classify_f16's argmax-only calls with N > 32: mode 1 = fused row-panel kernel where the call has at least half a 256-query panel per CU (default; env
 * PCLIP_CLASSIFY_PANEL), 2 = for every shape it can run (tests), 0 = the two stages, < 0 = query only.  Returns the previous setting (-1 = not decided yet). */
int pclip_classify_panel_config(int mode);
int pclip_classify_panel_dump_f16(const void* q, const void* zi, const void* zt, int Q, int N, int D, float* dump, int exact,
                                  void* ws, size_t ws_bytes, pclip_stream_t stream);

/* (alpha, beta) grid, main.py:142-146, 187-199, 419-430: from the two distance matrices evaluate all
 * na*nb pairs and accumulate correct[ia*nb + ib] += #{q : argmax_n p == labels[q]} (int32, the
 * caller zeroes it; lowest-index tie rule).  Replaces 3*na*nb `P` calls + host syncs. */
int pclip_hp_sweep(const float* d2i, const float* d2t, const int32_t* labels, int Q, int N, int ldd,
                   const float* alphas, const float* one_minus_alphas, int na, const float* betas,
                   int nb, int32_t* correct, pclip_stream_t stream);

/* ---- query adapters: model.py:12-95 ------------------------------------------------------ */

/* Adapter_FC.forward (model.py:81-95): y = r16(r16(ratio*LN_D(W2 LN_{D/r}(W1 x))) + r16((1-ratio)*x)).
 * x [B,D]; w1 [H,D]; g1,b1 [H]; w2 [D,H]; g2,b2 [D]; all fp16.  l2norm_out=1 also applies the
 * following row normalise (main.py:408-409).  y_sq nullable (fp32 ||y||^2). */
int pclip_adapter_fc_f16(const void* x, int B, int D, int H, const void* w1, const void* g1, const void* b1,
                         const void* w2, const void* g2, const void* b2, float ratio, float one_minus_ratio,
                         int l2norm_out, void* y, float* y_sq, void* ws, size_t ws_bytes,
                         pclip_stream_t stream);

/* The last stage of Adapter_FC.forward on its own (model.py:88, 92-95): y = r16(r16(ratio * LN_D(h)) + r16((1-ratio) * x))
 * with fp16 LayerNorm parameters — the training step launches the stages one by one to keep their activations
 * (proto_clip_amd/train.py) and must end with exactly the arithmetic of pclip_adapter_fc_f16.  h, x, y [R, D] fp16. */
int pclip_layernorm_blend_f16(const void* h, const void* gamma, const void* beta, float eps, const void* x, float ratio,
                              float one_minus_ratio, int l2norm_out, void* y, float* y_sq, int R, int D,
                              pclip_stream_t stream);

/* Adapter.forward (model.py:49-78), width 16: pad D -> s*s, conv1 1x1 -> LN[16,s,s] ->
 * (three_x: conv2 3x3 pad 1 -> LN[16,s,s]) -> conv3 1x1 -> LN[1,s,s] -> +identity -> crop.  No ReLU.
 * conv1 [16], ln1w/ln1b [16*s*s], conv2 [16*16*3*3], ln2w/ln2b [16*s*s], conv3 [16], ln3w/ln3b [s*s]. */
int pclip_adapter_conv_f16(const void* x, int B, int D, int three_x, const void* conv1, const void* ln1w,
                           const void* ln1b, const void* conv2, const void* ln2w, const void* ln2b,
                           const void* conv3, const void* ln3w, const void* ln3b, int l2norm_out, void* y,
                           float* y_sq, pclip_stream_t stream);

/* ---- CLIP encoder building blocks: clip/model.py:155-238, 338-354 ------------------------- */

/* C[M,N] = epilogue(A[M,K] . B[N,K]^T): fp16 operands, fp32-accumulate MFMA, fp16 output.
 *   bias (nullable, fp16 [N]) is added before rounding (nn.Linear, clip/model.py:176-178);
 *   act: 0 none, 1 QuickGELU x*sigmoid(1.702x) (clip/model.py:164-166) with per-op fp16 rounding;
 *   residual (nullable, fp16 [M,N], ldc): out = r16(residual + r16(...)) (clip/model.py:188-189).
 * lda/ldb/ldc in elements.  K % 64 == 0 required.  Calls with no more 128x64 output tiles than the device has CUs (a serving request, the class-token tail) run a latency-oriented kernel with the same arithmetic (bit-identical results). */
int pclip_gemm_f16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                   const void* bias, int act, const void* residual, pclip_stream_t stream);

/* The same linear on the FOUR-wave 256 x 256 tile with the hand-scheduled (inline-asm) K-loop of csrc/pclip_gemm4w.hip: one wave per
 * SIMD, 128 x 128 accumulators per wave, one barrier per K-tile over a ring of five 32 KB half-tile slots.  Bit-identical to
 * pclip_gemm_f16 (same MFMA, operand roles and k order).  Requires N % 256 == 0, K % 64 == 0, K >= 192, 16-byte aligned rows;
 * a residual needs a bias and act == 0.  PCLIP_E_INVALID otherwise (pclip_gemm_f16 routes its 256 x 256 tiles to it by itself: pclip_gemm4w_config).  Replaces the same call sites as pclip_gemm_f16 (clip/model.py:176-190). */
int pclip_gemm4w_f16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                     const void* bias, int act, const void* residual, pclip_stream_t stream);
/* ... with the build of the K-loop chosen by the caller: 0 = the product loop, 1 = its RACE-STRESS build (an s_sleep pause of one wave, a different one each
 * time, in front of every counted s_waitcnt and every barrier: a wait that is too weak then reads stale LDS; tests demand bit-identity with variant 0),
 * 6 = ABLATION build that runs the K-loops and stores NOTHING (timing only: what a fully hidden epilogue would buy), 7 = K = 768 bias / QuickGELU tiles with the
 * second half of every tile stored from registers under the next tile's K-loop (bit-identical, measured slower; other shapes fall back to 0).  Test / tuning entry. */
int pclip_gemm4w_var_f16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                         const void* bias, int act, const void* residual, int var, pclip_stream_t stream);
/* Diagnostic (loop variant 8 of pclip_gemm4w_var_f16: the product loop in a kernel that keeps time stamps of its tile phases): device buffer of 260 unsigned that the
 * next variant-8 launches fill — stamps [4][64] of (workgroup 0 | gridDim / 2) x (wave 0 | 3), eight s_memrealtime stamps (100 MHz) per tile for the first eight
 * tiles, + one closing stamp per wave at [256 + w].  NULL switches it off.  tools/gemm4w_stamps.py. */
int pclip_gemm4w_stamp_buffer(void* stamps);
/* Routing of pclip_gemm_f16's 256 x 256 tiles: mode 1 = four-wave asm-loop kernel (default; env PCLIP_GEMM_4W), 0 = eight-wave kernel, < 0 = query only.
 * Returns the previous setting (-1 = not decided yet).  Same bits either way. */
int pclip_gemm4w_config(int mode);

/* The same linear for SMALL M (a serving request: M = 197 x batch rows; the class-token tail of the last block: M = batch),
 * where pclip_gemm_f16 has a dozen tiles for 256 CUs and a K-loop of 12 - 48 dependent round trips: the K range is cut into
 * up to 8 slices (a function of K only), one workgroup per (128 x 64 tile, slice) writing an fp32 slab into ws; a second launch
 * adds the slabs in slice order (deterministic) and applies bias / act.  Same arithmetic as pclip_gemm_f16 up to fp32 summation
 * order (no residual operand).
 *   pclip_gemm_splitk_workspace: bytes of `ws` for this shape on the current device, or 0 when the shape gains nothing
 *     (enough tiles, K < 512, N % 64 != 0) — then call pclip_gemm_f16.
 * Replaces the same nn.Linear call sites as pclip_gemm_f16 (clip/model.py:176-190, 236-238) on the serving path
 * (toolkit proto_clip_classifier.py:132-158). */
size_t pclip_gemm_splitk_workspace(int M, int N, int K);
int pclip_gemm_splitk_f16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                          const void* bias, int act, void* ws, size_t ws_bytes, pclip_stream_t stream);

/* Convolution-as-GEMM with the eval-mode BatchNorm (+ReLU) that follows it in the ModifiedResNet tower (clip/model.py:43-52,
 * 138-142): C = relu?( r16( r16(A B^T) * scale[n] + shift[n] ) ), scale/shift fp32 [N] = the folded running statistics and
 * affine.  Same rounding points as conv (fp16 tensor) followed by pclip_bn_act_f16. */
int pclip_gemm_bn_f16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                      const float* scale, const float* shift, int relu, pclip_stream_t stream);

/* conv3 + bn3 + `out += identity` + ReLU of a bottleneck (clip/model.py:49-52) in one launch:
 * C = relu( r16( r16( r16(A B^T) * scale[n] + shift[n] ) + residual ) ), residual fp16 [M, N] with the row stride ldc of C.
 * Identical to pclip_gemm_f16 followed by pclip_bn_act_f16(residual, relu).  N % 64 == 0, K % 64 == 0, 16-byte aligned operands
 * (PCLIP_E_INVALID otherwise: use the two calls). */
int pclip_gemm_bn_res_f16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                          const float* scale, const float* shift, const void* residual, pclip_stream_t stream);

/* 3x3 convolution, stride 1, padding 1, on NHWC fp16 activations x [B, H, W, Cin] + eval BatchNorm (+ReLU), as an implicit
 * GEMM: the im2col matrix is never materialised (each K-tile is gathered by buffer LDS-DMA; taps outside the image are out-of-range
 * offsets, i.e. zeros — `zero_line`, >= 128 zero bytes in device memory, is still required non-null for ABI stability but no longer read).  w [Cout, 3, 3, Cin] fp16; y [B*H*W, Cout].
 * Cin % 64 == 0, or Cin = 8 / 16 / 32 (the stem, clip/model.py:138-142) with every row of w zero-padded to a multiple of 64
 * halves; Cout % 64 == 0, or Cout = 32.  Identical to pclip_im2col3x3_f16 + pclip_gemm_bn_f16 (clip/model.py:20-22, 45-46). */
int pclip_conv3x3_bn_f16(const void* x, const void* w, const void* zero_line, int B, int H, int W, int Cin, int Cout,
                         const float* scale, const float* shift, int relu, void* y, pclip_stream_t stream);

/* The narrow layers of the tower (Cin, Cout in {32, 64}; H % 8 == 0, W % 56 == 0: the stem's conv2 / conv3 at 112 x 112, layer1's conv2 at 56 x 56; any batch) are routed by pclip_conv3x3_bn_f16 to a kernel of their own (csrc/pclip_conv_strip.hip: weights in registers, a tile's input block with its halo once
 * in LDS, zero padding by out-of-range buffer loads).  Same rounding points; the fp32 accumulation order differs from the implicit GEMM's ((dx, chunk, dy) instead of
 * (dy, dx, chunk)), so single fp16 results may differ by one ulp.  pclip_conv3x3_strip_applies: would this shape take it (1 / 0); pclip_conv3x3_strip_config: -1 the
 * environment's choice (PCLIP_CONV_STRIP, default on), 0 off, 1 on — returns the previous mode (tests, A/B runs). */
int pclip_conv3x3_strip_applies(int B, int H, int W, int Cin, int Cout);
int pclip_conv3x3_strip_config(int mode);

/* The stem's first convolution (3 -> Cout channels, 3x3, stride 2, pad 1) + eval BatchNorm (+ReLU) straight from the NCHW images `img` [B, 3, R, R] (fp32 if
 * img_is_f32 — rounded to fp16 on the way, as a separate cast would — else fp16) into NHWC fp16 y [B * Ho * Ho, Cout], Ho = (R - 1) / 2 + 1: no im2col matrix, no
 * cast pass (clip/model.py:100-102, 138).  w [Cout][64] fp16: the im2col column order (ky, kx, channel), zero beyond column 27 — the operand pclip_gemm_bn_f16 takes
 * after pclip_im2col3x3_f16, and the same arithmetic.  R even, Ho a multiple of 56, Cout 32 or 64 (pclip_stem_conv_applies; PCLIP_CONV_STEM=0 turns the routing of
 * the python model off); PCLIP_E_INVALID otherwise. */
/* relu(bn(conv3x3(x))) followed by nn.AvgPool2d(2) in one launch (the stem's conv3 / bn3 / relu / avgpool, clip/model.py:104-105, 142-143): y [B * (H / 2) * (W / 2), Cout].
 * The same bits as pclip_conv3x3_bn_f16 + pclip_avgpool_nhwc_f16 without writing and re-reading the unpooled activation.  Cin 32, Cout 64, H % 8 == 0, W % 56 == 0
 * (pclip_conv3x3_pool_applies, which also honours PCLIP_CONV_STRIP / pclip_conv3x3_strip_config); PCLIP_E_INVALID otherwise. */
int pclip_conv3x3_pool_applies(int H, int W, int Cin, int Cout);
int pclip_conv3x3_bn_pool_f16(const void* x, const void* w, int B, int H, int W, int Cin, int Cout, const float* scale, const float* shift, void* y,
                              pclip_stream_t stream);

int pclip_stem_conv_applies(int R, int Cout);
int pclip_stem_conv_bn_f16(const void* img, int img_is_f32, int B, int R, const void* w, int Cout, const float* scale, const float* shift, int relu, void* y,
                           pclip_stream_t stream);

/* LayerNorm over the last dim with fp32 statistics and fp32 affine parameters, fp16 in/out
 * (clip/model.py:155-161).  x rows are ld_x elements apart (lets ln_post read only the CLS rows). */
int pclip_layernorm_f16(const void* x, int ld_x, const float* gamma, const float* beta, float eps, void* y,
                        int R, int D, pclip_stream_t stream);

/* Residual add fused into the following LayerNorm (clip/model.py:188-189 then ln_2 / next ln_1 / ln_post /
 * ln_final): xs = r16(x + delta); x_out (nullable, may alias x; row stride ld) receives xs;
 * y [R, D] = r16(LayerNorm(xs)).  x and delta rows are ld elements apart. */
int pclip_add_layernorm_f16(const void* x, const void* delta, int ld, void* x_out, const float* gamma,
                            const float* beta, float eps, void* y, int R, int D, pclip_stream_t stream);

/* Multi-head self-attention core on a fused QKV buffer (nn.MultiheadAttention inside
 * ResidualAttentionBlock.attention, clip/model.py:183-185): qkv [B, L, 3*H*dh] fp16 (q|k|v blocks,
 * heads contiguous inside each) -> out [B, L, H*dh] fp16 = softmax(q k^T / sqrt(dh) [+causal]) v.
 * dh must be 64; L <= 288. */
int pclip_attention_f16(const void* qkv, void* out, int B, int L, int H, int dh, int causal,
                        pclip_stream_t stream);

/* The same attention with separate operands: queries = the first Lq tokens of every sequence (q + b*q_batch_stride + row*ldq,
 * head h at column h*64), keys / values in kv [B*L rows, row stride ldkv] at column offsets k_off / v_off; out [B*Lq, H*64].
 * Lq < L serves the last vision block, whose output is only read at the class token (clip/model.py:233); causal needs Lq == L. */
int pclip_attention_q_f16(const void* q, int ldq, long q_batch_stride, const void* kv, int ldkv, int k_off, int v_off, void* out,
                          int B, int L, int Lq, int H, int dh, int causal, pclip_stream_t stream);

/* Kernel choice behind the two attention entry points (same results bit for bit either way): mode -1 / 0 = one workgroup per
 * (image, head) pair (default); 1 = where its shape conditions hold (Lq == L, L <= 256), the persistent kernel that prefetches the
 * next pair's K / V / Q rows by LDS-DMA while the current one is multiplied — built and kept as a measured experiment (faster in
 * isolation, not inside the encoder).  max_grid > 0 caps the persistent grid (test hook: several items per
 * workgroup on small problems).  Process-wide, not thread-safe; PCLIP_ATT_PIPE=0/1 sets the initial mode.  No reference
 * counterpart (nn.MultiheadAttention picks its own kernels, clip/model.py:176-178). */
int pclip_attention_config(int mode, int max_grid);

/* ViT stem (clip/model.py:222-227): patch-conv as an im2col gather of [B,3,R,R] fp16 images into
 * [B*G*G, ld >= 3*P*P] rows, zero beyond 3*P*P (the GEMM against conv1.weight follows), and the token assembly
 * x = [class_emb ; patches] + pos -> fp16 [B, 1+G*G, W]. */
int pclip_im2col_patches_f16(const void* img, int B, int R, int P, void* cols, int ld, pclip_stream_t stream);
/* the same gather from fp32 images, fused with their cast to fp16 (`image.type(self.dtype)`, clip/model.py:339) */
int pclip_im2col_patches_f32(const float* img, int B, int R, int P, void* cols, int ld, pclip_stream_t stream);
int pclip_vit_assemble_tokens_f16(const void* patch_emb, const void* class_emb, const void* pos_emb, int B,
                                  int G2, int W, void* tokens, pclip_stream_t stream);
/* The same assembly fused with ln_pre and the first block's ln_1 (clip/model.py:225-227, 188): x0 = ln_pre(tokens) (the residual
 * stream entering the transformer) and h = ln_1(x0), one pass per token row, bit-identical to the three separate calls.
 * h may be NULL (then gamma_1 / beta_1 are unused): only x0 is produced. */
int pclip_vit_embed_ln_f16(const void* patch_emb, const void* class_emb, const void* pos_emb, int B, int G2, int W,
                           const float* gamma_pre, const float* beta_pre, const float* gamma_1, const float* beta_1, float eps,
                           void* x0, void* h, pclip_stream_t stream);

/* Text stem (clip/model.py:342-344): x = token_embedding[text] + positional_embedding, fp16. */
int pclip_text_embed_f16(const int64_t* tokens, const void* tok_emb, const void* pos_emb, int B, int L, int W,
                         int vocab, void* x, pclip_stream_t stream);
/* Row gather x[b, idx[b], :] (EOT token, clip/model.py:352) ; idx computed by argmax over tokens. */
int pclip_gather_eot_f16(const void* x, const int64_t* tokens, int B, int L, int W, void* out,
                         pclip_stream_t stream);

/* ---- ModifiedResNet tower pieces: clip/model.py:10-152 (activations NHWC fp16) ------------------------ */

/* im2col for a 3x3 / padding 1 / stride 1|2 convolution (clip/model.py:20, 109-113): element (b,y,x,c) of the
 * input sits at x[b*sb + y*sh + x*sw + c*sc] (NHWC activations or the NCHW image);
 * cols[((b*Ho+oy)*Wo+ox)*ld + (ky*3+kx)*C + c], zero outside the image and for columns >= 9*C. */
int pclip_im2col3x3_f16(const void* x, long sb, long sh, long sw, long sc, int B, int H, int W, int C, int stride,
                        void* cols, int ld, pclip_stream_t stream);

/* Eval-mode BatchNorm2d (scale/shift folded from the fp32 statistics) + optional residual add + optional ReLU on
 * [rows, C] fp16: y = relu(r16(r16(x*scale + shift) + residual))  (clip/model.py:43-52). */
int pclip_bn_act_f16(const void* x, const float* scale, const float* shift, const void* residual, int relu, void* y,
                     size_t rows, int C, pclip_stream_t stream);

/* nn.AvgPool2d(k) on NHWC fp16 (clip/model.py:23, 35, 115). */
int pclip_avgpool_nhwc_f16(const void* x, int B, int H, int W, int C, int k, void* y, pclip_stream_t stream);

/* AttentionPool2d token assembly (clip/model.py:68-70): [mean token ; tokens] + positional embedding. */
int pclip_attnpool_tokens_f16(const void* x, const void* pos, int B, int HW, int C, void* tokens, pclip_stream_t stream);

/* fp32 -> fp16 cast (image.type(self.dtype), clip/model.py:339). */
int pclip_cast_f32_f16(const float* x, void* y, size_t n, pclip_stream_t stream);

/* ---- episodic training step (reference main.py:216-310, utils.py:80-109; torch autograd + torch.optim.AdamW there) -------- */

/* fp16 -> fp32 copy (`.float()`, main.py:263, 270, 279). */
int pclip_cast_f16_f32(const void* x, float* y, size_t n, pclip_stream_t stream);

/* C[M,N] (ldc) = alpha * op(A) op(B) + beta * C on v_mfma_f32_32x32x2_f32, element (m,k) of op(A) at A[m*rsa + k*csa], element
 * (k,n) of op(B) at B[k*rsb + n*csb]; a_f16 / b_f16 != 0: that operand is fp16 in memory (exact in fp32).  Serves every
 * matmul autograd issues in the training step: cdist backward (dq = -2 G Z, dz = -2 G^T q), InfoNCE logits and their
 * gradients, Linear weight / input gradients of Adapter_FC. */
int pclip_gemm_f32(const void* A, int a_f16, long rsa, long csa, const void* B, int b_f16, long rsb, long csb, float* C, int ldc,
                   int M, int N, int K, float alpha, float beta, void* ws, size_t ws_bytes, pclip_stream_t stream);
/* ws (nullable): with >= 32 * M * N * 4 bytes, a call with at most 128 output tiles and K >= 1024 is cut into K slices whose
 * partial products are added in slice order (deterministic; differs from the unsplit sum in fp32 order only). */

/* out[c] (+)= scale * sum_r x[r, c]  (fixed summation order; accumulate != 0 adds to out).  ws (nullable, >= 64 * C * 4 bytes):
 * with it, more than 128 rows are summed as up to 64 row blocks in parallel and the block sums added in block order. */
int pclip_colsum_f32(const float* x, int ldx, int R, int C, float scale, float* out, int accumulate, void* ws, size_t ws_bytes,
                     pclip_stream_t stream);

/* C[r, :] += s * rowscale[r] * X[r, :]  (the 2*rowsum(G)*q and 2*colsum(G)*z terms of the cdist backward). */
int pclip_addscaled_rows_f32(float* C, int ldc, const float* X, int ldx, const float* rowscale, float s, int R, int D,
                             pclip_stream_t stream);

/* L = mean_q -log p[q, y_q] with p as in pclip_fuse_probs (utils.py:90-93 `NLLLoss()(torch.log(p), target)`): per-query terms
 * nll[q] = -log p[q,y_q], pmax[q], argmax[q] (utils.py:84-85) and the gradients gi/gt [Q, ldd] of L wrt the two squared-distance
 * rows; rowsum[q] = sum_c (gi + gt)[q, c].  q_total >= Q is the number of queries the mean runs over (> Q when the queries of a
 * step are sharded over ranks). */
int pclip_nll_grad(const float* d2i, const float* d2t, const int32_t* labels, int Q, int q_total, int N, int ldd, float alpha,
                   float one_minus_alpha, float beta, float* gi, float* gt, float* rowsum, float* nll, float* pmax,
                   int32_t* argmax, pclip_stream_t stream);

/* Backward of P (utils.py:225-244) for an ARBITRARY upstream gradient dp [Q, N] (leading dimension ldp) — the autograd-
 * transparent `utils.P` the reference's own loop drives (main.py:281-310: NLLLoss(torch.log(p)).backward()): gradients wrt both
 * squared-distance rows (gi, gt [Q, ldd], columns >= N untouched) and rowsum[q] = sum_c (gi + gt)[q, c]. */
int pclip_fuse_probs_backward(const float* d2i, const float* d2t, const float* dp, int ldp, int Q, int N, int ldd, float alpha,
                              float one_minus_alpha, float beta, float* gi, float* gt, float* rowsum, pclip_stream_t stream);

/* Backward of nn.NLLLoss()(torch.log(p), labels) (utils.py:90-93) wrt p: dp[q, c] = -g[0] / (Q p[q, c]) at c == labels[q], else 0;
 * g = upstream gradient of the scalar loss (device pointer). */
int pclip_nll_mean_backward(const float* p, int ldp, const int32_t* labels, int Q, int N, const float* g, float* dp, int lddp,
                            pclip_stream_t stream);

/* The same per-query terms from a materialised p [Q, N] (leading dimension ldp): utils.compute_loss_and_matches as a
 * forward-only drop-in (utils.py:84-93). */
int pclip_nll_rows(const float* p, int ldp, const int32_t* labels, int Q, int N, float* nll, float* pmax, int32_t* argmax,
                   pclip_stream_t stream);

/* Cross entropy of the rows of S [R, C>=R] against the diagonal (InfoNCE of utils.py:72-77 with the info-nce-pytorch
 * defaults): loss[r] = logsumexp(S[r,:]) - S[r,r], dS[r,c] = scale * (softmax(S[r,:])[c] - [c == r]). */
int pclip_softmax_ce_rows(const float* S, int lds, int R, int C, float scale, float* dS, int ldds, float* loss,
                          pclip_stream_t stream);

/* F.normalize(x, dim=-1, eps) on fp32 rows and its backward gx (+)= (gy - y (y.gy)) / max(|x|, eps): InfoNCE normalises
 * both of its arguments (utils.py:72-77). */
int pclip_l2norm_rows_f32(const float* x, float* y, int R, int D, float eps, pclip_stream_t stream);
int pclip_l2norm_rows_backward_f32(const float* x, const float* gy, float* gx, int R, int D, float eps, int accumulate,
                                   pclip_stream_t stream);

/* Backward of the prototype chain (main.py:260-264 image bank; 276-279 text bank with K = 1, final_norm = 0; the query rows
 * `adapter(x).float()` / norm with K = 1, per_shot_norm = 0): g [N, D] fp32 is the gradient wrt the chain's fp32 output, dmem
 * [N*K, D] fp16 the gradient wrt the fp16 rows `mem`; fp16 stages are rounded where autograd materialises fp16 tensors. */
int pclip_proto_backward_f16(const void* mem, const float* g, int N, int K, int D, int per_shot_norm, int final_norm, void* dmem,
                             pclip_stream_t stream);

/* nn.LayerNorm backward on fp16 tensors (model.py:86, 88): dx [R, D]; part [nblk][2][D] fp32 receives the partial
 * (dgamma | dbeta) sums of the rows each of the nblk workgroups visits (reduce with pclip_colsum_f32).  dy_scale: factor applied
 * to dy first, rounded to fp16 (the `ratio * x` of model.py:93-94). */
int pclip_layernorm_backward_f16(const void* x, int ldx, const void* gamma, const void* dy, int lddy, int R, int D, float eps,
                                 float dy_scale, void* dx, int lddx, float* part, int nblk, pclip_stream_t stream);

/* Backward of pclip_adapter_conv_f16 for the training step (reference: autograd through model.py:49-78; the input rows
 * are constants, main.py:266).  x, g [B, D] fp16 (input rows, gradient wrt the adapter output).  Outputs are fp32 PARTIAL sums of the
 * parameter gradients, R rows of them, to be summed over those R rows with pclip_colsum_f32: pw1/pw3 [R,16] (conv1 / conv3),
 * pw2 [R, 16*16*9] (conv2, layout [co][ci][ky][kx]), pg1/pb1, pg2/pb2 [R, 16*s*s], pg3/pb3 [R, s*s] (LayerNorm weight | bias),
 * s = ceil(sqrt(D)).  R = pclip_adapter_conv_backward_partials(B, D, three_x) <= B: conv-3x with D <= 576 runs ONE persistent launch on the
 * matrix pipe whose workgroup w accumulates rows w, w + R, ... (R = min(B, #CU); deterministic: fixed assignment, fixed order); every other
 * case writes one partial per input row (R = B).  conv-2x: conv2 / ln2* / pw2 / pg2 / pb2 are NULL (those parameters receive no gradient). */
int pclip_adapter_conv_backward_partials(int B, int D, int three_x);
int pclip_adapter_conv_backward_f16(const void* x, const void* g, int B, int D, int three_x, const void* conv1, const void* ln1w,
                                    const void* ln1b, const void* conv2, const void* ln2w, const void* ln2b, const void* conv3,
                                    const void* ln3w, float* pw1, float* pw2, float* pw3, float* pg1, float* pb1, float* pg2,
                                    float* pb2, float* pg3, float* pb3, pclip_stream_t stream);

/* One torch.optim.AdamW step (main.py:134-135: eps 1e-4, weight_decay 0.05) on fp16 parameters with fp16 moments, every
 * intermediate rounded to fp16 where the single-tensor implementation materialises an fp16 tensor; step counts from 1. */
int pclip_adamw_f16(void* p, const void* g, void* m, void* v, size_t n, double lr, double beta1, double beta2, double eps,
                    double weight_decay, int step, pclip_stream_t stream);

/* ---- image pre-processing (clip/clip.py:77-84 `_transform`; datasets/imagenet.py:8-23 `get_random_train_tfm`) ------------- */

/* A batch of decoded RGB images (uint8, HWC, device memory; srcs = device array of B pointers) -> normalised CHW tensors
 * out [B, 3, n_px, n_px] (fp32, or fp16 when out_f16: the cast encode_image applies first, clip/model.py:339).  Per image a
 * 16-int32 descriptor (device array desc [B][16]):
 *   0 h, 1 w                  source size
 *   2 box_top, 3 box_left, 4 box_h, 5 box_w   region that is resized (whole image for Resize; the RandomResizedCrop box)
 *   6 rs_h, 7 rs_w            size the box is resized to (PIL bicubic with antialiasing, bit-identical to Image.resize)
 *   8 win_top, 9 win_left     n_px x n_px window of the resized box that is kept (CenterCrop offsets; 0, 0 for the train tfm)
 *   10 flip                   horizontal flip of the window (RandomHorizontalFlip)
 *   11 coef_off               offset (int32 units) of this image's coefficient tables in ws: n_px*(2+ks_h) + n_px*(2+ks_v) ints
 *   12 tmp_off                offset (bytes) of its box_h x n_px x 3 uint8 scratch image in ws
 *   13 ks_h, 14 ks_v          taps per output: ceil(2 * max(box / rs, 1)) * 2 + 1 per axis (1 when box == rs)
 * max_box_h = max over the batch of box_h.  mean / std: Normalize constants. */
int pclip_preprocess_u8(const void* const* srcs, const int32_t* desc, int B, int n_px, int max_box_h, float mean0, float mean1,
                        float mean2, float std0, float std1, float std2, void* out, int out_f16, void* ws, pclip_stream_t stream);

/* ---- workspace sizing -------------------------------------------------------------------- */
#define PCLIP_OP_SQDIST 1
#define PCLIP_OP_CLASSIFY 2
#define PCLIP_OP_ADAPTER_FC 3
size_t pclip_workspace_bytes(int op, int Q, int N, int D);

#ifdef __cplusplus
}
#endif
#endif /* PCLIP_H */
