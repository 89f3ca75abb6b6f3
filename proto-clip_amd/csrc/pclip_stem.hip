// ViT / text stems (clip/model.py:217-236, 341-354): patch gather (+ fp32 -> fp16 cast), token assembly + ln_pre, token embedding, EOT gather.
#include "pclip_encoder_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {
// ---- ViT / text stems ------------------------------------------------------------------------------
// conv1 (kernel = stride = P, no bias) == GEMM of im2col rows [B*G*G, ld >= 3*P*P] against weight
// [W, 3*P*P]; columns >= 3*P*P are zero (K padded to the GEMM's BK for ViT-L/14, 3*14*14 = 588 -> 640).
// IT = float: the image.type(self.dtype) cast of clip/model.py:339 happens on the way (one rounding per pixel, as the cast kernel's)
template <bool VEC, typename IT = half_t>
__global__ __launch_bounds__(256) void im2col_kernel(const IT* __restrict__ img, int B, int R, int P, int G,
                                                     int ld, half_t* __restrict__ cols) {
    const int KP = 3 * P * P;
    constexpr int V = VEC ? 8 : 1;
    const int ldv = ld / V;
    const size_t total = (size_t)B * G * G * ldv;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int k = (int)(i % ldv) * V;
        const size_t pr = i / ldv;
        half_t* dst = cols + pr * ld + k;
        if (k >= KP) {
            if (VEC) { half8_t z; for (int j = 0; j < 8; ++j) z[j] = (half_t)0.f; st_half8(dst, z); }
            else *dst = (half_t)0.f;
            continue;
        }
        const int gx = (int)(pr % G), gy = (int)((pr / G) % G), bb = (int)(pr / ((size_t)G * G));
        const int c = k / (P * P), py = (k / P) % P, px = k % P;
        const IT* src = img + (((size_t)bb * 3 + c) * R + gy * P + py) * R + gx * P + px;
        if (VEC) {
            half8_t o;
            if constexpr (sizeof(IT) == 2) o = ld_half8(reinterpret_cast<const half_t*>(src));
            else {
                const float4_t a = *reinterpret_cast<const float4_t*>(src), b = *reinterpret_cast<const float4_t*>(src + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { o[j] = (half_t)a[j]; o[j + 4] = (half_t)b[j]; }
            }
            st_half8(dst, o);
        } else {
            *dst = (half_t)*src;
        }
    }
}

// tokens[b, 0] = r16(class + pos[0]); tokens[b, 1+g] = r16(patch[b, g] + pos[1+g])   (clip/model.py:225-226)
__global__ __launch_bounds__(256) void assemble_tokens_kernel(const half_t* __restrict__ patch,
                                                              const half_t* __restrict__ cls,
                                                              const half_t* __restrict__ pos, int B, int G2, int W,
                                                              half_t* __restrict__ tokens) {
    const int L = G2 + 1, WV = W / 8;
    const size_t nvec = (size_t)B * L * WV;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const int d = (int)(i % WV) * 8;
        const size_t row = i / WV;
        const int l = (int)(row % L);
        const size_t bb = row / L;
        half8_t a = l == 0 ? ld_half8(cls + d) : ld_half8(patch + (bb * G2 + (l - 1)) * W + d);
        half8_t p = ld_half8(pos + (size_t)l * W + d);
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)a[j] + (float)p[j]);
        st_half8(tokens + row * W + d, o);
    }
}

// The ViT stem after the patch GEMM in ONE pass per token row: x0 = ln_pre(r16([class ; patch] + pos)) and h = ln_1 of the first
// block (clip/model.py:225-227, 188), both written — the three kernels it replaces (assemble, ln_pre, ln_1) re-read the
// residual stream twice.  Row arithmetic identical to assemble_tokens_kernel + layernorm_kernel<MODE 0> (same summation order).
template <int NCH>
__device__ __forceinline__ void ln_row_inplace(float (&v)[NCH][8], int D, int lane, const float* __restrict__ gamma,
                                               const float* __restrict__ beta, float eps) {
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (c * 512 + lane * 8 < D) {
#pragma unroll
            for (int j = 0; j < 8; ++j) s += v[c][j];
        }
    const float mean = wave_sum(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (c * 512 + lane * 8 < D) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float t = v[c][j] - mean; q = ln_sq_acc(t, q); }
        }
    const float rstd = 1.f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        const int d = c * 512 + lane * 8;
        if (d < D) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float o = ln_affine(v[c][j], mean, rstd, (float)gamma[d + j], (float)beta[d + j]);
                v[c][j] = r16(o);
            }
        }
    }
}

template <int NCH, bool GB_LDS = false>
__global__ __launch_bounds__(256) void vit_embed_ln_kernel(const half_t* __restrict__ patch, const half_t* __restrict__ cls,
                                                           const half_t* __restrict__ pos, int B, int G2, int W,
                                                           const float* __restrict__ g0, const float* __restrict__ b0,
                                                           const float* __restrict__ g1, const float* __restrict__ b1, float eps,
                                                           half_t* __restrict__ x0, half_t* __restrict__ h) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int L = G2 + 1;
    const size_t R = (size_t)B * L;
    // GB_LDS (whole batches): the four affine vectors once per workgroup into LDS, as in layernorm_pf_kernel — per 1.5 KB row they were 12 KB through the
    // vector-memory path
    __shared__ __attribute__((aligned(16))) float gb_s[4][GB_LDS ? NCH * 512 : 4];
    if (GB_LDS) {
        for (int i = threadIdx.x; i < NCH * 512; i += 256) {
            gb_s[0][i] = i < W ? g0[i] : 0.f;
            gb_s[1][i] = i < W ? b0[i] : 0.f;
            gb_s[2][i] = (h && i < W) ? g1[i] : 0.f;
            gb_s[3][i] = (h && i < W) ? b1[i] : 0.f;
        }
        __syncthreads();
        g0 = gb_s[0]; b0 = gb_s[1]; g1 = gb_s[2]; b1 = gb_s[3];
    }
    for (size_t row = (size_t)blockIdx.x * 4 + wave; row < R; row += (size_t)gridDim.x * 4) {
        const int l = (int)(row % L);
        const size_t bb = row / L;
        const half_t* src = l == 0 ? cls : patch + (bb * G2 + (l - 1)) * W;
        float v[NCH][8];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int d = c * 512 + lane * 8;
            if (d < W) {
                const half8_t a = ld_half8(src + d), p = ld_half8(pos + (size_t)l * W + d);
#pragma unroll
                for (int j = 0; j < 8; ++j) v[c][j] = (float)(half_t)((float)a[j] + (float)p[j]);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
            }
        }
        ln_row_inplace<NCH>(v, W, lane, g0, b0, eps);
        half8_t xh[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int d = c * 512 + lane * 8;
            if (d < W) {
#pragma unroll
                for (int j = 0; j < 8; ++j) xh[c][j] = (half_t)v[c][j];
                st_half8(x0 + row * W + d, xh[c]);
            }
        }
        if (!h) continue;
        ln_row_inplace<NCH>(v, W, lane, g1, b1, eps);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int d = c * 512 + lane * 8;
            if (d < W) {
                half8_t o;
#pragma unroll
                for (int j = 0; j < 8; ++j) o[j] = (half_t)v[c][j];
                st_half8(h + row * W + d, o);
            }
        }
    }
}

__global__ __launch_bounds__(256) void text_embed_kernel(const int64_t* __restrict__ tokens,
                                                         const half_t* __restrict__ emb,
                                                         const half_t* __restrict__ pos, int B, int L, int W, int vocab,
                                                         half_t* __restrict__ x) {
    const int WV = W / 8;
    const size_t nvec = (size_t)B * L * WV;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < nvec; i += (size_t)gridDim.x * 256) {
        const int d = (int)(i % WV) * 8;
        const size_t row = i / WV;
        const int l = (int)(row % L);
        int64_t tk = tokens[row];
        tk = tk < 0 ? 0 : (tk >= vocab ? vocab - 1 : tk);
        half8_t a = ld_half8(emb + (size_t)tk * W + d), p = ld_half8(pos + (size_t)l * W + d), o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)a[j] + (float)p[j]);
        st_half8(x + row * W + d, o);
    }
}

// out[b] = x[b, argmax_l tokens[b, l]]  (first maximum, like torch.argmax)
__global__ __launch_bounds__(64) void gather_eot_kernel(const half_t* __restrict__ x, const int64_t* __restrict__ tokens,
                                                        int L, int W, half_t* __restrict__ out) {
    const int b = blockIdx.x, lane = threadIdx.x;
    float bv = -1.f; int bi = 0x7fffffff;
    for (int l = lane; l < L; l += 64) {
        const float t = (float)tokens[(size_t)b * L + l];
        if (t > bv) { bv = t; bi = l; }
    }
    wave_argmax(bv, bi);
    for (int d = lane * 8; d < W; d += 512) st_half8(out + (size_t)b * W + d, ld_half8(x + ((size_t)b * L + bi) * W + d));
}

}  // namespace

extern "C" int pclip_im2col_patches_f16(const void* img, int B, int R, int P, void* cols, int ld,
                                        pclip_stream_t stream) {
    PCLIP_REQUIRE(img && cols, "pclip_im2col_patches_f16: null pointer");
    PCLIP_REQUIRE(B >= 0 && P > 0 && R > 0 && R % P == 0, "pclip_im2col_patches_f16: bad B=%d R=%d P=%d", B, R, P);
    PCLIP_REQUIRE(ld >= 3 * P * P, "pclip_im2col_patches_f16: ld=%d < 3*P*P", ld);
    if (B == 0) return PCLIP_OK;
    const int G = R / P;
    const bool vec = P % 8 == 0 && R % 8 == 0 && ld % 8 == 0;
    const size_t total = (size_t)B * G * G * (vec ? ld / 8 : ld);
    if (vec) im2col_kernel<true><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>((const half_t*)img, B, R, P, G, ld, (half_t*)cols);
    else im2col_kernel<false><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>((const half_t*)img, B, R, P, G, ld, (half_t*)cols);
    return pclip_check_launch("im2col");
}

extern "C" int pclip_im2col_patches_f32(const float* img, int B, int R, int P, void* cols, int ld, pclip_stream_t stream) {
    PCLIP_REQUIRE(img && cols, "pclip_im2col_patches_f32: null pointer");
    PCLIP_REQUIRE(B >= 0 && P > 0 && R > 0 && R % P == 0, "pclip_im2col_patches_f32: bad B=%d R=%d P=%d", B, R, P);
    PCLIP_REQUIRE(ld >= 3 * P * P, "pclip_im2col_patches_f32: ld=%d < 3*P*P", ld);
    if (B == 0) return PCLIP_OK;
    const int G = R / P;
    const bool vec = P % 8 == 0 && R % 8 == 0 && ld % 8 == 0 && ((uintptr_t)img & 15) == 0;
    const size_t total = (size_t)B * G * G * (vec ? ld / 8 : ld);
    if (vec) im2col_kernel<true, float><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>(img, B, R, P, G, ld, (half_t*)cols);
    else im2col_kernel<false, float><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>(img, B, R, P, G, ld, (half_t*)cols);
    return pclip_check_launch("im2col (fp32 images)");
}

extern "C" int pclip_vit_assemble_tokens_f16(const void* patch_emb, const void* class_emb, const void* pos_emb, int B,
                                             int G2, int W, void* tokens, pclip_stream_t stream) {
    PCLIP_REQUIRE(patch_emb && class_emb && pos_emb && tokens, "pclip_vit_assemble_tokens_f16: null pointer");
    PCLIP_REQUIRE(B >= 0 && G2 > 0 && W > 0 && W % 8 == 0, "pclip_vit_assemble_tokens_f16: bad shape");
    if (B == 0) return PCLIP_OK;
    assemble_tokens_kernel<<<flat_grid((size_t)B * (G2 + 1) * (W / 8)), 256, 0, (hipStream_t)stream>>>(
        (const half_t*)patch_emb, (const half_t*)class_emb, (const half_t*)pos_emb, B, G2, W, (half_t*)tokens);
    return pclip_check_launch("assemble_tokens");
}

extern "C" int pclip_vit_embed_ln_f16(const void* patch_emb, const void* class_emb, const void* pos_emb, int B, int G2, int W,
                                      const float* gamma_pre, const float* beta_pre, const float* gamma_1, const float* beta_1, float eps,
                                      void* x0, void* h, pclip_stream_t stream) {
    PCLIP_REQUIRE(patch_emb && class_emb && pos_emb && gamma_pre && beta_pre && x0 && (!h || (gamma_1 && beta_1)),
                  "pclip_vit_embed_ln_f16: null pointer");
    PCLIP_REQUIRE(B >= 0 && G2 > 0 && W > 0 && W % 8 == 0 && W <= 4096, "pclip_vit_embed_ln_f16: bad shape B=%d G2=%d W=%d", B, G2, W);
    if (B == 0) return PCLIP_OK;
    const size_t R = (size_t)B * (G2 + 1);
    int grid = (int)((R + 3) / 4 > 16384 ? 16384 : (R + 3) / 4);
    const int ln_grid = pclip_device_cus() * PCLIP_LN_BPC;
    if (PCLIP_LN_LDS && W <= 1024 && ln_grid > 0 && R >= (size_t)16 * ln_grid) {     // whole batch: resident-size grid, affine vectors from LDS (<= 16 KB per workgroup)
        grid = ln_grid;
        if (W <= 512) vit_embed_ln_kernel<1, true><<<grid, 256, 0, (hipStream_t)stream>>>((const half_t*)patch_emb, (const half_t*)class_emb, (const half_t*)pos_emb, B, G2, W, gamma_pre,
                                                                                          beta_pre, gamma_1, beta_1, eps, (half_t*)x0, (half_t*)h);
        else vit_embed_ln_kernel<2, true><<<grid, 256, 0, (hipStream_t)stream>>>((const half_t*)patch_emb, (const half_t*)class_emb, (const half_t*)pos_emb, B, G2, W, gamma_pre,
                                                                                 beta_pre, gamma_1, beta_1, eps, (half_t*)x0, (half_t*)h);
        return pclip_check_launch("vit_embed_ln");
    }
#define PCLIP_VEL(NCH) vit_embed_ln_kernel<NCH><<<grid, 256, 0, (hipStream_t)stream>>>((const half_t*)patch_emb, (const half_t*)class_emb, \
        (const half_t*)pos_emb, B, G2, W, gamma_pre, beta_pre, gamma_1, beta_1, eps, (half_t*)x0, (half_t*)h)
    if (W <= 512) PCLIP_VEL(1);
    else if (W <= 1024) PCLIP_VEL(2);
    else if (W <= 2048) PCLIP_VEL(4);
    else PCLIP_VEL(8);
#undef PCLIP_VEL
    return pclip_check_launch("vit_embed_ln");
}

extern "C" int pclip_text_embed_f16(const int64_t* tokens, const void* tok_emb, const void* pos_emb, int B, int L, int W,
                                    int vocab, void* x, pclip_stream_t stream) {
    PCLIP_REQUIRE(tokens && tok_emb && pos_emb && x, "pclip_text_embed_f16: null pointer");
    PCLIP_REQUIRE(B >= 0 && L > 0 && W > 0 && W % 8 == 0 && vocab > 0, "pclip_text_embed_f16: bad shape");
    if (B == 0) return PCLIP_OK;
    text_embed_kernel<<<flat_grid((size_t)B * L * (W / 8)), 256, 0, (hipStream_t)stream>>>(
        tokens, (const half_t*)tok_emb, (const half_t*)pos_emb, B, L, W, vocab, (half_t*)x);
    return pclip_check_launch("text_embed");
}

extern "C" int pclip_gather_eot_f16(const void* x, const int64_t* tokens, int B, int L, int W, void* out,
                                    pclip_stream_t stream) {
    PCLIP_REQUIRE(x && tokens && out, "pclip_gather_eot_f16: null pointer");
    PCLIP_REQUIRE(B >= 0 && L > 0 && W > 0 && W % 8 == 0, "pclip_gather_eot_f16: bad shape");
    if (B == 0) return PCLIP_OK;
    gather_eot_kernel<<<B, 64, 0, (hipStream_t)stream>>>((const half_t*)x, tokens, L, W, (half_t*)out);
    return pclip_check_launch("gather_eot");
}
