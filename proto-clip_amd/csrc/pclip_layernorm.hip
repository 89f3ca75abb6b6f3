// LayerNorm passes of the CLIP towers and adapters (clip/model.py:155-161; model.py:86-95): wave-per-row kernels with fp32 statistics, the whole-batch prefetching form,
// row statistics / finalisation and the weight fold of the opt-in LayerNorm fold, the fused add + LayerNorm of the split-K serving path.
#include "pclip_encoder_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {
// ---- LayerNorm, one wave per row ----------------------------------------------------------------
// MODE 0: y = r16(LN(x))                                   (clip/model.py:155-161, model.py:86,88)
// MODE 1: y = r16(r16(ratio*r16(LN(x))) + r16(omr*res))    (Adapter_FC blend, model.py:92-95)
//         followed, if l2norm, by the row normalise of main.py:408-409.
template <int NCH, typename PT, int MODE>
__global__ __launch_bounds__(256) void layernorm_kernel(const half_t* __restrict__ x, int ld_x,
                                                        const PT* __restrict__ gamma, const PT* __restrict__ beta,
                                                        float eps, half_t* __restrict__ y, int R, int D,
                                                        const half_t* __restrict__ res, float ratio, float omr,
                                                        int l2norm, float* __restrict__ sq_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < R; row += gridDim.x * 4) {
        const half_t* xr = x + (size_t)row * ld_x;
        float v[NCH][8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int d = c * 512 + lane * 8;
            if (d < D) {
                half8_t h = ld_half8(xr + d);
#pragma unroll
                for (int j = 0; j < 8; ++j) { v[c][j] = (float)h[j]; s += v[c][j]; }
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
            }
        }
        const float mean = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int d = c * 512 + lane * 8;
            if (d < D) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float t = v[c][j] - mean; q = ln_sq_acc(t, q); }
            }
        }
        const float rstd = 1.f / sqrtf(wave_sum(q) / (float)D + eps);
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int d = c * 512 + lane * 8;
            if (d < D) {
                half8_t rh;
                if (MODE == 1) rh = ld_half8(res + (size_t)row * D + d);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float o = ln_affine(v[c][j], mean, rstd, (float)gamma[d + j], (float)beta[d + j]);
                    o = r16(o);
                    if (MODE == 1) o = r16(r16(ratio * o) + r16(omr * (float)rh[j]));
                    v[c][j] = o;
                    ss += o * o;
                }
            }
        }
        float n = 1.f;
        if (MODE == 1 && l2norm) {
            n = r16(sqrtf(wave_sum(ss)));
            ss = 0.f;
        }
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int d = c * 512 + lane * 8;
            if (d < D) {
                half8_t o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    o[j] = (MODE == 1 && l2norm) ? (half_t)(v[c][j] / n) : (half_t)v[c][j];
                    const float f = (float)o[j];
                    if (MODE == 1 && l2norm) ss += f * f;
                }
                st_half8(y + (size_t)row * D + d, o);
            }
        }
        if (MODE == 1 && sq_out) {
            ss = wave_sum(ss);
            if (lane == 0) sq_out[row] = ss;
        }
    }
}

// The whole-batch LayerNorm pass of the towers (MODE 0, fp32 affine) with the NEXT row's loads requested before the current row's two
// wave reductions (each a chain of six ds_bpermute round trips): one more 16 / 24 bytes per lane in flight, 8 registers.  Same arithmetic
// and summation order per row as layernorm_kernel.
template <int NCH, bool GB_LDS = false>
__global__ __launch_bounds__(256) void layernorm_pf_kernel(const half_t* __restrict__ x, int ld_x, const float* __restrict__ gamma,
                                                           const float* __restrict__ beta, float eps, half_t* __restrict__ y, int R, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6, stride = gridDim.x * 4;
    int row = blockIdx.x * 4 + wave;
    // GB_LDS (the whole-batch pass of the towers): gamma / beta once per workgroup into LDS.  Per row they are 4 x the bytes of the row
    // itself through the vector-memory path (8 dwordx4 loads per lane against 2 for x); as ds_read_b128 they use the LDS pipe instead
    // (256 B/clk against 64): [201 728, 768] 129 -> 111 us, same bits (profiles/r03_ab_ln_lds.txt)
    // (layout: the lane's eight values of a chunk as two 16-byte halves in two PLANES, [plane][chunk * 256 + lane * 4 ..]: a ds_read_b128 of the wave is 1 KB
    // contiguous — with the eight values adjacent (32-byte lane stride) the PMC pass counted 22 % of the LDS cycles as bank conflicts)
    __shared__ __attribute__((aligned(16))) float gb_s[2][2][GB_LDS ? NCH * 256 : 4];
    if (GB_LDS) {
        for (int i = threadIdx.x; i < NCH * 512; i += 256) {
            const int pos = (i >> 9) * 256 + ((i & 511) >> 3) * 4 + (i & 3), plane = (i >> 2) & 1;
            gb_s[0][plane][pos] = i < D ? gamma[i] : 0.f;
            gb_s[1][plane][pos] = i < D ? beta[i] : 0.f;
        }
        __syncthreads();
    }
    half8_t cur[NCH], nxt[NCH];
    auto load = [&](half8_t (&h)[NCH], int r) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (c * 512 + lane * 8 < D) h[c] = ld_half8(x + (size_t)r * ld_x + c * 512 + lane * 8);
    };
    if (row < R) load(cur, row);
    for (; row < R; row += stride) {
        if (row + stride < R) load(nxt, row + stride);
        half8_t o[NCH];
        ln_row_pf<NCH>(cur, D, lane, eps, [&](int c, int j) {
            return GB_LDS ? float2_t{gb_s[0][j >> 2][c * 256 + lane * 4 + (j & 3)], gb_s[1][j >> 2][c * 256 + lane * 4 + (j & 3)]}
                          : float2_t{gamma[c * 512 + lane * 8 + j], beta[c * 512 + lane * 8 + j]};
        }, o);
#pragma unroll
        for (int c = 0; c < NCH; ++c)
            if (c * 512 + lane * 8 < D) st_half8(y + (size_t)row * D + c * 512 + lane * 8, o[c]);
#pragma unroll
        for (int c = 0; c < NCH; ++c) cur[c] = nxt[c];
    }
}


// Residual add fused into the next LayerNorm (clip/model.py:188-189 followed by ln_2 / the next block's ln_1 /
// ln_post / ln_final): xs = r16(x + delta) is (optionally) stored back and y = r16(LN(xs)).  Keeping the
// residual out of the GEMM epilogues lets those run without a single ordinary vector load.
template <int NCH>
__global__ __launch_bounds__(256) void add_layernorm_kernel(const half_t* __restrict__ x, const half_t* __restrict__ delta,
                                                            int ld, half_t* x_out, const float* __restrict__ gamma,
                                                            const float* __restrict__ beta, float eps,
                                                            half_t* __restrict__ y, int R, int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < R; row += gridDim.x * 4) {
        const half_t* xr = x + (size_t)row * ld;
        const half_t* dr = delta + (size_t)row * ld;
        float v[NCH][8];
        float s = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int d = c * 512 + lane * 8;
            if (d < D) {
                const half8_t a = ld_half8(xr + d), b = ld_half8(dr + d);
                half8_t o;
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    o[j] = (half_t)((float)a[j] + (float)b[j]);
                    v[c][j] = (float)o[j];
                    s += v[c][j];
                }
                if (x_out) st_half8(x_out + (size_t)row * ld + d, o);
            } else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
            }
        }
        const float mean = wave_sum(s) / (float)D;
        float q = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int d = c * 512 + lane * 8;
            if (d < D) {
#pragma unroll
                for (int j = 0; j < 8; ++j) { const float t = v[c][j] - mean; q = ln_sq_acc(t, q); }
            }
        }
        const float rstd = 1.f / sqrtf(wave_sum(q) / (float)D + eps);
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int d = c * 512 + lane * 8;
            if (d < D) {
                const float4_t g0 = *reinterpret_cast<const float4_t*>(gamma + d), g1 = *reinterpret_cast<const float4_t*>(gamma + d + 4);
                const float4_t b0 = *reinterpret_cast<const float4_t*>(beta + d), b1 = *reinterpret_cast<const float4_t*>(beta + d + 4);
                half8_t o;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o[j] = (half_t)ln_affine(v[c][j], mean, rstd, g0[j], b0[j]);
                    o[j + 4] = (half_t)ln_affine(v[c][j + 4], mean, rstd, g1[j], b1[j]);
                }
                st_half8(y + (size_t)row * D + d, o);
            }
        }
    }
}

}  // namespace

extern "C" int pclip_layernorm_f16(const void* x, int ld_x, const float* gamma, const float* beta, float eps, void* y,
                                   int R, int D, pclip_stream_t stream) {
    PCLIP_REQUIRE(x && gamma && beta && y, "pclip_layernorm_f16: null pointer");
    PCLIP_REQUIRE(D > 0 && D % 8 == 0 && D <= 4096 && ld_x >= D && ld_x % 8 == 0 && R >= 0,
                  "pclip_layernorm_f16: bad shape R=%d D=%d ld=%d", R, D, ld_x);
    if (R == 0) return PCLIP_OK;
    // whole-batch pass (>= 16 rows per workgroup of a PCLIP_LN_BPC-per-CU grid): gamma / beta from the workgroup's LDS copy
    const int ln_grid = pclip_device_cus() * PCLIP_LN_BPC;          // (0 if the CU count is unknown: the old paths)
    if (PCLIP_LN_PF && PCLIP_LN_LDS && ln_grid > 0 && R >= 16 * ln_grid) {
        DISPATCH_NCH(D, (layernorm_pf_kernel<NCH, true><<<ln_grid, 256, 0, (hipStream_t)stream>>>((const half_t*)x, ld_x, gamma, beta, eps, (half_t*)y, R, D)));
        return pclip_check_launch("layernorm");
    }
    if (PCLIP_LN_PF && R > 4 * 16384) {                      // more rows than waves in the grid: the row loop iterates, prefetch pays
        DISPATCH_NCH(D, (layernorm_pf_kernel<NCH, false><<<row_grid(R), 256, 0, (hipStream_t)stream>>>((const half_t*)x, ld_x, gamma, beta, eps, (half_t*)y, R, D)));
        return pclip_check_launch("layernorm");
    }
    DISPATCH_NCH(D, (layernorm_kernel<NCH, float, 0><<<row_grid(R), 256, 0, (hipStream_t)stream>>>(
                        (const half_t*)x, ld_x, gamma, beta, eps, (half_t*)y, R, D, nullptr, 0.f, 0.f, 0, nullptr)));
    return pclip_check_launch("layernorm");
}

extern "C" int pclip_add_layernorm_f16(const void* x, const void* delta, int ld, void* x_out, const float* gamma,
                                       const float* beta, float eps, void* y, int R, int D, pclip_stream_t stream) {
    PCLIP_REQUIRE(x && delta && gamma && beta && y, "pclip_add_layernorm_f16: null pointer");
    PCLIP_REQUIRE(D > 0 && D % 8 == 0 && D <= 4096 && ld >= D && ld % 8 == 0 && R >= 0,
                  "pclip_add_layernorm_f16: bad shape R=%d D=%d ld=%d", R, D, ld);
    if (R == 0) return PCLIP_OK;
    DISPATCH_NCH(D, (add_layernorm_kernel<NCH><<<row_grid(R), 256, 0, (hipStream_t)stream>>>(
                        (const half_t*)x, (const half_t*)delta, ld, (half_t*)x_out, gamma, beta, eps, (half_t*)y, R, D)));
    return pclip_check_launch("add_layernorm");
}

// internal (used by pclip_adapter.hip): LayerNorm with fp16 affine parameters, optional Adapter_FC blend
int pclip_layernorm_f16p(const void* x, const void* gamma, const void* beta, float eps, void* y, int R, int D,
                         const void* res, float ratio, float omr, int l2norm, float* sq_out, hipStream_t s) {
    if (R == 0) return PCLIP_OK;
    if (res) {
        DISPATCH_NCH(D, (layernorm_kernel<NCH, half_t, 1><<<row_grid(R), 256, 0, s>>>(
                            (const half_t*)x, D, (const half_t*)gamma, (const half_t*)beta, eps, (half_t*)y, R, D,
                            (const half_t*)res, ratio, omr, l2norm, sq_out)));
    } else {
        DISPATCH_NCH(D, (layernorm_kernel<NCH, half_t, 0><<<row_grid(R), 256, 0, s>>>(
                            (const half_t*)x, D, (const half_t*)gamma, (const half_t*)beta, eps, (half_t*)y, R, D,
                            nullptr, 0.f, 0.f, 0, nullptr)));
    }
    return pclip_check_launch("layernorm_f16p");
}

// Kernel choice of pclip_attention_*: mode -1 automatic (whole batches take the persistent kernel), 0 never, 1 always (when its
// shape conditions hold); max_grid > 0 caps its grid (tests: several items per workgroup on small problems).