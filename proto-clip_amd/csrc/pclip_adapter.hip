// Query adapters: model.py:12-95.
//  - Adapter_FC: two skinny MFMA GEMMs + two LayerNorms, the 0.2/0.8 blend and the following row
//    normalise fused into the second LayerNorm pass.
//  - Adapter (conv-2x / conv-3x): one workgroup per feature vector; the whole [16, s, s] activation
//    stack lives in LDS (never in HBM — the reference materialises 5 such tensors per row), the 3x3
//    convolution runs on packed-fp16 dot products with fp32 accumulation, and the three whole-tensor
//    LayerNorms ([C,s,s]-shaped affine, model.py:37-45) are block reductions.
#include "pclip_common.h"
#include <stdlib.h>
#include <type_traits>

int pclip_layernorm_f16p(const void* x, const void* gamma, const void* beta, float eps, void* y, int R, int D,
                         const void* res, float ratio, float omr, int l2norm, float* sq_out, hipStream_t s);

namespace {

constexpr int CW = 16;          // adapter width (model.py:23)

__device__ __forceinline__ float block_sum(float v, float* red) {
    v = wave_sum(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    __syncthreads();                       // protect red[] from the previous use
    if (lane == 0) red[wave] = v;
    __syncthreads();
    return red[0] + red[1] + red[2] + red[3];
}

// LDS layout (bytes), s2 = s*s, hp = (s+2)*(s+2):
//   xs   float[1024]                      padded input row
//   a1p  half2[8][hp]                     LN1 output, channel pairs, zero halo (conv-3x only)
//   t2   half[16][s2]                     conv2 output before LN2 (conv-3x only)
//   w2p  half2[8][9][16]                  conv2 weights as (ci, ci+1) pairs
//   u    float[1024]                      conv3 output
//   red  float[4]
template <bool THREE_X>
__global__ __launch_bounds__(256) void adapter_conv_kernel(const half_t* __restrict__ x, int D, int s,
                                                           const half_t* __restrict__ conv1, const half_t* __restrict__ ln1w,
                                                           const half_t* __restrict__ ln1b, const half_t* __restrict__ conv2,
                                                           const half_t* __restrict__ ln2w, const half_t* __restrict__ ln2b,
                                                           const half_t* __restrict__ conv3, const half_t* __restrict__ ln3w,
                                                           const half_t* __restrict__ ln3b, int l2norm,
                                                           half_t* __restrict__ y, float* __restrict__ y_sq) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int s2 = s * s, sp = s + 2, hp = sp * sp;
    float* xs = reinterpret_cast<float*>(smem);
    float* u = xs + 1024;
    float* red = u + 1024;
    half2_t* a1p = reinterpret_cast<half2_t*>(red + 4);
    half_t* t2 = reinterpret_cast<half_t*>(a1p + 8 * hp);
    half2_t* w2p = reinterpret_cast<half2_t*>(t2 + CW * s2);
    const int tid = threadIdx.x;
    const size_t row = blockIdx.x;
    const float eps = 1e-5f;

    for (int p = tid; p < s2; p += 256) xs[p] = p < D ? (float)x[row * D + p] : 0.f;
    float w1[CW], w3[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) { w1[c] = (float)conv1[c]; w3[c] = (float)conv3[c]; }
    if (THREE_X) {
        for (int i = tid; i < 8 * hp; i += 256) a1p[i] = half2_t{(half_t)0.f, (half_t)0.f};
        for (int i = tid; i < 8 * 9 * CW; i += 256) {
            const int co = i % CW, tap = (i / CW) % 9, cp = i / (CW * 9);
            w2p[i] = half2_t{conv2[(co * CW + 2 * cp) * 9 + tap], conv2[(co * CW + 2 * cp + 1) * 9 + tap]};
        }
    }
    __syncthreads();

    // conv1 (1x1, 1 -> 16, model.py:63) + LN1 over [16, s, s] (model.py:64): t = r16(w1[c] * x[p])
    const int n1 = CW * s2;
    float sm = 0.f;
    for (int p = tid; p < s2; p += 256) {
        const float xv = xs[p];
#pragma unroll
        for (int c = 0; c < CW; ++c) sm += r16(w1[c] * xv);
    }
    const float mean1 = block_sum(sm, red) / (float)n1;
    float sq = 0.f;
    for (int p = tid; p < s2; p += 256) {
        const float xv = xs[p];
#pragma unroll
        for (int c = 0; c < CW; ++c) { const float t = r16(w1[c] * xv) - mean1; sq += t * t; }
    }
    const float rstd1 = 1.f / sqrtf(block_sum(sq, red) / (float)n1 + eps);

    if (THREE_X) {
        // a1 = r16(LN1) into the halo buffer as channel pairs
        for (int p = tid; p < s2; p += 256) {
            const float xv = xs[p];
            const int py = p / s, px = p - py * s;
            const int hpos = (py + 1) * sp + px + 1;
#pragma unroll
            for (int cp = 0; cp < 8; ++cp) {
                const int c0 = 2 * cp, c1 = c0 + 1;
                const float a0 = (r16(w1[c0] * xv) - mean1) * rstd1 * (float)ln1w[c0 * s2 + p] + (float)ln1b[c0 * s2 + p];
                const float a1 = (r16(w1[c1] * xv) - mean1) * rstd1 * (float)ln1w[c1 * s2 + p] + (float)ln1b[c1 * s2 + p];
                a1p[cp * hp + hpos] = half2_t{(half_t)a0, (half_t)a1};
            }
        }
        __syncthreads();
        // conv2 3x3 pad 1, 16 -> 16 (model.py:67): packed-fp16 dot2 with fp32 accumulate
        float sm2 = 0.f;
        for (int p = tid; p < s2; p += 256) {
            const int py = p / s, px = p - py * s;
            float acc[CW];
#pragma unroll
            for (int c = 0; c < CW; ++c) acc[c] = 0.f;
#pragma unroll 1
            for (int cp = 0; cp < 8; ++cp) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dy = tap / 3, dx = tap - dy * 3;
                    const half2_t a = a1p[cp * hp + (py + dy) * sp + px + dx];
                    const half2_t* wv = w2p + (cp * 9 + tap) * CW;
#pragma unroll
                    for (int co = 0; co < CW; ++co) acc[co] = __builtin_amdgcn_fdot2(a, wv[co], acc[co], false);
                }
            }
#pragma unroll
            for (int co = 0; co < CW; ++co) {
                const half_t h = (half_t)acc[co];
                t2[co * s2 + p] = h;
                sm2 += (float)h;
            }
        }
        const float mean2 = block_sum(sm2, red) / (float)n1;   // block_sum's barriers also publish t2
        float sq2 = 0.f;
        for (int i = tid; i < n1; i += 256) { const float t = (float)t2[i] - mean2; sq2 += t * t; }
        const float rstd2 = 1.f / sqrtf(block_sum(sq2, red) / (float)n1 + eps);
        // conv3 (1x1, 16 -> 1, model.py:70) on a2 = r16(LN2(t2))
        for (int p = tid; p < s2; p += 256) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                const float a2 = r16(((float)t2[c * s2 + p] - mean2) * rstd2 * (float)ln2w[c * s2 + p] + (float)ln2b[c * s2 + p]);
                acc = fmaf(w3[c], a2, acc);
            }
            u[p] = r16(acc);
        }
    } else {
        // conv-2x: conv3 directly on a1 = r16(LN1)
        for (int p = tid; p < s2; p += 256) {
            const float xv = xs[p];
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                const float a1 = r16((r16(w1[c] * xv) - mean1) * rstd1 * (float)ln1w[c * s2 + p] + (float)ln1b[c * s2 + p]);
                acc = fmaf(w3[c], a1, acc);
            }
            u[p] = r16(acc);
        }
    }
    // LN3 over [1, s, s] (model.py:71), + identity (model.py:73), crop to D (model.py:75-76)
    float s3 = 0.f;
    for (int p = tid; p < s2; p += 256) s3 += u[p];       // own writes only: no barrier needed yet
    const float mean3 = block_sum(s3, red) / (float)s2;
    float q3 = 0.f;
    for (int p = tid; p < s2; p += 256) { const float t = u[p] - mean3; q3 += t * t; }
    const float rstd3 = 1.f / sqrtf(block_sum(q3, red) / (float)s2 + eps);
    float ss = 0.f;
    for (int p = tid; p < D; p += 256) {
        const float o = r16((u[p] - mean3) * rstd3 * (float)ln3w[p] + (float)ln3b[p]);
        const float v = r16(o + xs[p]);
        u[p] = v;
        ss += v * v;
    }
    if (l2norm) {
        const float n = r16(sqrtf(block_sum(ss, red)));
        ss = 0.f;
        for (int p = tid; p < D; p += 256) {
            const half_t h = (half_t)(u[p] / n);
            y[row * D + p] = h;
            ss += (float)h * (float)h;
        }
    } else {
        for (int p = tid; p < D; p += 256) y[row * D + p] = (half_t)u[p];
    }
    if (y_sq) {
        const float t = block_sum(ss, red);
        if (tid == 0) y_sq[row] = t;
    }
}

// ---- conv-3x forward with the 3x3 convolution on the matrix pipe ---------------------------------------------------------------
// adapter_conv_kernel<true> spends its time in 609 k packed dot products per row on the VALU (37 TFLOP/s: PMC VALU busy 43 %, LDS-bound)
// and re-reads 68 KB of LayerNorm parameters per row from L2.  Here
//  * conv2 (3x3, 16 -> 16 on s x s, model.py:67) is an implicit GEMM per row: D[co, p] = sum_k W[co, k] A[k, p], k = tap * 16 + ci (9 taps,
//    padded to K = 160 = 5 k-steps of v_mfma_f32_16x16x32_f16).  The A operand of the instruction is the WEIGHT fragment (16 co x 32 k,
//    20 registers per lane, loaded once per workgroup), the B operand the pixels: LN1's output lives in LDS as [halo pixel][16 channels]
//    (32 B per pixel, zero halo), so a lane's fragment — 8 consecutive channels of pixel (y + dy, x + dx) — is ONE ds_read_b128.
//    A 16-pixel tile costs 5 reads + 5 MFMAs; a wave owns tiles wave, wave + 4, ...; a lane ends up with channels 4q .. 4q+3 of pixel
//    (tile, lane & 15) in four fp32 registers per tile — t2 never touches memory: LN2's two passes, its affine and conv3 run on them.
//  * the workgroup is PERSISTENT over rows and every thread keeps the LayerNorm parameters of the (channel, pixel) positions it owns in
//    registers (NT <= PREG_MAX): no parameter traffic per row at all.
//  * conv1 + LN1 + LN3 + residual + row-normalise run in a row-linear layout (thread t owns pixels t, t + 256, ...): x comes straight
//    from global memory into registers, r16(w1[c] x) is a packed fp16 multiply (one rounding of the exact product, as r16 of the fp32 product).
// Rounding points as adapter_conv_kernel: conv outputs r16 of an fp32 accumulation, LayerNorms fp32 statistics over
// the fp16 tensor -> affine -> r16.  What differs is fp32 SUMMATION ORDER (MFMA's k order; conv3's 16 channels as four in-lane chains
// + (q0 + q1) + (q2 + q3); block sums): <= 1 fp16 ulp on isolated elements, inside every adapter tolerance (tests/test_gpu_parity.py).
typedef float float4v_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ float pair_sum16(float v) {      // v + (lane ^ 16)'s v   (v_permlane16_swap: VALU, no LDS crossbar)
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned a = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(a, a, false, false);
    const unsigned x = r[0], y = r[1];
    return __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
#else
    return v;
#endif
}
__device__ __forceinline__ float pair_sum32(float v) {      // v + (lane ^ 32)'s v
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned a = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    const unsigned x = r[0], y = r[1];
    return __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
#else
    return v;
#endif
}

// Sum over the 64 lanes on the VALU (DPP inside a 16-lane row, then v_permlane16_swap / v_permlane32_swap) instead of six ds_bpermute round
// trips through the LDS crossbar (__shfl_xor): every lane ends with the same total.  After each level the lanes of an aligned group hold the
// group's sum (a + b == b + a bit for bit), so the mirror patterns supply the partner group's value (pclip_encoder.hip stats_level).
__device__ __forceinline__ float wave_sum_valu(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    auto dpp = [](float x, auto ctrl) {
        const int i = __builtin_bit_cast(int, x);
        return x + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, decltype(ctrl)::value, 0xF, 0xF, false));
    };
    v = dpp(v, std::integral_constant<int, 0xB1>());      // quad_perm [1,0,3,2]: lane ^ 1
    v = dpp(v, std::integral_constant<int, 0x4E>());      // quad_perm [2,3,0,1]: lane ^ 2
    v = dpp(v, std::integral_constant<int, 0x141>());     // row_half_mirror: the other quad of the 8
    v = dpp(v, std::integral_constant<int, 0x140>());     // row_mirror: the other half of the 16
    return pair_sum32(pair_sum16(v));
#else
    return v;
#endif
}
// Block sum on ONE barrier: the per-wave totals go to red[par][wave] with `par` alternating call by call — the slot written two calls later is
// separated from this call's readers by the barrier of the call in between.
__device__ __forceinline__ float block_sum1(float v, float* red /* [2][4] */, int& par) {
    v = wave_sum_valu(v);
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    float* r = red + 4 * par;
    par ^= 1;
    if (lane == 0) r[wave] = v;
    __syncthreads();
    return (r[0] + r[1]) + (r[2] + r[3]);
}

constexpr int TWO_WG_MAX = 9;    // NT up to which two workgroups share a CU (256 registers per lane: D <= 576); larger rows get the whole register file (one per CU)

template <int NT>                // 64-pixel groups of the row: s2 <= 64 * NT; a wave owns NT 16-pixel tiles, a thread (NT + 3) / 4 pixels of the linear passes
__global__ __launch_bounds__(256, NT <= TWO_WG_MAX ? 2 : 1) void adapter_conv3x_mfma_kernel(const half_t* __restrict__ x, int B, int D, int s,
                                                                     const half_t* __restrict__ conv1, const half_t* __restrict__ ln1w,
                                                                     const half_t* __restrict__ ln1b, const half_t* __restrict__ conv2,
                                                                     const half_t* __restrict__ ln2w, const half_t* __restrict__ ln2b,
                                                                     const half_t* __restrict__ conv3, const half_t* __restrict__ ln3w,
                                                                     const half_t* __restrict__ ln3b, int l2norm,
                                                                     half_t* __restrict__ y, float* __restrict__ y_sq) {
    constexpr int NPX = (NT + 3) / 4;
    constexpr bool PREG = true;      // LayerNorm parameters of the thread's positions in registers (false: re-read per row, A/B only)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int s2 = s * s, sp = s + 2, hp = sp * sp, n1 = CW * s2;
    half_t* a1h = reinterpret_cast<half_t*>(smem);                               // [hp][16] fp16: LN1 output, channels contiguous, zero halo
    float* u = reinterpret_cast<float*>(smem + ((hp * 32 + 15) & ~15));          // [256 * NPX] conv3 output
    float* red = u + 256 * NPX;                                                  // [2][4]
    int par = 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4;
    const float eps = 1e-5f;

    // ---- once per workgroup ----
    for (int i = tid; i < hp * 8; i += 256) reinterpret_cast<unsigned*>(a1h)[i] = 0u;          // the halo stays zero: rows only rewrite the interior
    half8_t wf[5];                                                                              // conv2 as the MFMA A operand: row co = lane & 15, k = 32 ks + 8 q ..
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const int tap = 2 * ks + (q >> 1), ci0 = 8 * (q & 1), co = lane & 15;
#pragma unroll
        for (int j = 0; j < 8; ++j) wf[ks][j] = tap < 9 ? conv2[(co * CW + ci0 + j) * 9 + tap] : (half_t)0.f;
    }
    int tapoff[5];                                                                              // byte offset of this lane's tap (and channel half) inside a1h
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const int tap = 2 * ks + (q >> 1) < 9 ? 2 * ks + (q >> 1) : 8;                        // k >= 144: zero weights on finite data
        tapoff[ks] = ((tap / 3) * sp + tap % 3) * 32 + (q & 1) * 16;
    }
    half2_t w1p[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) w1p[c] = half2_t{conv1[2 * c], conv1[2 * c + 1]};
    float w3q[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) w3q[r] = (float)conv3[4 * q + r];
    // tiles of this wave: mt = wave + 4 i, pixel p = 16 mt + (lane & 15)
    int poff[NT];
    const int p0 = 16 * wave + (lane & 15);                                                    // pixel of tile 0; tile i: p0 + 64 i
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int pp = p0 + 64 * i;
        const int pc = pp < s2 ? pp : 0, py = pc / s, px = pc - py * s;
        poff[i] = (py * sp + px) * 32;                                                          // tap (0, 0) of pixel pc: halo position (py + 0, px + 0)
    }
    half2_t g1r[PREG ? NPX : 1][8], b1r[PREG ? NPX : 1][8];                                     // LN1 parameters of the thread's pixels, channel pairs
    half2_t g2r[PREG ? NT : 1][2], b2r[PREG ? NT : 1][2];                                       // LN2: channels (4q, 4q+1), (4q+2, 4q+3) of pixel (tile i, lane & 15)
    half2_t gb3r[PREG ? NPX : 1];                                                               // LN3: (gamma, beta)
    if (PREG) {
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const int pp = tid + 256 * j, pc = pp < s2 ? pp : 0;
#pragma unroll
            for (int c = 0; c < 8; ++c) {
                g1r[j][c] = half2_t{ln1w[(2 * c) * s2 + pc], ln1w[(2 * c + 1) * s2 + pc]};
                b1r[j][c] = half2_t{ln1b[(2 * c) * s2 + pc], ln1b[(2 * c + 1) * s2 + pc]};
            }
            gb3r[j] = half2_t{ln3w[pc], ln3b[pc]};
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int pp = p0 + 64 * i, pc = pp < s2 ? pp : 0;
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                g2r[i][r] = half2_t{ln2w[(4 * q + 2 * r) * s2 + pc], ln2w[(4 * q + 2 * r + 1) * s2 + pc]};
                b2r[i][r] = half2_t{ln2b[(4 * q + 2 * r) * s2 + pc], ln2b[(4 * q + 2 * r + 1) * s2 + pc]};
            }
        }
    }
    __syncthreads();

    // the NEXT row's values are requested a whole row ahead: a row is 1 KB straight from HBM, and its ~2 us would otherwise open every iteration
    half_t xnext[NPX];
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        const int pp = tid + 256 * j;
        xnext[j] = (pp < D && (int)blockIdx.x < B) ? x[(size_t)blockIdx.x * D + pp] : (half_t)0.f;
    }
    for (int row = blockIdx.x; row < B; row += gridDim.x) {
#if defined(__HIP_DEVICE_COMPILE__)
        // the packed parameters are made opaque once per row: hipcc otherwise hoists their fp16 -> fp32 conversions out of the row loop and keeps
        // TWICE the registers live (256 + 188 B of scratch at D = 512)
        if (PREG) {
#pragma unroll
            for (int j = 0; j < NPX; ++j) {
#pragma unroll
                for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(g1r[j][c]), "+v"(b1r[j][c]));
                asm volatile("" : "+v"(gb3r[j]));
            }
#pragma unroll
            for (int i = 0; i < NT; ++i) asm volatile("" : "+v"(g2r[i][0]), "+v"(g2r[i][1]), "+v"(b2r[i][0]), "+v"(b2r[i][1]));
        }
#endif
        // ---- conv1 (1x1, 1 -> 16, model.py:63): t1 = r16(w1[c] x[p]) as packed fp16 products; LN1 statistics over [16, s, s] (model.py:64) ----
        // (t1 is recomputed in each of the three passes: 8 packed multiplies per pixel against 8 registers per pixel held across two block reductions)
        half_t xh[NPX];
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const int pp = tid + 256 * j;
            xh[j] = xnext[j];
            const int nrow = row + (int)gridDim.x;
            xnext[j] = (pp < D && nrow < B) ? x[(size_t)nrow * D + pp] : (half_t)0.f;
            const half2_t xx = {xh[j], xh[j]};
            if (pp < s2) {
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const half2_t t = w1p[c] * xx;                                               // v_pk_mul_f16: the exact product rounded once
                    sm += (float)t[0] + (float)t[1];
                }
            }
        }
        const float mean1 = block_sum1(sm, red, par) / (float)n1;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < NPX; ++j)
            if (tid + 256 * j < s2) {
                const half2_t xx = {xh[j], xh[j]};
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const half2_t t = w1p[c] * xx;
                    const float d0 = (float)t[0] - mean1, d1 = (float)t[1] - mean1;
                    sq = fmaf(d0, d0, sq);
                    sq = fmaf(d1, d1, sq);
                }
            }
        const float rstd1 = 1.f / sqrtf(block_sum1(sq, red, par) / (float)n1 + eps);
        // a1 = r16(LN1(t1)) -> a1h interior, 16 channels (32 B) per pixel
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const int pp = tid + 256 * j;
            if (pp < s2) {
                const int py = pp / s, px = pp - py * s;
                const half2_t xx = {xh[j], xh[j]};
                half2_t o[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const half2_t gg = PREG ? g1r[j][c] : half2_t{ln1w[(2 * c) * s2 + pp], ln1w[(2 * c + 1) * s2 + pp]};
                    const half2_t bb = PREG ? b1r[j][c] : half2_t{ln1b[(2 * c) * s2 + pp], ln1b[(2 * c + 1) * s2 + pp]};
                    const half2_t t = w1p[c] * xx;
                    const float a0 = ((float)t[0] - mean1) * rstd1 * (float)gg[0] + (float)bb[0];
                    const float a1 = ((float)t[1] - mean1) * rstd1 * (float)gg[1] + (float)bb[1];
                    o[c] = half2_t{(half_t)a0, (half_t)a1};
                }
                half8_t* dst = reinterpret_cast<half8_t*>(a1h + ((py + 1) * sp + px + 1) * CW);
                dst[0] = half8_t{o[0][0], o[0][1], o[1][0], o[1][1], o[2][0], o[2][1], o[3][0], o[3][1]};
                dst[1] = half8_t{o[4][0], o[4][1], o[5][0], o[5][1], o[6][0], o[6][1], o[7][0], o[7][1]};
            }
        }
        __syncthreads();
        // ---- conv2 on the matrix pipe: acc[i][r] = t2 pre-rounding, channel 4q + r of pixel (tile i, lane & 15) ----
        float4v_t acc[NT];
        const char* abase = reinterpret_cast<const char*>(a1h);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            acc[i] = float4v_t{0.f, 0.f, 0.f, 0.f};
            if (16 * wave + 64 * i < s2) {                                                       // wave-uniform: the tile exists
#pragma unroll
                for (int ks = 0; ks < 5; ++ks) {
                    const half8_t bfrag = *reinterpret_cast<const half8_t*>(abase + poff[i] + tapoff[ks]);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks], bfrag, acc[i], 0, 0, 0);
                }
            }
        }
        // t2 = r16(acc); LN2 statistics over [16, s, s] (model.py:68)
        float sm2 = 0.f;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[i][r] = r16(acc[i][r]);
                if (p0 + 64 * i < s2) sm2 += acc[i][r];
            }
        const float mean2 = block_sum1(sm2, red, par) / (float)n1;
        float sq2 = 0.f;
#pragma unroll
        for (int i = 0; i < NT; ++i)
            if (p0 + 64 * i < s2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = acc[i][r] - mean2; sq2 = fmaf(d, d, sq2); }
            }
        const float rstd2 = 1.f / sqrtf(block_sum1(sq2, red, par) / (float)n1 + eps);
        // a2 = r16(LN2(t2)); conv3 (1x1, 16 -> 1, model.py:70): four channels in the lane, then the four lane groups
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int pp = p0 + 64 * i, pc = pp < s2 ? pp : 0;
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float gg = PREG ? (float)g2r[i][r >> 1][r & 1] : (float)ln2w[(4 * q + r) * s2 + pc];
                const float bb = PREG ? (float)b2r[i][r >> 1][r & 1] : (float)ln2b[(4 * q + r) * s2 + pc];
                const float a2 = r16((acc[i][r] - mean2) * rstd2 * gg + bb);
                v = fmaf(w3q[r], a2, v);
            }
            v = pair_sum32(pair_sum16(v));                                                       // (q0 + q1) + (q2 + q3), every lane of the four
            if (q == 0 && pp < s2) u[pp] = r16(v);
        }
        __syncthreads();                                                                         // u complete; every conv2 read of a1h is done
        // ---- LN3 over [1, s, s] (model.py:71), + identity (model.py:73), crop to D (model.py:75-76), row normalise (main.py:408-409) ----
        float uu[NPX];
        float s3 = 0.f;
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const int pp = tid + 256 * j;
            uu[j] = pp < s2 ? u[pp] : 0.f;
            s3 += uu[j];
        }
        const float mean3 = block_sum1(s3, red, par) / (float)s2;
        float q3 = 0.f;
#pragma unroll
        for (int j = 0; j < NPX; ++j)
            if (tid + 256 * j < s2) { const float d = uu[j] - mean3; q3 = fmaf(d, d, q3); }
        const float rstd3 = 1.f / sqrtf(block_sum1(q3, red, par) / (float)s2 + eps);
        float ss = 0.f, vv[NPX];
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const int pp = tid + 256 * j;
            vv[j] = 0.f;
            if (pp < D) {
                const float gg = PREG ? (float)gb3r[j][0] : (float)ln3w[pp], bb = PREG ? (float)gb3r[j][1] : (float)ln3b[pp];
                const float o = r16((uu[j] - mean3) * rstd3 * gg + bb);
                vv[j] = r16(o + (float)xh[j]);
                ss += vv[j] * vv[j];
            }
        }
        if (l2norm) {
            const float n = r16(sqrtf(block_sum1(ss, red, par)));
            ss = 0.f;
#pragma unroll
            for (int j = 0; j < NPX; ++j) {
                const int pp = tid + 256 * j;
                if (pp < D) {
                    const half_t h = (half_t)(vv[j] / n);
                    y[(size_t)row * D + pp] = h;
                    ss += (float)h * (float)h;
                }
            }
        } else {
#pragma unroll
            for (int j = 0; j < NPX; ++j)
                if (tid + 256 * j < D) y[(size_t)row * D + tid + 256 * j] = (half_t)vv[j];
        }
        if (y_sq) {
            const float t = block_sum1(ss, red, par);
            if (tid == 0) y_sq[row] = t;
        }
    }
}

// ---- backward of the conv adapter for the training step (main.py:267 `adapter(zq_imgs)`; autograd in the reference) ----
// One workgroup per row again: the forward is recomputed into LDS (nothing was saved), then the chain
//   +identity <- LN3 <- conv3 <- [LN2 <- conv2 <-] LN1 <- conv1
// is walked backwards with every gradient tensor rounded to fp16 where autograd materialises one.  The input rows are
// constants (main.py:266), so only parameter gradients leave the kernel, as PER-ROW contributions (fp32) that the host
// reduces over rows with pclip_colsum_f32 — deterministic, no atomics:
//   pw1/pw3 [B,16], pw2 [B,2304], pg1/pb1/pg2/pb2 [B,16*s2], pg3/pb3 [B,s2]   (dgamma | dbeta of the three LayerNorms)
// Extra LDS over the forward: dt2 as channel pairs with a zero halo (the transposed 3x3 convolution reads it the way the
// forward reads a1) and conv2's weights paired over the OUTPUT channel.
__device__ __forceinline__ void block_sum16(float (&v)[CW], float* red /* [4][16] */) {
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
#pragma unroll
    for (int c = 0; c < CW; ++c) v[c] = wave_sum(v[c]);
    __syncthreads();
    if (lane == 0) {
#pragma unroll
        for (int c = 0; c < CW; ++c) red[wave * CW + c] = v[c];
    }
    __syncthreads();
#pragma unroll
    for (int c = 0; c < CW; ++c) v[c] = red[c] + red[CW + c] + red[2 * CW + c] + red[3 * CW + c];
}

template <bool THREE_X>
__global__ __launch_bounds__(256) void adapter_conv_backward_kernel(
    const half_t* __restrict__ x, const half_t* __restrict__ g, int D, int s, const half_t* __restrict__ conv1,
    const half_t* __restrict__ ln1w, const half_t* __restrict__ ln1b, const half_t* __restrict__ conv2,
    const half_t* __restrict__ ln2w, const half_t* __restrict__ ln2b, const half_t* __restrict__ conv3,
    const half_t* __restrict__ ln3w, float* __restrict__ pw1, float* __restrict__ pw2, float* __restrict__ pw3,
    float* __restrict__ pg1, float* __restrict__ pb1, float* __restrict__ pg2, float* __restrict__ pb2,
    float* __restrict__ pg3, float* __restrict__ pb3) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int s2 = s * s, sp = s + 2, hp = sp * sp, n1 = CW * s2;
    float* xs = reinterpret_cast<float*>(smem);
    float* u = xs + 1024;
    float* red = u + 1024;                                       // [4] scalar reductions, then [64] for block_sum16
    float* red16 = red + 4;
    half2_t* a1p = reinterpret_cast<half2_t*>(red16 + 64);
    half_t* t2 = reinterpret_cast<half_t*>(a1p + 8 * hp);        // conv2 output, later reused for da1
    half2_t* d2p = reinterpret_cast<half2_t*>(t2 + CW * s2);     // dt2, channel pairs over co, zero halo
    half2_t* w2p = d2p + 8 * hp;                                 // [ci pair][tap][co]
    half2_t* w2t = w2p + 8 * 9 * CW;                             // [co pair][tap][ci]
    const int tid = threadIdx.x;
    const size_t row = blockIdx.x;
    const float eps = 1e-5f;

    for (int p = tid; p < s2; p += 256) xs[p] = p < D ? (float)x[row * D + p] : 0.f;
    float w1[CW], w3[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) { w1[c] = (float)conv1[c]; w3[c] = (float)conv3[c]; }
    if (THREE_X) {
        for (int i = tid; i < 8 * hp; i += 256) { a1p[i] = half2_t{(half_t)0.f, (half_t)0.f}; d2p[i] = half2_t{(half_t)0.f, (half_t)0.f}; }
        for (int i = tid; i < 8 * 9 * CW; i += 256) {
            const int o = i % CW, tap = (i / CW) % 9, pr = i / (CW * 9);
            w2p[i] = half2_t{conv2[(o * CW + 2 * pr) * 9 + tap], conv2[(o * CW + 2 * pr + 1) * 9 + tap]};          // o = co, pr = ci pair
            w2t[i] = half2_t{conv2[((2 * pr) * CW + o) * 9 + tap], conv2[((2 * pr + 1) * CW + o) * 9 + tap]};      // o = ci, pr = co pair
        }
    }
    __syncthreads();

    // ---------------- forward recomputation (identical arithmetic to adapter_conv_kernel) ----------------
    float sm = 0.f;
    for (int p = tid; p < s2; p += 256) {
        const float xv = xs[p];
#pragma unroll
        for (int c = 0; c < CW; ++c) sm += r16(w1[c] * xv);
    }
    const float mean1 = block_sum(sm, red) / (float)n1;
    float sq = 0.f;
    for (int p = tid; p < s2; p += 256) {
        const float xv = xs[p];
#pragma unroll
        for (int c = 0; c < CW; ++c) { const float t = r16(w1[c] * xv) - mean1; sq += t * t; }
    }
    const float rstd1 = 1.f / sqrtf(block_sum(sq, red) / (float)n1 + eps);
    float mean2 = 0.f, rstd2 = 0.f;
    if (THREE_X) {
        for (int p = tid; p < s2; p += 256) {
            const float xv = xs[p];
            const int py = p / s, px = p - py * s;
            const int hpos = (py + 1) * sp + px + 1;
#pragma unroll
            for (int cp = 0; cp < 8; ++cp) {
                const int c0 = 2 * cp, c1 = c0 + 1;
                const float a0 = (r16(w1[c0] * xv) - mean1) * rstd1 * (float)ln1w[c0 * s2 + p] + (float)ln1b[c0 * s2 + p];
                const float a1 = (r16(w1[c1] * xv) - mean1) * rstd1 * (float)ln1w[c1 * s2 + p] + (float)ln1b[c1 * s2 + p];
                a1p[cp * hp + hpos] = half2_t{(half_t)a0, (half_t)a1};
            }
        }
        __syncthreads();
        float sm2 = 0.f;
        for (int p = tid; p < s2; p += 256) {
            const int py = p / s, px = p - py * s;
            float acc[CW];
#pragma unroll
            for (int c = 0; c < CW; ++c) acc[c] = 0.f;
#pragma unroll 1
            for (int cp = 0; cp < 8; ++cp) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dy = tap / 3, dx = tap - dy * 3;
                    const half2_t a = a1p[cp * hp + (py + dy) * sp + px + dx];
                    const half2_t* wv = w2p + (cp * 9 + tap) * CW;
#pragma unroll
                    for (int co = 0; co < CW; ++co) acc[co] = __builtin_amdgcn_fdot2(a, wv[co], acc[co], false);
                }
            }
#pragma unroll
            for (int co = 0; co < CW; ++co) {
                const half_t h = (half_t)acc[co];
                t2[co * s2 + p] = h;
                sm2 += (float)h;
            }
        }
        mean2 = block_sum(sm2, red) / (float)n1;
        float sq2 = 0.f;
        for (int i = tid; i < n1; i += 256) { const float t = (float)t2[i] - mean2; sq2 += t * t; }
        rstd2 = 1.f / sqrtf(block_sum(sq2, red) / (float)n1 + eps);
        for (int p = tid; p < s2; p += 256) {
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                const float a2 = r16(((float)t2[c * s2 + p] - mean2) * rstd2 * (float)ln2w[c * s2 + p] + (float)ln2b[c * s2 + p]);
                acc = fmaf(w3[c], a2, acc);
            }
            u[p] = r16(acc);
        }
    } else {
        for (int p = tid; p < s2; p += 256) {
            const float xv = xs[p];
            float acc = 0.f;
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                const float a1 = r16((r16(w1[c] * xv) - mean1) * rstd1 * (float)ln1w[c * s2 + p] + (float)ln1b[c * s2 + p]);
                acc = fmaf(w3[c], a1, acc);
            }
            u[p] = r16(acc);
        }
    }
    float s3 = 0.f;
    for (int p = tid; p < s2; p += 256) s3 += u[p];
    const float mean3 = block_sum(s3, red) / (float)s2;
    float q3 = 0.f;
    for (int p = tid; p < s2; p += 256) { const float t = u[p] - mean3; q3 += t * t; }
    const float rstd3 = 1.f / sqrtf(block_sum(q3, red) / (float)s2 + eps);

    // ---------------- LN3 backward: upstream = g on the first D positions (crop + identity add pass it through) ----------
    float sa = 0.f, sb = 0.f;
    for (int p = tid; p < s2; p += 256) {
        const float go = p < D ? (float)g[row * D + p] : 0.f;
        const float xh = (u[p] - mean3) * rstd3, gy = go * (float)ln3w[p];
        sa += gy;
        sb += gy * xh;
        pg3[row * s2 + p] = go * xh;
        pb3[row * s2 + p] = go;
    }
    const float A3 = block_sum(sa, red) / (float)s2, B3 = block_sum(sb, red) / (float)s2;
    for (int p = tid; p < s2; p += 256) {
        const float go = p < D ? (float)g[row * D + p] : 0.f;
        const float xh = (u[p] - mean3) * rstd3, gy = go * (float)ln3w[p];
        u[p] = r16(rstd3 * (gy - A3 - xh * B3));                    // du, fp16 like autograd's grad of conv3's output
    }
    // ---------------- conv3 backward: dW3[c] = sum_p du[p] * a_last[c,p];  da_last = r16(w3[c] * du[p]) ----------------
    float acc16[CW];
#pragma unroll
    for (int c = 0; c < CW; ++c) acc16[c] = 0.f;
    for (int p = tid; p < s2; p += 256) {
        const float du = u[p], xv = xs[p];
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            const float al = THREE_X ? r16(((float)t2[c * s2 + p] - mean2) * rstd2 * (float)ln2w[c * s2 + p] + (float)ln2b[c * s2 + p])
                                     : r16((r16(w1[c] * xv) - mean1) * rstd1 * (float)ln1w[c * s2 + p] + (float)ln1b[c * s2 + p]);
            acc16[c] = fmaf(du, al, acc16[c]);
        }
    }
    block_sum16(acc16, red16);
    if (tid < CW) pw3[row * CW + tid] = acc16[tid];

    if (THREE_X) {
        // ------------ LN2 backward: da2 = r16(w3[c] du[p]); dt2 = rstd2 (gy - mean gy - xh mean(gy xh)) ------------
        sa = sb = 0.f;
        for (int p = tid; p < s2; p += 256) {
            const float du = u[p];
#pragma unroll
            for (int c = 0; c < CW; ++c) {
                const float da = r16(w3[c] * du);
                const float xh = ((float)t2[c * s2 + p] - mean2) * rstd2, gy = da * (float)ln2w[c * s2 + p];
                sa += gy;
                sb += gy * xh;
                pg2[row * n1 + c * s2 + p] = da * xh;
                pb2[row * n1 + c * s2 + p] = da;
            }
        }
        const float A2 = block_sum(sa, red) / (float)n1, B2 = block_sum(sb, red) / (float)n1;
        for (int p = tid; p < s2; p += 256) {
            const float du = u[p];
            const int py = p / s, px = p - py * s;
            const int hpos = (py + 1) * sp + px + 1;
#pragma unroll
            for (int cp = 0; cp < 8; ++cp) {
                float d[2];
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const int c = 2 * cp + j;
                    const float da = r16(w3[c] * du);
                    const float xh = ((float)t2[c * s2 + p] - mean2) * rstd2, gy = da * (float)ln2w[c * s2 + p];
                    d[j] = rstd2 * (gy - A2 - xh * B2);
                }
                d2p[cp * hp + hpos] = half2_t{(half_t)d[0], (half_t)d[1]};
            }
        }
        __syncthreads();
        // ------------ conv2 weight gradient: dW2[co,ci,tap] = sum_p dt2[co,p] * a1[ci, p + tap] ------------
        // A work item is (co pair, ci pair, tap): per pixel ONE 4-byte read of each operand feeds four fmaf chains (the
        // one-output-per-thread version read two halfs per multiply-add: 2304 x 529 of them per row, most of this kernel's time).
        // Every output keeps its (py, px)-ordered fp32 chain.
        for (int it = tid; it < 8 * 8 * 9; it += 256) {
            const int tap = it % 9, cip = (it / 9) & 7, cop = it / 72;
            const int dy = tap / 3, dx = tap - dy * 3;
            const half2_t* dsrc = d2p + cop * hp;
            const half2_t* asrc = a1p + cip * hp;
            float a00 = 0.f, a01 = 0.f, a10 = 0.f, a11 = 0.f;       // [co parity][ci parity]
            for (int py = 0; py < s; ++py)
                for (int px = 0; px < s; ++px) {
                    const half2_t dv = dsrc[(py + 1) * sp + px + 1], av = asrc[(py + dy) * sp + px + dx];
                    a00 = fmaf((float)dv[0], (float)av[0], a00);
                    a01 = fmaf((float)dv[0], (float)av[1], a01);
                    a10 = fmaf((float)dv[1], (float)av[0], a10);
                    a11 = fmaf((float)dv[1], (float)av[1], a11);
                }
            float* dst = pw2 + row * (CW * CW * 9);
            dst[((2 * cop) * CW + 2 * cip) * 9 + tap] = a00;
            dst[((2 * cop) * CW + 2 * cip + 1) * 9 + tap] = a01;
            dst[((2 * cop + 1) * CW + 2 * cip) * 9 + tap] = a10;
            dst[((2 * cop + 1) * CW + 2 * cip + 1) * 9 + tap] = a11;
        }
        // ------------ conv2 input gradient (transposed conv): da1[ci,p] = r16(sum_co,tap dt2[co, p - tap + 1] w2[co,ci,tap]) ----
        __syncthreads();                                            // t2 is overwritten with da1 below: all readers are done
        for (int p = tid; p < s2; p += 256) {
            const int py = p / s, px = p - py * s;
            float acc[CW];
#pragma unroll
            for (int c = 0; c < CW; ++c) acc[c] = 0.f;
#pragma unroll 1
            for (int cp = 0; cp < 8; ++cp) {
#pragma unroll
                for (int tap = 0; tap < 9; ++tap) {
                    const int dy = tap / 3, dx = tap - dy * 3;
                    const half2_t dv = d2p[cp * hp + (py + 2 - dy) * sp + px + 2 - dx];
                    const half2_t* wv = w2t + (cp * 9 + tap) * CW;
#pragma unroll
                    for (int ci = 0; ci < CW; ++ci) acc[ci] = __builtin_amdgcn_fdot2(dv, wv[ci], acc[ci], false);
                }
            }
#pragma unroll
            for (int ci = 0; ci < CW; ++ci) t2[ci * s2 + p] = (half_t)acc[ci];
        }
        __syncthreads();
    }
    // ---------------- LN1 backward + conv1 weight gradient ----------------
    sa = sb = 0.f;
    for (int p = tid; p < s2; p += 256) {
        const float du = u[p], xv = xs[p];
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            const float da = THREE_X ? (float)t2[c * s2 + p] : r16(w3[c] * du);
            const float xh = (r16(w1[c] * xv) - mean1) * rstd1, gy = da * (float)ln1w[c * s2 + p];
            sa += gy;
            sb += gy * xh;
            pg1[row * n1 + c * s2 + p] = da * xh;
            pb1[row * n1 + c * s2 + p] = da;
        }
    }
    const float A1 = block_sum(sa, red) / (float)n1, B1 = block_sum(sb, red) / (float)n1;
#pragma unroll
    for (int c = 0; c < CW; ++c) acc16[c] = 0.f;
    for (int p = tid; p < s2; p += 256) {
        const float du = u[p], xv = xs[p];
#pragma unroll
        for (int c = 0; c < CW; ++c) {
            const float da = THREE_X ? (float)t2[c * s2 + p] : r16(w3[c] * du);
            const float xh = (r16(w1[c] * xv) - mean1) * rstd1, gy = da * (float)ln1w[c * s2 + p];
            const float dt1 = r16(rstd1 * (gy - A1 - xh * B1));
            acc16[c] = fmaf(dt1, xv, acc16[c]);
        }
    }
    block_sum16(acc16, red16);
    if (tid < CW) pw1[row * CW + tid] = acc16[tid];
}

// ---- conv-3x backward on the matrix pipe, ONE launch for the whole batch -------------------------------------------------------------
// adapter_conv_backward_kernel<true> writes 149 KB of per-row parameter contributions (fp32) per row — the host walked the batch in chunks of
// 512 rows, each a launch + nine column sums (120 launches for an ImageNet episode) — and runs the two 1.2 MFLOP contractions per row (conv2's
// weight gradient and its transposed convolution) as scalar / packed-dot chains.  Here a PERSISTENT workgroup (one per CU, the whole register file)
// owns rows w, w + G, ... and keeps every parameter-gradient accumulator across its rows: ONE partial row per workgroup leaves the kernel
// ([G, n] fp32 -> one column sum per parameter over G <= #CU rows; deterministic: fixed row -> workgroup assignment, fixed order).
//  * forward recomputation = adapter_conv3x_mfma_kernel's arithmetic (a1 in LDS pixel-major [halo pixel][16 channels], conv2 on MFMA, t2 / a2 in registers);
//  * conv2 weight gradient dW2[co, ci, tap] = sum_p dt2[co, p] a1[ci, p + tap]: a GEMM over the PIXELS (K = halo-linear pixel index, 32 per
//    v_mfma_f32_16x16x32_f16).  dt2 is written pixel-major with a zero halo like a1, so both operands are [k = pixel][16 channels] in LDS and
//    their fragments (8 consecutive pixels of one channel per lane) are hardware transpose-reads (ds_read_b64_tr_b16), the tap a whole-pixel
//    offset of the a1 operand; wave w owns taps w, w + 4, w + 8 and its accumulators simply keep running over the workgroup's rows;
//  * conv2 input gradient (transposed convolution) = the forward's implicit GEMM with dt2 as the pixel operand and w2 transposed / taps mirrored;
//  * LN2 / LN1 backward, conv3 / conv1 weight gradients in the MFMA layout (a lane owns channels 4q .. 4q+3 of the pixels (tile i, lane & 15)),
//    LN3 backward in the row-linear layout.  Gradient tensors are rounded to fp16 where autograd materialises one, as in the VALU kernel.
template <int NT>
__global__ __launch_bounds__(256, 1) void adapter_conv3x_bwd_mfma_kernel(
    const half_t* __restrict__ x, const half_t* __restrict__ g, int B, int D, int s, const half_t* __restrict__ conv1,
    const half_t* __restrict__ ln1w, const half_t* __restrict__ ln1b, const half_t* __restrict__ conv2, const half_t* __restrict__ ln2w,
    const half_t* __restrict__ ln2b, const half_t* __restrict__ conv3, const half_t* __restrict__ ln3w, float* __restrict__ pw1,
    float* __restrict__ pw2, float* __restrict__ pw3, float* __restrict__ pg1, float* __restrict__ pb1, float* __restrict__ pg2,
    float* __restrict__ pb2, float* __restrict__ pg3, float* __restrict__ pb3) {
    constexpr int NPX = (NT + 3) / 4;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int s2 = s * s, sp = s + 2, hp = sp * sp, n1 = CW * s2;
    const int HK = (hp + 31) & ~31, PADP = sp + 1;                               // pixels of the K range of the weight-gradient GEMM; zero pixels in front of / behind a1
    half_t* a1h = reinterpret_cast<half_t*>(smem) + PADP * CW;                   // [-PADP, HK + PADP) x 16 fp16, interior rewritten per row, everything else zero
    half_t* d2h = reinterpret_cast<half_t*>(smem) + (HK + 2 * PADP) * CW;        // [HK][16] fp16: dt2, zero halo / tail
    float* u = reinterpret_cast<float*>(d2h + HK * CW);                          // [256 * NPX] conv3 output, then its gradient
    half_t* xs = reinterpret_cast<half_t*>(u + 256 * NPX);                       // [256 * NPX] the row (zero-padded to s2)
    float* red = reinterpret_cast<float*>(xs + 256 * NPX);                       // [2][8] block reductions (pairs), then [4][16] for the final 16-wide ones
    float* lg1 = red + 128;                                                      // [16 s2] LN1 gamma / beta gradient accumulators over the workgroup's rows
    float* lb1 = lg1 + n1;                                                       //         (LN2's stay in registers: all four would not fit the register file)
    int par = 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, q = lane >> 4;
    const float eps = 1e-5f;

    // ---- once per workgroup ----
    for (int i = tid; i < (2 * HK + 2 * PADP) * 8; i += 256) reinterpret_cast<unsigned*>(smem)[i] = 0u;
    for (int i = tid; i < 2 * n1; i += 256) lg1[i] = 0.f;
    half8_t wf[5], wt[5];               // conv2 as the MFMA A operand: forward (row co, k = tap * 16 + ci) and transposed (row ci, k = tap * 16 + co)
    int tapoff[5], tapoff_t[5];
#pragma unroll
    for (int ks = 0; ks < 5; ++ks) {
        const int tap = 2 * ks + (q >> 1), c0 = 8 * (q & 1), r = lane & 15;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            wf[ks][j] = tap < 9 ? conv2[(r * CW + c0 + j) * 9 + tap] : (half_t)0.f;
            wt[ks][j] = tap < 9 ? conv2[((c0 + j) * CW + r) * 9 + tap] : (half_t)0.f;
        }
        const int tc = tap < 9 ? tap : 8, dy = tc / 3, dx = tc - dy * 3;
        tapoff[ks] = (dy * sp + dx) * 32 + (q & 1) * 16;                        // forward: pixel (py + dy, px + dx) of the halo image
        tapoff_t[ks] = ((2 - dy) * sp + (2 - dx)) * 32 + (q & 1) * 16;          // transposed: pixel (py + 2 - dy, px + 2 - dx)
    }
    half2_t w1p[8];
#pragma unroll
    for (int c = 0; c < 8; ++c) w1p[c] = half2_t{conv1[2 * c], conv1[2 * c + 1]};
    float w3q[4];
    half_t w1q[4];
#pragma unroll
    for (int r = 0; r < 4; ++r) { w3q[r] = (float)conv3[4 * q + r]; w1q[r] = conv1[4 * q + r]; }
    int poff[NT], hidx[NT];
    const int p0 = 16 * wave + (lane & 15);
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int pp = p0 + 64 * i, pc = pp < s2 ? pp : 0, py = pc / s, px = pc - py * s;
        poff[i] = (py * sp + px) * 32;
        hidx[i] = (py + 1) * sp + px + 1;                                        // halo-linear index of the pixel itself
    }
    half2_t g1r[NPX][8], b1r[NPX][8];
    half_t g3r[NPX];
    half2_t g2r[NT][2], b2r[NT][2], g1m[NT][2];
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        const int pp = tid + 256 * j, pc = pp < s2 ? pp : 0;
#pragma unroll
        for (int c = 0; c < 8; ++c) {
            g1r[j][c] = half2_t{ln1w[(2 * c) * s2 + pc], ln1w[(2 * c + 1) * s2 + pc]};
            b1r[j][c] = half2_t{ln1b[(2 * c) * s2 + pc], ln1b[(2 * c + 1) * s2 + pc]};
        }
        g3r[j] = ln3w[pc];
    }
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int pp = p0 + 64 * i, pc = pp < s2 ? pp : 0;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            g2r[i][r] = half2_t{ln2w[(4 * q + 2 * r) * s2 + pc], ln2w[(4 * q + 2 * r + 1) * s2 + pc]};
            b2r[i][r] = half2_t{ln2b[(4 * q + 2 * r) * s2 + pc], ln2b[(4 * q + 2 * r + 1) * s2 + pc]};
            g1m[i][r] = half2_t{ln1w[(4 * q + 2 * r) * s2 + pc], ln1w[(4 * q + 2 * r + 1) * s2 + pc]};
        }
    }
    // accumulators over the workgroup's rows
    float4v_t accw[3] = {float4v_t{0.f, 0.f, 0.f, 0.f}, float4v_t{0.f, 0.f, 0.f, 0.f}, float4v_t{0.f, 0.f, 0.f, 0.f}};   // dW2[co = 4q + r][ci = lane & 15] of taps wave, wave + 4, wave + 8
    float aw1[4] = {0.f, 0.f, 0.f, 0.f}, aw3[4] = {0.f, 0.f, 0.f, 0.f};                                                    // conv1 / conv3 weight gradients, this lane's pixels
    float ag2[NT][4], ab2[NT][4], ag3[NPX], ab3[NPX];
#pragma unroll
    for (int i = 0; i < NT; ++i)
#pragma unroll
        for (int r = 0; r < 4; ++r) ag2[i][r] = ab2[i][r] = 0.f;
#pragma unroll
    for (int j = 0; j < NPX; ++j) ag3[j] = ab3[j] = 0.f;
    // transpose-read addressing (tools/probe/tr_probe.hip, pclip_encoder.hip attn_voff): inside a 16-lane group lane i supplies the address of 4
    // consecutive halves and lane l receives element (l & 3) of the words addressed by lanes 4 jj + ((l & 15) >> 2) — with lane i pointing at
    // M[k0 + (i >> 2)][4 (i & 3) ..] of a [k][16] fp16 matrix, lane l receives M[k0 + jj][l & 15], jj = 0 .. 3: four consecutive k of ITS column.
    // The MFMA operand wants k = 8 q .. 8 q + 7 for lane group q: two reads, k0 = 8 q and 8 q + 4.
    const int troff = (8 * q + ((lane & 15) >> 2)) * 32 + (lane & 3) * 8;
    auto tr8 = [&](const half_t* m, int k_pix) {                                 // column (lane & 15) of rows k_pix + 8 q .. + 7 of the [.][16] matrix m
        const char* b = reinterpret_cast<const char*>(m) + k_pix * 32 + troff;
        const half4_t lo = lds_tr_read4(b), hi4 = lds_tr_read4(b + 4 * 32);
        return half8_t{lo[0], lo[1], lo[2], lo[3], hi4[0], hi4[1], hi4[2], hi4[3]};
    };
    // block reduction of a PAIR on one barrier (block_sum1's parity scheme)
    auto block_sum_pair = [&](float& a, float& b) {
        a = wave_sum_valu(a);
        b = wave_sum_valu(b);
        float* r = red + 8 * par;
        par ^= 1;
        if (lane == 0) { r[wave] = a; r[4 + wave] = b; }
        __syncthreads();
        a = (r[0] + r[1]) + (r[2] + r[3]);
        b = (r[4] + r[5]) + (r[6] + r[7]);
    };
    auto block_sum_one = [&](float a) { float z = 0.f; block_sum_pair(a, z); return a; };
    // The packed parameters / t2 are made opaque at the head of every phase: hipcc otherwise converts them to fp32 ONCE per row (or once per
    // kernel) and keeps the copies live across the phases — 100+ registers (512 + scratch at D = 512).
    auto opaque_mfma_params = [&]() {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < NT; ++i) asm volatile("" : "+v"(g2r[i][0]), "+v"(g2r[i][1]), "+v"(b2r[i][0]), "+v"(b2r[i][1]), "+v"(g1m[i][0]), "+v"(g1m[i][1]));
#endif
    };
    __syncthreads();

    for (int row = blockIdx.x; row < B; row += gridDim.x) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
#pragma unroll
            for (int c = 0; c < 8; ++c) asm volatile("" : "+v"(g1r[j][c]), "+v"(b1r[j][c]));
        }
#pragma unroll
        for (int i = 0; i < NT; ++i) asm volatile("" : "+v"(g2r[i][0]), "+v"(g2r[i][1]), "+v"(b2r[i][0]), "+v"(b2r[i][1]), "+v"(g1m[i][0]), "+v"(g1m[i][1]));
#endif
        // ================= forward recomputation (adapter_conv3x_mfma_kernel) =================
        half_t xh[NPX];
        float sm = 0.f;
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const int pp = tid + 256 * j;
            xh[j] = pp < D ? x[(size_t)row * D + pp] : (half_t)0.f;
            xs[pp] = xh[j];
            const half2_t xx = {xh[j], xh[j]};
            if (pp < s2) {
#pragma unroll
                for (int c = 0; c < 8; ++c) { const half2_t t = w1p[c] * xx; sm += (float)t[0] + (float)t[1]; }
            }
        }
        const float mean1 = block_sum_one(sm) / (float)n1;
        float sq = 0.f;
#pragma unroll
        for (int j = 0; j < NPX; ++j)
            if (tid + 256 * j < s2) {
                const half2_t xx = {xh[j], xh[j]};
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const half2_t t = w1p[c] * xx;
                    const float d0 = (float)t[0] - mean1, d1 = (float)t[1] - mean1;
                    sq = fmaf(d0, d0, sq);
                    sq = fmaf(d1, d1, sq);
                }
            }
        const float rstd1 = 1.f / sqrtf(block_sum_one(sq) / (float)n1 + eps);
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const int pp = tid + 256 * j;
            if (pp < s2) {
                const int py = pp / s, px = pp - py * s;
                const half2_t xx = {xh[j], xh[j]};
                half2_t o[8];
#pragma unroll
                for (int c = 0; c < 8; ++c) {
                    const half2_t t = w1p[c] * xx;
                    const float a0 = ((float)t[0] - mean1) * rstd1 * (float)g1r[j][c][0] + (float)b1r[j][c][0];
                    const float a1 = ((float)t[1] - mean1) * rstd1 * (float)g1r[j][c][1] + (float)b1r[j][c][1];
                    o[c] = half2_t{(half_t)a0, (half_t)a1};
                }
                half8_t* dst = reinterpret_cast<half8_t*>(a1h + ((py + 1) * sp + px + 1) * CW);
                dst[0] = half8_t{o[0][0], o[0][1], o[1][0], o[1][1], o[2][0], o[2][1], o[3][0], o[3][1]};
                dst[1] = half8_t{o[4][0], o[4][1], o[5][0], o[5][1], o[6][0], o[6][1], o[7][0], o[7][1]};
            }
        }
        __syncthreads();
        float4v_t acc[NT];
        const char* abase = reinterpret_cast<const char*>(a1h);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            acc[i] = float4v_t{0.f, 0.f, 0.f, 0.f};
            if (16 * wave + 64 * i < s2) {
#pragma unroll
                for (int ks = 0; ks < 5; ++ks) {
                    const half8_t bfrag = *reinterpret_cast<const half8_t*>(abase + poff[i] + tapoff[ks]);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wf[ks], bfrag, acc[i], 0, 0, 0);
                }
            }
        }
        float sm2 = 0.f;
#pragma unroll
        for (int i = 0; i < NT; ++i)
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                acc[i][r] = r16(acc[i][r]);                                       // t2
                if (p0 + 64 * i < s2) sm2 += acc[i][r];
            }
        const float mean2 = block_sum_one(sm2) / (float)n1;
        float sq2 = 0.f;
#pragma unroll
        for (int i = 0; i < NT; ++i)
            if (p0 + 64 * i < s2) {
#pragma unroll
                for (int r = 0; r < 4; ++r) { const float d = acc[i][r] - mean2; sq2 = fmaf(d, d, sq2); }
            }
        const float rstd2 = 1.f / sqrtf(block_sum_one(sq2) / (float)n1 + eps);
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int pp = p0 + 64 * i;
            float v = 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float a2 = r16((acc[i][r] - mean2) * rstd2 * (float)g2r[i][r >> 1][r & 1] + (float)b2r[i][r >> 1][r & 1]);
                v = fmaf(w3q[r], a2, v);
            }
            v = pair_sum32(pair_sum16(v));
            if (q == 0 && pp < s2) u[pp] = r16(v);
        }
        __syncthreads();
        // ================= LN3 (statistics + backward), row-linear layout =================
        float uu[NPX], s3 = 0.f;
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const int pp = tid + 256 * j;
            uu[j] = pp < s2 ? u[pp] : 0.f;
            s3 += uu[j];
        }
        const float mean3 = block_sum_one(s3) / (float)s2;
        float q3 = 0.f;
#pragma unroll
        for (int j = 0; j < NPX; ++j)
            if (tid + 256 * j < s2) { const float d = uu[j] - mean3; q3 = fmaf(d, d, q3); }
        const float rstd3 = 1.f / sqrtf(block_sum_one(q3) / (float)s2 + eps);
        // upstream g on the first D positions (the crop and the identity add pass it through)
        float go[NPX], sa = 0.f, sb = 0.f;
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const int pp = tid + 256 * j;
            go[j] = pp < D ? (float)g[(size_t)row * D + pp] : 0.f;
            if (pp < s2) {
                const float xh3 = (uu[j] - mean3) * rstd3, gy = go[j] * (float)g3r[j];
                sa += gy;
                sb += gy * xh3;
                ag3[j] += go[j] * xh3;
                ab3[j] += go[j];
            }
        }
        block_sum_pair(sa, sb);
        const float A3 = sa / (float)s2, B3 = sb / (float)s2;
#pragma unroll
        for (int j = 0; j < NPX; ++j) {
            const int pp = tid + 256 * j;
            if (pp < s2) {
                const float xh3 = (uu[j] - mean3) * rstd3, gy = go[j] * (float)g3r[j];
                u[pp] = r16(rstd3 * (gy - A3 - xh3 * B3));                        // du, fp16 like autograd's gradient of conv3's output
            }
        }
        __syncthreads();
        // ================= conv3 / LN2 backward, MFMA layout =================
        float duv[NT];
        sa = sb = 0.f;
        opaque_mfma_params();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < NT; ++i) asm volatile("" : "+v"(acc[i]));
#endif
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int pp = p0 + 64 * i;
            duv[i] = pp < s2 ? u[pp] : 0.f;
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const float gg = (float)g2r[i][r >> 1][r & 1];
                const float xh2 = (acc[i][r] - mean2) * rstd2;
                const float a2 = r16(xh2 * gg + (float)b2r[i][r >> 1][r & 1]);
                const float da = r16(w3q[r] * duv[i]);
                if (pp < s2) {
                    aw3[r] = fmaf(duv[i], a2, aw3[r]);
                    const float gy = da * gg;
                    sa += gy;
                    sb += gy * xh2;
                    ag2[i][r] += da * xh2;
                    ab2[i][r] += da;
                }
            }
        }
        block_sum_pair(sa, sb);
        const float A2 = sa / (float)n1, B2 = sb / (float)n1;
        opaque_mfma_params();
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < NT; ++i) asm volatile("" : "+v"(acc[i]));
#endif
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int pp = p0 + 64 * i;
            if (pp < s2) {
                half4_t d4;
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float gg = (float)g2r[i][r >> 1][r & 1];
                    const float xh2 = (acc[i][r] - mean2) * rstd2;
                    const float gy = r16(w3q[r] * duv[i]) * gg;
                    d4[r] = (half_t)(rstd2 * (gy - A2 - xh2 * B2));
                }
                *reinterpret_cast<half4_t*>(d2h + hidx[i] * CW + 4 * q) = d4;      // dt2[4q .. 4q+3] of this pixel
            }
        }
        __syncthreads();
        // ================= conv2 weight gradient: accw[t][r] += sum_k dt2[k][co = 4q + r] a1[k + toff][ci = lane & 15] =================
#pragma unroll
        for (int t = 0; t < 3; ++t) {
            const int tap = wave + 4 * t;
            if (tap < 9) {                                                        // wave-uniform
                const int dy = tap / 3, dx = tap - dy * 3, toff = (dy - 1) * sp + (dx - 1);
                for (int k0 = 0; k0 < HK; k0 += 32) {
                    const half8_t af = tr8(d2h, k0), bf = tr8(a1h, k0 + toff);
                    accw[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(af, bf, accw[t], 0, 0, 0);
                }
            }
        }
        // ================= conv2 input gradient (transposed convolution) + LN1 backward + conv1 weight gradient =================
        const char* dbase = reinterpret_cast<const char*>(d2h);
        sa = sb = 0.f;
        opaque_mfma_params();
        float da1[NT][4];
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            float4v_t ad = float4v_t{0.f, 0.f, 0.f, 0.f};
            if (16 * wave + 64 * i < s2) {
#pragma unroll
                for (int ks = 0; ks < 5; ++ks) {
                    const half8_t bfrag = *reinterpret_cast<const half8_t*>(dbase + poff[i] + tapoff_t[ks]);
                    ad = __builtin_amdgcn_mfma_f32_16x16x32_f16(wt[ks], bfrag, ad, 0, 0, 0);
                }
            }
            const int pp = p0 + 64 * i;
            const half_t xv = xs[pp < s2 ? pp : 0];
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                da1[i][r] = r16(ad[r]);                                           // da1[ci = 4q + r][pixel]
                if (pp < s2) {
                    const float t1 = (float)(half_t)(w1q[r] * xv);
                    const float xh1 = (t1 - mean1) * rstd1, gy = da1[i][r] * (float)g1m[i][r >> 1][r & 1];
                    sa += gy;
                    sb += gy * xh1;
                    lg1[(4 * q + r) * s2 + pp] += da1[i][r] * xh1;                 // this lane is the only writer of (channel, pixel)
                    lb1[(4 * q + r) * s2 + pp] += da1[i][r];
                }
            }
        }
        block_sum_pair(sa, sb);
        const float A1 = sa / (float)n1, B1 = sb / (float)n1;
        opaque_mfma_params();
#pragma unroll
        for (int i = 0; i < NT; ++i) {
            const int pp = p0 + 64 * i;
            if (pp < s2) {
                const half_t xv = xs[pp];
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    const float t1 = (float)(half_t)(w1q[r] * xv);
                    const float xh1 = (t1 - mean1) * rstd1, gy = da1[i][r] * (float)g1m[i][r >> 1][r & 1];
                    const float dt1 = r16(rstd1 * (gy - A1 - xh1 * B1));
                    aw1[r] = fmaf(dt1, (float)xv, aw1[r]);
                }
            }
        }
        __syncthreads();                                                          // a1h / d2h / u / xs are rewritten by the next row
    }

    // ---- one partial row per workgroup ----
    const size_t wg = blockIdx.x;
#pragma unroll
    for (int i = 0; i < NT; ++i) {
        const int pp = p0 + 64 * i;
        if (pp < s2) {
#pragma unroll
            for (int r = 0; r < 4; ++r) {
                const size_t o = wg * n1 + (size_t)(4 * q + r) * s2 + pp;
                pg2[o] = ag2[i][r];
                pb2[o] = ab2[i][r];
            }
        }
    }
    for (int i = tid; i < n1; i += 256) { pg1[wg * n1 + i] = lg1[i]; pb1[wg * n1 + i] = lb1[i]; }   // (own writes + the row loop's last barrier: visible)
#pragma unroll
    for (int j = 0; j < NPX; ++j) {
        const int pp = tid + 256 * j;
        if (pp < s2) { pg3[wg * s2 + pp] = ag3[j]; pb3[wg * s2 + pp] = ab3[j]; }
    }
#pragma unroll
    for (int t = 0; t < 3; ++t) {
        const int tap = wave + 4 * t;
        if (tap < 9) {
#pragma unroll
            for (int r = 0; r < 4; ++r) pw2[wg * (CW * CW * 9) + ((4 * q + r) * CW + (lane & 15)) * 9 + tap] = accw[t][r];
        }
    }
    // conv1 / conv3 weight gradients: this lane's pixels -> the 16 lanes of a channel group (DPP) -> the four waves (LDS)
    __syncthreads();
    float* r16w = red;                                                            // [2][4][16]
#pragma unroll
    for (int r = 0; r < 4; ++r) {
        float a = aw1[r], b = aw3[r];
#if defined(__HIP_DEVICE_COMPILE__)
        auto dpp = [](float v, auto ctrl) {
            const int iv = __builtin_bit_cast(int, v);
            return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, decltype(ctrl)::value, 0xF, 0xF, false));
        };
        a = dpp(dpp(dpp(dpp(a, std::integral_constant<int, 0xB1>()), std::integral_constant<int, 0x4E>()), std::integral_constant<int, 0x141>()), std::integral_constant<int, 0x140>());
        b = dpp(dpp(dpp(dpp(b, std::integral_constant<int, 0xB1>()), std::integral_constant<int, 0x4E>()), std::integral_constant<int, 0x141>()), std::integral_constant<int, 0x140>());
#endif
        if ((lane & 15) == 0) { r16w[wave * 16 + 4 * q + r] = a; r16w[64 + wave * 16 + 4 * q + r] = b; }
    }
    __syncthreads();
    if (tid < CW) {
        pw1[wg * CW + tid] = (r16w[tid] + r16w[16 + tid]) + (r16w[32 + tid] + r16w[48 + tid]);
        pw3[wg * CW + tid] = (r16w[64 + tid] + r16w[80 + tid]) + (r16w[96 + tid] + r16w[112 + tid]);
    }
}

}  // namespace

static bool adapter_mfma_enabled() {
    static const bool on = !(getenv("PCLIP_ADAPTER_MFMA") && getenv("PCLIP_ADAPTER_MFMA")[0] == '0');     // A/B switch: 0 = the VALU kernels
    return on;
}
static int adapter_bwd_workgroups(int B, int D, int three_x) {
    // rows of partial sums the backward writes: the persistent MFMA kernel (conv-3x, D <= 576) one per workgroup, the VALU kernel one per input row
    int s = 1;
    while (s * s < D) ++s;
    const int NT = (s * s + 63) / 64;
    if (!three_x || NT > TWO_WG_MAX || !adapter_mfma_enabled()) return B;
    int cus = pclip_device_cus();
    if (cus <= 0) cus = 256;
    return B < cus ? B : cus;
}
extern "C" int pclip_adapter_conv_backward_partials(int B, int D, int three_x) { return adapter_bwd_workgroups(B, D, three_x); }

extern "C" int pclip_adapter_conv_backward_f16(const void* x, const void* g, int B, int D, int three_x, const void* conv1,
                                               const void* ln1w, const void* ln1b, const void* conv2, const void* ln2w,
                                               const void* ln2b, const void* conv3, const void* ln3w, float* pw1, float* pw2,
                                               float* pw3, float* pg1, float* pb1, float* pg2, float* pb2, float* pg3, float* pb3,
                                               pclip_stream_t stream) {
    PCLIP_REQUIRE(x && g && conv1 && ln1w && ln1b && conv3 && ln3w && pw1 && pw3 && pg1 && pb1 && pg3 && pb3,
                  "pclip_adapter_conv_backward_f16: null pointer");
    PCLIP_REQUIRE(!three_x || (conv2 && ln2w && ln2b && pw2 && pg2 && pb2), "pclip_adapter_conv_backward_f16: conv-3x needs conv2 / bn2 and their outputs");
    PCLIP_REQUIRE(B >= 0 && D > 0 && D <= 1024, "pclip_adapter_conv_backward_f16: D=%d must be in [1, 1024]", D);
    if (B == 0) return PCLIP_OK;
    int s = 1;
    while (s * s < D) ++s;
    const int s2 = s * s, hp = (s + 2) * (s + 2);
    const size_t lds = (size_t)(1024 + 1024 + 4 + 64) * 4 + (three_x ? (size_t)2 * 8 * hp * 4 + (size_t)CW * s2 * 2 + (size_t)2 * 8 * 9 * CW * 4 : 0);
    hipStream_t st = (hipStream_t)stream;
    static DevOnce attr[2];
    const int R = adapter_bwd_workgroups(B, D, three_x);
    if (three_x && (s2 + 63) / 64 <= TWO_WG_MAX && adapter_mfma_enabled()) {                 // persistent MFMA kernel: R partial rows
        const int NT = (s2 + 63) / 64, NPX = (NT + 3) / 4, HK = (hp + 31) & ~31, PADP = s + 3;
        const size_t lds_m = (size_t)(2 * HK + 2 * PADP) * 32 + (size_t)256 * NPX * 6 + 128 * 4 + (size_t)2 * CW * s2 * 4;
        static DevOnce attr_m;
        const void* fns[] = {(const void*)adapter_conv3x_bwd_mfma_kernel<1>, (const void*)adapter_conv3x_bwd_mfma_kernel<2>, (const void*)adapter_conv3x_bwd_mfma_kernel<3>,
                             (const void*)adapter_conv3x_bwd_mfma_kernel<4>, (const void*)adapter_conv3x_bwd_mfma_kernel<5>, (const void*)adapter_conv3x_bwd_mfma_kernel<6>,
                             (const void*)adapter_conv3x_bwd_mfma_kernel<7>, (const void*)adapter_conv3x_bwd_mfma_kernel<8>, (const void*)adapter_conv3x_bwd_mfma_kernel<9>};
        if (!attr_m.done()) {
            for (const void* f : fns)
                if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                    pclip_set_error("pclip_adapter_conv_backward_f16: cannot raise the dynamic LDS limit");
                    return PCLIP_E_LAUNCH;
                }
            attr_m.set();
        }
#define PCLIP_ADAPTER_BWD(N)                                                                                                                  \
        case N: adapter_conv3x_bwd_mfma_kernel<N><<<R, 256, lds_m, st>>>((const half_t*)x, (const half_t*)g, B, D, s, (const half_t*)conv1,    \
                    (const half_t*)ln1w, (const half_t*)ln1b, (const half_t*)conv2, (const half_t*)ln2w, (const half_t*)ln2b,                  \
                    (const half_t*)conv3, (const half_t*)ln3w, pw1, pw2, pw3, pg1, pb1, pg2, pb2, pg3, pb3); break;
        switch (NT) {
            PCLIP_ADAPTER_BWD(1) PCLIP_ADAPTER_BWD(2) PCLIP_ADAPTER_BWD(3) PCLIP_ADAPTER_BWD(4) PCLIP_ADAPTER_BWD(5) PCLIP_ADAPTER_BWD(6)
            PCLIP_ADAPTER_BWD(7) PCLIP_ADAPTER_BWD(8) PCLIP_ADAPTER_BWD(9)
            default: pclip_set_error("pclip_adapter_conv_backward_f16: s2=%d out of range", s2); return PCLIP_E_INVALID;
        }
#undef PCLIP_ADAPTER_BWD
        return pclip_check_launch("adapter_conv_backward (mfma)");
    }
    if (three_x) {
        if (!attr[1].done()) {
            if (hipFuncSetAttribute((const void*)adapter_conv_backward_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                pclip_set_error("pclip_adapter_conv_backward_f16: cannot raise the dynamic LDS limit");
                return PCLIP_E_LAUNCH;
            }
            attr[1].set();
        }
        adapter_conv_backward_kernel<true><<<B, 256, lds, st>>>((const half_t*)x, (const half_t*)g, D, s, (const half_t*)conv1, (const half_t*)ln1w,
            (const half_t*)ln1b, (const half_t*)conv2, (const half_t*)ln2w, (const half_t*)ln2b, (const half_t*)conv3, (const half_t*)ln3w,
            pw1, pw2, pw3, pg1, pb1, pg2, pb2, pg3, pb3);
    } else {
        adapter_conv_backward_kernel<false><<<B, 256, lds, st>>>((const half_t*)x, (const half_t*)g, D, s, (const half_t*)conv1, (const half_t*)ln1w,
            (const half_t*)ln1b, nullptr, nullptr, nullptr, (const half_t*)conv3, (const half_t*)ln3w, pw1, nullptr, pw3, pg1, pb1, nullptr, nullptr, pg3, pb3);
    }
    return pclip_check_launch("adapter_conv_backward");
}

namespace {
}  // namespace

extern "C" int pclip_adapter_fc_f16(const void* x, int B, int D, int H, const void* w1, const void* g1, const void* b1,
                                    const void* w2, const void* g2, const void* b2, float ratio, float one_minus_ratio,
                                    int l2norm_out, void* y, float* y_sq, void* ws, size_t ws_bytes,
                                    pclip_stream_t stream) {
    PCLIP_REQUIRE(x && w1 && g1 && b1 && w2 && g2 && b2 && y, "pclip_adapter_fc_f16: null pointer");
    PCLIP_REQUIRE(B >= 0 && D > 0 && H > 0 && D % 64 == 0 && H % 64 == 0 && D <= 4096 && H <= 4096,
                  "pclip_adapter_fc_f16: D=%d and H=%d must be multiples of 64 (<= 4096)", D, H);
    if (B == 0) return PCLIP_OK;
    const size_t need = pclip_workspace_bytes(PCLIP_OP_ADAPTER_FC, B, H, D);
    PCLIP_REQUIRE(ws != nullptr, "pclip_adapter_fc_f16: workspace required");
    if (ws_bytes < need) { pclip_set_error("pclip_adapter_fc_f16: workspace %zu < %zu", ws_bytes, need); return PCLIP_E_WORKSPACE; }
    char* b = (char*)ws;
    void* h1 = b;
    void* h1n = b + align_up((size_t)B * H * 2, 256);
    void* h2 = b + 2 * align_up((size_t)B * H * 2, 256);
    hipStream_t s = (hipStream_t)stream;
    int e;
    if ((e = pclip_gemm_f16(x, D, w1, D, h1, H, B, H, D, nullptr, 0, nullptr, stream))) return e;          // fc.0
    if ((e = pclip_layernorm_f16p(h1, g1, b1, 1e-5f, h1n, B, H, nullptr, 0.f, 0.f, 0, nullptr, s))) return e;  // fc.1
    if ((e = pclip_gemm_f16(h1n, H, w2, H, h2, D, B, D, H, nullptr, 0, nullptr, stream))) return e;        // fc.2
    return pclip_layernorm_f16p(h2, g2, b2, 1e-5f, y, B, D, x, ratio, one_minus_ratio, l2norm_out, y_sq, s);   // fc.3 + blend
}

extern "C" int pclip_layernorm_blend_f16(const void* h, const void* gamma, const void* beta, float eps, const void* x, float ratio,
                                         float one_minus_ratio, int l2norm_out, void* y, float* y_sq, int R, int D,
                                         pclip_stream_t stream) {
    PCLIP_REQUIRE(h && gamma && beta && x && y, "pclip_layernorm_blend_f16: null pointer");
    PCLIP_REQUIRE(R >= 0 && D > 0 && D % 8 == 0 && D <= 4096, "pclip_layernorm_blend_f16: bad shape R=%d D=%d", R, D);
    return pclip_layernorm_f16p(h, gamma, beta, eps, y, R, D, x, ratio, one_minus_ratio, l2norm_out, y_sq, (hipStream_t)stream);
}

extern "C" int pclip_adapter_conv_f16(const void* x, int B, int D, int three_x, const void* conv1, const void* ln1w,
                                      const void* ln1b, const void* conv2, const void* ln2w, const void* ln2b,
                                      const void* conv3, const void* ln3w, const void* ln3b, int l2norm_out, void* y,
                                      float* y_sq, pclip_stream_t stream) {
    PCLIP_REQUIRE(x && conv1 && ln1w && ln1b && conv3 && ln3w && ln3b && y, "pclip_adapter_conv_f16: null pointer");
    PCLIP_REQUIRE(!three_x || (conv2 && ln2w && ln2b), "pclip_adapter_conv_f16: conv-3x needs conv2/bn2 parameters");
    PCLIP_REQUIRE(B >= 0 && D > 0 && D <= 1024, "pclip_adapter_conv_f16: D=%d must be in (0, 1024]", D);
    if (B == 0) return PCLIP_OK;
    int s = 1;
    while (s * s < D) ++s;          // ceil(sqrt(D)), model.py:30
    const int s2 = s * s, hp = (s + 2) * (s + 2);
    size_t lds = (1024 + 1024 + 4) * 4;
    if (three_x) lds += (size_t)8 * hp * 4 + (size_t)CW * s2 * 2 + 8 * 9 * CW * 4;
    hipStream_t st = (hipStream_t)stream;
    if (three_x && adapter_mfma_enabled()) {
        // persistent workgroups, two per CU (register budget of __launch_bounds__(256, 2)); LDS: a1 halo image + u + the reduction scratch
        const int NT = (s2 + 63) / 64, NPX = (NT + 3) / 4;
        const size_t lds_m = (size_t)((hp * 32 + 15) & ~15) + (size_t)(256 * NPX + 8) * 4;
        int cus = pclip_device_cus();
        if (cus <= 0) cus = 256;
        const int wgs = NT <= TWO_WG_MAX ? 2 * cus : cus;
        const int grid = B < wgs ? B : wgs;
#define PCLIP_ADAPTER_LAUNCH(N)                                                                                                               \
        case N: adapter_conv3x_mfma_kernel<N><<<grid, 256, lds_m, st>>>((const half_t*)x, B, D, s, (const half_t*)conv1, (const half_t*)ln1w,  \
                    (const half_t*)ln1b, (const half_t*)conv2, (const half_t*)ln2w, (const half_t*)ln2b, (const half_t*)conv3,                \
                    (const half_t*)ln3w, (const half_t*)ln3b, l2norm_out, (half_t*)y, y_sq); break;
        switch (NT) {
            PCLIP_ADAPTER_LAUNCH(1) PCLIP_ADAPTER_LAUNCH(2) PCLIP_ADAPTER_LAUNCH(3) PCLIP_ADAPTER_LAUNCH(4) PCLIP_ADAPTER_LAUNCH(5) PCLIP_ADAPTER_LAUNCH(6)
            PCLIP_ADAPTER_LAUNCH(7) PCLIP_ADAPTER_LAUNCH(8) PCLIP_ADAPTER_LAUNCH(9) PCLIP_ADAPTER_LAUNCH(10) PCLIP_ADAPTER_LAUNCH(11) PCLIP_ADAPTER_LAUNCH(12)
            PCLIP_ADAPTER_LAUNCH(13) PCLIP_ADAPTER_LAUNCH(14) PCLIP_ADAPTER_LAUNCH(15) PCLIP_ADAPTER_LAUNCH(16)
            default: pclip_set_error("pclip_adapter_conv_f16: s2=%d out of range", s2); return PCLIP_E_INVALID;
        }
#undef PCLIP_ADAPTER_LAUNCH
    } else if (three_x) {
        static DevOnce attr;
        if (!attr.done()) {
            (void)hipFuncSetAttribute((const void*)adapter_conv_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 96 * 1024);
            attr.set();
        }
        adapter_conv_kernel<true><<<B, 256, lds, st>>>((const half_t*)x, D, s, (const half_t*)conv1, (const half_t*)ln1w,
                                                       (const half_t*)ln1b, (const half_t*)conv2, (const half_t*)ln2w,
                                                       (const half_t*)ln2b, (const half_t*)conv3, (const half_t*)ln3w,
                                                       (const half_t*)ln3b, l2norm_out, (half_t*)y, y_sq);
    } else {
        adapter_conv_kernel<false><<<B, 256, lds, st>>>((const half_t*)x, D, s, (const half_t*)conv1, (const half_t*)ln1w,
                                                        (const half_t*)ln1b, nullptr, nullptr, nullptr, (const half_t*)conv3,
                                                        (const half_t*)ln3w, (const half_t*)ln3b, l2norm_out, (half_t*)y, y_sq);
    }
    return pclip_check_launch("adapter_conv");
}
