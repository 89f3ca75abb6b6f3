// Pieces shared by the translation units of the CLIP encoder (pclip_linear.hip, pclip_layernorm.hip, pclip_attention.hip, pclip_stem.hip — one file, pclip_encoder.hip,
// until round 5): the linear epilogue descriptor, the row-statistics association order, the LayerNorm row arithmetic, launch-grid helpers.  Everything lives in an
// anonymous namespace: each translation unit gets its own copy.
#pragma once
#include "pclip_gemm.h"
#include "pclip_epilogue.h"

namespace {
#ifndef PCLIP_EPI_PIPE
#define PCLIP_EPI_PIPE 1         // 256 x 256 tiles: the LDS-staged epilogue as a four-slab pipeline (pgemm::epilogue_pipe)
#endif
using CfgBigT = pgemm::Cfg<256, 256, 2, 4>;
// ---- nn.Linear on MFMA ------------------------------------------------------------------------
// Rounding points follow the reference's fp16 tensors: r16(acc + bias); QuickGELU as three fp16
// elementwise ops (clip/model.py:166); residual add rounds once more (clip/model.py:188-189).
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
struct LinearEpi {
    const half_t* __restrict__ bias;
    const half_t* __restrict__ residual;
    half_t* __restrict__ C;
    int ldc;
    int act;                          // 0 none, 1 QuickGELU, 2 per-column affine (eval BatchNorm), 3 affine + ReLU
    const float* __restrict__ scale;  // act 2 / 3 / 5: y = r16(r16(acc) * scale[n] + shift[n])
    const float* __restrict__ shift;
};

// The two places of a LayerNorm row where a multiply is followed by an add: hipcc contracted them into an fma in some instantiations and not in others (NCH = 1 and
// NCH = 2 of the SAME source differed — one fp16 ulp on 1e-5 of the elements — as soon as the code around them changed), and "a row alone == the row in a batch"
// needs every LayerNorm kernel to round alike.  Spelled out, contraction off: the squared deviations accumulate by fma, the affine is two rounded multiplies and a
// rounded add (what the D = 768 / 1024 instantiations had compiled to).
__device__ __forceinline__ float ln_sq_acc(float t, float q) { return __builtin_fmaf(t, t, q); }
__device__ __forceinline__ float ln_affine_dev(float t, float rstd, float g, float b) {      // t = v - mean
#pragma clang fp contract(off)
    return (t * rstd) * g + b;
}
__device__ __forceinline__ float ln_affine(float v, float mean, float rstd, float g, float b) { return ln_affine_dev(v - mean, rstd, g, b); }

// ... and its last four levels (lane ^ 8, ^ 4, ^ 2, ^ 1: inside a DPP row of 16 lanes) on their own
__device__ __forceinline__ float row16_sum_x(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    int i = __builtin_bit_cast(int, v);
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x128, 0xF, 0xF, false));            // row_ror:8: lane ^ 8
    i = __builtin_bit_cast(int, v);
    int t = __builtin_amdgcn_update_dpp(i, i, 0x104, 0xF, 0x5, false);                                         // row_shl:4 into lanes 0-3, 8-11 of a row: lane + 4
    t = __builtin_amdgcn_update_dpp(t, i, 0x114, 0xF, 0xA, false);                                             // row_shr:4 into lanes 4-7, 12-15: lane - 4
    v = v + __builtin_bit_cast(float, t);
    i = __builtin_bit_cast(int, v);
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0x4E, 0xF, 0xF, false));             // quad_perm [2,3,0,1]: lane ^ 2
    i = __builtin_bit_cast(int, v);
    v = v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(i, i, 0xB1, 0xF, 0xF, false));             // quad_perm [1,0,3,2]: lane ^ 1
#endif
    return v;
}

// wave_sum(v) — v += v[lane ^ 32], ^ 16, ^ 8, ^ 4, ^ 2, ^ 1 in that order — without its six ds_bpermute round trips through the LDS crossbar: the same additions
// (a + b is commutative, so every lane forms the same values level by level: same bits) on the VALU: v_permlane32_swap, v_permlane16_swap, DPP row_ror:8,
// two bank-masked DPP row shifts for ^ 4 (gfx9 has no row_xmask), quad_perm for ^ 2 and ^ 1.
__device__ __forceinline__ float wave_sum_x(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    {
        const unsigned a = __builtin_bit_cast(unsigned, v);
        const auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
        v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    }
    {
        const unsigned a = __builtin_bit_cast(unsigned, v);
        const auto r = __builtin_amdgcn_permlane16_swap(a, a, false, false);
        v = __builtin_bit_cast(float, (unsigned)r[0]) + __builtin_bit_cast(float, (unsigned)r[1]);
    }
#endif
    return row16_sum_x(v);
}

// One row of the whole-batch LayerNorm: the lane's chunks `cur` (eight halves per 512-column chunk) -> `out`; affine(c, j) returns (gamma, beta) of column
// c * 512 + lane * 8 + j.  (layernorm_pf_kernel; a row's bits do not depend on the batch around it.)
template <int NCH, class Affine>
__device__ __forceinline__ void ln_row_pf(const half8_t (&cur)[NCH], int D, int lane, float eps, const Affine& affine, half8_t (&out)[NCH]) {
    float v[NCH][8];
    float s = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c * 512 + lane * 8 < D) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { v[c][j] = (float)cur[c][j]; s += v[c][j]; }
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) v[c][j] = 0.f;
        }
    }
    const float mean = wave_sum_x(s) / (float)D;
    float q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
        if (c * 512 + lane * 8 < D) {
#pragma unroll
            for (int j = 0; j < 8; ++j) { const float t = v[c][j] - mean; q = ln_sq_acc(t, q); }
        }
    const float rstd = 1.f / sqrtf(wave_sum_x(q) / (float)D + eps);
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        if (c * 512 + lane * 8 < D) {
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float2_t gb = affine(c, j);
                float t = ln_affine(v[c][j], mean, rstd, gb[0], gb[1]);
                t = r16(t);
                out[c][j] = (half_t)t;
            }
        }
    }
}

#ifndef PCLIP_LN_PF
#define PCLIP_LN_PF 1
#endif
#ifndef PCLIP_LN_LDS
#define PCLIP_LN_LDS 1
#endif
#ifndef PCLIP_LN_BPC
#define PCLIP_LN_BPC 32
#endif

// (Measured and rejected, profiles/r03_ab_rejected.txt: TWO ADJACENT rows of D = 768 as three full wave loads, row B brought into this
// kernel's lane layout with v_permlane32_swap — the half-empty second load of a 1.5 KiB row is NOT what holds the pass at 4.5 TB/s
// stand-alone: 136.7 vs 137.0 us; the 6.5 TB/s of [65 792, 1024] is the Infinity Cache (270 MB footprint).  Also:
// two rows per wave + gamma / beta hoisted into registers + non-temporal stores of h
// — 152 vs 139 us on [201 728, 768] stand-alone, 123 vs 43 us on [65 792, 1024]: the 32 extra registers cost more occupancy than
// the extra loads in flight bring.)
inline int row_grid(int R) { int g = ceil_div(R, 4); return g < 1 ? 1 : (g > 16384 ? 16384 : g); }
inline int flat_grid(size_t n) { size_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g)); }
}  // namespace

#define DISPATCH_NCH(D, CALL)                                  \
    do {                                                       \
        if ((D) <= 512) { constexpr int NCH = 1; CALL; }       \
        else if ((D) <= 1024) { constexpr int NCH = 2; CALL; } \
        else if ((D) <= 2048) { constexpr int NCH = 4; CALL; } \
        else { constexpr int NCH = 8; CALL; }                  \
    } while (0)
