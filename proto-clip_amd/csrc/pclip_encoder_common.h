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
    const float* __restrict__ scale;  // act 2 / 3 / 5: y = r16(r16(acc) * scale[n] + shift[n]);  act 7 / 8: column sums of the folded weight
    const float* __restrict__ shift;  //                                                           act 7 / 8: folded bias
    const float* __restrict__ rowstats = nullptr;   // act 7 / 8: (mean, rstd) per row of A, [round_up(M, 256) + 256][2] fp32
    float* __restrict__ partials = nullptr;         // act 9: (sum, sum of squares) per row and 64 output columns, [M][N / 64][2] fp32
    // act 10 = act 6 + y = LayerNorm(updated rows) by the workgroup that completes a row panel (see linear_fast_kernel)
    const float* ln_gamma = nullptr;
    const float* ln_beta = nullptr;
    half_t* ln_y = nullptr;                         // [M][N], contiguous rows
    int* ln_cnt = nullptr;                          // arrival counters, one per 128 rows of C, zero before and after every launch
    float ln_eps = 0.f;
};
struct LnPanel { const float* gamma; const float* beta; half_t* y; int* cnt; float eps; };

// LayerNorm folded into the linear that consumes it (act 7; 8 = + QuickGELU):  LN(x) W^T + b with LN(x) = (x - mu) rstd g + beta
//   = rstd (x (g . W)^T - mu colsum(g . W)) + (beta W^T + b):  the GEMM runs on the UN-normalised rows x against the folded weight
// Wf = r16(g . W) and the epilogue applies the row's (mu, rstd) and the column's (colsum(Wf), beta W^T + b) — the LayerNorm pass
// (read x, write h: 4 bytes per element) and the h tensor disappear; what is left of it is pclip_row_stats_f16 (read x once).
// Rounding points: h = r16(LN(x)) is no longer formed and Wf is rounded instead of W; the result is rounded to fp16
// where the reference rounds the linear's output.  ONE expression for every kernel: the persistent and the ring kernel agree bit
// for bit (a row alone == the row in a batch).
__device__ __forceinline__ float ln_fold(float acc, float mu, float rstd, float cs, float bf) {
    return fmaf(rstd, fmaf(-mu, cs, acc), bf);
}

// ---- row statistics in ONE association order, whoever produces them -------------------------------------------------------------
// The (mean, rstd) pairs ln_fold consumes come from (sum, sum of squares) of the row's fp16 values.  They are produced either by
// the standalone pass (row_stats_kernel: reads x) or, for free, by the epilogue of the residual GEMM that writes x (act 9: the
// row-major store pass already holds the final values) — as PARTIALS per 64 columns, finished by stats_finalize_kernel.  A row's
// statistics must not depend on the producer (a row alone == the row in a batch, ring kernel == persistent kernel, any tile
// width), so the association order is fixed:
//   chunk (8 consecutive columns): v_dot2_f32_f16 chains over its four column pairs, in column order;
//   64-column group: butterfly over its 8 chunks (xor 1, 2, 4); 256-column block: the tree (g0 + g1) + (g2 + g3) of its groups
//   (= the butterfly continued with xor 8, 16: fp32 addition commutes, so every lane of the butterfly holds the tree's value);
//   row: the blocks added left to right.
// The GEMM epilogues write one partial per 64-column group whatever their tile width (three butterfly levels inside 8 lanes,
// one 8-byte store per row segment and group); stats_finalize_kernel and row_stats_kernel continue the same tree.
__device__ __forceinline__ void stats_chunk(const half8_t& h, float& s, float& q) {
    // two columns per instruction, straight from the packed halves: v_dot2_f32_f16 (fp32 accumulate), pairs in column order
    const half2_t one = {(half_t)1.f, (half_t)1.f};
    s = 0.f;
    q = 0.f;
#pragma unroll
    for (int j = 0; j < 8; j += 2) {
        const half2_t p = {h[j], h[j + 1]};
#if defined(__HIP_DEVICE_COMPILE__)
        s = __builtin_amdgcn_fdot2(p, one, s, false);
        q = __builtin_amdgcn_fdot2(p, p, q, false);
#endif
    }
}
// One butterfly level on the VALU (DPP / v_permlane16_swap) instead of a ds_bpermute through the LDS pipe (as __shfl_xor compiles:
// 320 of them per tile made the act-9 epilogue cost what the statistics pass it replaces cost).  Levels 4 and 8 use the mirror
// patterns: after the lower levels every lane of an aligned group holds the group's sum (identical bits: a + b == b + a), so
// "lane 7 - i" / "lane 15 - i" supply exactly the partner group's value that "lane i ^ 4" / "lane i ^ 8" would.
template <int OFF>
__device__ __forceinline__ float stats_level(float v) {
#if defined(__HIP_DEVICE_COMPILE__)
    const int iv = __builtin_bit_cast(int, v);
    if (OFF == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)iv, (unsigned)iv, false, false);   // {own, partner row} / {partner row, own}
        const unsigned a = r[0], b = r[1];
        return __builtin_bit_cast(float, a) + __builtin_bit_cast(float, b);
    }
    constexpr int ctrl = OFF == 1 ? 0xB1 : OFF == 2 ? 0x4E : OFF == 4 ? 0x141 : 0x140;   // quad_perm [1,0,3,2] / [2,3,0,1], row_half_mirror, row_mirror
    return v + __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(iv, iv, ctrl, 0xF, 0xF, false));
#else
    return v;
#endif
}
template <int LANES>   // 8, 16 or 32 consecutive lanes hold the chunks of one row segment
__device__ __forceinline__ void stats_butterfly(float& s, float& q) {
    s = stats_level<1>(s); q = stats_level<1>(q);
    s = stats_level<2>(s); q = stats_level<2>(q);
    s = stats_level<4>(s); q = stats_level<4>(q);
    if (LANES >= 16) { s = stats_level<8>(s); q = stats_level<8>(q); }
    if (LANES >= 32) { s = stats_level<16>(s); q = stats_level<16>(q); }
}
__device__ __forceinline__ float2_t stats_from_sums(float s, float q, int D, float eps) {
    const float mean = s / (float)D;
    const float var = fmaxf(fmaf(-mean, mean, q / (float)D), 0.f);
    return float2_t{mean, 1.f / sqrtf(var + eps)};
}

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
// c * 512 + lane * 8 + j.  ONE definition for layernorm_pf_kernel and for the row-panel LayerNorm inside the residual GEMM (linear_fast_kernel act 10): same
// operations in the same order, so both produce the same bits (tests/test_gpu_encoder.py::test_gemm_res_ln_equals_two_launches).
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

// act 10: the updated residual rows are read again by ANOTHER workgroup (possibly on another XCD, whose L2 is not coherent with this one's) inside the same launch:
// device-scope write-through stores (sc1) put them where every XCD sees them once the store has completed (vmcnt).
// (Inline assembly because no builtin stores 16 bytes with a scope.  hipcc's hazard recogniser does not look inside the statement: a VALU write to the data registers of a
// store of more than 8 bytes needs two wait states behind it on gfx940+, and the compiler re-used them in the very next instructions — every 16-byte chunk's first dword
// came out as an address fragment.  Hence the s_nop 1.)
__device__ __forceinline__ void st_out_dev(half_t* p, half8_t v) {
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("global_store_dwordx4 %0, %1, off sc1\n\ts_nop 1" ::"v"(p), "v"(v));
#endif
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
// (mean, rstd) of a row for a LayerNorm folded into the consuming linear (ln_fold).  NOT layernorm_kernel's arithmetic: that one is
// two-pass fp32 (mean, then the sum of squared deviations); these are ONE-pass sums (sum, sum of squares) in the canonical
// association order of stats_chunk / stats_butterfly and var = E[x^2] - mean^2, clamped at 0 — so that the GEMM epilogue that writes
// x can produce them from the tile it holds.  In fp32 the cancellation costs 2^-24 (E[x^2] / var) relative: measured on rows with
// a mean of 3 sigma and 50 sigma outlier channels (tests/test_gpu_encoder.py::test_gemm_ln_fold "trained") it stays below 1e-5.
// hv: the row's values (fp16-representable floats), lane-major chunks of 8 as every row kernel here holds them (chunk c*64 + lane =
// columns c*512 + 8*lane ..): the canonical (sum, sum of squares) of the row (see stats_chunk) and from them (mean, rstd).
// Lanes 0-31 / 32-63 of chunk set c are the 256-column blocks 2c / 2c+1; columns >= D contribute exact zeros.
template <int NCH>
__device__ __forceinline__ float2_t row_mean_rstd(const half8_t (&hv)[NCH], int D, int lane, float eps) {
    float S = 0.f, Q = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        float s = 0.f, q = 0.f;
        if (c * 512 + lane * 8 < D) stats_chunk(hv[c], s, q);
        stats_butterfly<32>(s, q);
        const float s0 = __shfl(s, 0, WAVE), q0 = __shfl(q, 0, WAVE), s1 = __shfl(s, 32, WAVE), q1 = __shfl(q, 32, WAVE);
        if (c * 512 < D) { S += s0; Q += q0; }
        if (c * 512 + 256 < D) { S += s1; Q += q1; }
    }
    return stats_from_sums(S, Q, D, eps);
}

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
