// CLIP encoder building blocks: the fp16 transformer towers of
// clip/model.py:169-238 and 341-354 as hand-written gfx950 kernels — MFMA linear layers with fused
// bias / QuickGELU / residual epilogues, fp32-statistics LayerNorm, and a whole-sequence attention
// kernel (the CLIP sequences, 50..257 tokens, fit one workgroup, so no online softmax is needed).
#include "pclip_gemm.h"
#include "pclip_epilogue.h"
#include <stdlib.h>
#include <type_traits>

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

// ---- fast kernel: N % BN == 0, 16-byte aligned C rows, no residual -----------------------------------------
// Persistent: one launch = at most `slots` resident workgroups; each walks output tiles round by round
// (round r covers tiles [r*G, (r+1)*G), XCD-remapped inside the round so that one XCD's L2 sees
// neighbouring tiles).  Around a tile boundary nothing drains the vector-memory counter:
//   K-loop(i) -> glds of K-tile 0 of tile i+1 -> epilogue(i) on LDS-only barriers (stores stay in flight)
//   -> bias glds(i+1) -> first barrier of K-loop(i+1) waits with vmcnt(#stores + 1): only the K-tile glds.
// Every vector-memory operation of this kernel is an LDS-DMA or a store (the bias row of the tile also
// travels by global_load_lds into a small double-buffered LDS strip), because hipcc answers any ordinary
// VGPR load issued beside an LDS-DMA with a full vmcnt(0) drain at its use (guide §5, trap (b)).
template <class C, bool HAS_BIAS, int ACT>
__global__ __launch_bounds__(C::NTHREADS, 2) void linear_fast_kernel(const half_t* __restrict__ A, int lda,
                                                                     const half_t* __restrict__ B, int ldb, int M, int N,
                                                                     int K, const half_t* __restrict__ bias,
                                                                     const float* __restrict__ scale,
                                                                     const float* __restrict__ shift,
                                                                     half_t* Cout, int ldc, int tiles_n,
                                                                     int ntiles, const half_t* residual = nullptr,
                                                                     const float* __restrict__ rowstats = nullptr,
                                                                     float* __restrict__ partials = nullptr, int band = 0,
                                                                     LnPanel lnp = LnPanel{}) {
    // ACT 10: ACT 6, and the LayerNorm that follows the residual add in a transformer block (clip/model.py:188-189: ln_2 behind `x + attn`, the next block's ln_1
    // behind `x + mlp`) WITHOUT a pass of its own over x: a row panel (BM rows x N) is complete when its N / BN tiles — computed by as many workgroups at about the same
    // time, which is what lets them share the A rows in L2 — have all been stored; every workgroup counts its finished tiles into the panel's counter (device-scope
    // atomic), and the one whose increment completes the panel normalises the panel's rows: it reads them back (device-scope loads; they are a few tens of microseconds
    // old and still on the chip) and writes y with ordinary streaming stores that drain while the launch multiplies on.  The separate pass was bound by HBM (620 MB at
    // 6 TB/s = 102 us); here the 310 MB of reads never reach HBM and the writes overlap the K-loops.  Row arithmetic = ln_row_pf, the pass's own: same bits.
    // No workgroup ever WAITS for another (nothing spins), so there is no forward-progress assumption.  A tile is counted two tiles late — after the K-loop of the
    // NEXT tile, whose last iteration drained the vector-memory counter of every wave (wait_vm<0> + barrier), i.e. without a wait of its own for the stores — and the
    // returned count is looked at another tile later; the last two tiles of a workgroup are settled behind the loop.
    // ACT 5: relu(r16(r16(r16(acc) * scale + shift) + residual)) — bn3 + `out += identity` + ReLU of a bottleneck (clip/model.py:49-52)
    // in the epilogue of its conv3 GEMM; the residual rows are read row-major in the coalesced store pass.
    // ACT 9: ACT 6 + the row-statistics partials of the updated rows (stats_chunk / stats_butterfly) into `partials` [M][N/64][2]
    // ACT 6: r16(residual + r16(acc + bias)) — `x = x + attn(..)` / `x = x + mlp(..)` of a transformer block (clip/model.py:188-189)
    // in the epilogue of out_proj / c_proj; Cout may BE residual (the residual stream is updated in place: every 16-byte chunk is
    // read and then written by the same thread)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool M16 = true;                                               // 16x16x32 MFMAs (accumulator layout of pgemm::mainloop_sr)
    half_t* bias_lds = reinterpret_cast<half_t*>(smem + C::LDS_BYTES);       // [2][BN] fp16
    float* affine_lds = reinterpret_cast<float*>(smem + C::LDS_BYTES);       // ACT >= 2: [2][ scale BN | shift BN ] fp32
    constexpr bool LNF = ACT == 7 || ACT == 8;                                // LayerNorm folded into this linear (ln_fold)
    constexpr bool AFFINE = ACT == 2 || ACT == 3 || ACT == 5 || LNF;          // LNF: the strips hold colsum(Wf) | folded bias
    constexpr int STRIP_BYTES = AFFINE ? 2 * 2 * C::BN * 4 : 2 * C::BN * 2;   // then 256 bytes of scrap for the L2 prefetch
    constexpr int NSTAT = LNF ? (C::BM * 8 + 1023) / 1024 : 0;                // LDS-DMA pieces of a tile's (mean, rstd) rows
    float* stats_lds = reinterpret_cast<float*>(smem + C::LDS_BYTES + STRIP_BYTES + 256);   // LNF: [2][BM][2] fp32
    constexpr bool LNP = ACT == 10;
    // LNP: [2][2] panels to normalise behind this tile's epilogue (-1: none) | thread 0's bookkeeping: [8] owned panels not yet complete, their number, the one whose
    // count is on its way back | [12] + number: the panels settled behind the loop
    int* lnp_flag = reinterpret_cast<int*>(smem + C::LDS_BYTES + STRIP_BYTES + 256);
    int* lnp_own = lnp_flag + 4;
    float* lnp_gb = reinterpret_cast<float*>(smem + C::LDS_BYTES + STRIP_BYTES + 256 + 128);   // LNP: [gamma | beta][plane][2 * 256] fp32 for the whole launch (layernorm_pf_kernel's layout)
    constexpr int LNP_OQ = 8, LNP_NOWN = 8, LNP_CHK = 9, LNP_LIST = 10, LNP_NL = 22, LNP_ORPH = 1 << 16;
    const int G = gridDim.x;
    int tile = pgemm::xcd_remap(blockIdx.x, G);
    if (tile >= ntiles) return;
    // Linear tile id -> (row block, column tile).  band == 0: column tiles fastest (a round covers whole rows of tiles).  band > 0
    // (PCLIP_GEMM_BAND, tools/ab_band.py): the output is walked in BANDS of `band` column tiles, row blocks fastest inside a band, so
    // that for half of the launch every XCD multiplies against the same `band` weight panels (N = 3072, band 6: 2.4 MB of the 4 MiB L2
    // instead of 4.7) at the price of reading the activations once per band.
    const int tiles_m_all = ntiles / tiles_n;
    auto decomp = [&](int t, int& tm, int& tn) {
        if (band < 0) t = ntiles - 1 - t;                   // band == -1: the tiles in DESCENDING order (PCLIP_GEMM_REV: the rows the producer wrote last are read first)
        if (band <= 0 || band >= tiles_n) { tm = t / tiles_n; tn = t - tm * tiles_n; return; }
        const int per_band = tiles_m_all * band, bnd = t / per_band, r = t - bnd * per_band;
        const int w = band < tiles_n - bnd * band ? band : tiles_n - bnd * band;
        tm = r / w;
        tn = bnd * band + r - tm * w;
    };
    int p = 0;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wn = wave % C::WN;
    // M16: buffer-descriptor staging + pipelined K-loop; eight-wave tiles split the DMA issue by wave role (pgemm::TilePairR)
    using TP = std::conditional_t<(PCLIP_DMA_ROLES && C::NWAVES == 8), pgemm::TilePairR<C>, pgemm::TilePair<C>>;
    TP tp;
    // The bias enters as the INITIAL VALUE of the accumulators (fp32 copy of the fp16 bias: r16(bias + sum) instead of
    // r16(sum + bias), same value up to fp32 summation order), so the epilogue has no bias pass.  Its strip is copied one
    // tile ahead (double-buffered); every wave copies the same BN values: uniform vmcnt bookkeeping.
    auto copy_bias = [&](int t, int par) {
        int tm_, tn;
        decomp(t, tm_, tn);
        if (lane < C::BN / 8)
            __builtin_amdgcn_global_load_lds((pgemm::gbl_ptr_t)(bias + tn * C::BN + lane * 8), (pgemm::lds_ptr_t)(bias_lds + par * C::BN), 16, 0, 0);
    };
    // eval-mode BatchNorm (+ReLU) of the ResNet tower (clip/model.py:43-52) as the epilogue of the convolution's GEMM: the
    // per-column scale / shift strips travel like the bias strip, one tile ahead
    auto copy_affine = [&](int t, int par) {
        int tm_, tn;
        decomp(t, tm_, tn);
        if (lane < C::BN / 4) {
            __builtin_amdgcn_global_load_lds((pgemm::gbl_ptr_t)(scale + tn * C::BN + lane * 4), (pgemm::lds_ptr_t)(affine_lds + par * 2 * C::BN), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((pgemm::gbl_ptr_t)(shift + tn * C::BN + lane * 4), (pgemm::lds_ptr_t)(affine_lds + par * 2 * C::BN + C::BN), 16, 0, 0);
        }
    };
    // LNF: the (mean, rstd) pairs of the tile's BM rows, one tile ahead like the strips; rowstats is allocated in whole 256-row
    // blocks, so the last tile reads (never used) padding instead of running off the end
    auto copy_stats = [&](int t, int par) {
        int tm, tn_;
        decomp(t, tm, tn_);
#pragma unroll
        for (int i = 0; i < NSTAT; ++i)
            if (NSTAT * 128 == C::BM || lane < (C::BM - i * 128) / 2)
                __builtin_amdgcn_global_load_lds((pgemm::gbl_ptr_t)(rowstats + ((size_t)tm * C::BM + i * 128 + lane * 2) * 2),
                                                 (pgemm::lds_ptr_t)(stats_lds + par * C::BM * 2 + i * 256), 16, 0, 0);
    };
    if constexpr (LNP) {
        for (int i = tid; i < 2 * 512; i += C::NTHREADS) {
            const int pos = (i >> 9) * 256 + ((i & 511) >> 3) * 4 + (i & 3), plane = (i >> 2) & 1;
            lnp_gb[plane * 2 * 256 + pos] = i < N ? lnp.gamma[i] : 0.f;
            lnp_gb[(2 + plane) * 2 * 256 + pos] = i < N ? lnp.beta[i] : 0.f;
        }
    }
    if (HAS_BIAS || AFFINE) {
        if (AFFINE) copy_affine(tile, 0); else copy_bias(tile, 0);
        if (LNF) copy_stats(tile, 0);
        pgemm::wait_vm<0>();
        pgemm::lds_barrier();
    }
    {
        int tm, tn;
        decomp(tile, tm, tn);
        tp.prepare(A, lda, B, ldb, M, N, tm * C::BM, tn * C::BN, wave, lane);
        tp.stage(0, smem + p * C::STAGE_BYTES, wave);
    }
    constexpr int PST = ACT == 9 ? 1 : 0;                                     // act 9: one store of statistics partials per pass
    // vector-memory operations a wave issues between a tile's K-tile 0 pieces and the first wait of its K-loop: the previous tile's stores + the strip copies
    constexpr int YOUNGER = C::NH * C::NPASS * (1 + PST) + (AFFINE ? 2 : (HAS_BIAS ? 1 : 0)) + NSTAT;
    bool prev_full = false;
    int parity = 0;
    // LNP: `stored` = panel of the tile whose stores are in flight (not yet counted), `counted` = panel of the tile whose count is on its way back in `ticket` (thread 0)
    int stored = -1, counted = -1, ticket = 0, chkv = 0;
    if (LNP && tid == 0) { lnp_own[LNP_NOWN] = 0; lnp_own[LNP_CHK] = -1; }
    // WHO normalises a complete panel.  "The workgroup whose count completes it" piles the work up: a workgroup that has normalised one panel is late from then on, so
    // it is the last to arrive at its following panels too and normalises those as well (measured: 0 - 7 panels per workgroup, the launch as slow as the busiest).  So
    // every panel has an OWNER — the workgroup that computes one designated tile of it (below), a third of everybody's tiles at N = 768 — which looks at
    // the counter of its oldest unfinished panel once per tile (a device-scope load, requested behind one K-loop and read behind the next) and normalises the panel
    // once it reads the full count.  Still nobody waits: behind its last tile an owner adds LNP_ORPH to the counters of the panels it still holds — complete ones
    // it normalises on the spot, incomplete ones are now ORPHANS, normalised by the workgroup whose count completes them (it sees the flag in the value its
    // atomic returns).  The atomics on one counter are totally ordered, so exactly one of the two happens.
    // (Which tile: in round r — r = (LAST tile of the panel) / G — the column tile (r (G % tiles_n + 1)) % tiles_n.  In the ascending order a workgroup's column
    // tile advances by G % tiles_n per round, so it owns a panel exactly every tiles_n-th round: 3 of its 9 tiles at N = 768, nobody more.  With the round of the
    // panel's FIRST tile the two workgroups at a round's seam owned 7 and 0.)
    auto owner_tile = [&](int tm, int tn) { return tn == (((tm * tiles_n + tiles_n - 1) / G) * (G % tiles_n + 1)) % tiles_n; };
    // rows of panel lp (complete: every tile of it has been counted) -> lnp.y.  No barrier: a wave reads x, the launch's gamma / beta copy in LDS, and writes y.
    auto ln_panel = [&](int lp) {
        if constexpr (LNP) {
            constexpr int NCH = 2;                                      // 512 <= N <= 1024, N % 128 == 0 (launcher)
            const float* gb = lnp_gb;
            int lt = tid;                                             // opaque copy: the constants below are formed here, not hoisted across the K-loop (they spilled)
#if defined(__HIP_DEVICE_COMPILE__)
            asm volatile("" : "+v"(lt));
#endif
            const int ln = lt & 63;
            const int r0 = lp * C::BM, rows = M - r0 < C::BM ? M - r0 : C::BM;
            // FOUR rows per wave instruction: the 16 lanes of a DPP row own one row of x, lane p of them the columns of ln_row_pf's lanes p, p + 16, p + 32, p + 48
            // ("slots" 0 - 3: 8 columns of the first 512 each, 8 more of the columns beyond for the slots those reach) — every lane is busy at N = 768 (a quarter
            // idles in the one-row-per-wave form), the per-row scalars (two divisions, a square root) and the reductions cost a quarter, and a reduction is two
            // in-lane levels + four DPP levels.  SAME BITS as ln_row_pf: a slot's partial sums run over its columns in ln_row_pf's order, and
            // (P0 + P2) + (P1 + P3) followed by row16_sum_x is wave_sum's tree (levels ^ 32, ^ 16 pair slots, ^ 8 ... ^ 1 pair lanes of the row).
            // Rows and columns beyond the panel / N fail the descriptors' bounds checks (loads answer zeros, stores are dropped): every quad issues the SAME eight
            // loads and eight stores, so the waits are counted.  Loads: inline assembly, device scope (sc1) — issued through builtins hipcc serialised them with
            // vmcnt(0); their destination registers are not touched before the counted wait and the pin behind it.
            uint4_t rx;
            pgemm::rsrc_t ry;
#if defined(__HIP_DEVICE_COMPILE__)
            {
                const uint64_t ax = (uint64_t)(Cout + (size_t)r0 * ldc), ay = (uint64_t)(lnp.y + (size_t)r0 * N);
                rx = uint4_t{(uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)ax), (uint32_t)__builtin_amdgcn_readfirstlane((uint32_t)(ax >> 32)) & 0xffffu,
                             (uint32_t)__builtin_amdgcn_readfirstlane(rows * ldc * 2), 0x00020000u};
                const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)ay), hi = __builtin_amdgcn_readfirstlane((uint32_t)(ay >> 32));
                ry = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, __builtin_amdgcn_readfirstlane(rows * N * 2), 0x00020000);
            }
#endif
            constexpr int NIT = C::BM / (C::NWAVES * 4), NOP = 8;       // quads per wave; vector-memory operations of a quad (loads, and stores)
            static_assert(C::BM % (C::NWAVES * 8) == 0, "quads are walked in pairs (two register buffers)");
            const int ns1 = (N - 512) >> 7;                           // slots that reach beyond column 512 (uniform): 2 at N = 768
            const int p16 = ln & 15, g4 = ln >> 4;
            const float fN = (float)N;
            half8_t bufa[8], bufb[8];
            auto load = [&](half8_t (&h)[8], int it) {
                const int r = (it * C::NWAVES + wave) * 4 + g4;
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int d = (q >> 2) * 512 + ((q & 3) * 16 + p16) * 8;
                    const int off = d < N ? (r * ldc + d) * 2 : 0x7ffffff0;
#if defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("buffer_load_dwordx4 %0, %1, %2, 0 offen sc1" : "=v"(h[q]) : "v"(off), "s"(rx));
#endif
                }
            };
            auto quad = [&](half8_t (&h)[8], int it) {
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                for (int q = 0; q < 8; ++q) asm volatile("" : "+v"(h[q]));                 // the values exist from here on (behind the wait)
#endif
                const int r = (it * C::NWAVES + wave) * 4 + g4;
                float P[4];
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    float a = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) a += (float)h[sl][j];
                    if (sl < ns1) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) a += (float)h[4 + sl][j];
                    }
                    P[sl] = a;
                }
                const float mean = row16_sum_x((P[0] + P[2]) + (P[1] + P[3])) / fN;
                float t[8][8];
#pragma unroll
                for (int sl = 0; sl < 4; ++sl) {
                    float a = 0.f;
#pragma unroll
                    for (int j = 0; j < 8; ++j) { t[sl][j] = (float)h[sl][j] - mean; a = ln_sq_acc(t[sl][j], a); }
                    if (sl < ns1) {
#pragma unroll
                        for (int j = 0; j < 8; ++j) { t[4 + sl][j] = (float)h[4 + sl][j] - mean; a = ln_sq_acc(t[4 + sl][j], a); }
                    }
                    P[sl] = a;
                }
                const float rstd = 1.f / sqrtf(row16_sum_x((P[0] + P[2]) + (P[1] + P[3])) / fN + lnp.eps);
#pragma unroll
                for (int q = 0; q < 8; ++q) {
                    const int sl = q & 3, c = q >> 2, d = c * 512 + (sl * 16 + p16) * 8;
                    half8_t o;
                    if (c == 0 || sl < ns1) {
                        const float* gp = gb + c * 256 + (sl * 16 + p16) * 4;
                        const float4_t ga = *reinterpret_cast<const float4_t*>(gp), gc = *reinterpret_cast<const float4_t*>(gp + NCH * 256);
                        const float4_t ba = *reinterpret_cast<const float4_t*>(gp + 2 * NCH * 256), bc = *reinterpret_cast<const float4_t*>(gp + 3 * NCH * 256);
#pragma unroll
                        for (int j = 0; j < 4; ++j) {
                            o[j] = (half_t)r16(ln_affine_dev(t[q][j], rstd, ga[j], ba[j]));
                            o[j + 4] = (half_t)r16(ln_affine_dev(t[q][j + 4], rstd, gc[j], bc[j]));
                        }
                    }
#if defined(__HIP_DEVICE_COMPILE__)
                    __builtin_amdgcn_raw_buffer_store_b128(__builtin_bit_cast(uint4_t, o), ry, d < N ? (r * N + d) * 2 : 0x7ffffff0, 0, PCLIP_NT_STORE ? 2 : 0);
#endif
                }
            };
            // issue order per wave: L0 L1 | quad 0, S0, L2 | quad 1, S1, L3 | ...: counted waits — first quad vmcnt(NOP) (only L1 is younger), middle quads
            // vmcnt(2 NOP) (S(b-1) and L(b+1)), last quad vmcnt(NOP) (S(b-1))
            load(bufa, 0);
#pragma unroll 1
            for (int it = 0; it < NIT; it += 2) {
                load(bufb, it + 1);
                if (it == 0) pgemm::wait_vm<NOP>(); else pgemm::wait_vm<2 * NOP>();
                quad(bufa, it);
                if (it + 2 < NIT) { load(bufa, it + 2); pgemm::wait_vm<2 * NOP>(); } else pgemm::wait_vm<NOP>();
                quad(bufb, it + 1);
            }
            if (lt == 0) lnp.cnt[lp] = 0;                            // the counter is this workgroup's now: zero for the next launch
        }
    };
    for (; tile < ntiles; tile += G, parity ^= 1) {
        int tile_m, tile_n;
        decomp(tile, tile_m, tile_n);
        const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;
        const bool full = m0 + C::BM <= M;                    // workgroup-uniform: every row of the tile exists
        pgemm::Acc<C> acc;
        if (HAS_BIAS) {
            // the strip of this tile was copied one tile ago; with two or more K-tiles the K-loop's vmcnt(0) + barrier in
            // between made it visible, a single K-tile (K = 64) only has the counted wait: close that case explicitly
            if (K == pgemm::BK) { pgemm::wait_vm<0>(); pgemm::lds_barrier(); }
            const half_t* bl = bias_lds + parity * C::BN + wn * (C::BN / C::WN);
#pragma unroll
            for (int j = 0; j < C::TN; ++j)
#pragma unroll
                for (int g = 0; g < 4; ++g) {
                    const int coff = (g & 1) * 16 + 4 * (lane >> 4);                               // columns of elements 4g .. 4g+3
                    const half4_t b = *reinterpret_cast<const half4_t*>(bl + j * 32 + coff);
#pragma unroll
                    for (int e = 0; e < 4; ++e)
#pragma unroll
                        for (int i = 0; i < C::TM; ++i) acc.v[i][j][4 * g + e] = (float)b[e];
                }
            // next tile's strip (the last tile re-copies its own: the count of younger operations stays the same)
            copy_bias(tile + G < ntiles ? tile + G : tile, parity ^ 1);
        }
        if (AFFINE) copy_affine(tile + G < ntiles ? tile + G : tile, parity ^ 1);
        if (LNF) copy_stats(tile + G < ntiles ? tile + G : tile, parity ^ 1);
        // (Measured and rejected, profiles/r03_ab_rejected.txt: pulling the residual tile's 1024 lines into L2 during the K-loop with one
        // 4-byte LDS-DMA per line — out_proj 301 -> 341 us, c_proj 854 -> 879 us: 1024 more requests per tile in the queue the operand
        // DMAs wait in.)
        const int next = tile + G;
        pgemm::mainloop_sr<C, YOUNGER, !HAS_BIAS, TP>(tp, K / pgemm::BK, smem, acc, p, prev_full, wave, lane);
        if constexpr (LNP) {
            // K >= 2 K-tiles (launcher): the last iteration began with wait_vm<0> + barrier — every store of the previous tile, from every wave, has completed
            if (tid == 0) {
                // the count of tile i - 2 is back: did it complete a panel nobody owns any more?
                lnp_flag[2 * parity] = counted >= 0 && (ticket & 0xffff) == tiles_n - 1 && (ticket & LNP_ORPH) ? counted : -1;
                int no = lnp_own[LNP_NOWN], ck = lnp_own[LNP_CHK], f1 = -1;
                if (ck >= 0 && (chkv & 0xffff) == tiles_n) {          // my oldest panel is complete (it cannot be an orphan: only I could have made it one)
                    f1 = ck;
                    for (int k = 0; k + 1 < no; ++k) lnp_own[k] = lnp_own[k + 1];
                    --no;
                }
                lnp_flag[2 * parity + 1] = f1;
                if (owner_tile(tile_m, tile_n)) {
                    if (no < LNP_OQ) lnp_own[no++] = tile_m;
                    else __hip_atomic_fetch_add(lnp.cnt + tile_m, LNP_ORPH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);   // (list full: this tile is not counted yet, so the panel cannot be complete)
                }
                lnp_own[LNP_NOWN] = no;
                if (stored >= 0) ticket = __hip_atomic_fetch_add(lnp.cnt + stored, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                ck = no > 0 ? lnp_own[0] : -1;
                if (ck >= 0) chkv = __hip_atomic_load(lnp.cnt + ck, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                lnp_own[LNP_CHK] = ck;
            }
            counted = stored;
            stored = tile_m;
        }
        if (next < ntiles) {                                  // buffer p is free: prefetch the next tile's K-tile 0
            int tm, tn;
            decomp(next, tm, tn);
            tp.prepare(A, lda, B, ldb, M, N, tm * C::BM, tn * C::BN, wave, lane);
            tp.stage(0, smem + p * C::STAGE_BYTES, wave);
        }
        char* stg = smem + (p ^ 1) * C::STAGE_BYTES;          // buffer of the last K-tile, reused after a barrier
        int etid = tid;                                       // opaque copy: the epilogue's lane constants are recomputed per tile (pgemm::epilogue_f16)
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(etid));
#endif
        const int col = n0 + 8 * (etid % C::CPR);
        // LNF: the lane's column strips (colsum | folded bias for its 4 columns of every (j, g & 1)) and row statistics ((mean, rstd)
        // of its row in every (i, g >> 1)) are read from LDS ONCE per tile into registers (the K-loop's fragment registers are free
        // here): read inside `pre` they were 96 LDS reads per lane and slab, re-issued behind every staging write, and the epilogue
        // cost as much as the LayerNorm pass it replaces.
        float4_t lcs[LNF ? C::TN : 1][2], lbf[LNF ? C::TN : 1][2];
        float2_t lms[LNF ? C::TM : 1][2];
        if (LNF) {
            const int cq = 4 * (lane >> 4), rq = lane & 15;
#pragma unroll
            for (int j = 0; j < C::TN; ++j)
#pragma unroll
                for (int b = 0; b < 2; ++b) {
                    const float* st = affine_lds + parity * 2 * C::BN + wn * (C::BN / C::WN) + j * 32 + b * 16 + cq;
                    lcs[j][b] = *reinterpret_cast<const float4_t*>(st);
                    lbf[j][b] = *reinterpret_cast<const float4_t*>(st + C::BN);
                }
#pragma unroll
            for (int i = 0; i < C::TM; ++i)
#pragma unroll
                for (int a = 0; a < 2; ++a)
                    lms[i][a] = *reinterpret_cast<const float2_t*>(stats_lds + (parity * C::BM + (wave / C::WN) * (C::BM / C::WM) + i * 32 + a * 16 + rq) * 2);
        }
        auto pre = [&](int i, int j, int coff, float4_t v, int rl, int g) {
            if (ACT == 1) return quick_gelu16x4(v);
            half4_t h;
            if (LNF) {
                const float4_t cs = lcs[j][g & 1], bf = lbf[j][g & 1];
                const float2_t ms = lms[i][g >> 1];
                float4_t y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = ln_fold(v[e], ms[0], ms[1], cs[e], bf[e]);
                if (ACT == 8) return quick_gelu16x4(y);
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (half_t)y[e];
                return h;
            }
            if (AFFINE) {
                const float* st = affine_lds + parity * 2 * C::BN + wn * (C::BN / C::WN) + j * 32 + coff;
                const float4_t sc = *reinterpret_cast<const float4_t*>(st), sh = *reinterpret_cast<const float4_t*>(st + C::BN);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float y = r16(r16(v[e]) * sc[e] + sh[e]);              // bn(conv(x)): the conv output is an fp16 tensor
                    if (ACT == 3) y = fmaxf(y, 0.f);
                    h[e] = (half_t)y;
                }
                return h;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = (half_t)v[e];
            return h;
        };
        // Residual operand: the NPASS 16-byte chunks a thread adds to in slab h are requested TOGETHER in the slab hook, before
        // the slab is staged (they fly during the LDS write pass) — Cout may alias residual (in-place residual stream), so the
        // compiler cannot hoist a later pass's load above an earlier pass's store by itself: load -> wait -> store per pass was
        // 16 dependent round trips per tile.
        constexpr bool RES = ACT == 5 || ACT == 6 || ACT == 9 || ACT == 10;
        // (measured, profiles/r03_ab_epilogue_pipe.txt: c_fc + QuickGELU 1001 -> 972 us; the bias-only and residual epilogues do not profit — their phases
        // are bound by the LDS write rate / the stores' address path / the residual loads' latency one after the other either way — and keep epilogue_f16)
        constexpr bool PIPE = PCLIP_EPI_PIPE && (ACT == 1 || ACT == 8 || PCLIP_EPI_PIPE == 2) && !LNP && C::BM == 256 && C::BN == 256 && C::WM == 2 && C::WN == 4;
        half8_t rr[RES ? C::NPASS : 1];
        // PIPE: slab k = 32-row block k of both wave rows, four passes of 16 rows; its residual chunks go to rr[(k & 1) * 4 + ps], requested one interval ahead
        auto ahead = [&](int k) {
            if (!RES) return;
#pragma unroll
            for (int ps = 0; ps < 4; ++ps) {
                const int rs = etid / C::CPR + ps * 16, r = (rs >> 5) * 128 + k * 32 + (rs & 31);
                if (full || m0 + r < M) rr[RES ? (k & 1) * 4 + ps : 0] = ld_half8(residual + (size_t)(m0 + r) * ldc + col);
            }
        };
        auto slab = [&](int h) {
            if (!RES) return;
#pragma unroll
            for (int ps = 0; ps < C::NPASS; ++ps) {
                const int r = h * C::HR + etid / C::CPR + ps * C::ROWS_PER_PASS;
                if (full || m0 + r < M) rr[ps] = ld_half8(residual + (size_t)(m0 + r) * ldc + col);
            }
        };
        auto add_res = [&](int pass, half8_t h) {
            const half8_t x = rr[RES ? pass % C::NPASS : 0];     // (PIPE: pass = 4 k + ps -> (k & 1) * 4 + ps = pass % 8, NPASS = 8)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const float y = r16((float)x[j] + (float)h[j]);
                h[j] = (half_t)(ACT == 5 ? fmaxf(y, 0.f) : y);
            }
            return h;
        };
        // act 9: (sum, sum of squares) of the row segment this tile covers, from the values just stored; the CPR lanes of a row are
        // consecutive, lane c == 0 of each writes the segment's slots (every lane takes part in the butterfly: `valid` only
        // predicates the store)
        auto put_partials = [&](int r, int c, const half8_t& hv, bool valid) {
            float ps, pq;
            stats_chunk(hv, ps, pq);
            stats_butterfly<8>(ps, pq);                        // the 8 lanes of a 64-column group
            if (valid && (c & 7) == 0)
                *reinterpret_cast<float2_t*>(partials + ((size_t)(m0 + r) * (N >> 6) + (n0 >> 6) + (c >> 3)) * 2) = float2_t{ps, pq};
        };
        if constexpr (PIPE) {
            static_assert(C::NPASS == 8, "rr[pass % NPASS] pairs slab parity and pass");
            if (full)
                pgemm::epilogue_pipe<C>(acc, stg, ahead, pre, [&](int r, int c, int pass, half8_t h) {
                    const size_t o = (size_t)(m0 + r) * ldc + col;
                    if (RES) h = add_res(pass, h);
                    st_out(Cout + o, h);
                    if (ACT == 9) put_partials(r, c, h, true);
                });
            else
                pgemm::epilogue_pipe<C>(acc, stg, ahead, pre, [&](int r, int c, int pass, half8_t h) {
                    const size_t o = (size_t)(m0 + r) * ldc + col;
                    if (RES) h = add_res(pass, h);
                    if (m0 + r < M) st_out(Cout + o, h);
                    if (ACT == 9) put_partials(r, c, h, m0 + r < M);
                });
            prev_full = full;
            continue;
        }
        if (full)
            pgemm::epilogue_f16<C, M16>(acc, stg, slab, pre, [&](int r, int c, int pass, half8_t h) {
                const size_t o = (size_t)(m0 + r) * ldc + col;
                if (RES) h = add_res(pass, h);
                if (LNP) st_out_dev(Cout + o, h); else st_out(Cout + o, h);
                if (ACT == 9) put_partials(r, c, h, true);
            });
        else
            pgemm::epilogue_f16<C, M16>(acc, stg, slab, pre, [&](int r, int c, int pass, half8_t h) {
                const size_t o = (size_t)(m0 + r) * ldc + col;
                if (RES) h = add_res(pass, h);
                if (m0 + r < M) { if (LNP) st_out_dev(Cout + o, h); else st_out(Cout + o, h); }
                if (ACT == 9) put_partials(r, c, h, m0 + r < M);
            });
        prev_full = full;
        if constexpr (LNP) {
#pragma unroll 1
            for (int k = 0; k < 2; ++k) {
                const int lp = __builtin_amdgcn_readfirstlane(lnp_flag[2 * parity + k]);       // written before this epilogue's barriers
                if (lp >= 0) ln_panel(lp);
            }
        }
    }
    if constexpr (LNP) {
        // behind the loop: the last tile's stores, then both outstanding counts
        pgemm::wait_vm<0>();
        __syncthreads();
        if (tid == 0) {
            int* list = lnp_own + LNP_LIST;
            int nl = 0;
            auto mine = [&](int t) { return (t & 0xffff) == tiles_n - 1 && (t & LNP_ORPH); };
            if (counted >= 0 && mine(ticket)) list[nl++] = counted;
            if (stored >= 0 && mine(__hip_atomic_fetch_add(lnp.cnt + stored, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT))) list[nl++] = stored;
            // the panels this workgroup still owns: complete -> normalised here; incomplete -> orphans from now on
            const int no = lnp_own[LNP_NOWN];
            for (int k = 0; k < no; ++k)
                if ((__hip_atomic_fetch_add(lnp.cnt + lnp_own[k], LNP_ORPH, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) & 0xffff) == tiles_n) list[nl++] = lnp_own[k];
            lnp_own[LNP_NL] = nl;
        }
        __syncthreads();
        const int nl = __builtin_amdgcn_readfirstlane(lnp_own[LNP_NL]);
#pragma unroll 1
        for (int k = 0; k < nl; ++k) ln_panel(__builtin_amdgcn_readfirstlane(lnp_own[LNP_LIST + k]));
    }
}

// ---- 3x3 convolution (stride 1, pad 1, NHWC) + eval BatchNorm (+ReLU) as an implicit GEMM -----------------------------------
// Same persistent structure as linear_fast_kernel; the A operand is gathered by pgemm::ConvGather instead of read from an
// im2col matrix (clip/model.py:20-22, 45-46: conv2 / bn2 / relu of every bottleneck).  w is [Cout, ky, kx, Cin].
template <class C, int ACT>
__global__ __launch_bounds__(C::NTHREADS, 2) void conv3x3_fast_kernel(const half_t* __restrict__ x, const half_t* __restrict__ zero,
                                                                      const half_t* __restrict__ w, int H, int W, int Cin, int M,
                                                                      int N, const float* __restrict__ scale,
                                                                      const float* __restrict__ shift, half_t* __restrict__ Cout,
                                                                      int tiles_n, int ntiles) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* affine_lds = reinterpret_cast<float*>(smem + C::LDS_BYTES);       // [2][ scale BN | shift BN ] fp32
    const int G = gridDim.x;
    int tile = pgemm::xcd_remap(blockIdx.x, G);
    if (tile >= ntiles) return;
    const int nt = (9 * Cin + pgemm::BK - 1) / pgemm::BK, ldb = nt * pgemm::BK;    // w rows are zero-padded to the K-tile (Cin < 64)
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wn = wave % C::WN;
    pgemm::ConvGather<C> ga{x, zero, H, W, Cin, M, {}, {}};
    auto copy_affine = [&](int t, int par) {
        const int tn = t - (t / tiles_n) * tiles_n;
        if (lane < C::BN / 4) {
            __builtin_amdgcn_global_load_lds((pgemm::gbl_ptr_t)(scale + tn * C::BN + lane * 4), (pgemm::lds_ptr_t)(affine_lds + par * 2 * C::BN), 16, 0, 0);
            __builtin_amdgcn_global_load_lds((pgemm::gbl_ptr_t)(shift + tn * C::BN + lane * 4), (pgemm::lds_ptr_t)(affine_lds + par * 2 * C::BN + C::BN), 16, 0, 0);
        }
    };
    auto stage0 = [&](int t, int pbuf) {                       // K-tile 0 of tile t into buffer pbuf (ga prepared for t)
        const int tm = t / tiles_n, tn = t - tm * tiles_n;
        char* a = smem + pbuf * C::STAGE_BYTES;
        ga.stage(0, a);
        pgemm::stage_tile<C::BN, C::NWAVES>(w, ldb, tn * C::BN, N, 0, a + C::A_BYTES, wave, lane);
    };
    copy_affine(tile, 0);
    pgemm::wait_vm<0>();
    pgemm::lds_barrier();
    int p = 0;
    ga.prepare((tile / tiles_n) * C::BM);
    stage0(tile, p);
    constexpr int YOUNGER = C::NH * C::NPASS + 2;
    bool prev_full = false;
    int parity = 0;
    for (; tile < ntiles; tile += G, parity ^= 1) {
        const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
        const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;
        const bool full = m0 + C::BM <= M;
        pgemm::Acc<C> acc;
        copy_affine(tile + G < ntiles ? tile + G : tile, parity ^ 1);
        pgemm::mainloop_g<C, YOUNGER, true, true>([&](int t, char* dst) { ga.stage(t, dst); }, w, ldb, N, nt, n0, smem, acc, p, prev_full);
        const int next = tile + G;
        if (next < ntiles) {                                  // buffer p is free: prefetch the next tile's K-tile 0
            ga.prepare((next / tiles_n) * C::BM);
            stage0(next, p);
        }
        char* stg = smem + (p ^ 1) * C::STAGE_BYTES;
        const int col = n0 + 8 * (tid % C::CPR);
        auto pre = [&](int, int j, int coff, float4_t v, int rl, int g) {
            half4_t h;
            const float* st = affine_lds + parity * 2 * C::BN + wn * (C::BN / C::WN) + j * 32 + coff;
            const float4_t sc = *reinterpret_cast<const float4_t*>(st), sh = *reinterpret_cast<const float4_t*>(st + C::BN);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float y = r16(r16(v[e]) * sc[e] + sh[e]);
                if (ACT == 3) y = fmaxf(y, 0.f);
                h[e] = (half_t)y;
            }
            return h;
        };
        if (full)
            pgemm::epilogue_f16<C, true>(acc, stg, [](int) {}, pre,
                                   [&](int r, int, int, half8_t h) { st_half8(Cout + (size_t)(m0 + r) * N + col, h); });
        else
            pgemm::epilogue_f16<C, true>(acc, stg, [](int) {}, pre, [&](int r, int, int, half8_t h) {
                if (m0 + r < M) st_half8(Cout + (size_t)(m0 + r) * N + col, h);
            });
        prev_full = full;
    }
}

// ---- generic kernel: any M, N, leading dimensions; optional residual; one 128x128 tile per workgroup ------
__global__ __launch_bounds__(256, 2) void linear_generic_kernel(const half_t* __restrict__ A, int lda,
                                                                const half_t* __restrict__ B, int ldb, int M, int N,
                                                                int K, LinearEpi epi, int tiles_n) {
    using C = pgemm::CfgSmall;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int swz = pgemm::xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = swz / tiles_n, tile_n = swz - tile_m * tiles_n;
    const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;
    const int tid = threadIdx.x, wave = tid >> 6, wn = wave % C::WN;
    int p = 0;
    pgemm::stage_first<C>(A, lda, B, ldb, M, N, m0, n0, smem, p);
    pgemm::Acc<C> acc;
    pgemm::mainloop<C, 0>(A, lda, B, ldb, M, N, K, m0, n0, smem, acc, p, false);
    const half_t* __restrict__ bias = epi.bias;
    const half_t* __restrict__ residual = epi.residual;
    const int act = epi.act, ldc = epi.ldc;
    const int col = n0 + 8 * (tid % C::CPR);
    pgemm::epilogue_f16<C>(
        acc, smem + (p ^ 1) * C::STAGE_BYTES, [](int) {},
        [&](int, int j, int coff, float4_t v, int rl, int g) {
            const int n = n0 + wn * (C::BN / C::WN) + j * 32 + coff;
            half4_t h;
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                float x = v[e];
                if (bias) x += (float)bias[n + e < N ? n + e : N - 1];
                x = r16(x);
                if (act == 1) x = quick_gelu16(x);
                if (act >= 2) {
                    const int nn = n + e < N ? n + e : N - 1;
                    x = r16(x * epi.scale[nn] + epi.shift[nn]);
                    if (act == 3) x = fmaxf(x, 0.f);
                }
                h[e] = (half_t)x;
            }
            return h;
        },
        [&](int r, int, int, half8_t h) {
            const int row = m0 + r;
            if (row >= M || col >= N) return;
            const size_t o = (size_t)row * ldc + col;
            for (int e = 0; e < 8 && col + e < N; ++e) {
                float x = (float)h[e];
                if (residual) x = (float)residual[o + e] + x;
                epi.C[o + e] = (half_t)x;
            }
        });
}

using CfgBig = pgemm::Cfg<256, 256, 2, 4>;
using CfgWide = pgemm::Cfg<256, 128, 4, 2>;
using CfgNarrow = pgemm::Cfg<256, 64, 4, 2>;          // 64-channel convolutions of the ResNet tower
using CfgThin = pgemm::Cfg<256, 32, 4, 1>;            // its 32-channel stem (4 waves, two workgroups per CU)
using CfgSmall = pgemm::CfgSmall;

// Tile-order switches (PCLIP_GEMM_BAND, PCLIP_GEMM_REV): read from the environment ONCE; only under PCLIP_GEMM_CFG_LIVE (the A/B tools flip
// them between calls of one process) are they re-read per launch — no getenv on the product's launch path.
struct TileOrder { int band, rev; };
static TileOrder read_tile_order() {
    const char* b = getenv("PCLIP_GEMM_BAND");
    const char* r = getenv("PCLIP_GEMM_REV");
    return TileOrder{b ? atoi(b) : 0, r ? atoi(r) : 2};
}
static const TileOrder& tile_order() {
    static const bool live = getenv("PCLIP_GEMM_CFG_LIVE") != nullptr;
    static TileOrder order = read_tile_order();
    if (live) order = read_tile_order();
    return order;
}

template <class C, bool HAS_BIAS, int ACT>
static int launch_fast2(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const LinearEpi& epi,
                        int slots, hipStream_t s) {
    static DevOnce attr;
    constexpr bool LNF = ACT == 7 || ACT == 8;
    constexpr int LDS = C::LDS_BYTES + ((ACT == 2 || ACT == 3 || ACT == 5 || LNF) ? 2 * 2 * C::BN * 4 : 2 * C::BN * 2) + 256 +
                        (LNF ? 2 * C::BM * 8 : 0) + (ACT == 10 ? 128 + 8192 : 0);   // K-tile ring + double-buffered bias / affine strips + prefetch scrap + (mean, rstd) rows | panel flags
    if (!attr.done()) {
        if (hipFuncSetAttribute((const void*)linear_fast_kernel<C, HAS_BIAS, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize,
                                LDS) != hipSuccess) {
            pclip_set_error("pclip_gemm_f16: cannot raise the dynamic LDS limit to %d", LDS);
            return PCLIP_E_LAUNCH;
        }
        attr.set();
    }
    const int tiles_m = ceil_div(M, C::BM), tiles_n = N / C::BN, ntiles = tiles_m * tiles_n;
    const int grid = ntiles < slots ? ntiles : slots;
    const TileOrder& order = tile_order();
    const int band = order.band;
    // Tile order against the Infinity Cache (256 MiB, memory-side): a LayerNorm / attention pass writes its 310 MB output in ascending row order, so what is still
    // cached when the consuming GEMM starts are its LAST rows — walking the tiles in descending order reads those first (and leaves the GEMM's own first-written, high
    // rows to be evicted, its low rows fresh for the ascending pass behind it).  Same bits (tile order only); bench +0.4 % (profiles/r03_bench_rev.txt).  Default 2.
    const int rev_mode = order.rev;
    // 1: every launch descending; 2: only the launches that read a LayerNorm / attention output (K <= 1024: in_proj, c_fc, out_proj), c_proj ascending behind the descending c_fc
    // act 10 writes the LayerNorm output its consumer reads next (descending): ascending here, whatever it reads itself (PCLIP_LNP_REV=1: as act 6)
    static const bool lnp_rev = getenv("PCLIP_LNP_REV") && getenv("PCLIP_LNP_REV")[0] == '1';
    const bool rev = (rev_mode == 1 || (rev_mode == 2 && K <= 1024)) && (ACT != 10 || lnp_rev);
    linear_fast_kernel<C, HAS_BIAS, ACT><<<grid, C::NTHREADS, LDS, s>>>(
        (const half_t*)A, lda, (const half_t*)B, ldb, M, N, K, epi.bias, epi.scale, epi.shift, epi.C, epi.ldc, tiles_n, ntiles, epi.residual,
        epi.rowstats, epi.partials, rev ? -1 : (tiles_n >= 8 ? band : 0), LnPanel{epi.ln_gamma, epi.ln_beta, epi.ln_y, epi.ln_cnt, epi.ln_eps});
    return pclip_check_launch("gemm_f16");
}

template <class C>
static int launch_fast(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const LinearEpi& epi,
                       int slots, hipStream_t s) {
    if (epi.act == 2) return launch_fast2<C, false, 2>(A, lda, B, ldb, M, N, K, epi, slots, s);
    if (epi.act == 3) return launch_fast2<C, false, 3>(A, lda, B, ldb, M, N, K, epi, slots, s);
    if (epi.act == 5) return launch_fast2<C, false, 5>(A, lda, B, ldb, M, N, K, epi, slots, s);
    if (epi.act == 6) return launch_fast2<C, true, 6>(A, lda, B, ldb, M, N, K, epi, slots, s);
    if (epi.act == 9) return launch_fast2<C, true, 9>(A, lda, B, ldb, M, N, K, epi, slots, s);
    if (epi.act == 10) {
        // instantiated for the two tiles the N <= 1024 residual GEMMs of a tower run on (256 x 256 rounds + 128 x 128 tail); gemm_dispatch sends every other choice the two-launch way
        if constexpr (std::is_same_v<C, CfgBig> || std::is_same_v<C, CfgSmall>) return launch_fast2<C, true, 10>(A, lda, B, ldb, M, N, K, epi, slots, s);
        else { pclip_set_error("pclip_gemm_res_ln_f16: no fused form for this tile"); return PCLIP_E_INVALID; }
    }
    if (epi.act == 7) return launch_fast2<C, false, 7>(A, lda, B, ldb, M, N, K, epi, slots, s);
    if (epi.act == 8) return launch_fast2<C, false, 8>(A, lda, B, ldb, M, N, K, epi, slots, s);
    if (epi.bias) {
        if (epi.act == 1) return launch_fast2<C, true, 1>(A, lda, B, ldb, M, N, K, epi, slots, s);
        return launch_fast2<C, true, 0>(A, lda, B, ldb, M, N, K, epi, slots, s);
    }
    if (epi.act == 1) return launch_fast2<C, false, 1>(A, lda, B, ldb, M, N, K, epi, slots, s);
    return launch_fast2<C, false, 0>(A, lda, B, ldb, M, N, K, epi, slots, s);
}

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
#ifndef PCLIP_LN_PF
#define PCLIP_LN_PF 1
#endif
#ifndef PCLIP_LN_LDS
#define PCLIP_LN_LDS 1
#endif
#ifndef PCLIP_LN_BPC
#define PCLIP_LN_BPC 32
#endif
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

// (mean, rstd) of every row: what is left of a LayerNorm whose affine part has been folded into the consuming linear (ln_fold).
// One wave per row: the values do not depend on how many rows the call carries, and they are the values stats_finalize_kernel
// derives from the partials a residual GEMM wrote (same association order, stats_chunk).
template <int NCH>
__global__ __launch_bounds__(256) void row_stats_kernel(const half_t* __restrict__ x, int ld_x, float eps, float* __restrict__ stats, int R,
                                                        int D) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < R; row += gridDim.x * 4) {
        const half_t* xr = x + (size_t)row * ld_x;
        half8_t hv[NCH];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            const int d = c * 512 + lane * 8;
            if (d < D) hv[c] = ld_half8(xr + d);
        }
        const float2_t ms = row_mean_rstd<NCH>(hv, D, lane, eps);
        if (lane == 0) *reinterpret_cast<float2_t*>(stats + (size_t)row * 2) = ms;
    }
}

// partials [R][D / 64][2] (sum, sum of squares per 64 columns, written by the act-9 epilogues) -> stats [R][2] = (mean, rstd):
// per 256-column block the tree (g0 + g1) + (g2 + g3), blocks left to right.  One thread per row.
__global__ __launch_bounds__(256) void stats_finalize_kernel(const float* __restrict__ partials, int R, int D, float eps,
                                                             float* __restrict__ stats) {
    const int row = blockIdx.x * 256 + threadIdx.x;
    if (row >= R) return;
    const int ns = D >> 6;
    const float2_t* p = reinterpret_cast<const float2_t*>(partials) + (size_t)row * ns;
    float S = 0.f, Q = 0.f;
    for (int b = 0; b < ns; b += 4) {
        float2_t g[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) g[k] = b + k < ns ? p[b + k] : float2_t{0.f, 0.f};
        S += (g[0][0] + g[1][0]) + (g[2][0] + g[3][0]);
        Q += (g[0][1] + g[1][1]) + (g[2][1] + g[3][1]);
    }
    *reinterpret_cast<float2_t*>(stats + (size_t)row * 2) = stats_from_sums(S, Q, D, eps);
}

// Wf[n, :] = r16(gamma . W[n, :]),  colsum[n] = sum_k Wf[n, k] (of the ROUNDED values: it cancels the mean against exactly the
// weights the GEMM multiplies),  bfold[n] = sum_k beta[k] W[n, k] + bias[n];  fp32 sums, one wave per output row, fixed order.
__global__ __launch_bounds__(256) void ln_fold_weights_kernel(const half_t* __restrict__ W, int ldw, int N, int K, const float* __restrict__ gamma,
                                                              const float* __restrict__ beta, const half_t* __restrict__ bias,
                                                              half_t* __restrict__ Wf, float* __restrict__ colsum, float* __restrict__ bfold) {
    const int lane = threadIdx.x & 63, n = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (n >= N) return;
    float cs = 0.f, bs = 0.f;
    for (int k = lane * 8; k < K; k += 512) {
        const half8_t w = ld_half8(W + (size_t)n * ldw + k);
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            o[j] = (half_t)(gamma[k + j] * (float)w[j]);
            cs += (float)o[j];
            bs = fmaf(beta[k + j], (float)w[j], bs);
        }
        st_half8(Wf + (size_t)n * K + k, o);
    }
    cs = wave_sum(cs);
    bs = wave_sum(bs);
    if (lane == 0) {
        colsum[n] = cs;
        bfold[n] = bs + (bias ? (float)bias[n] : 0.f);
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

// ---- attention: one workgroup per (image, head), whole K/V of the head resident in LDS ----------------
// The CLIP sequences (50 .. 257 tokens) fit one workgroup.  Both contractions are computed TRANSPOSED so
// that a lane owns ONE query row throughout:
//   S^T = K Q^T      (A = K rows from LDS, B = Q rows in registers)  -> lane (q = lane&31) holds 16 keys
//   O^T = V^T P^T    (A = V^T rows from LDS, B = P^T = the S^T registers, already in B-operand order)
// so the softmax max / sum / rescale are in-lane scalars (one cross-half shuffle), no LDS round trip for P,
// and the k-order of the second contraction is whatever the first one produced (a contraction does not
// care, as long as A and B agree).  Keys are walked in 32-wide tiles with an online softmax, which keeps
// the register footprint at ~100 VGPRs (2 workgroups per CU) for any L <= 288.
constexpr int ATT_DH = 64;
constexpr int ATT_MAX_L = 288;
// Softmax variants of attn_query_tile (bit mask VAR; same-process A/B of the seven combinations, tools/ab_multi.py attn,
// profiles/r03_ab_attention_var.txt — ViT-B/16, B = 1024: 336 us -> 306 us with all four, each contributing):
//   1  deferred maximum: a row's running maximum only moves when the row outgrew it by more than 2^kAttDefer; in between the
//      probabilities are taken against the OLD maximum (they reach 2^kAttDefer instead of 1: exact in fp32, and the fp16 rounding of
//      P is relative) and the rescale of the 32 output accumulators (+ its v_exp) is skipped.  On N(0,1) data the maximum of a later
//      key tile practically never exceeds the first tiles' by a factor 4, so the rescale runs once per query tile instead of 4 times.
//   2  the row sum as two interleaved partial sums (v_pk_add_f32: 16 instead of 32 dependent adds per pair of key tiles)
//   4  scale-and-shift of two scores per instruction (v_pk_fma_f32)
//   8  s_setprio(1) around the MFMA clusters (four waves per SIMD at different phases: the guide's T5 regime)
// The eight-wave kernel (long sequences: ViT-B/16, ViT-L/14) takes all four; the four-wave kernel (ViT-B/32, the text tower) only the
// deferred maximum — the packed forms and the priority flips cost the causal L = 77 kernel 4 %.  Not bit-identical to round 2's
// kernel: outputs differ by one fp16 ulp on ~1e-4 of the elements, error against fp32 attention unchanged (tests/test_gpu_encoder.py).
constexpr float kAttDefer = 2.f;
#ifndef PCLIP_ATT_VAR_LONG
#define PCLIP_ATT_VAR_LONG 15
#endif
#ifndef PCLIP_ATT_VAR_SHORT
#define PCLIP_ATT_VAR_SHORT 1
#endif

// ds_read_b64_tr_b16: 64 bits per lane, 16-bit elements transposed inside each 16-lane group (see attention_kernel)
typedef __fp16 fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ half4_t tr_read4(const char* lds_addr) {
    const fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4_t*)(lds_addr));
    return __builtin_bit_cast(half4_t, v);
}

// Transpose-read addressing (probed on gfx950, tools/probe/tr_probe.hip): inside a 16-lane group, lane i supplies the address
// of 4 consecutive halfs and lane l receives element (l & 3) of the words addressed by lanes 4*jj + ((l & 15) >> 2), jj = 0..3.
// With lane i pointing at V[key0 + (i >> 2)][d0 + 4*(i & 3) ..], lane l therefore receives V[key0 + jj][d0 + (l & 15)]: four
// consecutive keys of ITS output dimension — the A-operand fragment of O^T = V^T P^T, without a transposed copy of V.
// voff[j]: byte offset of this lane's word for the output halves j = 0, 1.
__device__ __forceinline__ void attn_voff(int lane, int (&voff)[2]) {
    const int hi = lane >> 5;
    const int i16 = lane & 15, vrow = hi * 4 + (i16 >> 2), vd = ((lane >> 4) & 1) * 16 + 4 * (i16 & 3), vswz = ((vrow >> 1) & 1) << 2;
#pragma unroll
    for (int j = 0; j < 2; ++j) voff[j] = vrow * (ATT_DH * 2) + ((((j * 4 + (vd >> 3)) ^ vswz)) << 4) + (vd & 7) * 2;
}

// One 32-query tile (query row q = qb*32 + (lane & 31), fragments qf) against every key tile of the sequence resident in LDS:
// Ks [>= L rows][64] with the 16-byte chunks XOR-swizzled by swz_key(row) — rows >= L may hold ANYTHING, their scores are
// overwritten by the mask; Vs [NT*32 rows][64] with chunk ^ 4*((row >> 1) & 1) — rows >= L must be finite (their probabilities
// are exact zeros).  Returns O^T (unnormalised) and the row sum.  Shared by attention_kernel and attention_pipe_kernel: one
// instruction order, bit-identical results.
// max / sum of a value with its partner lane (lane ^ 32) through v_permlane32_swap (a VALU instruction) instead of the LDS round
// trip of a ds_bpermute: swap(v, v) leaves {own, partner} in the lower half-wave and {partner, own} in the upper one, and both
// operations are commutative, so every lane gets exactly the value of `x op shfl_xor(x, 32)`.
// (The two results are copied into scalars before the bit casts: __builtin_bit_cast(float, r[1]) applied to the builtin's result
// directly reads element 0 under this hipcc — the max / add of the pair silently became max(r0, r0).)
__device__ __forceinline__ void half_wave_pair(float v, float& r0, float& r1) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned a = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    const unsigned x = r[0], y = r[1];
    r0 = __builtin_bit_cast(float, x);
    r1 = __builtin_bit_cast(float, y);
#else
    r0 = r1 = v;
#endif
}
__device__ __forceinline__ float half_wave_max(float v) { float a, b; half_wave_pair(v, a, b); return fmaxf(a, b); }
__device__ __forceinline__ float half_wave_sum(float v) { float a, b; half_wave_pair(v, a, b); return a + b; }

// DEEP (the persistent kernel: two waves per SIMD, registers to spare): the K fragments of the NEXT pair of key tiles and the
// V^T fragments of THIS pair are requested right after the pair's score MFMAs, so their LDS latency passes under the softmax
// arithmetic instead of in front of every MFMA (+64 VGPRs).  Same operations in the same order per accumulator: same bits.
// VBAR: the caller has only made K visible so far (V is still landing); the first pair of key tiles waits for V — own pieces, then a workgroup barrier —
// between its softmax and its second contraction, so V's arrival passes under the first scores.  Every wave of the workgroup must pass that barrier once.
#ifndef PCLIP_ATT_QF4
#define PCLIP_ATT_QF4 1           // query-first form of the four-wave kernel for short non-causal sequences (0: A/B)
#endif
#ifndef PCLIP_ATT_EDGE
#define PCLIP_ATT_EDGE 1          // a lone last key tile with at most 24 valid keys skips its fully masked groups (tile_edge below; 0: A/B)
#endif
template <bool DEEP = false, int VAR = 0, bool VBAR = false>
__device__ __forceinline__ void attn_query_tile(const half_t* Ks, const half_t* Vs, const half8_t (&qf)[4], int q, int qb, int L, int causal,
                                                int NT, int hi, int ql, const int (&voff)[2], float16_t (&o)[2], float& lrun_out) {
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int e = 0; e < 16; ++e) o[j][e] = 0.f;
    // scores are kept in the log2 domain: s2 = (q.k) * (1/sqrt(64)) * log2(e), p = exp2(s2 - max2) — one
    // v_exp_f32 per probability; masks are applied only on the tiles that need them (last key tile, causal
    // diagonal); the running output is rescaled only when some row's maximum actually moved.
    constexpr float kScale = 0.125f * 1.4426950408889634f;
    float mrun = -__builtin_inff(), lrun = 0.f;
    const int tend = causal ? (qb + 1 < NT ? qb + 1 : NT) : NT;      // causal: keys beyond the block's last query are all masked
    // Key tiles are taken two at a time: the two score accumulators are independent MFMA chains (a single
    // 32x32x16 chain is issue-limited by its own accumulator dependency), and one max / rescale serves 64 keys.
    auto k_frag = [&](int t, int sidx) {
        const int kr = t * 32 + ql;                                // key row this lane feeds as the A operand
        return *reinterpret_cast<const half8_t*>(Ks + kr * ATT_DH + (((sidx * 2 + hi) ^ pgemm::swz_key(kr)) << 3));
    };
    auto v_frag = [&](int t, int sidx, int j) {
        // V^T fragment: row d = j*32 + ql, keys t*32 + 16s + 4hi + {0..3} and the same + 8: two transpose-reads
        const char* vb = reinterpret_cast<const char*>(Vs) + (t * 32 + sidx * 16) * (ATT_DH * 2) + voff[j];
        const half4_t v0 = tr_read4(vb);
        const half4_t v1 = tr_read4(vb + 8 * (ATT_DH * 2));
        return half8_t{v0[0], v0[1], v0[2], v0[3], v1[0], v1[1], v1[2], v1[3]};
    };
    half8_t kpre[2][4];                                            // DEEP: K fragments of the pair about to be multiplied
    auto k_prefetch = [&](int t0) {                                // always two tiles (the second clamped: one shape of code, no select between register sets)
#pragma unroll
        for (int u = 0; u < 2; ++u) {
            const int tt = t0 + u < NT ? t0 + u : NT - 1;
#pragma unroll
            for (int sidx = 0; sidx < 4; ++sidx) kpre[u][sidx] = k_frag(tt, sidx);
        }
    };
    auto tiles = [&](auto NTILE_C, int t0) {
        constexpr int NTILE = decltype(NTILE_C)::value;
        float16_t st[NTILE];
#pragma unroll
        for (int u = 0; u < NTILE; ++u)
#pragma unroll
            for (int e = 0; e < 16; ++e) st[u][e] = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
        if (VAR & 8) __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx)
#pragma unroll
            for (int u = 0; u < NTILE; ++u) {
                const half8_t kf = DEEP ? kpre[u][sidx] : k_frag(t0 + u, sidx);
                st[u] = __builtin_amdgcn_mfma_f32_32x32x16_f16(kf, qf[sidx], st[u], 0, 0, 0);
            }
#if defined(__HIP_DEVICE_COMPILE__)
        if (VAR & 8) __builtin_amdgcn_s_setprio(0);
#endif
        half8_t vpre[NTILE][2][2];
        if (DEEP) {
#pragma unroll
            for (int u = 0; u < NTILE; ++u)
#pragma unroll
                for (int sidx = 0; sidx < 2; ++sidx)
#pragma unroll
                    for (int j = 0; j < 2; ++j) vpre[u][sidx][j] = v_frag(t0 + u, sidx, j);
            const int tn = t0 + NTILE;
            if (tn < tend) k_prefetch(tn);
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);                     // keep the requests ahead of the softmax arithmetic
#endif
        }
        float tmax = -__builtin_inff();
#pragma unroll
        for (int u = 0; u < NTILE; ++u) {
            const int t = t0 + u;
            if ((t * 32 + 32 > L) || (causal && t == qb)) {        // wave-uniform: only edge tiles pay for the mask
                // key k = t*32 + c_e + 4*hi is valid iff k < L and (causal) k <= q, i.e. k < min(L, q + 1): ONE per-lane limit
                // against the compile-time c_e — a compare + select per element (the two-condition form was 12 instructions each)
                const int kend = causal ? (q + 1 < L ? q + 1 : L) : L;
                const int lim = kend - t * 32 - 4 * hi;
#pragma unroll
                for (int e = 0; e < 16; ++e)
                    if (!((e & 3) + 8 * (e >> 2) < lim)) st[u][e] = -__builtin_inff();
            }
        }
        {   // four independent maximum chains instead of one 32-deep dependent one (max is exact: same value)
            float m4[4] = {tmax, tmax, tmax, tmax};
#pragma unroll
            for (int u = 0; u < NTILE; ++u)
#pragma unroll
                for (int e = 0; e < 16; ++e) m4[e & 3] = fmaxf(m4[e & 3], st[u][e]);
            tmax = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
        }
        tmax = half_wave_max(tmax) * kScale;                       // kScale > 0: max commutes with the scaling
        // VAR & 1: deferred maximum (see above), decided PER ROW — a row's bits must not depend on the rows that share its wave (the
        // one-query form of the last block == the full attention); -inf + kAttDefer = -inf, so the first tile always sets the maximum.
        // The rescale below is skipped when no row of the wave moved (rows that did not move multiply by exp2(0) = 1 exactly).
        const bool moved = (VAR & 1) ? tmax > mrun + kAttDefer : fmaxf(mrun, tmax) != mrun;
        const float mnew = ((VAR & 1) && !moved) ? mrun : fmaxf(mrun, tmax);    // finite from the first tile on: key 0 is never masked
        const bool grow = __any(moved);
        float psum = 0.f;
        if (VAR & 6) {
            float2_t ps2 = {0.f, 0.f};
            const float2_t ks2 = {kScale, kScale}, nm2 = {-mnew, -mnew};
#pragma unroll
            for (int u = 0; u < NTILE; ++u)
#pragma unroll
                for (int e = 0; e < 16; e += 2) {
                    float2_t v = {st[u][e], st[u][e + 1]};
                    if (VAR & 4) {
                        v = v * ks2 + nm2;                         // v_pk_fma_f32
                        v = float2_t{__builtin_amdgcn_exp2f(v[0]), __builtin_amdgcn_exp2f(v[1])};
                    } else
                        v = float2_t{__builtin_amdgcn_exp2f(fmaf(v[0], kScale, -mnew)), __builtin_amdgcn_exp2f(fmaf(v[1], kScale, -mnew))};
                    st[u][e] = v[0];
                    st[u][e + 1] = v[1];
                    if (VAR & 2) ps2 += v;                         // v_pk_add_f32: two partial sums
                    else { psum += v[0]; psum += v[1]; }
                }
            psum += ps2[0] + ps2[1];
        } else {
#pragma unroll
            for (int u = 0; u < NTILE; ++u)
#pragma unroll
                for (int e = 0; e < 16; ++e) { st[u][e] = __builtin_amdgcn_exp2f(fmaf(st[u][e], kScale, -mnew)); psum += st[u][e]; }
        }
        psum = half_wave_sum(psum);
        if (grow) {
            const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
            lrun *= alpha;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[j][e] *= alpha;
        }
        lrun += psum;
        mrun = mnew;
        if (VBAR && t0 == 0) { pgemm::wait_vm<0>(); pgemm::lds_barrier(); }
#pragma unroll
        for (int u = 0; u < NTILE; ++u)
#pragma unroll
            for (int sidx = 0; sidx < 2; ++sidx) {
                half8_t pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[e] = (half_t)st[u][sidx * 8 + e];
#if defined(__HIP_DEVICE_COMPILE__)
                if (VAR & 8) __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    const half8_t vf = DEEP ? vpre[u][sidx][j] : v_frag(t0 + u, sidx, j);
                    o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(vf, pf, o[j], 0, 0, 0);
                }
#if defined(__HIP_DEVICE_COMPILE__)
                if (VAR & 8) __builtin_amdgcn_s_setprio(0);
#endif
            }
    };
    // A LAST key tile on its own whose keys beyond L are masked (non-causal; ViT-B/16: keys 192 .. 196 of tile 6, ViT-L/14: key 256 alone in tile 8): the groups of
    // eight keys (elements 4g .. 4g + 3 of both half-waves) without a single valid key are not computed at all — no mask, maximum, exponential, sum or conversion
    // for them, and no second contraction over keys 16 .. 31 when those are all masked.  Their probabilities are exact zeros in `tiles` (exp2(-inf)), which add
    // nothing to the sum and to O: same bits (the valid elements keep their order in the maximum chains and the partial sums).
    auto tile_edge = [&](int t0) {
        const int ng = (L - t0 * 32 + 7) >> 3;                          // groups with a valid key: 1 .. 3 (wave-uniform)
        float16_t st;
#pragma unroll
        for (int e = 0; e < 16; ++e) st[e] = 0.f;
#if defined(__HIP_DEVICE_COMPILE__)
        if (VAR & 8) __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
        for (int sidx = 0; sidx < 4; ++sidx) st = __builtin_amdgcn_mfma_f32_32x32x16_f16(k_frag(t0, sidx), qf[sidx], st, 0, 0, 0);
#if defined(__HIP_DEVICE_COMPILE__)
        if (VAR & 8) __builtin_amdgcn_s_setprio(0);
#endif
        const int lim = L - t0 * 32 - 4 * hi;
        float m4[4] = {-__builtin_inff(), -__builtin_inff(), -__builtin_inff(), -__builtin_inff()};
#pragma unroll
        for (int g = 0; g < 3; ++g)
            if (g < ng) {
#pragma unroll
                for (int r = 0; r < 4; ++r) {
                    if (!(r + 8 * g < lim)) st[4 * g + r] = -__builtin_inff();
                    m4[r] = fmaxf(m4[r], st[4 * g + r]);
                }
            }
        float tmax = fmaxf(fmaxf(m4[0], m4[1]), fmaxf(m4[2], m4[3]));
        tmax = half_wave_max(tmax) * kScale;
        const bool moved = (VAR & 1) ? tmax > mrun + kAttDefer : fmaxf(mrun, tmax) != mrun;
        const float mnew = ((VAR & 1) && !moved) ? mrun : fmaxf(mrun, tmax);
        const bool grow = __any(moved);
        float psum = 0.f;
        float2_t ps2 = {0.f, 0.f};
        const float2_t ks2 = {kScale, kScale}, nm2 = {-mnew, -mnew};
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            if (g < 3 && g < ng) {
#pragma unroll
                for (int e = 4 * g; e < 4 * g + 4; e += 2) {
                    float2_t v = {st[e], st[e + 1]};
                    if (VAR & 6) {
                        if (VAR & 4) {
                            v = v * ks2 + nm2;
                            v = float2_t{__builtin_amdgcn_exp2f(v[0]), __builtin_amdgcn_exp2f(v[1])};
                        } else
                            v = float2_t{__builtin_amdgcn_exp2f(fmaf(v[0], kScale, -mnew)), __builtin_amdgcn_exp2f(fmaf(v[1], kScale, -mnew))};
                        if (VAR & 2) ps2 += v;
                        else { psum += v[0]; psum += v[1]; }
                    } else {
                        v[0] = __builtin_amdgcn_exp2f(fmaf(v[0], kScale, -mnew)); psum += v[0];
                        v[1] = __builtin_amdgcn_exp2f(fmaf(v[1], kScale, -mnew)); psum += v[1];
                    }
                    st[e] = v[0];
                    st[e + 1] = v[1];
                }
            } else {
#pragma unroll
                for (int r = 0; r < 4; ++r) st[4 * g + r] = 0.f;
            }
        }
        if (VAR & 6) psum += ps2[0] + ps2[1];
        psum = half_wave_sum(psum);
        if (grow) {
            const float alpha = __builtin_amdgcn_exp2f(mrun - mnew);
            lrun *= alpha;
#pragma unroll
            for (int j = 0; j < 2; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) o[j][e] *= alpha;
        }
        lrun += psum;
        mrun = mnew;
        if (VBAR && t0 == 0) { pgemm::wait_vm<0>(); pgemm::lds_barrier(); }
#pragma unroll
        for (int sidx = 0; sidx < 2; ++sidx)
            if (sidx == 0 || ng > 2) {
                half8_t pf;
#pragma unroll
                for (int e = 0; e < 8; ++e) pf[e] = (half_t)st[sidx * 8 + e];
#if defined(__HIP_DEVICE_COMPILE__)
                if (VAR & 8) __builtin_amdgcn_s_setprio(1);
#endif
#pragma unroll
                for (int j = 0; j < 2; ++j) o[j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(v_frag(t0, sidx, j), pf, o[j], 0, 0, 0);
#if defined(__HIP_DEVICE_COMPILE__)
                if (VAR & 8) __builtin_amdgcn_s_setprio(0);
#endif
            }
    };
    int t = 0;
    if (DEEP) k_prefetch(0);
    for (; t + 1 < tend; t += 2) tiles(std::integral_constant<int, 2>{}, t);
    if (t < tend) {
        if (PCLIP_ATT_EDGE && VBAR && !DEEP && !causal && L - t * 32 <= 24) tile_edge(t);      // (the query-first kernels only: measured neutral to - 2 % in the looping eight-wave kernel at L = 257)
        else tiles(std::integral_constant<int, 1>{}, t);
    }
    lrun_out = lrun;
}

// O^T tile -> the query's 128-byte output row segment: lane (ql, hi) holds d = j*32 + 8g + 4hi + (e & 3), i.e. each output row is
// split across the two half-waves in 8-byte pieces.  v_permlane32_swap pairs the pieces of column groups (2k, 2k+1) so that every
// lane owns 16 contiguous bytes: four dwordx4 stores per lane instead of sixteen dwordx2 (the store tail is issue-bound; guide T21).
// `orow` = this lane's output row (+ head offset); every lane executes the swaps, `valid` only predicates the stores.
__device__ __forceinline__ void attn_store_tile(half_t* orow, const float16_t (&o)[2], float lrun, int hi, bool valid) {
    const float inv = 1.f / lrun;
#pragma unroll
    for (int j = 0; j < 2; ++j)
#pragma unroll
        for (int k = 0; k < 2; ++k) {
            unsigned a[2], bq[2];
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const half2_t ha = {(half_t)(o[j][8 * k + 2 * w] * inv), (half_t)(o[j][8 * k + 2 * w + 1] * inv)};          // group g = 2k
                const half2_t hb = {(half_t)(o[j][8 * k + 4 + 2 * w] * inv), (half_t)(o[j][8 * k + 4 + 2 * w + 1] * inv)};  // group g = 2k + 1
                a[w] = __builtin_bit_cast(unsigned, ha);
                bq[w] = __builtin_bit_cast(unsigned, hb);
            }
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int w = 0; w < 2; ++w) {
                const auto r = __builtin_amdgcn_permlane32_swap(a[w], bq[w], false, false);   // upper half of a <-> lower half of b
                a[w] = r[0];
                bq[w] = r[1];
            }
#endif
            // lanes 0-31: [own g=2k | partner's g=2k] = d 16k .. 16k+7; lanes 32-63: [partner's g=2k+1 | own g=2k+1] = d 16k+8 .. 16k+15
            if (valid) *reinterpret_cast<uint4_t*>(orow + j * 32 + 16 * k + 8 * hi) = uint4_t{a[0], a[1], bq[0], bq[1]};
        }
}

// General operand form: queries q [B][Lq rows, row stride ldq] (the FIRST Lq tokens of each sequence), keys / values in
// kv [B*L rows, row stride ldkv] at column offsets k_off / v_off; the fused-QKV case is q = kv = qkv, ldq = ldkv = 3W,
// k_off = W, v_off = 2W, Lq = L.  Lq < L serves the last vision block, whose output is only read at the class token.
// QF (round 4): every wave has AT MOST ONE query tile (the host guarantees ceil(Lq / 32) <= NW, hence LP <= 256: at most four pieces per wave and operand) and
// requests its query fragments BEFORE the K / V stages, so their round trip passes under the staging instead of opening the compute phase behind the barrier
// (ViT-B/16 354 -> 320 us stand-alone); and the workgroup barrier only waits for K — V is awaited (own pieces + a second LDS-only barrier) between the first
// pair of key tiles' softmax and its second contraction (-> 303 us).  Same bits (profiles/r04_ab_attention_qfirst.txt).  Not for waves that loop over several tiles (ViT-L/14: + 6 %) nor the short causal text sequences (+ 7 %).
template <int NW, int VAR, bool QF = false>   // waves per workgroup (__launch_bounds__'s second argument = waves per SIMD: two workgroups per CU); softmax variant
__global__ __launch_bounds__(NW * 64, NW / 2) void attention_kernel(const half_t* __restrict__ qp, int ldq, long q_batch,
                                                           const half_t* __restrict__ kvp, int ldkv, int k_off, int v_off,
                                                           half_t* __restrict__ out, int L, int Lq, int H, int causal, int NT,
                                                           int LV) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int LP = NT * 32;
    half_t* Ks = reinterpret_cast<half_t*>(smem);             // [LP][64], 16-byte chunks XOR-swizzled by swz_key(row)
    half_t* Vs = Ks + LP * ATT_DH;                            // [LP][64]: V row-major, chunk-swizzled (see the staging loop)
    const int b = blockIdx.x / H, h = blockIdx.x % H;
    const int W = H * ATT_DH;
    const half_t* kbase = kvp + (size_t)b * L * ldkv + h * ATT_DH + k_off;
    const half_t* vbase = kvp + (size_t)b * L * ldkv + h * ATT_DH + v_off;
    const half_t* qbase = qp + (size_t)b * q_batch + h * ATT_DH;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    half8_t qf0[4];
    if (QF) {
        const int q0 = wave * 32 + (lane & 31), qc0 = q0 < Lq ? q0 : Lq - 1;
#pragma unroll
        for (int s_ = 0; s_ < 4; ++s_) qf0[s_] = ld_half8(qbase + (size_t)qc0 * ldq + s_ * 16 + (lane >> 5) * 8);
    }

    // K: global_load_lds, 8 rows x 128 B per wave instruction, swizzle on the source chunk (as the GEMM tiles)
    for (int r0 = wave * 8; r0 < LP; r0 += NW * 8) {
        const int r = r0 + (lane >> 3);
        const int c = (lane & 7) ^ pgemm::swz_key(r);
        const int rc = r < L ? r : L - 1;                     // rows >= L are masked in the scores
        __builtin_amdgcn_global_load_lds((pgemm::gbl_ptr_t)(kbase + (size_t)rc * ldkv + c * 8),
                                         (pgemm::lds_ptr_t)(Ks + r0 * ATT_DH), 16, 0, 0);
    }
    // V: row-major like K, by LDS-DMA (no registers, no transposing ds_writes); the 16-byte chunks of key row r are XORed with
    // 4 * ((r >> 1) & 1) so that the four keys of a transpose-read (ds_read_b64_tr_b16, below) land on 4 x 16 distinct banks.
    // Rows >= L re-read row L-1: their probabilities are exact zeros (masked scores), so they contribute 0 * finite = 0.
    for (int r0 = wave * 8; r0 < LP; r0 += NW * 8) {
        const int r = r0 + (lane >> 3);
        const int c = (lane & 7) ^ (((r >> 1) & 1) << 2);
        const int rc = r < L ? r : L - 1;
        __builtin_amdgcn_global_load_lds((pgemm::gbl_ptr_t)(vbase + (size_t)rc * ldkv + c * 8),
                                         (pgemm::lds_ptr_t)(Vs + r0 * ATT_DH), 16, 0, 0);
    }
#ifndef PCLIP_ATT_VBAR
#define PCLIP_ATT_VBAR 1          // QF kernels: barrier on K alone, V awaited between the first scores and the first second contraction (334.7 -> 303.5 us stand-alone, same bits)
#endif
    constexpr bool VB = PCLIP_ATT_VBAR && QF;
    if (VB) {
        // the wave's V pieces (the youngest operations: rows wave * 8 + NW * 8 k < LP, one to four of them) stay in flight: K and the query fragments have landed
        // once no more than those are outstanding
        const int nv = (LP - wave * 8 + NW * 8 - 1) / (NW * 8);
        // (EXACTLY nv: with "<= 2 -> vmcnt(2)" a wave of the four-wave form that stages ONE piece per operand (L <= 32) went through with its K piece still in flight —
        // caught by a small-tower image -> logits fixture failing in two of four runs)
        if (nv <= 1) pgemm::wait_vm<1>(); else if (nv == 2) pgemm::wait_vm<2>(); else if (nv == 3) pgemm::wait_vm<3>(); else pgemm::wait_vm<4>();
        pgemm::lds_barrier();
    } else
        __syncthreads();

    const int hi = lane >> 5, ql = lane & 31;
    int voff[2];
    attn_voff(lane, voff);
    const int NTq = (Lq + 31) >> 5;
    auto process = [&](int qb, const half8_t (&qf)[4]) {
        const int q = qb * 32 + ql;
        float16_t o[2];
        float lrun;
        attn_query_tile<false, VAR, VB>(Ks, Vs, qf, q, qb, L, causal, NT, hi, ql, voff, o, lrun);
        attn_store_tile(out + ((size_t)b * Lq + q) * W + h * ATT_DH, o, lrun, hi, q < Lq);
    };
    if (QF) {
        if (wave < NTq) process(wave, qf0);
        else if (VB) { pgemm::wait_vm<0>(); pgemm::lds_barrier(); }      // the barrier inside the first pair of key tiles
        return;
    }
    for (int qb = wave; qb < NTq; qb += NW) {
        const int q = qb * 32 + ql;                     // this lane's query row
        const int qc = q < Lq ? q : Lq - 1;
        half8_t qf[4];
#pragma unroll
        for (int s = 0; s < 4; ++s) qf[s] = ld_half8(qbase + (size_t)qc * ldq + s * 16 + hi * 8);
        process(qb, qf);
    }
}

// ---- persistent, double-buffered form of the same attention (whole batches) --------------------------------------------------
// attention_kernel is a chain of dependent phases per (image, head): K/V by LDS-DMA -> barrier -> query loads -> compute -> stores,
// and two co-resident workgroups fall into lockstep, so the memory pipe idles while the SIMDs work and vice versa (ablation,
// the parts ADD).  Here a workgroup walks items (image, head) i, i + G, ...: while item i is multiplied out of LDS
// buffer i & 1, the K/V rows of item i + G arrive in the other buffer and its query rows in a third region, all by LDS-DMA, and
// the output stores of item i drain during item i + G.  Per wave and iteration the vector-memory stream is
//   DMA(next: K, V, Q pieces) | 4 output stores (this)
// so the wait at the top of the next iteration is the counted vmcnt(4): everything but this item's stores.
// The LDS-DMA instructions are INLINE ASM.  hipcc's wait-count pass treats a pending LDS-DMA it knows about as a pending LDS
// write: it put s_waitcnt vmcnt(0) in front of the first ds_read_b64_tr_b16 of the compute phase (the transpose-read intrinsic
// carries no address it could disambiguate), i.e. it drained the prefetch right where it was meant to overlap.  An asm LDS-DMA
// has no register destination (register-safe, guide §5.7 item 1); its completion is ordered by the explicit vmcnt + barrier
// below.  M0 (the LDS destination) is saved and restored inside the statement; the descriptor and M0 come from readfirstlane,
// hence the leading s_nop 4 (SALU write -> VMEM read of an SGPR).
// One buffer descriptor per item and operand (base = the image's first row at this head) with per-lane byte offsets that are
// the same for every item (row * ld + swizzled chunk) and the K / V column offset in the scalar offset.
// Arithmetic per query tile = attn_query_tile: bit-identical to attention_kernel.
// Requires Lq == L, NT <= NW (one query tile per wave; waves without a tile only stage) and 2 x (K + V) + Q rows <= 160 KiB.
__device__ __forceinline__ void attn_dma16(uint4_t rs, int voff, int soff, unsigned lds_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
    unsigned keep;
    asm volatile(
        "s_nop 4\n\t"
        "s_mov_b32 %0, m0\n\t"
        "s_mov_b32 m0, %3\n\t"
        "s_nop 0\n\t"
        "buffer_load_dwordx4 %1, %2, %4 offen lds\n\t"
        "s_mov_b32 m0, %0"
        : "=&s"(keep)
        : "v"(voff), "s"(rs), "s"(lds_addr), "s"(soff)
        : "memory");
#endif
}
__device__ __forceinline__ uint4_t attn_rsrc(const void* base) {
    const uint64_t addr = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)addr), hi = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
    return uint4_t{lo, hi & 0xffffu, 0x7fffffffu, 0x00020000u};   // stride 0, num_records 2 GiB, raw 32-bit data format
}

template <int NW, int WPS, int VAR>   // waves per workgroup, waves per SIMD the register budget must allow, softmax variant
__global__ __launch_bounds__(NW * 64, WPS) void attention_pipe_kernel(const half_t* __restrict__ qp, int ldq, long q_batch,
                                                                      const half_t* __restrict__ kvp, int ldkv, int k_off, int v_off,
                                                                      half_t* __restrict__ out, int L, int H, int causal, int NT, int KR,
                                                                      int nitems) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int NST = 4;                                    // output stores per wave and item (attn_store_tile)
    constexpr int MAXP = 4;                                   // LDS-DMA pieces (8 rows each) per wave and operand: NT*32 <= NW*8*MAXP
    constexpr int RB = ATT_DH * 2;                            // bytes per row
    const int LP = NT * 32;
    const int BUF = (KR + LP) * RB;                           // bytes per K/V buffer: K rows [0, KR) | V rows [0, LP)
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int hi = lane >> 5, ql = lane & 31;
    const int W = H * ATT_DH, G = gridDim.x;
    const bool has_tile = wave < NT;
    const unsigned lds0 = (unsigned)(size_t)(pgemm::lds_ptr_t)smem;
    char* Qs = smem + 2 * BUF;                                // [KR rows][64] query rows of the item about to be computed (K swizzle)
    int voff[2];
    attn_voff(lane, voff);
    // per-lane source offsets of this wave's pieces, identical for every item
    int kvo[MAXP], vvo[MAXP], qvo[MAXP];
#pragma unroll
    for (int i = 0; i < MAXP; ++i) {
        const int r = wave * 8 + i * NW * 8 + (lane >> 3);
        const int rc = r < L ? r : L - 1;                     // K / Q rows >= L: masked / never stored; V rows >= L: finite filler (their probabilities are exact zeros)
        const int ck = ((lane & 7) ^ pgemm::swz_key(r)) << 3, cv = ((lane & 7) ^ (((r >> 1) & 1) << 2)) << 3;
        kvo[i] = (rc * ldkv + ck) * 2;
        vvo[i] = (rc * ldkv + cv) * 2;
        qvo[i] = (rc * ldq + ck) * 2;
    }
    const int q = wave * 32 + ql;                             // this lane's query row (has_tile)
    auto stage = [&](int item, int buf) {
        const int b = item / H, h = item - b * H;
        const uint4_t rkv = attn_rsrc(kvp + (size_t)b * L * ldkv + h * ATT_DH);
        const uint4_t rq = attn_rsrc(qp + (size_t)b * q_batch + h * ATT_DH);
        const unsigned kb = lds0 + buf * BUF, vb = kb + KR * RB, qb = lds0 + 2 * BUF;
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int r0 = wave * 8 + i * NW * 8;
            if (r0 < KR) attn_dma16(rkv, kvo[i], k_off * 2, kb + r0 * RB);
        }
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int r0 = wave * 8 + i * NW * 8;
            if (r0 < LP) attn_dma16(rkv, vvo[i], v_off * 2, vb + r0 * RB);
        }
#pragma unroll
        for (int i = 0; i < MAXP; ++i) {
            const int r0 = wave * 8 + i * NW * 8;
            if (r0 < KR) attn_dma16(rq, qvo[i], 0, qb + r0 * RB);
        }
    };
    int item = blockIdx.x;
    if (item >= nitems) return;
    stage(item, 0);
    for (int it = 0; item < nitems; item += G, ++it) {
        const int cur = it & 1;
        // this item's K / V / Q pieces have landed (the previous item's stores may still be in flight)
        if (has_tile && it > 0) pgemm::wait_vm<NST>(); else pgemm::wait_vm<0>();
        pgemm::lds_barrier();                                 // everyone's pieces are visible; everyone is done with the other K/V buffer
        half8_t qf[4];
        if (has_tile) {
            const int qr = q < KR ? q : KR - 1;               // rows of the last tile beyond the staged ones: never stored
#pragma unroll
            for (int s = 0; s < 4; ++s)
                qf[s] = *reinterpret_cast<const half8_t*>(Qs + qr * RB + (((s * 2 + hi) ^ pgemm::swz_key(qr)) << 4));
        }
        pgemm::lds_barrier();                                 // every wave holds its query fragments: the Q region is free
#ifndef PCLIP_ATT_ABL
#define PCLIP_ATT_ABL 0          // ablation builds (tools/ablate_attention.py): 1 no prefetch DMA in the loop, 2 no compute, 4 no stores
#endif
        const int next = item + G;
        if (next < nitems && !(PCLIP_ATT_ABL & 1)) stage(next, cur ^ 1);
        if (has_tile) {
            const half_t* Ks = reinterpret_cast<const half_t*>(smem + cur * BUF);
            const half_t* Vs = Ks + KR * ATT_DH;
            float16_t o[2];
            float lrun;
#ifndef PCLIP_ATT_STAGGER
#define PCLIP_ATT_STAGGER 0
#endif
#if PCLIP_ATT_STAGGER && defined(__HIP_DEVICE_COMPILE__)
            if (wave >= NW / 2) __builtin_amdgcn_s_sleep(PCLIP_ATT_STAGGER);     // second wave of each SIMD: start out of phase with the first
#endif
            if (PCLIP_ATT_ABL & 2) {
#pragma unroll
                for (int j = 0; j < 2; ++j)
#pragma unroll
                    for (int e = 0; e < 16; ++e) o[j][e] = (float)qf[e & 3][e & 7];
                lrun = 1.f;
            } else
                attn_query_tile<true, VAR>(Ks, Vs, qf, q, wave, L, causal, NT, hi, ql, voff, o, lrun);
            const int b = item / H, h = item - b * H;
            attn_store_tile(out + ((size_t)b * L + q) * W + h * ATT_DH, o, lrun, hi, q < L && !(PCLIP_ATT_ABL & 4));
        }
    }
}

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
                                                           half_t* __restrict__ x0, half_t* __restrict__ h, float* __restrict__ stats) {
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
        if (stats) {                                         // the first block's ln_1 is folded into its in_proj: (mean, rstd) of x0 instead of h
            const float2_t ms = row_mean_rstd<NCH>(xh, W, lane, eps);
            if (lane == 0) *reinterpret_cast<float2_t*>(stats + row * 2) = ms;
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

inline int row_grid(int R) { int g = ceil_div(R, 4); return g < 1 ? 1 : (g > 16384 ? 16384 : g); }
inline int flat_grid(size_t n) { size_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g)); }

}  // namespace

// Tile choice and row split.  The fast kernels are persistent (one workgroup per resident slot), so a launch costs
// ceil(tiles / slots) ROUNDS of one tile each; when the last round is mostly empty (M=50432, N=768: 591 tiles of
// 256x256 on 256 slots = 2.31 rounds, paid as 3) the rows of the partial round are split off and dispatched again,
// where a smaller tile spreads them over every CU.  Estimated time of a configuration = rounds x tile area x
// (slots / CUs) / relative K-loop rate (measured on MI355X, tools/ab_cfg.py), in units of 128x128 tile areas.
#ifndef PCLIP_GEMM_4W_DEFAULT
#define PCLIP_GEMM_4W_DEFAULT 1
#endif
static long g_gemm_launches = 0;
static int g_use4w = -1;                    // 256 x 256 tiles on the four-wave asm-loop kernel: -1 = PCLIP_GEMM_4W / the default, decided at the first launch
extern "C" int pclip_gemm4w_config(int mode) {
    const int before = g_use4w;
    if (mode >= 0) g_use4w = mode != 0;
    return before;
}
extern "C" long pclip_gemm_kernel_launches(void) { return g_gemm_launches; }

namespace {
struct TileCfg { int bm, bn, wg_per_cu; double eff; };
constexpr int kNumCfgs = 5;                  // configurations the cost model chooses from
constexpr TileCfg kTileCfgs[kNumCfgs] = {{128, 128, 2, 0.85}, {256, 128, 1, 0.85}, {256, 256, 1, 1.0}, {256, 64, 1, 0.6}, {256, 32, 2, 0.4}};
constexpr double kLaunchCost = 0.5;          // extra launch of a split, in the same units

thread_local int g_min_bn = 0;               // act 9 (statistics partials per 64 columns): tiles narrower than 64 columns are excluded
inline double tile_cost(const TileCfg& c, long M, int N, int cus) {
    if (N % c.bn || c.bn < g_min_bn) return 1e30;
    const long slots = (long)c.wg_per_cu * cus, nt = ((M + c.bm - 1) / c.bm) * (N / c.bn);
    // two workgroups per CU share its matrix pipe — unless the launch has no more tiles than CUs: then every workgroup has a CU to itself
    // (the 60-tile tail of the N = 768 GEMMs as 240 tiles of 128 x 128: 10.6 / 28.8 us against 11.7 / 33.1 us as 256 x 64, K = 768 / 3072)
    const double share = nt <= cus ? 1.0 : (double)c.wg_per_cu;
    return (double)((nt + slots - 1) / slots) * (c.bm / 128.0) * (c.bn / 128.0) * share / c.eff;
}
inline int best_cfg(long M, int N, int cus, double* cost_out) {
    int pick = -1;
    double best = 1e29;
    for (int i = 0; i < kNumCfgs; ++i) {
        const double c = tile_cost(kTileCfgs[i], M, N, cus);
        if (c < best) { best = c; pick = i; }
    }
    if (cost_out) *cost_out = best;
    return pick;
}

// fewer 128x64 tiles than CUs: the latency-oriented ring kernel (defined below), bit-identical to the persistent kernels
bool small_applies(int M, int N, int cus);
int launch_small_one(const half_t* A, int lda, const half_t* B, int ldb, int M, int N, int K, const LinearEpi& epi, hipStream_t s);

}  // namespace
// four-wave 256 x 256 tile with the asm K-loop (pclip_gemm4w.hip)
bool pclip_gemm4w_supports(int M, int N, int K, int lda, int ldb, int ldc, const void* C, const void* bias, const void* residual, int act);
int pclip_gemm4w_launch(const half_t* A, int lda, const half_t* B, int ldb, int M, int N, int K, const half_t* bias, half_t* C, int ldc, int act,
                        const half_t* residual, int slots, int rev, hipStream_t s);
namespace {
int gemm_dispatch(const half_t* A, int lda, const half_t* B, int ldb, int M, int N, int K, LinearEpi epi, int cus, int forced,
                  bool may_split, hipStream_t s) {
    struct MinBn { int old; MinBn(int v) : old(g_min_bn) { g_min_bn = v; } ~MinBn() { g_min_bn = old; } } min_bn(epi.act == 9 ? 64 : 0);
    const bool aligned = (!epi.residual || ((epi.act == 5 || epi.act == 6 || epi.act == 9 || epi.act == 10) && ((uintptr_t)epi.residual & 15) == 0)) && epi.ldc % 8 == 0 && ((uintptr_t)epi.C & 15) == 0 &&
                         (!epi.bias || ((uintptr_t)epi.bias & 15) == 0);
    static const bool small_on = !(getenv("PCLIP_GEMM_SMALL") && getenv("PCLIP_GEMM_SMALL")[0] == '0');
    // act 10 where the panel LayerNorm has no fused form (ring / generic / other tiles, a single K-tile, N not 512 ... 1024 in steps of 128, 32-bit panel offsets): the residual GEMM, then the
    // LayerNorm pass over the same rows — the same bits either way (ln_row_pf)
    auto two_launches = [&](int forced_, bool may_split_) {
        LinearEpi e6 = epi;
        e6.act = 6;
        const int rc = gemm_dispatch(A, lda, B, ldb, M, N, K, e6, cus, forced_, may_split_, s);
        if (rc != PCLIP_OK) return rc;
        return pclip_layernorm_f16(epi.C, epi.ldc, epi.ln_gamma, epi.ln_beta, epi.ln_eps, epi.ln_y, M, N, (pclip_stream_t)s);
    };
    if (epi.act == 10 && (!aligned || !epi.ln_cnt || K < 2 * pgemm::BK || N < 512 || N > 1024 || N % 128 || (long)pgemm::Cfg<256, 256, 2, 4>::BM * epi.ldc * 2 >= 0x7fffffffL ||
                          (forced == -1 && small_on && small_applies(M, N, cus))))
        return two_launches(forced, may_split);
    if (aligned && forced == -1 && small_on && (epi.act <= 1 || epi.act == 6 || epi.act == 9 || (((uintptr_t)epi.scale | (uintptr_t)epi.shift) & 15) == 0) && small_applies(M, N, cus))
        return launch_small_one(A, lda, B, ldb, M, N, K, epi, s);
    double cost = 1e30;
    int pick = aligned ? best_cfg(M, N, cus, &cost) : -1;
    if (forced == -2) { may_split = false; pick = -1; }        // generic kernel
    if (epi.act == 5 && pick < 0) { pclip_set_error("pclip_gemm_bn_res_f16: N=%d / alignment not supported by the fused epilogue", N); return PCLIP_E_INVALID; }
    if (epi.act == 9 && pick < 0) { pclip_set_error("pclip_gemm_res_stats_f16: N=%d / alignment not supported by the fused epilogue", N); return PCLIP_E_INVALID; }
    if (epi.act >= 7 && pick < 0) { pclip_set_error("pclip_gemm_ln_f16: N=%d / alignment not supported by the fused epilogue", N); return PCLIP_E_INVALID; }
    if (forced >= 0) {
        may_split = false;
        if (aligned && forced < kNumCfgs && N % kTileCfgs[forced].bn == 0) pick = forced;
    }
    if (pick >= 0 && may_split) {
        long split_rows = 0;                                   // rows given to the full rounds of configuration split_cfg
        int split_cfg = -1;
        for (int i = 0; i < kNumCfgs; ++i) {
            const TileCfg& c = kTileCfgs[i];
            if (N % c.bn) continue;
            const long slots = (long)c.wg_per_cu * cus, tiles_n = N / c.bn, nt = ((M + c.bm - 1) / c.bm) * tiles_n;
            const long full_rows = (nt / slots) * slots / tiles_n * c.bm;
            if (nt <= slots || nt % slots == 0 || full_rows >= M) continue;
            double rest = 1e30;
            best_cfg(M - full_rows, N, cus, &rest);
            const double split = tile_cost(c, full_rows, N, cus) + rest + kLaunchCost;
            if (split < cost) { cost = split; split_cfg = i; split_rows = full_rows; }
        }
        if (split_cfg >= 0) {
            int rc = gemm_dispatch(A, lda, B, ldb, (int)split_rows, N, K, epi, cus, split_cfg, false, s);
            if (rc != PCLIP_OK) return rc;
            LinearEpi tail = epi;
            tail.C = epi.C + (size_t)split_rows * epi.ldc;
            if (epi.residual) tail.residual = epi.residual + (size_t)split_rows * epi.ldc;   // act 5 / 6: same row stride as C
            if (epi.rowstats) tail.rowstats = epi.rowstats + (size_t)split_rows * 2;         // act 7 / 8 (split_rows is a multiple of 128: 16-byte aligned)
            if (epi.partials) tail.partials = epi.partials + (size_t)split_rows * (N / 64) * 2;   // act 9
            if (epi.act == 10) {                                                                  // the first launch's panels are at most split_rows / 128
                tail.ln_y = epi.ln_y + (size_t)split_rows * N;
                tail.ln_cnt = epi.ln_cnt + split_rows / 128;
            }
            return gemm_dispatch(A + (size_t)split_rows * lda, lda, B, ldb, M - (int)split_rows, N, K, tail, cus, -1, true, s);
        }
    }
    if (epi.act == 10 && pick != 2 && pick != 0) return two_launches(pick < 0 ? -2 : pick, false);
    ++g_gemm_launches;
    if (pick < 0 && epi.act == 6) epi.act = 0;                  // generic kernel: bias + residual operands, same roundings
    if (pick == 2) {
        // the same tile on four waves with the hand-scheduled K-loop (bit-identical): PCLIP_GEMM_4W=1 (default: see DESIGN §3)
        if (g_use4w < 0) { const char* e = getenv("PCLIP_GEMM_4W"); g_use4w = e ? (atoi(e) != 0) : PCLIP_GEMM_4W_DEFAULT; }
        if (g_use4w && pclip_gemm4w_supports(M, N, K, lda, ldb, epi.ldc, epi.C, epi.bias, epi.residual, epi.act)) {
            const TileOrder& order = tile_order();
            const bool rev = order.rev == 1 || (order.rev == 2 && K <= 1024);
            return pclip_gemm4w_launch(A, lda, B, ldb, M, N, K, epi.bias, epi.C, epi.ldc, epi.act, epi.residual, cus, rev ? 1 : 0, s);
        }
        return launch_fast<CfgBig>(A, lda, B, ldb, M, N, K, epi, cus, s);
    }
    if (pick == 1) return launch_fast<CfgWide>(A, lda, B, ldb, M, N, K, epi, cus, s);
    if (pick == 0) return launch_fast<CfgSmall>(A, lda, B, ldb, M, N, K, epi, 2 * cus, s);
    if (pick == 3) return launch_fast<CfgNarrow>(A, lda, B, ldb, M, N, K, epi, cus, s);
    if (pick == 4) return launch_fast<CfgThin>(A, lda, B, ldb, M, N, K, epi, 2 * cus, s);
    const int tiles_m = ceil_div(M, 128), tiles_n = ceil_div(N, 128);
    linear_generic_kernel<<<tiles_m * tiles_n, 256, CfgSmall::LDS_BYTES, s>>>(A, lda, B, ldb, M, N, K, epi, tiles_n);
    return pclip_check_launch("gemm_f16 (generic)");
}
}  // namespace

extern "C" int pclip_gemm_f16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                              const void* bias, int act, const void* residual, pclip_stream_t stream) {
    PCLIP_REQUIRE(A && B && C, "pclip_gemm_f16: null pointer");
    PCLIP_REQUIRE(M >= 0 && N > 0 && K > 0, "pclip_gemm_f16: bad shape M=%d N=%d K=%d", M, N, K);
    PCLIP_REQUIRE(K % pgemm::BK == 0, "pclip_gemm_f16: K=%d must be a multiple of %d", K, pgemm::BK);
    PCLIP_REQUIRE(lda >= K && ldb >= K && ldc >= N && lda % 8 == 0 && ldb % 8 == 0, "pclip_gemm_f16: bad leading dims");
    PCLIP_REQUIRE(act == 0 || act == 1, "pclip_gemm_f16: unknown activation %d", act);
    if (M == 0) return PCLIP_OK;
    LinearEpi epi{(const half_t*)bias, (const half_t*)residual, (half_t*)C, ldc, act, nullptr, nullptr};
    if (residual && bias && act == 0) epi.act = 6;              // fused residual epilogue of the persistent / ring kernels
    int cus = pclip_device_cus();                          // per device (a process may drive several GPUs): cached per device id in pclip_api.hip
    if (cus <= 0) cus = 256;
    static int forced = -1;
    static bool live = false, nosplit = false, init = false;
    if (!init || live) {
        init = true;
        const char* f = getenv("PCLIP_GEMM_CFG");          // tuning override: 0 small, 1 wide, 2 big, 3 generic, 4 narrow (256x64)
        forced = f ? atoi(f) : -1;
        if (forced == 3) forced = -2;                       // generic kernel
        else if (forced == 4) forced = 3;                   // index of the 256x64 configuration
        else if (forced >= 5) forced = -1;
        live = getenv("PCLIP_GEMM_CFG_LIVE") != nullptr;   // tools/ab_cfg.py: re-read the overrides on every call
        nosplit = getenv("PCLIP_GEMM_NOSPLIT") != nullptr;
    }
    return gemm_dispatch((const half_t*)A, lda, (const half_t*)B, ldb, M, N, K, epi, cus, forced, !nosplit, (hipStream_t)stream);
}

// x += A W^T + bias in place of C = residual (pclip_gemm_f16 with `residual`), and the statistics partials of the updated rows.
extern "C" int pclip_gemm_res_stats_f16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                                        const void* bias, const void* residual, float* partials, pclip_stream_t stream) {
    PCLIP_REQUIRE(A && B && C && bias && residual && partials, "pclip_gemm_res_stats_f16: null pointer");
    PCLIP_REQUIRE(M >= 0 && N > 0 && K > 0, "pclip_gemm_res_stats_f16: bad shape M=%d N=%d K=%d", M, N, K);
    PCLIP_REQUIRE(K % pgemm::BK == 0 && N % 64 == 0, "pclip_gemm_res_stats_f16: K=%d / N=%d must be multiples of 64", K, N);
    PCLIP_REQUIRE(lda >= K && ldb >= K && ldc >= N && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0, "pclip_gemm_res_stats_f16: bad leading dims");
    PCLIP_REQUIRE((((uintptr_t)C | (uintptr_t)residual | (uintptr_t)bias | (uintptr_t)partials) & 15) == 0, "pclip_gemm_res_stats_f16: operands must be 16-byte aligned");
    if (M == 0) return PCLIP_OK;
    LinearEpi epi{(const half_t*)bias, (const half_t*)residual, (half_t*)C, ldc, 9, nullptr, nullptr, nullptr, partials};
    int cus = pclip_device_cus();
    if (cus <= 0) cus = 256;
    static const bool nosplit = getenv("PCLIP_GEMM_NOSPLIT") != nullptr;
    return gemm_dispatch((const half_t*)A, lda, (const half_t*)B, ldb, M, N, K, epi, cus, -1, !nosplit, (hipStream_t)stream);
}

// x += A W^T + bias in place, and y = LayerNorm(x) of the updated rows without a pass of its own (linear_fast_kernel act 10): `panel_counters` = one int per 128 rows
// of x (+ 2), zero on entry and zero again on return (the kernel resets what it counted); NULL, or PCLIP_RES_LN=0, selects the two launches it replaces — same bits.
extern "C" int pclip_gemm_res_ln_f16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K, const void* bias,
                                     const float* gamma, const float* beta, float eps, void* y, int32_t* panel_counters, pclip_stream_t stream) {
    PCLIP_REQUIRE(A && B && C && bias && gamma && beta && y, "pclip_gemm_res_ln_f16: null pointer");
    PCLIP_REQUIRE(M >= 0 && N > 0 && K > 0, "pclip_gemm_res_ln_f16: bad shape M=%d N=%d K=%d", M, N, K);
    PCLIP_REQUIRE(K % pgemm::BK == 0 && N % 8 == 0 && N <= 4096, "pclip_gemm_res_ln_f16: K=%d must be a multiple of 64, N=%d of 8 (<= 4096)", K, N);
    PCLIP_REQUIRE(lda >= K && ldb >= K && ldc >= N && lda % 8 == 0 && ldb % 8 == 0, "pclip_gemm_res_ln_f16: bad leading dims");
    PCLIP_REQUIRE((((uintptr_t)y | (uintptr_t)panel_counters) & 15) == 0, "pclip_gemm_res_ln_f16: y / panel_counters must be 16-byte aligned");
    if (M == 0) return PCLIP_OK;
    static const bool fused = !(getenv("PCLIP_RES_LN") && getenv("PCLIP_RES_LN")[0] == '0');
    LinearEpi epi{(const half_t*)bias, (const half_t*)C, (half_t*)C, ldc, 10, nullptr, nullptr};
    epi.ln_gamma = gamma;
    epi.ln_beta = beta;
    epi.ln_y = (half_t*)y;
    epi.ln_cnt = fused ? panel_counters : nullptr;
    epi.ln_eps = eps;
    int cus = pclip_device_cus();
    if (cus <= 0) cus = 256;
    static const bool nosplit = getenv("PCLIP_GEMM_NOSPLIT") != nullptr;
    return gemm_dispatch((const half_t*)A, lda, (const half_t*)B, ldb, M, N, K, epi, cus, -1, !nosplit, (hipStream_t)stream);
}

// LayerNorm folded into the linear that consumes it (see ln_fold): y = act(LN(x) W^T + b) from the un-normalised rows x, their
// (mean, rstd) pairs and the folded weight / column sums / bias of pclip_ln_fold_weights_f16.
extern "C" int pclip_gemm_ln_f16(const void* x, int ldx, const float* rowstats, const void* Wf, int ldw, void* C, int ldc, int M, int N,
                                 int K, const float* colsum, const float* bfold, int act, pclip_stream_t stream) {
    PCLIP_REQUIRE(x && rowstats && Wf && C && colsum && bfold, "pclip_gemm_ln_f16: null pointer");
    PCLIP_REQUIRE(M >= 0 && N > 0 && K > 0, "pclip_gemm_ln_f16: bad shape M=%d N=%d K=%d", M, N, K);
    PCLIP_REQUIRE(K % pgemm::BK == 0 && N % 64 == 0, "pclip_gemm_ln_f16: K=%d / N=%d must be multiples of 64", K, N);
    PCLIP_REQUIRE(ldx >= K && ldw >= K && ldc >= N && ldx % 8 == 0 && ldw % 8 == 0 && ldc % 8 == 0, "pclip_gemm_ln_f16: bad leading dims");
    PCLIP_REQUIRE((((uintptr_t)colsum | (uintptr_t)bfold | (uintptr_t)rowstats | (uintptr_t)C) & 15) == 0, "pclip_gemm_ln_f16: operands must be 16-byte aligned");
    PCLIP_REQUIRE(act == 0 || act == 1, "pclip_gemm_ln_f16: unknown activation %d", act);
    if (M == 0) return PCLIP_OK;
    LinearEpi epi{nullptr, nullptr, (half_t*)C, ldc, act == 1 ? 8 : 7, colsum, bfold, rowstats};
    int cus = pclip_device_cus();
    if (cus <= 0) cus = 256;
    static const bool nosplit = getenv("PCLIP_GEMM_NOSPLIT") != nullptr;
    return gemm_dispatch((const half_t*)x, ldx, (const half_t*)Wf, ldw, M, N, K, epi, cus, -1, !nosplit, (hipStream_t)stream);
}

namespace {
// ---- split-K for small M (serving: M = 197 x batch rows, the class-token tail: M = batch) ------------------------------------
// A request of a few images gives every encoder linear 12 - 48 output tiles for 256 CUs and a K-loop of 12 - 48 dependent
// LDS-DMA round trips (c_proj at M = 197: 12 workgroups x 48 K-tiles = 50 us).  Here the K range is cut into S slices, one
// workgroup per (128 x 64 tile, slice), each writing its fp32 accumulators (valid rows only) as a [S][M][N] slab; a second,
// fully parallel launch adds the S slabs in slice order (deterministic), applies bias / QuickGELU and stores fp16.  The slabs
// are 32 KB per (tile, slice) — far beyond what a last-arriver reduction inside the first launch handles well (a first
// version with device-scope fences + a tile counter measured 3x SLOWER than the unsplit kernel: every workgroup's release
// writes back its XCD's L2) — so the combine sits at the launch boundary (guide §5: "combine in the next kernel").
using CfgSplit = pgemm::Cfg<128, 64, 4, 2>;       // 8 waves x 32x32: two waves per SIMD share the DMA set-up and the MFMAs of a K-tile

constexpr int kSmallStages = 4;                             // 6 slots measured no faster (7.5 vs 7.1 us at 12 K-tiles): not latency-limited any more
constexpr int kSmallLds = kSmallStages * CfgSplit::STAGE_BYTES;        // the K-tile ring of pgemm::mainloop_ring (96 KiB)

// S > 1: slice ks of the K range -> fp32 slab.  S == 1: the whole K range, bias / QuickGELU and the fp16 store right here.
template <int ACT>
__global__ __launch_bounds__(CfgSplit::NTHREADS, 1) void linear_small_kernel(const half_t* __restrict__ A, int lda,
                                                                           const half_t* __restrict__ B, int ldb, int M, int N,
                                                                           int K, int tiles_n, int S, int steps_per,
                                                                           float* __restrict__ ws, const half_t* __restrict__ bias,
                                                                           half_t* Cout, int ldc,
                                                                           const float* __restrict__ scale,
                                                                           const float* __restrict__ shift,
                                                                           const half_t* residual = nullptr,
                                                                           const float* __restrict__ rowstats = nullptr,
                                                                           float* __restrict__ partials = nullptr) {
    using C = CfgSplit;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile = blockIdx.x / S, ks = blockIdx.x - tile * S;
    const int tile_m = tile / tiles_n, tile_n = tile - tile_m * tiles_n;
    const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;
    const int k0 = ks * steps_per * pgemm::BK;
    const int klen = (K - k0 < steps_per * pgemm::BK) ? K - k0 : steps_per * pgemm::BK;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / C::WN, wn = wave % C::WN;
    pgemm::Acc<C> acc;
    // S == 1: the bias is the accumulators' initial value exactly as in linear_fast_kernel -> the same bits as that kernel
    pgemm::mainloop_ring<C, kSmallStages>(A + k0, lda, B + k0, ldb, M, N, klen / pgemm::BK, m0, n0, smem, acc, [&]() {
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                half4_t b = {(half_t)0.f, (half_t)0.f, (half_t)0.f, (half_t)0.f};
                if (S == 1 && bias) b = *reinterpret_cast<const half4_t*>(bias + n0 + wn * (C::BN / C::WN) + j * 32 + (g & 1) * 16 + 4 * (lane >> 4));
#pragma unroll
                for (int e = 0; e < 4; ++e)
#pragma unroll
                    for (int i = 0; i < C::TM; ++i) acc.v[i][j][4 * g + e] = (float)b[e];
            }
    });
    if (S == 1) {
        const int col = n0 + 8 * (tid % C::CPR);
        auto pre = [&](int i, int j, int coff, float4_t v, int rl, int g) {
            if (ACT == 1) return quick_gelu16x4(v);
            half4_t h;
            if (ACT == 7 || ACT == 8) {                     // LayerNorm folded into the linear: ln_fold, as linear_fast_kernel
                const int n = n0 + wn * (C::BN / C::WN) + j * 32 + coff;
                const float4_t cs = *reinterpret_cast<const float4_t*>(scale + n), bf = *reinterpret_cast<const float4_t*>(shift + n);
                const int m = m0 + wm * (C::BM / C::WM) + i * 32 + rl;
                const float2_t ms = *reinterpret_cast<const float2_t*>(rowstats + (size_t)(m < M ? m : M - 1) * 2);
                float4_t y;
#pragma unroll
                for (int e = 0; e < 4; ++e) y[e] = ln_fold(v[e], ms[0], ms[1], cs[e], bf[e]);
                if (ACT == 8) return quick_gelu16x4(y);
#pragma unroll
                for (int e = 0; e < 4; ++e) h[e] = (half_t)y[e];
                return h;
            }
            if (ACT == 2 || ACT == 3 || ACT == 5) {         // eval BatchNorm (+ReLU) as in linear_fast_kernel: same roundings
                const int n = n0 + wn * (C::BN / C::WN) + j * 32 + coff;
                const float4_t sc = *reinterpret_cast<const float4_t*>(scale + n), sh = *reinterpret_cast<const float4_t*>(shift + n);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    float y = r16(r16(v[e]) * sc[e] + sh[e]);
                    if (ACT == 3) y = fmaxf(y, 0.f);
                    h[e] = (half_t)y;
                }
                return h;
            }
#pragma unroll
            for (int e = 0; e < 4; ++e) h[e] = (half_t)v[e];
            return h;
        };
        // slot 0 of the ring is the staging buffer: the epilogue's first barrier comes after every wave's last K-tile
        pgemm::epilogue_f16<C, true>(acc, smem, [](int) {}, pre, [&](int r, int c, int, half8_t h) {
            const bool valid = m0 + r < M;
            const size_t o = (size_t)(m0 + r) * ldc + col;
            if ((ACT == 5 || ACT == 6 || ACT == 9) && valid) {
                const half8_t rr = ld_half8(residual + o);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float y = r16((float)rr[j] + (float)h[j]);
                    h[j] = (half_t)(ACT == 5 ? fmaxf(y, 0.f) : y);
                }
            }
            if (valid) st_half8(Cout + o, h);
            if (ACT == 9) {                                  // statistics partials of the updated row segment: as linear_fast_kernel (CPR = 8: one slot)
                float ps, pq;
                stats_chunk(h, ps, pq);
                stats_butterfly<8>(ps, pq);
                if (valid && c == 0) *reinterpret_cast<float2_t*>(partials + ((size_t)(m0 + r) * (N >> 6) + (n0 >> 6)) * 2) = float2_t{ps, pq};
            }
        });
        return;
    }
    // 16x16x32 accumulator layout: element group (i, j, g) of a lane = row wm*64 + i*32 + (g>>1)*16 + (lane&15),
    // columns wn*32 + j*32 + (g&1)*16 + 4*(lane>>4) .. +3
    float* slab = ws + (size_t)ks * M * N;
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int m = m0 + wm * (C::BM / C::WM) + i * 32 + (g >> 1) * 16 + (lane & 15);
                const int n = n0 + wn * (C::BN / C::WN) + j * 32 + (g & 1) * 16 + 4 * (lane >> 4);
                if (m < M)
                    *reinterpret_cast<float4_t*>(slab + (size_t)m * N + n) =
                        float4_t{acc.v[i][j][4 * g], acc.v[i][j][4 * g + 1], acc.v[i][j][4 * g + 2], acc.v[i][j][4 * g + 3]};
            }
}

// The implicit-GEMM 3x3 convolution (conv3x3_fast_kernel) for launches with no more tiles than CUs: the same gather, the ring
// K-loop, the same BatchNorm (+ReLU) epilogue — bit-identical to the persistent kernel.
template <int ACT>
__global__ __launch_bounds__(CfgSplit::NTHREADS, 1) void conv3x3_small_kernel(const half_t* __restrict__ x, const half_t* __restrict__ zero,
                                                                            const half_t* __restrict__ w, int H, int W, int Cin, int M,
                                                                            int N, const float* __restrict__ scale,
                                                                            const float* __restrict__ shift, half_t* __restrict__ Cout,
                                                                            int tiles_n) {
    using C = CfgSplit;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tile_m = blockIdx.x / tiles_n, tile_n = blockIdx.x - tile_m * tiles_n;
    const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;
    const int nt = (9 * Cin + pgemm::BK - 1) / pgemm::BK, K = nt * pgemm::BK;       // w rows are zero-padded to the K-tile (Cin < 64)
    const int tid = threadIdx.x, wave = tid >> 6, wn = wave % C::WN;
    pgemm::ConvGather<C> ga{x, zero, H, W, Cin, M, {}, {}};
    ga.prepare(m0);
    pgemm::Acc<C> acc;
    pgemm::mainloop_ring_g<C, kSmallStages>([&](int t, char* dst) { ga.stage(t, dst); }, w, K, N, nt, n0, smem, acc, [&]() {
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
            for (int j = 0; j < C::TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc.v[i][j][e] = 0.f;
    });
    const int col = n0 + 8 * (tid % C::CPR);
    auto pre = [&](int, int j, int coff, float4_t v, int rl, int g) {
        const int n = n0 + wn * (C::BN / C::WN) + j * 32 + coff;
        const float4_t sc = *reinterpret_cast<const float4_t*>(scale + n), sh = *reinterpret_cast<const float4_t*>(shift + n);
        half4_t h;
#pragma unroll
        for (int e = 0; e < 4; ++e) {
            float y = r16(r16(v[e]) * sc[e] + sh[e]);
            if (ACT == 3) y = fmaxf(y, 0.f);
            h[e] = (half_t)y;
        }
        return h;
    };
    pgemm::epilogue_f16<C, true>(acc, smem, [](int) {}, pre, [&](int r, int, int, half8_t h) {
        if (m0 + r < M) st_half8(Cout + (size_t)(m0 + r) * N + col, h);
    });
}

// out[m, n .. n+7] = act(sum_s slab[s][m][n ..] + bias[n ..]) — slices added in order, one thread per 8 columns
template <int ACT>
__global__ __launch_bounds__(256) void splitk_reduce_kernel(const float* __restrict__ ws, int S, int M, int N,
                                                            const half_t* __restrict__ bias, half_t* __restrict__ Cout, int ldc) {
    const int cpr = N >> 3;
    const size_t idx = (size_t)blockIdx.x * 256 + threadIdx.x;
    if (idx >= (size_t)M * cpr) return;
    const int m = (int)(idx / cpr), n = (int)(idx - (size_t)m * cpr) * 8;
    const float* src = ws + (size_t)m * N + n;
    const size_t slab = (size_t)M * N;
    float4_t lo = {0.f, 0.f, 0.f, 0.f}, hi = lo;
#pragma unroll 8
    for (int s = 0; s < S; ++s) {
        lo += *reinterpret_cast<const float4_t*>(src + s * slab);
        hi += *reinterpret_cast<const float4_t*>(src + s * slab + 4);
    }
    if (bias) {
        const half8_t b = ld_half8(bias + n);
#pragma unroll
        for (int e = 0; e < 4; ++e) { lo[e] += (float)b[e]; hi[e] += (float)b[e + 4]; }
    }
    half4_t h0, h1;
    if (ACT == 1) { h0 = quick_gelu16x4(lo); h1 = quick_gelu16x4(hi); }
    else {
#pragma unroll
        for (int e = 0; e < 4; ++e) { h0[e] = (half_t)lo[e]; h1[e] = (half_t)hi[e]; }
    }
    st_half8(Cout + (size_t)m * ldc + n, half8_t{h0[0], h0[1], h0[2], h0[3], h1[0], h1[1], h1[2], h1[3]});
}

inline int small_attr() {
    static DevOnce done;
    if (!done.done()) {
        const void* fns[] = {(const void*)linear_small_kernel<0>, (const void*)linear_small_kernel<1>, (const void*)linear_small_kernel<2>,
                             (const void*)linear_small_kernel<3>, (const void*)linear_small_kernel<5>, (const void*)linear_small_kernel<6>,
                             (const void*)linear_small_kernel<7>, (const void*)linear_small_kernel<8>, (const void*)linear_small_kernel<9>, (const void*)conv3x3_small_kernel<2>, (const void*)conv3x3_small_kernel<3>};
        for (const void* f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, kSmallLds) != hipSuccess) {
                pclip_set_error("gemm_f16 (small M): cannot raise the dynamic LDS limit to %d", kSmallLds);
                return PCLIP_E_LAUNCH;
            }
        done.set();
    }
    return PCLIP_OK;
}

bool small_applies(int M, int N, int cus) {
    return M > 0 && N % CfgSplit::BN == 0 && (long)ceil_div(M, CfgSplit::BM) * (N / CfgSplit::BN) <= cus;
}

int launch_small_one(const half_t* A, int lda, const half_t* B, int ldb, int M, int N, int K, const LinearEpi& epi, hipStream_t s) {
    if (int e = small_attr()) return e;
    const int tiles_n = N / CfgSplit::BN, grid = ceil_div(M, CfgSplit::BM) * tiles_n, steps = K / pgemm::BK;
    ++g_gemm_launches;
#define PCLIP_SMALL_LAUNCH(ACT)                                                                                                          \
    linear_small_kernel<ACT><<<grid, CfgSplit::NTHREADS, kSmallLds, s>>>(A, lda, B, ldb, M, N, K, tiles_n, 1, steps, nullptr, epi.bias, epi.C, \
                                                                        epi.ldc, epi.scale, epi.shift, epi.residual, epi.rowstats, epi.partials)
    if (epi.act == 5) PCLIP_SMALL_LAUNCH(5);
    else if (epi.act == 9) PCLIP_SMALL_LAUNCH(9);
    else if (epi.act == 7) PCLIP_SMALL_LAUNCH(7);
    else if (epi.act == 8) PCLIP_SMALL_LAUNCH(8);
    else if (epi.act == 6) PCLIP_SMALL_LAUNCH(6);
    else if (epi.act == 1) PCLIP_SMALL_LAUNCH(1);
    else if (epi.act == 2) PCLIP_SMALL_LAUNCH(2);
    else if (epi.act == 3) PCLIP_SMALL_LAUNCH(3);
    else PCLIP_SMALL_LAUNCH(0);
#undef PCLIP_SMALL_LAUNCH
    return pclip_check_launch("gemm_f16 (small M)");
}

struct SplitPlan { int tiles_m, tiles_n, S, steps_per; size_t bytes; };
// The slicing depends on K ONLY (up to 8 slices of >= 2 K-tiles), so that a row's result does not depend on how many other rows
// the call carries; M and N only decide whether the split is used at all: few tiles for the chip, a K-loop long enough to cut.
inline SplitPlan splitk_plan(int M, int N, int K, int cus) {
    SplitPlan pl{0, 0, 0, 0, 0};
    if (M <= 0 || N <= 0 || K <= 0 || N % CfgSplit::BN || K % pgemm::BK) return pl;
    pl.tiles_m = ceil_div(M, CfgSplit::BM);
    pl.tiles_n = N / CfgSplit::BN;
    const int tiles = pl.tiles_m * pl.tiles_n, steps = K / pgemm::BK;
    if (steps < 8) return pl;
    pl.steps_per = steps / 8 > 2 ? steps / 8 : 2;
    pl.S = ceil_div(steps, pl.steps_per);
    pl.bytes = (size_t)pl.S * M * N * sizeof(float);
    // measured model (tools/splitk_bench.py, us): one launch of the ring kernel = 3 + 0.34 per K-tile; split = 4.5 (two launches)
    // + 0.34 per K-tile of a slice + the slabs written and read back at ~3 TB/s
    static const int always = getenv("PCLIP_SPLITK_ALWAYS") ? atoi(getenv("PCLIP_SPLITK_ALWAYS")) : 0;
    const double t_one = 3.0 + 0.34 * steps, t_split = 4.5 + 0.34 * pl.steps_per + 2.0 * (double)pl.bytes / 3.0e6;
    if (tiles > cus || tiles * pl.S > 4 * cus || (!always && t_split + 0.5 > t_one)) { pl.S = 0; pl.bytes = 0; return pl; }
    return pl;
}

}  // namespace

extern "C" size_t pclip_gemm_splitk_workspace(int M, int N, int K) {
    int cus = pclip_device_cus();
    if (cus <= 0) cus = 256;
    return splitk_plan(M, N, K, cus).bytes;
}

extern "C" int pclip_gemm_splitk_f16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                                     const void* bias, int act, void* ws, size_t ws_bytes, pclip_stream_t stream) {
    PCLIP_REQUIRE(A && B && C && ws, "pclip_gemm_splitk_f16: null pointer");
    PCLIP_REQUIRE(act == 0 || act == 1, "pclip_gemm_splitk_f16: unknown activation %d", act);
    PCLIP_REQUIRE(lda >= K && ldb >= K && ldc >= N && lda % 8 == 0 && ldb % 8 == 0 && ldc % 8 == 0, "pclip_gemm_splitk_f16: bad leading dims");
    PCLIP_REQUIRE(((uintptr_t)A & 15) == 0 && ((uintptr_t)B & 15) == 0 && ((uintptr_t)C & 15) == 0 && (!bias || ((uintptr_t)bias & 15) == 0) &&
                      ((uintptr_t)ws & 15) == 0, "pclip_gemm_splitk_f16: operands must be 16-byte aligned");
    int cus = pclip_device_cus();
    if (cus <= 0) cus = 256;
    const SplitPlan pl = splitk_plan(M, N, K, cus);
    PCLIP_REQUIRE(pl.S >= 2, "pclip_gemm_splitk_f16: shape M=%d N=%d K=%d is not a split-K shape (pclip_gemm_splitk_workspace == 0)", M, N, K);
    if (ws_bytes < pl.bytes) { pclip_set_error("pclip_gemm_splitk_f16: workspace %zu < %zu", ws_bytes, pl.bytes); return PCLIP_E_WORKSPACE; }
    hipStream_t s = (hipStream_t)stream;
    if (int e = small_attr()) return e;
    linear_small_kernel<0><<<pl.tiles_m * pl.tiles_n * pl.S, CfgSplit::NTHREADS, kSmallLds, s>>>(
        (const half_t*)A, lda, (const half_t*)B, ldb, M, N, K, pl.tiles_n, pl.S, pl.steps_per, (float*)ws, nullptr, nullptr, 0, nullptr, nullptr);
    const int rgrid = (int)(((size_t)M * (N / 8) + 255) / 256);
    if (act == 1)
        splitk_reduce_kernel<1><<<rgrid, 256, 0, s>>>((const float*)ws, pl.S, M, N, (const half_t*)bias, (half_t*)C, ldc);
    else
        splitk_reduce_kernel<0><<<rgrid, 256, 0, s>>>((const float*)ws, pl.S, M, N, (const half_t*)bias, (half_t*)C, ldc);
    return pclip_check_launch("gemm_f16 (split-K)");
}

extern "C" int pclip_gemm_bn_res_f16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                                     const float* scale, const float* shift, const void* residual, pclip_stream_t stream) {
    PCLIP_REQUIRE(A && B && C && scale && shift && residual, "pclip_gemm_bn_res_f16: null pointer");
    PCLIP_REQUIRE(M >= 0 && N > 0 && K > 0, "pclip_gemm_bn_res_f16: bad shape M=%d N=%d K=%d", M, N, K);
    PCLIP_REQUIRE(K % pgemm::BK == 0 && N % 64 == 0, "pclip_gemm_bn_res_f16: K=%d, N=%d must be multiples of 64", K, N);
    PCLIP_REQUIRE(lda >= K && ldb >= K && ldc >= N && lda % 8 == 0 && ldb % 8 == 0, "pclip_gemm_bn_res_f16: bad leading dims");
    PCLIP_REQUIRE(((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0, "pclip_gemm_bn_res_f16: scale / shift must be 16-byte aligned");
    if (M == 0) return PCLIP_OK;
    LinearEpi epi{nullptr, (const half_t*)residual, (half_t*)C, ldc, 5, scale, shift};
    int cus = pclip_device_cus();
    if (cus <= 0) cus = 256;
    return gemm_dispatch((const half_t*)A, lda, (const half_t*)B, ldb, M, N, K, epi, cus, -1, true, (hipStream_t)stream);
}

extern "C" int pclip_gemm_bn_f16(const void* A, int lda, const void* B, int ldb, void* C, int ldc, int M, int N, int K,
                                 const float* scale, const float* shift, int relu, pclip_stream_t stream) {
    PCLIP_REQUIRE(A && B && C && scale && shift, "pclip_gemm_bn_f16: null pointer");
    PCLIP_REQUIRE(M >= 0 && N > 0 && K > 0, "pclip_gemm_bn_f16: bad shape M=%d N=%d K=%d", M, N, K);
    PCLIP_REQUIRE(K % pgemm::BK == 0, "pclip_gemm_bn_f16: K=%d must be a multiple of %d", K, pgemm::BK);
    PCLIP_REQUIRE(lda >= K && ldb >= K && ldc >= N && lda % 8 == 0 && ldb % 8 == 0, "pclip_gemm_bn_f16: bad leading dims");
    if (M == 0) return PCLIP_OK;
    LinearEpi epi{nullptr, nullptr, (half_t*)C, ldc, relu ? 3 : 2, scale, shift};
    int cus = pclip_device_cus();
    if (cus <= 0) cus = 256;
    const bool strips_ok = N % 4 == 0 && ((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0;
    return gemm_dispatch((const half_t*)A, lda, (const half_t*)B, ldb, M, N, K, epi, cus, strips_ok ? -1 : -2, true, (hipStream_t)stream);
}

namespace {
template <class C, int ACT>
int launch_conv2(const void* x, const void* zero, const void* w, int B, int H, int W, int Cin, int Cout, const float* scale,
                 const float* shift, void* y, int slots, hipStream_t s) {
    static DevOnce attr;
    constexpr int LDS = C::LDS_BYTES + 2 * 2 * C::BN * 4;
    if (!attr.done()) {
        if (hipFuncSetAttribute((const void*)conv3x3_fast_kernel<C, ACT>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
            pclip_set_error("pclip_conv3x3_bn_f16: cannot raise the dynamic LDS limit to %d", LDS);
            return PCLIP_E_LAUNCH;
        }
        attr.set();
    }
    const int M = B * H * W, tiles_m = ceil_div(M, C::BM), tiles_n = Cout / C::BN, ntiles = tiles_m * tiles_n;
    conv3x3_fast_kernel<C, ACT><<<ntiles < slots ? ntiles : slots, C::NTHREADS, LDS, s>>>(
        (const half_t*)x, (const half_t*)zero, (const half_t*)w, H, W, Cin, M, Cout, scale, shift, (half_t*)y, tiles_n, ntiles);
    return pclip_check_launch("conv3x3_bn");
}
template <class C>
int launch_conv(const void* x, const void* zero, const void* w, int B, int H, int W, int Cin, int Cout, const float* scale,
                const float* shift, int relu, void* y, int slots, hipStream_t s) {
    return relu ? launch_conv2<C, 3>(x, zero, w, B, H, W, Cin, Cout, scale, shift, y, slots, s)
                : launch_conv2<C, 2>(x, zero, w, B, H, W, Cin, Cout, scale, shift, y, slots, s);
}
}  // namespace

extern "C" int pclip_conv3x3_bn_f16(const void* x, const void* w, const void* zero_line, int B, int H, int W, int Cin, int Cout,
                                    const float* scale, const float* shift, int relu, void* y, pclip_stream_t stream) {
    PCLIP_REQUIRE(x && w && zero_line && scale && shift && y, "pclip_conv3x3_bn_f16: null pointer");
    PCLIP_REQUIRE(B >= 0 && H > 0 && W > 0 && H < 32768 && W < 32768, "pclip_conv3x3_bn_f16: bad shape B=%d H=%d W=%d", B, H, W);
    PCLIP_REQUIRE(Cin > 0 && (Cin % 64 == 0 || Cin == 8 || Cin == 16 || Cin == 32), "pclip_conv3x3_bn_f16: Cin=%d must be a multiple of 64, or 8 / 16 / 32 (use im2col + pclip_gemm_bn_f16 otherwise)", Cin);
    PCLIP_REQUIRE(Cout > 0 && (Cout % 64 == 0 || Cout == 32), "pclip_conv3x3_bn_f16: Cout=%d must be a multiple of 64, or 32", Cout);
    PCLIP_REQUIRE((long)B * H * W < (1L << 31) / 1, "pclip_conv3x3_bn_f16: too many output pixels");
    PCLIP_REQUIRE(((uintptr_t)scale & 15) == 0 && ((uintptr_t)shift & 15) == 0 && ((uintptr_t)y & 15) == 0 && ((uintptr_t)x & 15) == 0,
                  "pclip_conv3x3_bn_f16: pointers must be 16-byte aligned");
    if (B == 0) return PCLIP_OK;
    int cus = pclip_device_cus();
    if (cus <= 0) cus = 256;
    hipStream_t s = (hipStream_t)stream;
    if (Cout == 32)                                                             // the stem's 32 -> 32 convolution: 256 x 32 tiles
        return launch_conv<CfgThin>(x, zero_line, w, B, H, W, Cin, Cout, scale, shift, relu, y, 2 * cus, s);
    static const bool small_on = !(getenv("PCLIP_GEMM_SMALL") && getenv("PCLIP_GEMM_SMALL")[0] == '0');
    if (small_on && small_applies(B * H * W, Cout, cus)) {                     // a request of a few images: the ring kernel
        if (int e = small_attr()) return e;
        const int tiles_n = Cout / CfgSplit::BN, grid = ceil_div(B * H * W, CfgSplit::BM) * tiles_n;
        if (relu)
            conv3x3_small_kernel<3><<<grid, CfgSplit::NTHREADS, kSmallLds, s>>>((const half_t*)x, (const half_t*)zero_line, (const half_t*)w, H, W, Cin,
                                                                           B * H * W, Cout, scale, shift, (half_t*)y, tiles_n);
        else
            conv3x3_small_kernel<2><<<grid, CfgSplit::NTHREADS, kSmallLds, s>>>((const half_t*)x, (const half_t*)zero_line, (const half_t*)w, H, W, Cin,
                                                                           B * H * W, Cout, scale, shift, (half_t*)y, tiles_n);
        return pclip_check_launch("conv3x3_bn (small M)");
    }
    double cost;
    int pick = best_cfg((long)B * H * W, Cout, cus, &cost);
    if (pick == 4 || pick < 0) pick = 3;                        // the 4-wave thin tile has no gather variant; Cout % 64 == 0 always fits 256x64
    if (pick == 2) return launch_conv<CfgBig>(x, zero_line, w, B, H, W, Cin, Cout, scale, shift, relu, y, cus, s);
    if (pick == 1) return launch_conv<CfgWide>(x, zero_line, w, B, H, W, Cin, Cout, scale, shift, relu, y, cus, s);
    if (pick == 0) return launch_conv<CfgSmall>(x, zero_line, w, B, H, W, Cin, Cout, scale, shift, relu, y, 2 * cus, s);
    return launch_conv<CfgNarrow>(x, zero_line, w, B, H, W, Cin, Cout, scale, shift, relu, y, cus, s);
}

#define DISPATCH_NCH(D, CALL)                                  \
    do {                                                       \
        if ((D) <= 512) { constexpr int NCH = 1; CALL; }       \
        else if ((D) <= 1024) { constexpr int NCH = 2; CALL; } \
        else if ((D) <= 2048) { constexpr int NCH = 4; CALL; } \
        else { constexpr int NCH = 8; CALL; }                  \
    } while (0)

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

extern "C" int pclip_row_stats_f16(const void* x, int ld_x, float eps, float* stats, int R, int D, pclip_stream_t stream) {
    PCLIP_REQUIRE(x && stats, "pclip_row_stats_f16: null pointer");
    PCLIP_REQUIRE(R >= 0 && D > 0 && D % 8 == 0 && D <= 4096 && ld_x >= D && ld_x % 8 == 0, "pclip_row_stats_f16: bad R=%d D=%d ld=%d", R, D, ld_x);
    if (R == 0) return PCLIP_OK;
    hipStream_t s = (hipStream_t)stream;
    DISPATCH_NCH(D, (row_stats_kernel<NCH><<<row_grid(R), 256, 0, s>>>((const half_t*)x, ld_x, eps, stats, R, D)));
    return pclip_check_launch("row_stats_f16");
}

extern "C" int pclip_row_stats_finalize(const float* partials, int R, int D, float eps, float* stats, pclip_stream_t stream) {
    PCLIP_REQUIRE(partials && stats, "pclip_row_stats_finalize: null pointer");
    PCLIP_REQUIRE(R >= 0 && D > 0 && D % 64 == 0, "pclip_row_stats_finalize: bad R=%d D=%d", R, D);
    if (R == 0) return PCLIP_OK;
    stats_finalize_kernel<<<ceil_div(R, 256), 256, 0, (hipStream_t)stream>>>(partials, R, D, eps, stats);
    return pclip_check_launch("row_stats_finalize");
}

extern "C" int pclip_ln_fold_weights_f16(const void* W, int ldw, int N, int K, const float* gamma, const float* beta, const void* bias,
                                         void* Wf, float* colsum, float* bfold, pclip_stream_t stream) {
    PCLIP_REQUIRE(W && gamma && beta && Wf && colsum && bfold, "pclip_ln_fold_weights_f16: null pointer");
    PCLIP_REQUIRE(N > 0 && K > 0 && K % 8 == 0 && ldw >= K && ldw % 8 == 0, "pclip_ln_fold_weights_f16: bad N=%d K=%d ldw=%d", N, K, ldw);
    ln_fold_weights_kernel<<<ceil_div(N, 4), 256, 0, (hipStream_t)stream>>>((const half_t*)W, ldw, N, K, gamma, beta, (const half_t*)bias,
                                                                            (half_t*)Wf, colsum, bfold);
    return pclip_check_launch("ln_fold_weights_f16");
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
static int att_mode_from_env() {                 // PCLIP_ATT_PIPE=0 / 1: initial mode (A/B runs of whole programs); default automatic
    const char* e = getenv("PCLIP_ATT_PIPE");
    return e && (e[0] == '0' || e[0] == '1') ? e[0] - '0' : -1;
}
static int g_att_mode = att_mode_from_env(), g_att_grid = 0;
extern "C" int pclip_attention_config(int mode, int max_grid) {
    PCLIP_REQUIRE(mode >= -1 && mode <= 1 && max_grid >= 0, "pclip_attention_config: bad mode=%d max_grid=%d", mode, max_grid);
    g_att_mode = mode;
    g_att_grid = max_grid;
    return PCLIP_OK;
}

extern "C" int pclip_attention_q_f16(const void* q, int ldq, long q_batch_stride, const void* kv, int ldkv, int k_off, int v_off,
                                     void* out, int B, int L, int Lq, int H, int dh, int causal, pclip_stream_t stream) {
    PCLIP_REQUIRE(q && kv && out, "pclip_attention_q_f16: null pointer");
    PCLIP_REQUIRE(dh == ATT_DH, "pclip_attention_q_f16: head dim %d unsupported (must be 64)", dh);
    PCLIP_REQUIRE(B >= 0 && H > 0 && L > 0 && L <= ATT_MAX_L && Lq > 0 && Lq <= L, "pclip_attention_q_f16: bad B=%d H=%d L=%d Lq=%d (L <= %d)",
                  B, H, L, Lq, ATT_MAX_L);
    PCLIP_REQUIRE(!causal || Lq == L, "pclip_attention_q_f16: the causal mask needs all queries (Lq == L)");
    PCLIP_REQUIRE(ldq % 8 == 0 && ldkv % 8 == 0 && k_off % 8 == 0 && v_off % 8 == 0 && q_batch_stride % 8 == 0,
                  "pclip_attention_q_f16: strides / offsets must be multiples of 8 halves");
    if (B == 0) return PCLIP_OK;
    const int NT = ceil_div(L, 32), LP = NT * 32;
    const int LV = 0;                                       // (unused: V is kept row-major now)
    // mode 1 only: the persistent double-buffered kernel, one workgroup per CU (short sequences: as many as the LDS holds, at
    // most two: the register budget of the deep-prefetch loop).  Measured: 8 % faster than one workgroup per
    // item in isolation on N(0,1) data (348 vs 377 us, ViT-B/16 B = 1024), no faster inside the encoder (40.3 vs 40.2 ms per
    // step, same-box A/B) — the automatic mode does not select it.
    const long nitems = (long)B * H;
    const int cus = pclip_device_cus();
    const int KR = (L + 7) / 8 * 8;
    const size_t plds = (2 * (size_t)(KR + LP) + KR) * ATT_DH * 2;          // two K/V buffers + the query rows
    if (Lq == L && g_att_mode != 0 && NT <= 8 && plds <= 160 * 1024 && cus > 0 && g_att_mode == 1) {
        static DevOnce pipe_attr;
        if (!pipe_attr.done()) {
            if (hipFuncSetAttribute((const void*)attention_pipe_kernel<4, 2, PCLIP_ATT_VAR_SHORT>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess ||
                hipFuncSetAttribute((const void*)attention_pipe_kernel<8, 2, PCLIP_ATT_VAR_LONG>, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024) != hipSuccess) {
                pclip_set_error("pclip_attention_f16: cannot raise the dynamic LDS limit");
                return PCLIP_E_LAUNCH;
            }
            pipe_attr.set();
        }
        int per_cu = NT <= 4 ? (int)((160 * 1024) / plds) : 1;
        per_cu = per_cu < 1 ? 1 : (per_cu > 2 ? 2 : per_cu);
        long grid = (long)cus * per_cu;
        if (g_att_grid > 0 && g_att_grid < grid) grid = g_att_grid;
        if (grid > nitems) grid = nitems;
        if (NT <= 4)
            attention_pipe_kernel<4, 2, PCLIP_ATT_VAR_SHORT><<<(int)grid, 256, plds, (hipStream_t)stream>>>((const half_t*)q, ldq, q_batch_stride, (const half_t*)kv, ldkv,
                                                                                      k_off, v_off, (half_t*)out, L, H, causal, NT, KR, (int)nitems);
        else
            attention_pipe_kernel<8, 2, PCLIP_ATT_VAR_LONG><<<(int)grid, 512, plds, (hipStream_t)stream>>>((const half_t*)q, ldq, q_batch_stride, (const half_t*)kv, ldkv,
                                                                                      k_off, v_off, (half_t*)out, L, H, causal, NT, KR, (int)nitems);
        return pclip_check_launch("attention (pipelined)");
    }
    const size_t lds = 2 * (size_t)LP * ATT_DH * 2;
    // The softmax variant follows the SEQUENCE (more than four key tiles: the long form), not the kernel: the one-query form of the
    // last vision block (Lq = 1, four waves) must produce the bits of the full attention over the same keys.
    static DevOnce attr_set;
    if (!attr_set.done()) {
        const void* fns[] = {(const void*)attention_kernel<8, PCLIP_ATT_VAR_LONG>, (const void*)attention_kernel<4, PCLIP_ATT_VAR_LONG>,
                             (const void*)attention_kernel<4, PCLIP_ATT_VAR_SHORT>, (const void*)attention_kernel<8, PCLIP_ATT_VAR_LONG, true>,
                             (const void*)attention_kernel<4, PCLIP_ATT_VAR_SHORT, true>};
        for (const void* f : fns)
            if (hipFuncSetAttribute(f, hipFuncAttributeMaxDynamicSharedMemorySize, 80 * 1024) != hipSuccess) {
                pclip_set_error("pclip_attention_f16: cannot raise the dynamic LDS limit");
                return PCLIP_E_LAUNCH;
            }
        attr_set.set();
    }
    // more than four query tiles (ViT-B/16: 7, ViT-L/14: 9): eight waves, one tile each, two workgroups = four waves per SIMD
    // (VGPRs capped at 128); measured 438 -> 424 us (ViT-B/16), 224 -> 200 us (ViT-L/14), bit-identical.  Short sequences
    // (ViT-B/32: 2 tiles, text: 3) keep the four-wave workgroup, whose idle waves cost less.
#define PCLIP_ATT_LAUNCH(NW, VAR)                                                                                                         \
    attention_kernel<NW, VAR><<<B * H, NW * 64, lds, (hipStream_t)stream>>>((const half_t*)q, ldq, q_batch_stride, (const half_t*)kv, ldkv, k_off, \
                                                                             v_off, (half_t*)out, L, Lq, H, causal, NT, LV)
    static const bool qfirst = !(getenv("PCLIP_ATT_QFIRST") && getenv("PCLIP_ATT_QFIRST")[0] == '0');      // A/B switch
    // short NON-causal sequences (ViT-B/32: 50 tokens = 2 tiles): every wave of the four-wave workgroup has at most one tile too: 71.9 -> 68.0 us (B = 1024), same bits;
    // the causal text sequences (77 tokens) lose 14 % in this form (489 -> 558 us: their waves' work is triangular) and keep the looping kernel
    if (PCLIP_ATT_QF4 && qfirst && NT <= 4 && Lq == L && !causal)
        attention_kernel<4, PCLIP_ATT_VAR_SHORT, true><<<B * H, 4 * 64, lds, (hipStream_t)stream>>>((const half_t*)q, ldq, q_batch_stride, (const half_t*)kv, ldkv, k_off,
                                                                                                     v_off, (half_t*)out, L, Lq, H, causal, NT, LV);
    else if ((Lq + 31) / 32 > 4 && (Lq + 31) / 32 <= 8 && NT <= 8 && qfirst)      // NT <= 8: the kernel's counted waits assume at most four pieces per wave and operand
        attention_kernel<8, PCLIP_ATT_VAR_LONG, true><<<B * H, 8 * 64, lds, (hipStream_t)stream>>>((const half_t*)q, ldq, q_batch_stride, (const half_t*)kv, ldkv, k_off,
                                                                                                    v_off, (half_t*)out, L, Lq, H, causal, NT, LV);
    else if ((Lq + 31) / 32 > 4) PCLIP_ATT_LAUNCH(8, PCLIP_ATT_VAR_LONG);
    else if (NT > 4) PCLIP_ATT_LAUNCH(4, PCLIP_ATT_VAR_LONG);
    else PCLIP_ATT_LAUNCH(4, PCLIP_ATT_VAR_SHORT);
#undef PCLIP_ATT_LAUNCH
    return pclip_check_launch("attention");
}

extern "C" int pclip_attention_f16(const void* qkv, void* out, int B, int L, int H, int dh, int causal,
                                   pclip_stream_t stream) {
    PCLIP_REQUIRE(qkv && out, "pclip_attention_f16: null pointer");
    PCLIP_REQUIRE(H > 0 && L > 0, "pclip_attention_f16: bad H=%d L=%d", H, L);
    const int W = H * dh;
    return pclip_attention_q_f16(qkv, 3 * W, (long)L * 3 * W, qkv, 3 * W, W, 2 * W, out, B, L, L, H, dh, causal, stream);
}

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
                                      void* x0, void* h, float* stats, pclip_stream_t stream) {
    PCLIP_REQUIRE(patch_emb && class_emb && pos_emb && gamma_pre && beta_pre && x0 && (h || stats) && (!h || (gamma_1 && beta_1)),
                  "pclip_vit_embed_ln_f16: null pointer");
    PCLIP_REQUIRE(B >= 0 && G2 > 0 && W > 0 && W % 8 == 0 && W <= 4096, "pclip_vit_embed_ln_f16: bad shape B=%d G2=%d W=%d", B, G2, W);
    if (B == 0) return PCLIP_OK;
    const size_t R = (size_t)B * (G2 + 1);
    int grid = (int)((R + 3) / 4 > 16384 ? 16384 : (R + 3) / 4);
    const int ln_grid = pclip_device_cus() * PCLIP_LN_BPC;
    if (PCLIP_LN_LDS && W <= 1024 && ln_grid > 0 && R >= (size_t)16 * ln_grid) {     // whole batch: resident-size grid, affine vectors from LDS (<= 16 KB per workgroup)
        grid = ln_grid;
        if (W <= 512) vit_embed_ln_kernel<1, true><<<grid, 256, 0, (hipStream_t)stream>>>((const half_t*)patch_emb, (const half_t*)class_emb, (const half_t*)pos_emb, B, G2, W, gamma_pre,
                                                                                          beta_pre, gamma_1, beta_1, eps, (half_t*)x0, (half_t*)h, stats);
        else vit_embed_ln_kernel<2, true><<<grid, 256, 0, (hipStream_t)stream>>>((const half_t*)patch_emb, (const half_t*)class_emb, (const half_t*)pos_emb, B, G2, W, gamma_pre,
                                                                                 beta_pre, gamma_1, beta_1, eps, (half_t*)x0, (half_t*)h, stats);
        return pclip_check_launch("vit_embed_ln");
    }
#define PCLIP_VEL(NCH) vit_embed_ln_kernel<NCH><<<grid, 256, 0, (hipStream_t)stream>>>((const half_t*)patch_emb, (const half_t*)class_emb, \
        (const half_t*)pos_emb, B, G2, W, gamma_pre, beta_pre, gamma_1, beta_1, eps, (half_t*)x0, (half_t*)h, stats)
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
