// Attention of the CLIP towers (clip/model.py:183-185): one workgroup per (image, head), K / V resident in LDS, online softmax in registers.
#include "pclip_encoder_common.h"
#include <stdlib.h>
#include <type_traits>

namespace {
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

}  // namespace

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