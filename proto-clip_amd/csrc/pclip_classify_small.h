// classify_small: the whole of P (utils.py:225-244) for N <= 32 classes in one launch — the body of classify_small_kernel (pclip_classify.hip); sq8 is shared with the
// mid-N kernel (pclip_classify_mid.hip).
#pragma once
#include "pclip_common.h"

namespace {

// ---- small class counts (N <= 32: EuroSAT's 10 classes): the whole of P in ONE launch --------------------------
// The two-stage path is four launches (two norm passes, the distance GEMM, the softmax pass) — at EuroSAT's size (8.35 MB of traffic, 1.3 us of HBM time) that is
// all launch latency.  Here a group of 16 queries belongs to a PAIR of waves, one per bank (visual / textual): a wave's run time at this size is the issue of its own
// instruction stream (one wave per SIMD: profiles/r05_c2_phases.txt), so the two softmaxes run side by side on two SIMDs instead of one after the other.  Every
// operand goes straight from memory into the MFMA layout — query rows (lane = row l&15, k-chunk l>>4: 16-byte loads, every byte used once) and the wave's bank
// (lane = class l&15, same chunks; the 10 - 64 KB of a bank are L2 hits for every wave after the first) — no LDS staging, no barrier before the first MFMA.  The
// contraction is v_mfma_f32_16x16x32_f16 with the classes as the first operand, so a lane ends up with 4 consecutive classes (4*(l>>4)+e) of ONE query (l&15) per
// 16-class tile: the fp32 norms (accumulated from the very fragments the MFMAs consume), the cdist epilogue, the bank's softmax and its alpha / (1 - alpha) weight are
// finished in registers with VALU butterflies (lane_xor<16 / 32>).  The textual wave hands its 4 NT terms per lane to the visual wave through LDS (one workgroup
// barrier per group, buffers alternate), which adds them (visual + textual, the reference's order), writes p and finishes argmax / top-k.
// acc + sum of the squares of 8 halfs: v_dot2_f32_f16 (exact products, fp32 accumulate), 4 instructions
__device__ __forceinline__ float sq8(half8_t f, float acc) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const half2_t h = {f[2 * j], f[2 * j + 1]};
        acc = __builtin_amdgcn_fdot2(h, h, acc, false);
    }
    return acc;
}

// LDS bytes of a workgroup of `nwaves` waves: two exchange buffers of NT * 4 floats per lane and pair
__host__ __device__ constexpr int classify_small_lds(int nt, int nwaves) { return 2 * (nwaves / 2) * nt * 4 * 64 * 4; }

// `block` of `nblocks` workgroups run this body.
template <int NT, bool TWO>
__device__ __forceinline__ void classify_small_body(char* smem, const int block, const int nblocks, const half_t* __restrict__ q, const half_t* zi,
                                                    const half_t* __restrict__ zt, int Q, int N, int D, float alpha,
                                                    float oma, float beta, float* __restrict__ p,
                                                    int32_t* __restrict__ argmax, float* __restrict__ topk_p,
                                                    int32_t* __restrict__ topk_i, int k) {
    const int tid = threadIdx.x, nwaves = blockDim.x >> 6;
    const int lane = tid & 63, wave = tid >> 6, qr = lane & 15, kg = lane >> 4;
    const int steps = D >> 5, ngroups = (Q + 15) >> 4;
    const int gpw = TWO ? nwaves >> 1 : nwaves;                                       // groups a workgroup works on at a time
    const int slot = TWO ? wave >> 1 : wave, bank = TWO ? wave & 1 : 0;
    const half_t* z = bank ? zt : zi;
    const bool single = steps <= 16;                                                  // D <= 512: the bank fragments stay in registers for every group of the wave
    int g = block * gpw + slot;
    half8_t qf[16], zf[NT][16];
    float zn[NT];
    auto load_q = [&](int grp, int s0) {
        const int m = grp * 16 + qr;
        const half_t* qrow = q + (size_t)(m < Q ? m : Q - 1) * D + kg * 8;            // clamped: the load stays in bounds, the row is dropped
        if (steps - s0 >= 16) {
#pragma unroll
            for (int s = 0; s < 16; ++s) qf[s] = ld_half8(qrow + (s0 + s) * 32);
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s)
                if (s0 + s < steps) qf[s] = ld_half8(qrow + (s0 + s) * 32);
        }
    };
    auto load_z1 = [&](int c, int kk) -> half8_t { return ld_half8(z + (size_t)c * D + kk); };     // 8 halves of class row c at k = kk
    auto load_z = [&](int s0) {                                                       // rows of classes >= N are zero
#pragma unroll
        for (int t = 0; t < NT; ++t) {
            const int c = t * 16 + qr;
#pragma unroll
            for (int s = 0; s < 16; ++s) zf[t][s] = half8_t{};
            if (c < N) {
                if (steps - s0 >= 16) {
#pragma unroll
                    for (int s = 0; s < 16; ++s) zf[t][s] = load_z1(c, (s0 + s) * 32 + kg * 8);
                } else {
#pragma unroll
                    for (int s = 0; s < 16; ++s)
                        if (s0 + s < steps) zf[t][s] = load_z1(c, (s0 + s) * 32 + kg * 8);
                }
            }
        }
    };
    const bool mine = g < ngroups;
    if (mine) load_q(g, 0);
    if (single && mine) load_z(0);
    const int cls0 = 4 * kg;                                                          // first class of this lane inside a tile
    float* xch = reinterpret_cast<float*>(smem);
    int parity = 0;
    bool have = true, have_zn = false;
    for (int g0 = block * gpw; g0 < ngroups; g0 += nblocks * gpw, parity ^= 1) {      // the same trip count for every wave of the workgroup: one barrier per trip
        g = g0 + slot;
        const bool act = g < ngroups;
        const int m = g * 16 + qr;
        const bool mv = act && m < Q;
        float4_t acc[NT];
        float qs = 0.f;
#pragma unroll
        for (int t = 0; t < NT; ++t) acc[t] = float4_t{0.f, 0.f, 0.f, 0.f};
        float term[NT][4];
        if (act) {
            float znp[NT];
#pragma unroll
            for (int t = 0; t < NT; ++t) znp[t] = 0.f;
            for (int s0 = 0; s0 < steps; s0 += 16) {                                  // 512 k per pass: 16 query (+ 16 NT bank) loads in flight
                if (!have) load_q(g, s0);
                if (!single) load_z(s0);
                have = false;
                if (!have_zn) {                                                       // class norms: one v_dot2 chain per tile over the fragments, k ascending
#pragma unroll
                    for (int t = 0; t < NT; ++t)
#pragma unroll
                        for (int s = 0; s < 16; ++s)
                            if (s0 + s < steps) znp[t] = sq8(zf[t][s], znp[t]);
                }
                if (steps - s0 >= 16) {                                               // a full pass, free of per-step branches
#pragma unroll
                    for (int s = 0; s < 16; ++s) {
                        qs = sq8(qf[s], qs);
#pragma unroll
                        for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zf[t][s], qf[s], acc[t], 0, 0, 0);
                    }
                } else {
#pragma unroll
                    for (int s = 0; s < 16; ++s)
                        if (s0 + s < steps) {
                            qs = sq8(qf[s], qs);
#pragma unroll
                            for (int t = 0; t < NT; ++t) acc[t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zf[t][s], qf[s], acc[t], 0, 0, 0);
                        }
                }
            }
            if (!have_zn) {
#pragma unroll
                for (int t = 0; t < NT; ++t) {
                    float zsq = znp[t];                                               // ||z_c||^2 of class t*16 + qr ...
                    zsq += lane_xor<16>(zsq);
                    zsq += lane_xor<32>(zsq);
                    zn[t] = zsq;
                }
                have_zn = single;                                                     // the same fragments, the same chain: computed once per wave
            }
            qs += lane_xor<16>(qs);
            qs += lane_xor<32>(qs);
            // cdist epilogue + softmax over the classes of this query (utils.py:225-244), as in sqdist_kernel / fuse_probs_kernel
            float d2[NT][4], mn = __builtin_inff(), mx = -__builtin_inff();
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float zs = __shfl(zn[t], cls0 + e, WAVE);                   // ... moved to the accumulator layout
                    const float v = __fadd_rn(__fadd_rn(-2.f * acc[t][e], qs), zs);
                    const float d = sqrtf(fmaxf(v, 0.f));
                    d2[t][e] = __fmul_rn(d, d);
                    if (t * 16 + cls0 + e < N) { mn = fminf(mn, d2[t][e]); mx = fmaxf(mx, d2[t][e]); }
                }
            mn = fminf(mn, lane_xor<16>(mn)); mn = fminf(mn, lane_xor<32>(mn));
            mx = fmaxf(mx, lane_xor<16>(mx)); mx = fmaxf(mx, lane_xor<32>(mx));
            const float top = __fmul_rn(beta, beta >= 0.f ? -mn : -mx);
            float sum = 0.f;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    d2[t][e] = (t * 16 + cls0 + e < N) ? expf(__fsub_rn(__fmul_rn(beta, -d2[t][e]), top)) : 0.f;
                    sum += d2[t][e];
                }
            sum += lane_xor<16>(sum);
            sum += lane_xor<32>(sum);
            const float w = bank ? oma : alpha;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) term[t][e] = __fmul_rn(w, __fdiv_rn(d2[t][e], sum));
            // the next group's queries (single-pass shapes: requested before the hand-over so that they fly under it)
            if (g + nblocks * gpw < ngroups) { load_q(g + nblocks * gpw, 0); have = true; }
        }
        float pr[NT][4];
        if (TWO) {
            float* buf = xch + ((parity * gpw + slot) * NT * 4) * 64;
            if (bank && act) {
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e) buf[(t * 4 + e) * 64 + lane] = term[t][e];
            }
            __syncthreads();
            if (bank || !act) continue;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) pr[t][e] = __fadd_rn(term[t][e], buf[(t * 4 + e) * 64 + lane]);
        } else {
            if (!act) continue;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) pr[t][e] = term[t][e];
        }
        float best = -1.f;
        int besti = 0x7fffffff;
#pragma unroll
        for (int t = 0; t < NT; ++t)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const int c = t * 16 + cls0 + e;
                if (c < N) {
                    if (p && mv) p[(size_t)m * N + c] = pr[t][e];
                    if (pr[t][e] > best) { best = pr[t][e]; besti = c; }             // ascending c: first max kept
                } else {
                    pr[t][e] = -1.f;
                }
            }
        auto quad_argmax = [&](float& v, int& i) {
            { const float ov = lane_xor<16>(v); const int oi = lane_xor_i<16>(i); if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; } }
            { const float ov = lane_xor<32>(v); const int oi = lane_xor_i<32>(i); if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; } }
        };
        if (argmax) {
            quad_argmax(best, besti);
            if (kg == 0 && mv) argmax[m] = besti;
        }
        if (topk_p || topk_i) {
            for (int r = 0; r < k; ++r) {
                float bv = -2.f;
                int bi = 0x7fffffff;
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (pr[t][e] > bv) { bv = pr[t][e]; bi = t * 16 + cls0 + e; }
                quad_argmax(bv, bi);
                if (kg == 0 && mv) {
                    if (topk_p) topk_p[(size_t)m * k + r] = bv;
                    if (topk_i) topk_i[(size_t)m * k + r] = bi;
                }
#pragma unroll
                for (int t = 0; t < NT; ++t)
#pragma unroll
                    for (int e = 0; e < 4; ++e)
                        if (t * 16 + cls0 + e == bi) pr[t][e] = -2.f;                 // remove the winner
            }
        }
    }
}

}  // namespace
