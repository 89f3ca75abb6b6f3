// classify_small: the whole of P (utils.py:225-244) for N <= 32 classes in one launch — the body shared by classify_small_kernel (pclip_classify.hip) and the
// prototype-build + classification launch (pclip_proto_classify.hip).  ONE definition: the two entry points produce the same bits.
#pragma once
#include "pclip_common.h"

namespace {

// ---- small class counts (N <= 32: EuroSAT's 10 classes): the whole of P in ONE launch --------------------------
// The two-stage path above is four launches (two norm passes, the distance GEMM, the softmax pass) — at EuroSAT's size
// (8.35 MB of traffic, 1.3 us of HBM time) that is all launch latency.  Here a wave owns 16 queries: both banks sit in
// LDS (rows padded by 16 B so the 16 class rows of a fragment read land on different banks), the query rows go straight
// from HBM into the MFMA operand layout (lane = row l&15, k-chunk l>>4: 16-byte loads, every byte used once; the first
// 512 k of the wave's first group are requested BEFORE the banks are staged, so the two latencies overlap), the
// contraction is v_mfma_f32_16x16x32_f16 with the classes as the first operand, so a lane ends up with 4 consecutive
// classes (4*(l>>4)+e) of ONE query (l&15) per 16-class tile: the fp32 norms (accumulated from the very fragments the
// MFMAs consume), the cdist epilogue, both softmaxes, the alpha fusion and argmax / top-k are finished in registers with
// two xor-shuffles (16, 32) per reduction.  Workgroups are 2 waves when every group of 16 queries finds a free slot at
// once (latency-bound sizes) and 8 waves sharing one LDS copy of the banks otherwise.
// acc + sum of the squares of 8 halfs: v_dot2_f32_f16 (exact products, fp32 accumulate), 4 instructions
__device__ __forceinline__ float sq8(half8_t f, float acc) {
#pragma unroll
    for (int j = 0; j < 4; ++j) {
        const half2_t h = {f[2 * j], f[2 * j + 1]};
        acc = __builtin_amdgcn_fdot2(h, h, acc, false);
    }
    return acc;
}

// `block` of `nblocks` workgroups run this body.  FUSED (pclip_proto_classify.hip): `zi` is being written by the builder workgroups of the SAME launch —
// the textual bank and the first queries are requested first, then the workgroup waits for sync[0] == nbuilders (agent-scope acquire) and stages `zi`.
template <int NT, bool TWO, bool FUSED>
__device__ __forceinline__ void classify_small_body(char* smem, const int block, const int nblocks, const half_t* __restrict__ q, const half_t* zi,
                                                    const half_t* __restrict__ zt, int Q, int N, int D, float alpha,
                                                    float oma, float beta, float* __restrict__ p,
                                                    int32_t* __restrict__ argmax, float* __restrict__ topk_p,
                                                    int32_t* __restrict__ topk_i, int k, int* sync = nullptr, int nbuilders = 0, int wt = 0) {
    constexpr int NB = TWO ? 2 : 1, ROWS = NB * NT * 16;
    const int tid = threadIdx.x, nthreads = blockDim.x, nwaves = nthreads >> 6;
    const int lane = tid & 63, wave = tid >> 6, qr = lane & 15, kg = lane >> 4;
    const int units = D >> 3, row_bytes = D * 2 + 16, steps = D >> 5;
    const int ngroups = (Q + 15) >> 4;
    int g = block * nwaves + wave;
    half8_t qf[16];
    bool have = false;
    if (g < ngroups) {
        const int m = g * 16 + qr;
        const half_t* qrow = q + (size_t)(m < Q ? m : Q - 1) * D + kg * 8;
        if (steps >= 16) {
#pragma unroll
            for (int s = 0; s < 16; ++s) qf[s] = ld_half8(qrow + s * 32);
        } else {
#pragma unroll
            for (int s = 0; s < 16; ++s)
                if (s < steps) qf[s] = ld_half8(qrow + s * 32);
        }
        have = true;
    }
#if defined(__HIP_DEVICE_COMPILE__)
    const __amdgpu_buffer_rsrc_t zi_rs = __builtin_amdgcn_make_buffer_rsrc((void*)zi, 0, N * D * 2, 0x00020000);
#endif
    // banks -> LDS, eight 16-byte units per thread in flight; rows of classes >= N are zero
    auto stage_banks = [&](const int row_lo, const int row_hi) {
        const int total = row_hi * units;
        const float inv_units = 1.f / (float)units;
        for (int i0 = row_lo * units + tid; i0 < total; i0 += 8 * nthreads) {
            half8_t v[8];
            int off[8];
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int i = i0 + j * nthreads;
                int r = (int)(((float)i + 0.5f) * inv_units);                       // i / units (i < 2^20: exact after the fix-up)
                if (r * units > i) --r;
                if ((r + 1) * units <= i) ++r;
                const int u = i - r * units, c = r & (NT * 16 - 1) , bk = r / (NT * 16);
                static_assert((NT & (NT - 1)) == 0, "NT must be a power of two");
                v[j] = half8_t{};
                off[j] = i < total ? r * row_bytes + u * 16 : -1;
                if (i < total && c < N) {
                    if (FUSED && wt && !bk) {                                          // agent-coherent load (sc1): the row was written by another workgroup of this launch
#if defined(__HIP_DEVICE_COMPILE__)
                        v[j] = __builtin_bit_cast(half8_t, __builtin_amdgcn_raw_buffer_load_b128(zi_rs, (c * D + u * 8) * 2, 0, 16));
#endif
                    } else {
                        v[j] = ld_half8((bk ? zt : zi) + (size_t)c * D + u * 8);
                    }
                }
            }
#pragma unroll
            for (int j = 0; j < 8; ++j)
                if (off[j] >= 0) *reinterpret_cast<half8_t*>(smem + off[j]) = v[j];
        }
    };
    if (FUSED) {
        if (TWO) stage_banks(NT * 16, ROWS);                                          // the textual bank does not wait for anybody
        if (tid == 0) {
            int spins = 0;
            while (__hip_atomic_load(sync, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < nbuilders && ++spins < (1 << 22)) __builtin_amdgcn_s_sleep(1);
        }
        __syncthreads();
        if (!wt) __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");                   // this wave's later loads of zi see the builders' rows
        stage_banks(0, NT * 16);
    } else {
        stage_banks(0, ROWS);
    }
    __syncthreads();
    // every consumer counts itself once it is past the wait (the answer is looked at when the work is done: the round trip runs under the MFMAs); the last one
    // zeroes both words for the next launch (ordered behind this one on the stream)
    int ticket = -1;
    if (FUSED && tid == 0) ticket = __hip_atomic_fetch_add(sync + 1, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    const int cls0 = 4 * kg;                                                          // first class of this lane inside a tile
    for (; g < ngroups; g += nblocks * nwaves) {
        const int m = g * 16 + qr;
        const bool mv = m < Q;
        const half_t* qrow = q + (size_t)(mv ? m : Q - 1) * D + kg * 8;             // clamped: the load stays in bounds, the row is dropped
        float4_t acc[NB][NT];
        float zn[NB][NT];
#pragma unroll
        for (int b = 0; b < NB; ++b)
#pragma unroll
            for (int t = 0; t < NT; ++t) { acc[b][t] = float4_t{0.f, 0.f, 0.f, 0.f}; zn[b][t] = 0.f; }
        float qs = 0.f;
        for (int s0 = 0; s0 < steps; s0 += 16) {                                    // 512 k per pass: 16 query loads in flight
            if (steps - s0 >= 16) {
                // a full pass, free of per-step branches: the LDS fragment reads of four steps are issued together (a branch per step kept every read -> MFMA pair a
                // serial LDS round trip: 1.8 us of the 7.4 us EuroSAT launch; profiles/r05_c2_phases.txt)
                if (!have) {
#pragma unroll
                    for (int s = 0; s < 16; ++s) qf[s] = ld_half8(qrow + (s0 + s) * 32);
                }
                have = false;
#pragma unroll
                for (int s4 = 0; s4 < 16; s4 += 4) {
                    half8_t zf[4][NB][NT];
#pragma unroll
                    for (int u = 0; u < 4; ++u)
#pragma unroll
                        for (int b = 0; b < NB; ++b)
#pragma unroll
                            for (int t = 0; t < NT; ++t)
                                zf[u][b][t] = *reinterpret_cast<const half8_t*>(smem + ((b * NT + t) * 16 + qr) * row_bytes + (s0 + s4 + u) * 64 + kg * 16);
#pragma unroll
                    for (int u = 0; u < 4; ++u) {
                        qs = sq8(qf[s4 + u], qs);
#pragma unroll
                        for (int b = 0; b < NB; ++b)
#pragma unroll
                            for (int t = 0; t < NT; ++t) {
                                zn[b][t] = sq8(zf[u][b][t], zn[b][t]);
                                acc[b][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zf[u][b][t], qf[s4 + u], acc[b][t], 0, 0, 0);
                            }
                    }
                }
                continue;
            }
            if (!have) {
#pragma unroll
                for (int s = 0; s < 16; ++s)
                    if (s0 + s < steps) qf[s] = ld_half8(qrow + (s0 + s) * 32);
            }
            have = false;
#pragma unroll
            for (int s = 0; s < 16; ++s)
                if (s0 + s < steps) {
                    qs = sq8(qf[s], qs);
#pragma unroll
                    for (int b = 0; b < NB; ++b)
#pragma unroll
                        for (int t = 0; t < NT; ++t) {
                            const half8_t zf = *reinterpret_cast<const half8_t*>(
                                smem + ((b * NT + t) * 16 + qr) * row_bytes + (s0 + s) * 64 + kg * 16);
                            zn[b][t] = sq8(zf, zn[b][t]);
                            acc[b][t] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zf, qf[s], acc[b][t], 0, 0, 0);
                        }
                }
        }
        qs += lane_xor<16>(qs);
        qs += lane_xor<32>(qs);
        // cdist epilogue + softmax over the classes of this query (utils.py:225-244), as in sqdist_kernel / fuse_probs_kernel
        float pr[NT][4];
#pragma unroll
        for (int b = 0; b < NB; ++b) {
            float d2[NT][4], mn = __builtin_inff(), mx = -__builtin_inff();
#pragma unroll
            for (int t = 0; t < NT; ++t) {
                float zsq = zn[b][t];                                                 // ||z_c||^2 of class t*16 + qr ...
                zsq += lane_xor<16>(zsq);
                zsq += lane_xor<32>(zsq);
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float zs = __shfl(zsq, cls0 + e, WAVE);                     // ... moved to the accumulator layout
                    const float v = __fadd_rn(__fadd_rn(-2.f * acc[b][t][e], qs), zs);
                    const float d = sqrtf(fmaxf(v, 0.f));
                    d2[t][e] = __fmul_rn(d, d);
                    if (t * 16 + cls0 + e < N) { mn = fminf(mn, d2[t][e]); mx = fmaxf(mx, d2[t][e]); }
                }
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
            const float w = b ? oma : alpha;
#pragma unroll
            for (int t = 0; t < NT; ++t)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float term = __fmul_rn(w, __fdiv_rn(d2[t][e], sum));
                    pr[t][e] = b ? __fadd_rn(pr[t][e], term) : term;
                }
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
    if (FUSED && tid == 0 && ticket == nblocks - 1) {
        __hip_atomic_store(sync + 1, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __hip_atomic_store(sync, 0, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
    }
}

}  // namespace
