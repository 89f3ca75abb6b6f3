// Device functions of the prototype reduction (pclip_proto.hip: pclip_proto_build_f16 / partial sums; load_row_sq also serves pclip_classify_panel.hip)
#pragma once
#include "pclip_common.h"

namespace {

// A wave covers a row of D halves as NCH chunks of 512 (lane*8 .. lane*8+7 inside each chunk).
template <int NCH>
struct RowRegs {
    half8_t v[NCH];
};

template <int NCH>
__device__ __forceinline__ void load_row(const half_t* xr, int D, int lane, RowRegs<NCH>& r) {
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        int d = c * 512 + lane * 8;
        if (d < D) {
            r.v[c] = ld_half8(xr + d);
        } else {
#pragma unroll
            for (int j = 0; j < 8; ++j) r.v[c][j] = (half_t)0.f;
        }
    }
}

template <int NCH>
__device__ __forceinline__ float row_sq(const RowRegs<NCH>& r) {
    float ss = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = (float)r.v[c][j];
            ss += f * f;
        }
    return wave_sum(ss);
}

template <int NCH>
__device__ __forceinline__ float load_row_sq(const half_t* xr, int D, int lane, RowRegs<NCH>& r) {
    load_row<NCH>(xr, D, lane, r);
    return row_sq<NCH>(r);
}

// ---- shared tail: fp32 class sum -> z=r16(sum/cnt) -> fp16 / fp32 normalised prototype ----------
// Executed by ONE wave; acc[c][j] holds this lane's slice of the class sum.
template <int NCH>
__device__ __forceinline__ void finish_prototype(float (&acc)[NCH][8], float inv_or_cnt, int n, int D, int lane,
                                                 half_t* proto_f16, float* proto_f32, float* proto_sq) {
    float z[NCH][8];
    float ss = 0.f;
    const RowDiv dcnt(inv_or_cnt);
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            z[c][j] = r16(dcnt.fast ? dcnt.div_fast(acc[c][j]) : acc[c][j] / inv_or_cnt);      // mean over shots, rounded once (torch fp16 mean)
            ss += z[c][j] * z[c][j];
        }
    ss = wave_sum(ss);
    const float n16 = r16(sqrtf(ss));
    const float n32 = sqrtf(ss);
    const RowDiv d16(n16), d32(n32);
    float ss2 = 0.f;
#pragma unroll
    for (int c = 0; c < NCH; ++c) {
        int d = c * 512 + lane * 8;
        if (d < D) {
            if (proto_f16 || proto_sq) {
                half8_t o;
                if (d16.fast) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (half_t)d16.div_fast(z[c][j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (half_t)(z[c][j] / n16);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float f = (float)o[j];
                    ss2 += f * f;
                }
                if (proto_f16) st_half8(proto_f16 + (size_t)n * D + d, o);
            }
            if (proto_f32) {
                float4_t o0, o1;
#pragma unroll
                for (int j = 0; j < 4; ++j) {
                    o0[j] = d32.fast ? d32.div_fast(z[c][j]) : z[c][j] / n32;
                    o1[j] = d32.fast ? d32.div_fast(z[c][j + 4]) : z[c][j + 4] / n32;
                }
                *reinterpret_cast<float4_t*>(proto_f32 + (size_t)n * D + d) = o0;
                *reinterpret_cast<float4_t*>(proto_f32 + (size_t)n * D + d + 4) = o1;
            }
        }
    }
    if (proto_sq) {
        ss2 = wave_sum(ss2);
        if (lane == 0) proto_sq[n] = ss2;
    }
}

// Accumulate rows [lo, hi) of `mem` (optionally per-shot normalised) into per-lane fp32 sums; the four
// waves of the workgroup interleave rows and combine through LDS in wave order (deterministic).
template <int NCH>
__device__ __forceinline__ void class_sum(const half_t* __restrict__ mem, int lo, int hi, int D, int per_shot_norm,
                                          float (&acc)[NCH][8], float* red /* LDS [4][NCH*512] */) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[c][j] = 0.f;
    constexpr int PF = NCH <= 2 ? 4 : 2;                 // rows of one wave in flight: the loop is a latency chain otherwise
    for (int row0 = lo + wave; row0 < hi; row0 += 4 * PF) {
        RowRegs<NCH> rr[PF];
        if (row0 + 4 * (PF - 1) < hi) {
            // all PF rows exist: no branch between them, so their norm chains (butterfly, sqrt) run interleaved instead of one after the other (four rows of one
            // wave took 2.3 us of the 5 us EuroSAT launch; profiles/r05_c2_phases.txt); the accumulation order is the same: rows in index order
#pragma unroll
            for (int u = 0; u < PF; ++u) load_row<NCH>(mem + (size_t)(row0 + 4 * u) * D, D, lane, rr[u]);
            float nn[PF];
#pragma unroll
            for (int u = 0; u < PF; ++u) nn[u] = per_shot_norm ? r16(sqrtf(row_sq<NCH>(rr[u]))) : 1.f;

            bool allfast = true;
#pragma unroll
            for (int u = 0; u < PF; ++u) allfast = allfast && RowDiv(nn[u]).fast;
            if (!per_shot_norm) {
#pragma unroll
                for (int u = 0; u < PF; ++u)
#pragma unroll
                    for (int c = 0; c < NCH; ++c)
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[c][j] += (float)rr[u].v[c][j];
            } else if (allfast) {                                   // ONE wave-uniform branch for the PF rows: nothing between their division chains
#pragma unroll
                for (int u = 0; u < PF; ++u) {
                    const RowDiv dv(nn[u]);
#pragma unroll
                    for (int c = 0; c < NCH; ++c)
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[c][j] += r16(dv.div_fast((float)rr[u].v[c][j]));
                }
            } else {
#pragma unroll
                for (int u = 0; u < PF; ++u)
#pragma unroll
                    for (int c = 0; c < NCH; ++c)
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[c][j] += r16((float)rr[u].v[c][j] / nn[u]);
            }
            continue;
        }
#pragma unroll
        for (int u = 0; u < PF; ++u)
            if (row0 + 4 * u < hi) load_row<NCH>(mem + (size_t)(row0 + 4 * u) * D, D, lane, rr[u]);
#pragma unroll
        for (int u = 0; u < PF; ++u) {
            if (row0 + 4 * u >= hi) break;
            const RowRegs<NCH>& r = rr[u];
            if (per_shot_norm) {
                const float n = r16(sqrtf(row_sq<NCH>(r)));
                const RowDiv dn(n);
                if (dn.fast) {
#pragma unroll
                    for (int c = 0; c < NCH; ++c)
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[c][j] += r16(dn.div_fast((float)r.v[c][j]));
                } else {
#pragma unroll
                    for (int c = 0; c < NCH; ++c)
#pragma unroll
                        for (int j = 0; j < 8; ++j) acc[c][j] += r16((float)r.v[c][j] / n);
                }
            } else {
#pragma unroll
                for (int c = 0; c < NCH; ++c)
#pragma unroll
                    for (int j = 0; j < 8; ++j) acc[c][j] += (float)r.v[c][j];
            }
        }
    }
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) red[wave * (NCH * 512) + c * 512 + lane * 8 + j] = acc[c][j];
    __syncthreads();
    if (wave == 0) {
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                float s = red[c * 512 + lane * 8 + j];
                s += red[1 * (NCH * 512) + c * 512 + lane * 8 + j];
                s += red[2 * (NCH * 512) + c * 512 + lane * 8 + j];
                s += red[3 * (NCH * 512) + c * 512 + lane * 8 + j];
                acc[c][j] = s;
            }
    }
}

}  // namespace
