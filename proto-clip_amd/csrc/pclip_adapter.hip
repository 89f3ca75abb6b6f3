// Query adapters: model.py:12-95.
//  - Adapter_FC: two skinny MFMA GEMMs + two LayerNorms, the 0.2/0.8 blend and the following row
//    normalise fused into the second LayerNorm pass.
//  - Adapter (conv-2x / conv-3x): one workgroup per feature vector; the whole [16, s, s] activation
//    stack lives in LDS (never in HBM — the reference materialises 5 such tensors per row), the 3x3
//    convolution runs on packed-fp16 dot products with fp32 accumulation, and the three whole-tensor
//    LayerNorms ([C,s,s]-shaped affine, model.py:37-45) are block reductions.
#include "pclip_common.h"

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

}  // namespace

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
    if (three_x) {
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
