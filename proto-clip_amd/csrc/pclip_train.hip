// Episodic training step: the backward of the path
//   banks -> prototypes -> P -> NLL (+ InfoNCE alignment losses) and the fp16 AdamW update,
// reference main.py:260-310, utils.py:80-109, 225-244.  The reference leaves all of it to autograd on eager
// tensors; here the graph is static, so every stage is one explicit kernel and no tape is kept:
//   nll_grad          dL/d(d2_img), dL/d(d2_txt) of  L = mean_q -log p[q, y_q]  straight from the two distance rows
//   gemm_f32          strided fp32 MFMA GEMM: dq = -2 G.Z, dz = -2 G^T.q, InfoNCE logits and their gradients,
//                     weight gradients of the fc adapter
//   proto_backward    fp32 normalise <- .float() <- mean over shots <- fp16 per-shot normalise, one wave per class
//   layernorm_backward / softmax_ce_rows / colsum / adamw
// Sizes are small (an episode is <= 0.4 N classes x K queries), so the kernels favour exact restatement of the
// reference's rounding points over peak rate; each fp16 tensor of the reference's autograd is rounded where it is
// materialised there.
#include "pclip_common.h"

namespace {

// fp32 -> fp16 -> fp32 with the fp32 value pinned in a register first: hipcc otherwise folds an explicit fmaf + convert into
// v_fma_mixlo_f16, which rounds the exact fma ONCE to fp16, while torch materialises the fp32 result and then converts.
__device__ __forceinline__ float r16s(float x) {
    asm volatile("" : "+v"(x));
    return (float)(half_t)x;
}

inline int row_grid(int R, int cap) { int g = ceil_div(R, 4); return g < 1 ? 1 : (g > cap ? cap : g); }
inline int flat_grid(size_t n) { size_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 16384 ? 16384 : g)); }

__global__ __launch_bounds__(256) void cast_f16_f32_kernel(const half_t* __restrict__ x, float* __restrict__ y, size_t n) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) y[i] = (float)x[i];
}

// ---- strided fp32 GEMM: C = alpha * op(A) op(B) + beta * C -----------------------------------------------------
// A(m,k) = A[m*rsa + k*csa], B(k,n) = B[k*rsb + n*csb]; 64x64 tile, one 32x32 v_mfma_f32_32x32x2_f32 accumulator per
// wave (an fmaf chain in k order: the same arithmetic as pclip_sqdist_f32), K staged 32 at a time through LDS.  The
// staging loop walks whichever index is contiguous in memory fastest, so both transposes read coalesced.
template <typename TA, typename TB>
__global__ __launch_bounds__(256) void gemm_f32_kernel(const TA* __restrict__ A, long rsa, long csa, const TB* __restrict__ B,
                                                       long rsb, long csb, float* __restrict__ C, int ldc, int M, int N,
                                                       int K, float alpha, float beta, int kper, float* __restrict__ slabs) {
    __shared__ float As[64][33], Bs[64][33];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int wr = wave >> 1, wc = wave & 1, hi = lane >> 5, l31 = lane & 31;
    float16_t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    const bool a_k_fast = csa == 1, b_k_fast = rsb == 1;
    // register double buffering: the global loads of K-slab k0+32 fly while slab k0 is multiplied (the staging used to be a
    // load -> LDS -> barrier chain per slab: ~2 us of exposed latency each)
    // split-K (slabs != nullptr): blockIdx.z owns k in [z*kper, (z+1)*kper) and writes its raw sums to slabs[z][M][N]
    const int kbeg = slabs ? blockIdx.z * kper : 0;
    const int kend = slabs ? (kbeg + kper < K ? kbeg + kper : K) : K;
    float ra[8], rb[8];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + u * 256;
            int r, c;
            if (a_k_fast) { r = i >> 5; c = i & 31; } else { r = i & 63; c = i >> 6; }
            ra[u] = (m0 + r < M && k0 + c < kend) ? (float)A[(long)(m0 + r) * rsa + (long)(k0 + c) * csa] : 0.f;
            if (b_k_fast) { r = i >> 5; c = i & 31; } else { r = i & 63; c = i >> 6; }
            rb[u] = (n0 + r < N && k0 + c < kend) ? (float)B[(long)(k0 + c) * rsb + (long)(n0 + r) * csb] : 0.f;
        }
    };
    fetch(kbeg);
    for (int k0 = kbeg; k0 < kend; k0 += 32) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + u * 256;
            int r, c;
            if (a_k_fast) { r = i >> 5; c = i & 31; } else { r = i & 63; c = i >> 6; }
            As[r][c] = ra[u];
            if (b_k_fast) { r = i >> 5; c = i & 31; } else { r = i & 63; c = i >> 6; }
            Bs[r][c] = rb[u];
        }
        __syncthreads();
        if (k0 + 32 < kend) fetch(k0 + 32);
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2) {
            const float b = Bs[wc * 32 + l31][kk + hi], a = As[wr * 32 + l31][kk + hi];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc, 0, 0, 0);     // swapped: lane owns row m = lane & 31
        }
    }
    const int m = m0 + wr * 32 + l31;
    if (m >= M) return;
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int n = n0 + wc * 32 + 8 * (e >> 2) + 4 * hi + (e & 3);
        if (n < N) {
            if (slabs) { slabs[((size_t)blockIdx.z * M + m) * N + n] = acc[e]; continue; }
            float* c = C + (size_t)m * ldc + n;
            *c = beta == 0.f ? alpha * acc[e] : fmaf(beta, *c, alpha * acc[e]);
        }
    }
}

// C = alpha * sum_z slabs[z] + beta * C  (slices added in order: deterministic)
__global__ __launch_bounds__(256) void gemm_f32_reduce_kernel(const float* __restrict__ slabs, int S, int M, int N, float* __restrict__ C,
                                                              int ldc, float alpha, float beta) {
    const size_t total = (size_t)M * N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        float acc = 0.f;
        for (int z = 0; z < S; ++z) acc += slabs[(size_t)z * total + i];
        const int m = (int)(i / N), n = (int)(i - (size_t)m * N);
        float* c = C + (size_t)m * ldc + n;
        *c = beta == 0.f ? alpha * acc : fmaf(beta, *c, alpha * acc);
    }
}

// out[c] (+)= scale * sum_r x[r, c]  (rows summed in a fixed order: deterministic)
__global__ __launch_bounds__(256) void colsum_kernel(const float* __restrict__ x, int ldx, int R, int C, float scale,
                                                     float* __restrict__ out, int accumulate) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    float s = 0.f;
    if (c < C)
        for (int r = wave; r < R; r += 4) s += x[(size_t)r * ldx + c];
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < C) {
        const float t = scale * (((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane]);
        out[c] = accumulate ? out[c] + t : t;
    }
}

// part[blockIdx.y][c] = sum of the rows [blockIdx.y * rows_per, ...) of column c (same wave-interleaved fixed order as colsum_kernel)
__global__ __launch_bounds__(256) void colsum_part_kernel(const float* __restrict__ x, int ldx, int R, int C, int rows_per,
                                                          float* __restrict__ part) {
    __shared__ float red[4][64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int c = blockIdx.x * 64 + lane;
    const int r0 = blockIdx.y * rows_per, r1 = r0 + rows_per < R ? r0 + rows_per : R;
    float s = 0.f;
    if (c < C)
        for (int r = r0 + wave; r < r1; r += 4) s += x[(size_t)r * ldx + c];
    red[wave][lane] = s;
    __syncthreads();
    if (wave == 0 && c < C) part[(size_t)blockIdx.y * C + c] = ((red[0][lane] + red[1][lane]) + red[2][lane]) + red[3][lane];
}

// C[r, :] += s * rowscale[r] * X[r, :]
__global__ __launch_bounds__(256) void addscaled_rows_kernel(float* __restrict__ C, int ldc, const float* __restrict__ X, int ldx,
                                                             const float* __restrict__ rowscale, float s, int R, int D) {
    const size_t total = (size_t)R * D;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int r = (int)(i / D), d = (int)(i - (size_t)r * D);
        C[(size_t)r * ldc + d] = fmaf(s * rowscale[r], X[(size_t)r * ldx + d], C[(size_t)r * ldc + d]);
    }
}

// ---- NLL(log P) forward statistics + gradient wrt both distance rows, one wave per query -----------------------
// p = alpha*softmax(-beta d_i) + oma*softmax(-beta d_t) (utils.py:225-244), L = -(1/Q) sum_q log p[q, y_q]
// (utils.py:90-93).  dL/dp_y = -1/(Q p_y); through the two softmaxes and u = -beta d:
//   dL/dd_x[c] = beta * w_x * (s_x[c] * s_x[y] - [c == y] s_x[y]) ... written as  g_x[c] = k_x * ([c==y] - s_x[c]),
//   k_x = weight_x * beta * s_x[y] / (Q p_y).
__global__ __launch_bounds__(256) void nll_grad_kernel(const float* __restrict__ d2i, const float* __restrict__ d2t,
                                                       const int32_t* __restrict__ labels, int Q, int q_total, int N, int ldd,
                                                       float alpha, float oma, float beta, float* __restrict__ gi,
                                                       float* __restrict__ gt,
                                                       float* __restrict__ rowsum, float* __restrict__ nll,
                                                       float* __restrict__ pmax, int32_t* __restrict__ argmax) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = blockIdx.x * 4 + wave; q < Q; q += gridDim.x * 4) {
        const float* ri = d2i + (size_t)q * ldd;
        const float* rt = d2t + (size_t)q * ldd;
        float mi = 3.4e38f, mt = 3.4e38f, xi = -3.4e38f, xt = -3.4e38f;
        for (int c = lane; c < N; c += 64) {
            mi = fminf(mi, ri[c]); xi = fmaxf(xi, ri[c]);
            mt = fminf(mt, rt[c]); xt = fmaxf(xt, rt[c]);
        }
        mi = wave_min(mi); mt = wave_min(mt); xi = wave_max(xi); xt = wave_max(xt);
        const float mxi = __fmul_rn(beta, beta >= 0.f ? -mi : -xi), mxt = __fmul_rn(beta, beta >= 0.f ? -mt : -xt);
        float si = 0.f, st = 0.f;
        for (int c = lane; c < N; c += 64) {
            si += expf(__fsub_rn(__fmul_rn(beta, -ri[c]), mxi));
            st += expf(__fsub_rn(__fmul_rn(beta, -rt[c]), mxt));
        }
        si = wave_sum(si); st = wave_sum(st);
        const int y = labels[q];
        float best = -1.f;
        int bi = 0x7fffffff;
        for (int c = lane; c < N; c += 64) {
            const float a = expf(__fsub_rn(__fmul_rn(beta, -ri[c]), mxi)) / si, b = expf(__fsub_rn(__fmul_rn(beta, -rt[c]), mxt)) / st;
            const float p = __fadd_rn(__fmul_rn(alpha, a), __fmul_rn(oma, b));
            if (p > best) { best = p; bi = c; }
        }
        wave_argmax(best, bi);
        const float siy = expf(__fsub_rn(__fmul_rn(beta, -ri[y]), mxi)) / si, sty = expf(__fsub_rn(__fmul_rn(beta, -rt[y]), mxt)) / st;
        const float py = __fadd_rn(__fmul_rn(alpha, siy), __fmul_rn(oma, sty));
        const float gy = -1.f / ((float)q_total * py);                 // dL/dp at the label (mean over ALL queries of the step)
        // softmax backward with a one-hot upstream gradient: du[c] = s[c] * (g[c] - g_y s_y), then dd = -beta du
        const float kiy = alpha * gy * siy, kty = oma * gy * sty;       // g_y * s_y per bank
        float rs = 0.f;
        for (int c = lane; c < N; c += 64) {
            const float a = expf(__fsub_rn(__fmul_rn(beta, -ri[c]), mxi)) / si, b = expf(__fsub_rn(__fmul_rn(beta, -rt[c]), mxt)) / st;
            const float dui = (c == y ? alpha * gy * a : 0.f) - a * kiy;
            const float dut = (c == y ? oma * gy * b : 0.f) - b * kty;
            const float gic = -beta * dui, gtc = -beta * dut;
            gi[(size_t)q * ldd + c] = gic;
            gt[(size_t)q * ldd + c] = gtc;
            rs += gic + gtc;
        }
        rs = wave_sum(rs);
        if (lane == 0) {
            rowsum[q] = rs;
            nll[q] = -logf(py);
            pmax[q] = best;
            argmax[q] = bi;
        }
    }
}

// ---- backward of P for an ARBITRARY upstream gradient (the autograd-transparent utils.P) ----------------------------------------
// p = alpha s_i + oma s_t,  s_x = softmax(-beta d_x)  (utils.py:236-242).  Given dp = dL/dp [Q, N]:
//   dL/du_x[c] = s_x[c] (w_x dp[c] - sum_k w_x dp[k] s_x[k]),  u_x = -beta d_x  ->  dL/dd_x = -beta dL/du_x
// whatever loss the caller hangs off p (the reference's loop: NLLLoss(torch.log(p)), main.py:281-285).  One wave per query; the
// softmax arithmetic is nll_grad_kernel's.  Also rowsum[q] = sum_c (gi + gt)[q, c] for the cdist backward (train.py).
__global__ __launch_bounds__(256) void fuse_probs_backward_kernel(const float* __restrict__ d2i, const float* __restrict__ d2t,
                                                                  const float* __restrict__ dp, int ldp, int Q, int N, int ldd,
                                                                  float alpha, float oma, float beta, float* __restrict__ gi,
                                                                  float* __restrict__ gt, float* __restrict__ rowsum) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = blockIdx.x * 4 + wave; q < Q; q += gridDim.x * 4) {
        const float* ri = d2i + (size_t)q * ldd;
        const float* rt = d2t + (size_t)q * ldd;
        const float* rg = dp + (size_t)q * ldp;
        float mi = 3.4e38f, mt = 3.4e38f, xi = -3.4e38f, xt = -3.4e38f;
        for (int c = lane; c < N; c += 64) {
            mi = fminf(mi, ri[c]); xi = fmaxf(xi, ri[c]);
            mt = fminf(mt, rt[c]); xt = fmaxf(xt, rt[c]);
        }
        mi = wave_min(mi); mt = wave_min(mt); xi = wave_max(xi); xt = wave_max(xt);
        const float mxi = __fmul_rn(beta, beta >= 0.f ? -mi : -xi), mxt = __fmul_rn(beta, beta >= 0.f ? -mt : -xt);
        float si = 0.f, st = 0.f;
        for (int c = lane; c < N; c += 64) {
            si += expf(__fsub_rn(__fmul_rn(beta, -ri[c]), mxi));
            st += expf(__fsub_rn(__fmul_rn(beta, -rt[c]), mxt));
        }
        si = wave_sum(si); st = wave_sum(st);
        float ki = 0.f, kt = 0.f;                                        // sum_k dp[k] s_x[k]
        for (int c = lane; c < N; c += 64) {
            const float a = expf(__fsub_rn(__fmul_rn(beta, -ri[c]), mxi)) / si, b = expf(__fsub_rn(__fmul_rn(beta, -rt[c]), mxt)) / st;
            ki = fmaf(rg[c], a, ki);
            kt = fmaf(rg[c], b, kt);
        }
        ki = wave_sum(ki); kt = wave_sum(kt);
        float rs = 0.f;
        for (int c = lane; c < N; c += 64) {
            const float a = expf(__fsub_rn(__fmul_rn(beta, -ri[c]), mxi)) / si, b = expf(__fsub_rn(__fmul_rn(beta, -rt[c]), mxt)) / st;
            const float gic = -beta * (alpha * a * (rg[c] - ki)), gtc = -beta * (oma * b * (rg[c] - kt));
            gi[(size_t)q * ldd + c] = gic;
            gt[(size_t)q * ldd + c] = gtc;
            rs += gic + gtc;
        }
        rs = wave_sum(rs);
        if (lane == 0) rowsum[q] = rs;
    }
}

// ---- backward of  loss = -(1/Q) sum_q log p[q, y_q]  (nn.NLLLoss()(torch.log(p), y), utils.py:90-93) wrt p ------------------
// dp[q, c] = -g / (Q p[q, y_q]) at c == y_q, 0 elsewhere;  g = the upstream gradient of the loss (device scalar).
__global__ __launch_bounds__(256) void nll_mean_backward_kernel(const float* __restrict__ p, int ldp, const int32_t* __restrict__ labels,
                                                                int Q, int N, const float* __restrict__ g, float* __restrict__ dp,
                                                                int lddp) {
    const float gq = -g[0] / (float)Q;
    const size_t total = (size_t)Q * N;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int q = (int)(i / N), c = (int)(i - (size_t)q * N);
        dp[(size_t)q * lddp + c] = c == labels[q] ? gq / p[(size_t)q * ldp + c] : 0.f;
    }
}

// ---- utils.py:84-93 on a materialised p [Q, N]: pred_p, y_hat = p.max(1); nll[q] = -log p[q, y_q] ---------------------------
__global__ __launch_bounds__(256) void nll_rows_kernel(const float* __restrict__ p, int ldp, const int32_t* __restrict__ labels, int Q,
                                                       int N, float* __restrict__ nll, float* __restrict__ pmax,
                                                       int32_t* __restrict__ argmax) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int q = blockIdx.x * 4 + wave; q < Q; q += gridDim.x * 4) {
        const float* row = p + (size_t)q * ldp;
        float best = -3.4e38f;
        int bi = 0x7fffffff;
        for (int c = lane; c < N; c += 64)
            if (row[c] > best) { best = row[c]; bi = c; }
        wave_argmax(best, bi);
        if (lane == 0) {
            nll[q] = -logf(row[labels[q]]);
            pmax[q] = best;
            argmax[q] = bi;
        }
    }
}

// ---- cross entropy against the diagonal (InfoNCE, info-nce-pytorch defaults): rows of S are logits -----------
// loss[r] = logsumexp(S[r,:]) - S[r,r];  dS[r,c] = scale * (softmax(S[r,:])[c] - [c == r])
__global__ __launch_bounds__(256) void softmax_ce_rows_kernel(const float* __restrict__ S, int lds, int R, int C, float scale,
                                                              float* __restrict__ dS, int ldds, float* __restrict__ loss) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        const float* row = S + (size_t)r * lds;
        float mx = -3.4e38f;
        for (int c = lane; c < C; c += 64) mx = fmaxf(mx, row[c]);
        mx = wave_max(mx);
        float s = 0.f;
        for (int c = lane; c < C; c += 64) s += expf(row[c] - mx);
        s = wave_sum(s);
        for (int c = lane; c < C; c += 64) dS[(size_t)r * ldds + c] = scale * (expf(row[c] - mx) / s - (c == r ? 1.f : 0.f));
        if (lane == 0) loss[r] = (logf(s) + mx) - row[r];
    }
}

// ---- F.normalize(x, dim=-1) in fp32 and its backward (InfoNCE normalises both sides again), wave per row -----------
__global__ __launch_bounds__(256) void l2norm_rows_f32_kernel(const float* __restrict__ x, float* __restrict__ y, int R, int D,
                                                              float eps) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        float ss = 0.f;
        for (int d = lane; d < D; d += 64) { const float v = x[(size_t)r * D + d]; ss += v * v; }
        const float n = fmaxf(sqrtf(wave_sum(ss)), eps);
        for (int d = lane; d < D; d += 64) y[(size_t)r * D + d] = x[(size_t)r * D + d] / n;
    }
}

// gx (+)= (gy - y (y . gy)) / n with y = x / n, n = max(|x|, eps)
__global__ __launch_bounds__(256) void l2norm_rows_backward_f32_kernel(const float* __restrict__ x, const float* __restrict__ gy,
                                                                       float* __restrict__ gx, int R, int D, float eps,
                                                                       int accumulate) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int r = blockIdx.x * 4 + wave; r < R; r += gridDim.x * 4) {
        float ss = 0.f, dot = 0.f;
        for (int d = lane; d < D; d += 64) {
            const float v = x[(size_t)r * D + d];
            ss += v * v;
            dot += v * gy[(size_t)r * D + d];
        }
        const float n = fmaxf(sqrtf(wave_sum(ss)), eps);
        const float yg = wave_sum(dot) / n;                   // y . gy
        for (int d = lane; d < D; d += 64) {
            const size_t i = (size_t)r * D + d;
            const float t = (gy[i] - (x[i] / n) * yg) / n;
            gx[i] = accumulate ? gx[i] + t : t;
        }
    }
}

// ---- prototype chain backward, one wave per class (main.py:260-264 / 276-279) -------------------------------------
// forward:  zh_k = per_shot ? r16s(v_k / r16s(|v_k|)) : v_k ;  m = r16s(mean_k zh_k) ;  out = final ? m32 / |m32| : m32
// backward: g (fp32, wrt out) -> fp32 normalise -> r16 (.float()) -> /K (fp16) -> fp16 normalise of every shot.
// The fp16 stages follow autograd's op sequence for  x / x.norm(dim=-1, keepdim=True)  with fp16 tensors:
//   div:  dx1 = r16s(gk / n);  dn = r16s(sum_d r16s(-gk * r16s(r16s(x/n)/n)))      norm:  dx2 = r16s(x * r16s(dn / n))
// gk — the gradient of the fp16 mean, already divided by K — does not depend on the shot: it is computed once per class into an
// LDS row (with the shot norms), so the work per class is O(K D) instead of the O(K^2 D) of recomputing the mean per shot.
// LDSROWS: the K shot rows of the class are copied to LDS in one batch of loads first (K > 1: the three passes below are
// otherwise ~3 K dependent round trips per wave, with one wave per SIMD at N = 1000 classes nothing hides them).
template <bool LDSROWS>
__global__ __launch_bounds__(256) void proto_backward_kernel(const half_t* __restrict__ mem, const float* __restrict__ g, int N,
                                                             int K, int D, int per_shot, int final_norm,
                                                             half_t* __restrict__ dmem) {
    extern __shared__ __attribute__((aligned(16))) float pb_lds[];   // per wave: [D] m then gk | [32] shot norms | (LDSROWS) [K*D] halfs
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int per_wave = D + 32 + (LDSROWS ? (K * D + 1) / 2 : 0);  // floats
    float* gks = pb_lds + (size_t)wave * per_wave;
    float* nk = gks + D;
    half_t* rl = reinterpret_cast<half_t*>(nk + 32);
    for (int n = blockIdx.x * 4 + wave; n < N; n += gridDim.x * 4) {
        const half_t* grows = mem + (size_t)n * K * D;
        const float* gn = g + (size_t)n * D;
        half_t* drows = dmem + (size_t)n * K * D;
        if (LDSROWS) {                                       // D % 8 == 0 on this path (checked by the launcher)
            for (int i = lane; i < K * D / 8; i += 64) *reinterpret_cast<half8_t*>(rl + i * 8) = ld_half8(grows + (size_t)i * 8);
        }
        const half_t* rows = LDSROWS ? rl : grows;
        // exact forward recomputation: shot norms first (fp16-rounded, one wave reduction per shot)
        for (int k = 0; k < K; ++k) {
            const half_t* v = rows + (size_t)k * D;
            float ss = 0.f;
            for (int d = lane; d < D; d += 64) { const float x = (float)v[d]; ss += x * x; }
            ss = wave_sum(ss);
            if (lane == 0) nk[k] = r16s(sqrtf(ss));
        }
        // mean over shots (fp32 accumulate, one rounding), its fp32 norm and m.g
        float dot = 0.f, nrm2 = 0.f;
        for (int d = lane; d < D; d += 64) {
            float acc = 0.f;
            for (int k = 0; k < K; ++k) {
                const float x = (float)rows[(size_t)k * D + d];
                acc += per_shot ? r16s(x / nk[k]) : x;
            }
            const float m = r16s(acc / (float)K);
            gks[d] = m;
            nrm2 += m * m;
            dot += m * gn[d];
        }
        nrm2 = wave_sum(nrm2);
        dot = wave_sum(dot);
        const float nrm = sqrtf(nrm2);
        const float pg = dot / nrm;                        // p . g with p = m / nrm
        for (int d = lane; d < D; d += 64) {               // every lane touches only its own d: no barrier needed
            const float m = gks[d];
            const float gm = final_norm ? (gn[d] - (m / nrm) * pg) / nrm : gn[d];
            gks[d] = r16s(r16s(gm) / (float)K);
        }
        for (int k = 0; k < K; ++k) {
            const half_t* v = rows + (size_t)k * D;
            const float nkk = nk[k];
            float dn = 0.f;
            if (per_shot) {
                for (int d = lane; d < D; d += 64) {
                    const float x = (float)v[d];
                    dn += r16s(-gks[d] * r16s(r16s(x / nkk) / nkk));
                }
                dn = r16s(wave_sum(dn));
            }
            for (int d = lane; d < D; d += 64) {
                float out = gks[d];
                if (per_shot) {
                    const float x = (float)v[d];
                    out = r16s(r16s(out / nkk) + r16s(x * r16s(dn / nkk)));
                }
                drows[(size_t)k * D + d] = (half_t)out;
            }
        }
    }
}

// ---- LayerNorm backward (fp16 tensors, fp32 statistics), wave per row -------------------------------------------
// y = r16s(xh * gamma + beta), xh = (x - mu) * rstd.  dx = rstd * (gy - mean(gy) - xh * mean(gy * xh)), gy = dy * gamma.
// dy_scale: upstream factor applied (and rounded to fp16) first — the 0.2 of the Adapter_FC blend (model.py:93-94).
// Partial parameter gradients of the rows a workgroup visits go to part[blockIdx.x][0|1][D] (dgamma | dbeta).
template <int NI>
__global__ __launch_bounds__(256) void layernorm_backward_kernel(const half_t* __restrict__ x, int ldx, const half_t* __restrict__ gamma,
                                                                 const half_t* __restrict__ dy, int lddy, int R, int D, float eps,
                                                                 float dy_scale, half_t* __restrict__ dx, int lddx,
                                                                 float* __restrict__ part) {
    __shared__ float red[4][NI * 64];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    float dg[NI], db[NI];
#pragma unroll
    for (int i = 0; i < NI; ++i) dg[i] = db[i] = 0.f;
    for (int row = blockIdx.x * 4 + wave; row < R; row += gridDim.x * 4) {
        const half_t* xr = x + (size_t)row * ldx;
        const half_t* gr = dy + (size_t)row * lddy;
        float xv[NI], gv[NI];
        float s = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int d = i * 64 + lane;
            xv[i] = d < D ? (float)xr[d] : 0.f;
            gv[i] = d < D ? (dy_scale == 1.f ? (float)gr[d] : r16s(dy_scale * (float)gr[d])) : 0.f;
            s += xv[i];
        }
        const float mu = wave_sum(s) / (float)D;
        float ss = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int d = i * 64 + lane;
            const float c = d < D ? xv[i] - mu : 0.f;
            ss += c * c;
        }
        const float rstd = 1.f / sqrtf(wave_sum(ss) / (float)D + eps);
        float a = 0.f, b = 0.f;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int d = i * 64 + lane;
            if (d < D) {
                const float xh = (xv[i] - mu) * rstd, gy = gv[i] * (float)gamma[d];
                a += gy;
                b += gy * xh;
                dg[i] += gv[i] * xh;
                db[i] += gv[i];
            }
        }
        a = wave_sum(a) / (float)D;
        b = wave_sum(b) / (float)D;
#pragma unroll
        for (int i = 0; i < NI; ++i) {
            const int d = i * 64 + lane;
            if (d < D) {
                const float xh = (xv[i] - mu) * rstd, gy = gv[i] * (float)gamma[d];
                dx[(size_t)row * lddx + d] = (half_t)(rstd * (gy - a - xh * b));
            }
        }
    }
    for (int pass = 0; pass < 2; ++pass) {
        __syncthreads();
#pragma unroll
        for (int i = 0; i < NI; ++i) red[wave][i * 64 + lane] = pass ? db[i] : dg[i];
        __syncthreads();
        for (int d = threadIdx.x; d < D; d += 256)
            part[((size_t)blockIdx.x * 2 + pass) * D + d] = ((red[0][d] + red[1][d]) + red[2][d]) + red[3][d];
    }
}

// ---- AdamW on fp16 parameters with fp16 state (torch.optim.AdamW single-tensor sequence, main.py:134-135) --------
//   p = r16s(p * (1 - lr*wd));  m = lerp(m, g, 1 - b1);  v = r16s(fma((1 - b2) * g, g, r16s(v * b2)))
//   denom = r16s(r16s(r16s(sqrt(v)) / sqrt(bc2)) + eps);  p = r16s(p - step_size * m / denom)
__global__ __launch_bounds__(256) void adamw_f16_kernel(half_t* __restrict__ p, const half_t* __restrict__ g, half_t* __restrict__ m,
                                                        half_t* __restrict__ v, size_t n, float decay, float w1, float b2,
                                                        float omb2, float bc2_sqrt, float eps, float neg_step_size) {
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < n; i += (size_t)gridDim.x * 256) {
        const float gi = (float)g[i];
        float pi = r16s(__fmul_rn((float)p[i], decay));
        const float mi0 = (float)m[i];
        // torch lerp: weight < 0.5 ? start + weight * (end - start) : end - (end - start) * (1 - weight); ATen compiles it with
        // contraction (CPU and GPU), i.e. ONE rounding of the multiply-add — pinned against torch in tests/test_gpu_train.py
        const float diff = __fsub_rn(gi, mi0);
        const float mi = r16s(w1 < 0.5f ? fmaf(w1, diff, mi0) : fmaf(-diff, __fsub_rn(1.f, w1), gi));
        float vi = r16s(__fmul_rn((float)v[i], b2));
        vi = r16s(fmaf(__fmul_rn(omb2, gi), gi, vi));             // addcmul: self + value*t1*t2, contracted like ATen's kernel
        float den = r16s(sqrtf(vi));
        den = r16s(__fdiv_rn(den, bc2_sqrt));
        den = r16s(__fadd_rn(den, eps));
        pi = r16s(__fadd_rn(pi, __fdiv_rn(__fmul_rn(neg_step_size, mi), den)));
        p[i] = (half_t)pi;
        m[i] = (half_t)mi;
        v[i] = (half_t)vi;
    }
}

}  // namespace

extern "C" int pclip_cast_f16_f32(const void* x, float* y, size_t n, pclip_stream_t stream) {
    PCLIP_REQUIRE((x && y) || n == 0, "pclip_cast_f16_f32: null pointer");
    if (n == 0) return PCLIP_OK;
    cast_f16_f32_kernel<<<flat_grid(n), 256, 0, (hipStream_t)stream>>>((const half_t*)x, y, n);
    return pclip_check_launch("cast_f16_f32");
}

extern "C" int pclip_gemm_f32(const void* A, int a_f16, long rsa, long csa, const void* B, int b_f16, long rsb, long csb, float* C,
                              int ldc, int M, int N, int K, float alpha, float beta, void* ws, size_t ws_bytes, pclip_stream_t stream) {
    PCLIP_REQUIRE(A && B && C, "pclip_gemm_f32: null pointer");
    PCLIP_REQUIRE(M >= 0 && N >= 0 && K > 0 && ldc >= N, "pclip_gemm_f32: bad shape M=%d N=%d K=%d ldc=%d", M, N, K, ldc);
    if (M == 0 || N == 0) return PCLIP_OK;
    dim3 grid(ceil_div(N, 64), ceil_div(M, 64));
    hipStream_t s = (hipStream_t)stream;
    // few output tiles and a long K (weight gradients of the fc adapter: [D/4 x D] over thousands of query rows): K slices
    // in parallel -> slabs in the workspace -> added in slice order
    const int tiles = grid.x * grid.y;
    int S = 1, kper = K;
    if (ws && tiles <= 128 && K >= 1024) {
        S = 512 / tiles;
        if (S > K / 256) S = K / 256;
        if (S > 32) S = 32;
        kper = ceil_div(ceil_div(K, S), 32) * 32;
        S = ceil_div(K, kper);
        if (S < 2 || ws_bytes < (size_t)S * M * N * sizeof(float)) { S = 1; kper = K; }
    }
    float* slabs = S > 1 ? (float*)ws : nullptr;
    grid.z = S;
    if (a_f16 && b_f16) gemm_f32_kernel<half_t, half_t><<<grid, 256, 0, s>>>((const half_t*)A, rsa, csa, (const half_t*)B, rsb, csb, C, ldc, M, N, K, alpha, beta, kper, slabs);
    else if (a_f16) gemm_f32_kernel<half_t, float><<<grid, 256, 0, s>>>((const half_t*)A, rsa, csa, (const float*)B, rsb, csb, C, ldc, M, N, K, alpha, beta, kper, slabs);
    else if (b_f16) gemm_f32_kernel<float, half_t><<<grid, 256, 0, s>>>((const float*)A, rsa, csa, (const half_t*)B, rsb, csb, C, ldc, M, N, K, alpha, beta, kper, slabs);
    else gemm_f32_kernel<float, float><<<grid, 256, 0, s>>>((const float*)A, rsa, csa, (const float*)B, rsb, csb, C, ldc, M, N, K, alpha, beta, kper, slabs);
    if (S > 1) {
        const size_t total = (size_t)M * N;
        gemm_f32_reduce_kernel<<<(int)((total + 255) / 256 > 4096 ? 4096 : (total + 255) / 256), 256, 0, s>>>(slabs, S, M, N, C, ldc, alpha, beta);
    }
    return pclip_check_launch("gemm_f32");
}

extern "C" int pclip_colsum_f32(const float* x, int ldx, int R, int C, float scale, float* out, int accumulate, void* ws,
                                size_t ws_bytes, pclip_stream_t stream) {
    PCLIP_REQUIRE(x && out, "pclip_colsum_f32: null pointer");
    PCLIP_REQUIRE(R >= 0 && C > 0 && ldx >= C, "pclip_colsum_f32: bad shape R=%d C=%d ld=%d", R, C, ldx);
    hipStream_t s = (hipStream_t)stream;
    // many rows: row blocks in parallel -> partial sums [RB][C] in the workspace -> summed in block order (still deterministic);
    // one workgroup per 64 columns alone leaves the chip idle (C = 1000: 16 workgroups, C = 1 — a loss mean — one)
    int rb = R > 128 ? ceil_div(R, 32) : 1;
    if (rb > 64) rb = 64;
    if (rb > 1 && ws && ws_bytes >= (size_t)rb * C * sizeof(float)) {
        const int rows_per = ceil_div(R, rb);
        colsum_part_kernel<<<dim3(ceil_div(C, 64), rb), 256, 0, s>>>(x, ldx, R, C, rows_per, (float*)ws);
        colsum_kernel<<<ceil_div(C, 64), 256, 0, s>>>((const float*)ws, C, rb, C, scale, out, accumulate);
        return pclip_check_launch("colsum_f32 (two-pass)");
    }
    colsum_kernel<<<ceil_div(C, 64), 256, 0, s>>>(x, ldx, R, C, scale, out, accumulate);
    return pclip_check_launch("colsum_f32");
}

extern "C" int pclip_addscaled_rows_f32(float* C, int ldc, const float* X, int ldx, const float* rowscale, float s, int R, int D,
                                        pclip_stream_t stream) {
    PCLIP_REQUIRE(C && X && rowscale, "pclip_addscaled_rows_f32: null pointer");
    PCLIP_REQUIRE(R >= 0 && D > 0 && ldc >= D && ldx >= D, "pclip_addscaled_rows_f32: bad shape");
    if (R == 0) return PCLIP_OK;
    addscaled_rows_kernel<<<flat_grid((size_t)R * D), 256, 0, (hipStream_t)stream>>>(C, ldc, X, ldx, rowscale, s, R, D);
    return pclip_check_launch("addscaled_rows_f32");
}

extern "C" int pclip_nll_grad(const float* d2i, const float* d2t, const int32_t* labels, int Q, int q_total, int N, int ldd,
                              float alpha, float one_minus_alpha, float beta, float* gi, float* gt, float* rowsum, float* nll,
                              float* pmax, int32_t* argmax, pclip_stream_t stream) {
    PCLIP_REQUIRE(d2i && d2t && labels && gi && gt && rowsum && nll && pmax && argmax, "pclip_nll_grad: null pointer");
    PCLIP_REQUIRE(Q >= 0 && q_total >= Q && N > 0 && ldd >= N, "pclip_nll_grad: bad Q=%d q_total=%d N=%d ldd=%d", Q, q_total, N, ldd);
    if (Q == 0) return PCLIP_OK;
    nll_grad_kernel<<<row_grid(Q, 8192), 256, 0, (hipStream_t)stream>>>(d2i, d2t, labels, Q, q_total, N, ldd, alpha, one_minus_alpha, beta,
                                                                       gi, gt, rowsum, nll, pmax, argmax);
    return pclip_check_launch("nll_grad");
}

extern "C" int pclip_fuse_probs_backward(const float* d2i, const float* d2t, const float* dp, int ldp, int Q, int N, int ldd, float alpha,
                                         float one_minus_alpha, float beta, float* gi, float* gt, float* rowsum, pclip_stream_t stream) {
    PCLIP_REQUIRE(d2i && d2t && dp && gi && gt && rowsum, "pclip_fuse_probs_backward: null pointer");
    PCLIP_REQUIRE(Q >= 0 && N > 0 && ldd >= N && ldp >= N, "pclip_fuse_probs_backward: bad Q=%d N=%d ldd=%d ldp=%d", Q, N, ldd, ldp);
    if (Q == 0) return PCLIP_OK;
    fuse_probs_backward_kernel<<<row_grid(Q, 8192), 256, 0, (hipStream_t)stream>>>(d2i, d2t, dp, ldp, Q, N, ldd, alpha, one_minus_alpha, beta,
                                                                                  gi, gt, rowsum);
    return pclip_check_launch("fuse_probs_backward");
}

extern "C" int pclip_nll_mean_backward(const float* p, int ldp, const int32_t* labels, int Q, int N, const float* g, float* dp, int lddp,
                                       pclip_stream_t stream) {
    PCLIP_REQUIRE(p && labels && g && dp, "pclip_nll_mean_backward: null pointer");
    PCLIP_REQUIRE(Q >= 0 && N > 0 && ldp >= N && lddp >= N, "pclip_nll_mean_backward: bad Q=%d N=%d ldp=%d lddp=%d", Q, N, ldp, lddp);
    if (Q == 0) return PCLIP_OK;
    nll_mean_backward_kernel<<<flat_grid((size_t)Q * N), 256, 0, (hipStream_t)stream>>>(p, ldp, labels, Q, N, g, dp, lddp);
    return pclip_check_launch("nll_mean_backward");
}

extern "C" int pclip_nll_rows(const float* p, int ldp, const int32_t* labels, int Q, int N, float* nll, float* pmax, int32_t* argmax,
                              pclip_stream_t stream) {
    PCLIP_REQUIRE(p && labels && nll && pmax && argmax, "pclip_nll_rows: null pointer");
    PCLIP_REQUIRE(Q >= 0 && N > 0 && ldp >= N, "pclip_nll_rows: bad Q=%d N=%d ld=%d", Q, N, ldp);
    if (Q == 0) return PCLIP_OK;
    nll_rows_kernel<<<row_grid(Q, 8192), 256, 0, (hipStream_t)stream>>>(p, ldp, labels, Q, N, nll, pmax, argmax);
    return pclip_check_launch("nll_rows");
}

extern "C" int pclip_softmax_ce_rows(const float* S, int lds, int R, int C, float scale, float* dS, int ldds, float* loss,
                                     pclip_stream_t stream) {
    PCLIP_REQUIRE(S && dS && loss, "pclip_softmax_ce_rows: null pointer");
    PCLIP_REQUIRE(R >= 0 && C >= R && lds >= C && ldds >= C, "pclip_softmax_ce_rows: bad R=%d C=%d", R, C);
    if (R == 0) return PCLIP_OK;
    softmax_ce_rows_kernel<<<row_grid(R, 8192), 256, 0, (hipStream_t)stream>>>(S, lds, R, C, scale, dS, ldds, loss);
    return pclip_check_launch("softmax_ce_rows");
}

extern "C" int pclip_l2norm_rows_f32(const float* x, float* y, int R, int D, float eps, pclip_stream_t stream) {
    PCLIP_REQUIRE(x && y, "pclip_l2norm_rows_f32: null pointer");
    PCLIP_REQUIRE(R >= 0 && D > 0, "pclip_l2norm_rows_f32: bad shape R=%d D=%d", R, D);
    if (R == 0) return PCLIP_OK;
    l2norm_rows_f32_kernel<<<row_grid(R, 8192), 256, 0, (hipStream_t)stream>>>(x, y, R, D, eps);
    return pclip_check_launch("l2norm_rows_f32");
}

extern "C" int pclip_l2norm_rows_backward_f32(const float* x, const float* gy, float* gx, int R, int D, float eps, int accumulate,
                                              pclip_stream_t stream) {
    PCLIP_REQUIRE(x && gy && gx, "pclip_l2norm_rows_backward_f32: null pointer");
    PCLIP_REQUIRE(R >= 0 && D > 0, "pclip_l2norm_rows_backward_f32: bad shape R=%d D=%d", R, D);
    if (R == 0) return PCLIP_OK;
    l2norm_rows_backward_f32_kernel<<<row_grid(R, 8192), 256, 0, (hipStream_t)stream>>>(x, gy, gx, R, D, eps, accumulate);
    return pclip_check_launch("l2norm_rows_backward_f32");
}

extern "C" int pclip_proto_backward_f16(const void* mem, const float* g, int N, int K, int D, int per_shot_norm, int final_norm,
                                        void* dmem, pclip_stream_t stream) {
    PCLIP_REQUIRE(mem && g && dmem, "pclip_proto_backward_f16: null pointer");
    PCLIP_REQUIRE(N >= 0 && K > 0 && K <= 32 && D > 0 && D <= 3072, "pclip_proto_backward_f16: bad N=%d K=%d (<=32) D=%d (<=3072)", N, K, D);
    if (N == 0) return PCLIP_OK;
    const size_t base = 4 * (size_t)(D + 32) * sizeof(float), with_rows = 4 * ((size_t)(D + 32) + ((size_t)K * D + 1) / 2) * sizeof(float);
    if (K > 1 && D % 8 == 0 && with_rows <= 150 * 1024) {
        static DevOnce attr;
        if (!attr.done()) {
            if (hipFuncSetAttribute((const void*)proto_backward_kernel<true>, hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024) != hipSuccess) {
                pclip_set_error("pclip_proto_backward_f16: cannot raise the dynamic LDS limit");
                return PCLIP_E_LAUNCH;
            }
            attr.set();
        }
        proto_backward_kernel<true><<<row_grid(N, 8192), 256, with_rows, (hipStream_t)stream>>>((const half_t*)mem, g, N, K, D, per_shot_norm,
                                                                                           final_norm, (half_t*)dmem);
    } else {
        proto_backward_kernel<false><<<row_grid(N, 8192), 256, base, (hipStream_t)stream>>>((const half_t*)mem, g, N, K, D, per_shot_norm,
                                                                                       final_norm, (half_t*)dmem);
    }
    return pclip_check_launch("proto_backward");
}

extern "C" int pclip_layernorm_backward_f16(const void* x, int ldx, const void* gamma, const void* dy, int lddy, int R, int D,
                                            float eps, float dy_scale, void* dx, int lddx, float* part, int nblk,
                                            pclip_stream_t stream) {
    PCLIP_REQUIRE(x && gamma && dy && dx && part, "pclip_layernorm_backward_f16: null pointer");
    PCLIP_REQUIRE(R >= 0 && D > 0 && D <= 2048 && ldx >= D && lddy >= D && lddx >= D && nblk > 0,
                  "pclip_layernorm_backward_f16: bad R=%d D=%d (<=2048) nblk=%d", R, D, nblk);
    hipStream_t s = (hipStream_t)stream;
#define LNB(NI) layernorm_backward_kernel<NI><<<nblk, 256, 0, s>>>((const half_t*)x, ldx, (const half_t*)gamma, (const half_t*)dy, lddy, R, D, eps, dy_scale, (half_t*)dx, lddx, part)
    if (D <= 256) LNB(4);
    else if (D <= 512) LNB(8);
    else if (D <= 1024) LNB(16);
    else LNB(32);
#undef LNB
    return pclip_check_launch("layernorm_backward");
}

extern "C" int pclip_adamw_f16(void* p, const void* g, void* m, void* v, size_t n, double lr, double beta1, double beta2, double eps,
                               double weight_decay, int step, pclip_stream_t stream) {
    PCLIP_REQUIRE((p && g && m && v) || n == 0, "pclip_adamw_f16: null pointer");
    PCLIP_REQUIRE(step >= 1, "pclip_adamw_f16: step=%d must be >= 1", step);
    if (n == 0) return PCLIP_OK;
    // scalar prologue in double like the Python floats of torch/optim/adamw.py, cast once to the fp32 opmath type
    const double bc1 = 1.0 - pow(beta1, step), bc2 = 1.0 - pow(beta2, step);
    const double step_size = lr / bc1;
    adamw_f16_kernel<<<flat_grid(n), 256, 0, (hipStream_t)stream>>>((half_t*)p, (const half_t*)g, (half_t*)m, (half_t*)v, n,
                                                                   (float)(1.0 - lr * weight_decay), (float)(1.0 - beta1),
                                                                   (float)beta2, (float)(1.0 - beta2), (float)sqrt(bc2), (float)eps,
                                                                   (float)(-step_size));
    return pclip_check_launch("adamw_f16");
}
