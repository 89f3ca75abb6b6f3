// Classification against the two prototype banks: utils.py:225-244 `P`, the argmax/top-k consumers
// (main.py:190, toolkit proto_clip_classifier.py:146-147) and the (alpha, beta) grid (main.py:187-199,
// 419-430).  Split in two stages so that the distance rows — which do not depend on alpha/beta
//  — are produced once by the MFMA contraction and then consumed by cheap VALU passes.
#include "pclip_gemm.h"
#include "pclip_classify_small.h"
#include <stdlib.h>

// fused row-panel classification for large class counts (pclip_classify_panel.hip)
size_t pclip_classify_panel_workspace(int Q, int N, int D);
bool pclip_classify_panel_applies(int Q, int N, int D, float alpha, float one_minus_alpha, float beta);
bool pclip_classify_mid_applies(int Q, int N, int D, bool has_zt, bool topk);
int pclip_classify_mid_launch(const void* q, const void* zi, const void* zt, int Q, int N, int D, float alpha, float oma, float beta, float* p, int32_t* argmax,
                              hipStream_t s);
int pclip_classify_panel_launch(const void* q, const void* zi, const void* zt, int Q, int N, int D, const float* q_sq, const float* zi_sq, const float* zt_sq,
                                float alpha, float oma, float beta, int32_t* argmax, float* dump, bool dump_exact, void* ws, hipStream_t s);

namespace {

// ---- stage 1: squared distances on MFMA ---------------------------------------------------------
// torch.cdist (mm path) evaluates ||q||^2 + ||z||^2 - 2 q.z in fp32, clamps at 0, takes sqrt; the
// reference then squares it again (utils.py:230-233).  fp16 operands make every product exact in the
// fp32 accumulator, so only summation order differs from the reference's fp32 GEMM.
// Each lane finishes four consecutive classes of one query row: one 16-byte store.
__global__ __launch_bounds__(256, 2) void sqdist_kernel(const half_t* __restrict__ q, const half_t* __restrict__ zi,
                                                        const half_t* __restrict__ zt, int Q, int N, int D,
                                                        const float* __restrict__ q_sq, const float* __restrict__ zi_sq,
                                                        const float* __restrict__ zt_sq, float* __restrict__ d2i,
                                                        float* __restrict__ d2t, int ldd, int tiles_n) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int bank = blockIdx.y;
    const int swz = pgemm::xcd_remap(blockIdx.x, gridDim.x);
    const int tile_m = swz / tiles_n, tile_n = swz - tile_m * tiles_n;
    using C = pgemm::CfgSmall;
    const int m0 = tile_m * C::BM, n0 = tile_n * C::BN;
    pgemm::Acc<C> acc;
    int pbuf = 0;
    pgemm::stage_first<C>(q, D, bank ? zt : zi, D, Q, N, m0, n0, smem, 0);
    pgemm::mainloop<C>(q, D, bank ? zt : zi, D, Q, N, D, m0, n0, smem, acc, pbuf);
    const float* __restrict__ z_sq = bank ? zt_sq : zi_sq;
    float* __restrict__ out = bank ? d2t : d2i;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wr = wave >> 1, wc = wave & 1, hi = lane >> 5;
    // The fp32 tile (128 x 128 x 4 B = the whole 64 KiB of LDS, free after the K-loop) is staged so that the global stores
    // are whole 512-byte row segments: from the MFMA layout a store instruction would touch 32 rows x 32 bytes.
    // 16-byte units are XOR-swizzled with the row so that the 32 rows a wave writes at once land on different banks.
    float* stg = reinterpret_cast<float*>(smem);
    pgemm::lds_barrier();                                   // every wave is done reading the K-tiles
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ml = wr * 64 + i * 32 + (lane & 31);
        const int m = m0 + ml;
        const float qs = m < Q ? q_sq[m] : 0.f;
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = wc * 64 + j * 32 + 8 * g + 4 * hi;
                float4_t o;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    const float zs = n0 + nl + e < N ? z_sq[n0 + nl + e] : 0.f;
                    const float v = __fadd_rn(__fadd_rn(-2.f * acc.v[i][j][4 * g + e], qs), zs);
                    const float d = sqrtf(fmaxf(v, 0.f));
                    o[e] = __fmul_rn(d, d);
                }
                *reinterpret_cast<float4_t*>(stg + ml * 128 + (((nl >> 2) ^ (ml & 31)) << 2)) = o;
            }
    }
    pgemm::lds_barrier();
    const int u = tid & 31;                                 // 16-byte unit of the row: 32 units = 128 columns
#pragma unroll
    for (int ps = 0; ps < 16; ++ps) {
        const int r = (tid >> 5) + ps * 8, m = m0 + r, n = n0 + 4 * u;
        if (m >= Q) continue;
        const float4_t o = *reinterpret_cast<const float4_t*>(stg + r * 128 + ((u ^ (r & 31)) << 2));
        float* dst = out + (size_t)m * ldd + n;
        if (n + 3 < ldd) *reinterpret_cast<float4_t*>(dst) = o;          // columns in [N, ldd) are padding
        else {
#pragma unroll
            for (int e = 0; e < 4; ++e) if (n + e < ldd) dst[e] = o[e];
        }
    }
}

// Large problems (ImageNet: 50 000 x 1000 x 512 per bank): the persistent 256x256 tile of the encoder GEMM — the 128x128
// one-shot tiles above spend most of their 8 K-tiles waiting for LDS-DMA.  Tiles of both banks form one index space; the
// fp32 result leaves through LDS in four 64-row slabs (64 x 256 x 4 B = one 64 KiB stage buffer) as whole 1 KiB row segments.
__global__ __launch_bounds__(512, 2) void sqdist_big_kernel(const half_t* __restrict__ q, const half_t* __restrict__ zi,
                                                            const half_t* __restrict__ zt, int Q, int N, int D,
                                                            const float* __restrict__ q_sq, const float* __restrict__ zi_sq,
                                                            const float* __restrict__ zt_sq, float* __restrict__ d2i,
                                                            float* __restrict__ d2t, int ldd, int tiles_n, int tiles_per_bank,
                                                            int ntiles) {
    using C = pgemm::Cfg<256, 256, 2, 4>;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int G = gridDim.x;
    int tile = pgemm::xcd_remap(blockIdx.x, G);
    if (tile >= ntiles) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6, wm = wave / C::WN, wn = wave % C::WN, hi = lane >> 5;
    int p = 0;
    auto coords = [&](int t, int& bank, int& m0, int& n0) {
        bank = t / tiles_per_bank;
        const int r = t - bank * tiles_per_bank, tm = r / tiles_n;
        m0 = tm * C::BM;
        n0 = (r - tm * tiles_n) * C::BN;
    };
    {
        int bank, m0, n0;
        coords(tile, bank, m0, n0);
        pgemm::stage_first<C>(q, D, bank ? zt : zi, D, Q, N, m0, n0, smem, p);
    }
    constexpr int NSTORE = 32;                                   // 16-byte stores per thread per tile: 4 slabs x 8 passes
    // squared norms of the tile's 256 query rows / 256 prototypes travel by LDS-DMA too (an ordinary VGPR load beside the
    // K-tile prefetch would make hipcc drain vmcnt(0) at its use); every wave copies both strips: uniform vmcnt bookkeeping.
    // Q % 4 == 0 and N % 4 == 0 (checked by the launcher): a lane's four norms are all valid or all beyond the end (clamped).
    float* norms = reinterpret_cast<float*>(smem + C::LDS_BYTES);          // [2][ q: 256 | z: 256 ]
    bool prev_full = false;
    int par = 0;
    for (; tile < ntiles; tile += G, par ^= 1) {
        int bank, m0, n0;
        coords(tile, bank, m0, n0);
        const half_t* z = bank ? zt : zi;
        const float* __restrict__ z_sq = bank ? zt_sq : zi_sq;
        float* __restrict__ out = bank ? d2t : d2i;
        const bool full = m0 + C::BM <= Q && n0 + C::BN <= ldd;
        float* qn = norms + par * 512;
        float* zn = qn + 256;
        {
            int qi = m0 + 4 * lane, zi4 = n0 + 4 * lane;
            qi = qi + 4 <= Q ? qi : Q - 4;
            zi4 = zi4 + 4 <= N ? zi4 : N - 4;
            // (buffer LDS-DMA, like the tiles: behind a FLAT-encoded global_load_lds hipcc answers every LDS wait of the first K-tile with lgkmcnt(0): pclip_gemm.h make_rsrc)
            const pgemm::rsrc_t rq = pgemm::make_rsrc(q_sq, 0x7fffffffu), rz = pgemm::make_rsrc(z_sq, 0x7fffffffu);
            (void)rq; (void)rz; (void)qi; (void)zi4;
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rq, (pgemm::lds_ptr_t)qn, 16, qi * 4, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rz, (pgemm::lds_ptr_t)zn, 16, zi4 * 4, 0, 0, 0);
#endif
        }
        pgemm::Acc<C> acc;
        pgemm::mainloop<C, NSTORE + 2>(q, D, z, D, Q, N, D, m0, n0, smem, acc, p, prev_full);
        pgemm::wait_vm<0>();
        const int next = tile + G;
        if (next < ntiles) {
            int nb, nm0, nn0;
            coords(next, nb, nm0, nn0);
            pgemm::stage_first<C>(q, D, nb ? zt : zi, D, Q, N, nm0, nn0, smem, p);
        }
        float* stg = reinterpret_cast<float*>(smem + (p ^ 1) * C::STAGE_BYTES);
        // wave (wm, wn) owns rows wm*128 + i*32 + (lane&31), i < 4: slab h (64 rows) = row tiles i = 2*(h&1), 2*(h&1)+1 of wm = h>>1
#pragma unroll
        for (int h = 0; h < 4; ++h) {
            pgemm::lds_barrier();
            if (wm == (h >> 1)) {
#pragma unroll
                for (int ii = 0; ii < 2; ++ii) {
                    const int i = 2 * (h & 1) + ii;
                    const int ml = ii * 32 + (lane & 31);          // row inside the slab
                    const float qs = qn[h * 64 + ml];
#pragma unroll
                    for (int j = 0; j < C::TN; ++j)
#pragma unroll
                        for (int g = 0; g < 4; ++g) {
                            const int nl = wn * (C::BN / C::WN) + j * 32 + 8 * g + 4 * hi;
                            float4_t o;
#pragma unroll
                            for (int e = 0; e < 4; ++e) {
                                const float zs = zn[nl + e];
                                const float v = __fadd_rn(__fadd_rn(-2.f * acc.v[i][j][4 * g + e], qs), zs);
                                const float d = sqrtf(fmaxf(v, 0.f));
                                o[e] = __fmul_rn(d, d);
                            }
                            *reinterpret_cast<float4_t*>(stg + ml * 256 + (((nl >> 2) ^ (ml & 63)) << 2)) = o;
                        }
                }
            }
            pgemm::lds_barrier();
            const int u = tid & 63;                              // 16-byte unit of the 1 KiB row: 64 units = 256 columns
#pragma unroll
            for (int ps = 0; ps < 8; ++ps) {
                const int r = (tid >> 6) + ps * 8, m = m0 + h * 64 + r, n = n0 + 4 * u;
                const float4_t o = *reinterpret_cast<const float4_t*>(stg + r * 256 + ((u ^ (r & 63)) << 2));
                if (full) *reinterpret_cast<float4_t*>(out + (size_t)m * ldd + n) = o;
                else if (m < Q) {
                    float* dst = out + (size_t)m * ldd + n;
                    if (n + 3 < ldd) *reinterpret_cast<float4_t*>(dst) = o;
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e) if (n + e < ldd) dst[e] = o[e];
                    }
                }
            }
        }
        prev_full = full;
    }
}

// fp32-operand variant (training path, main.py:262-281: fp32 prototypes / adapted queries).  Exact fp32
// products and accumulation on v_mfma_f32_32x32x2_f32 (bitwise an fmaf chain, guide §3); 64x64 output tile per
// workgroup, one 32x32 accumulator per wave, K staged 32 columns at a time through LDS (padded rows: no
// conflicts on the per-lane column reads).  The problem sizes of that path are small (<= a few thousand rows).
__global__ __launch_bounds__(256) void sqdist_f32_kernel(const float* __restrict__ q, const float* __restrict__ z, int Q, int N,
                                                         int D, float* __restrict__ out, int ldd) {
    __shared__ float As[64][33], Bs[64][33];
    __shared__ float qn[64], zn[64];
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int m0 = blockIdx.y * 64, n0 = blockIdx.x * 64;
    const int wr = wave >> 1, wc = wave & 1, hi = lane >> 5, l31 = lane & 31;
    float16_t acc;
#pragma unroll
    for (int e = 0; e < 16; ++e) acc[e] = 0.f;
    float nrm = 0.f;                                   // threads 0..63: ||q_row||^2, 64..127: ||z_row||^2
    // register double buffering: the loads of K-slab k0+32 fly while slab k0 is multiplied (same arithmetic, same order)
    float ra[8], rb[8];
    auto fetch = [&](int k0) {
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + u * 256, r = i >> 5, c = i & 31;
            ra[u] = (m0 + r < Q && k0 + c < D) ? q[(size_t)(m0 + r) * D + k0 + c] : 0.f;
            rb[u] = (n0 + r < N && k0 + c < D) ? z[(size_t)(n0 + r) * D + k0 + c] : 0.f;
        }
    };
    fetch(0);
    for (int k0 = 0; k0 < D; k0 += 32) {
        __syncthreads();
#pragma unroll
        for (int u = 0; u < 8; ++u) {
            const int i = tid + u * 256, r = i >> 5, c = i & 31;
            As[r][c] = ra[u];
            Bs[r][c] = rb[u];
        }
        __syncthreads();
        if (k0 + 32 < D) fetch(k0 + 32);
        if (tid < 128) {
            const float* row = tid < 64 ? As[tid] : Bs[tid - 64];
#pragma unroll 8
            for (int c = 0; c < 32; ++c) nrm = fmaf(row[c], row[c], nrm);
        }
#pragma unroll
        for (int kk = 0; kk < 32; kk += 2) {
            // swapped operands as in the fp16 kernel: D[n][m] so that a lane owns row m = lane & 31
            const float b = Bs[wc * 32 + l31][kk + hi], a = As[wr * 32 + l31][kk + hi];
            acc = __builtin_amdgcn_mfma_f32_32x32x2f32(b, a, acc, 0, 0, 0);
        }
    }
    if (tid < 64) qn[tid] = nrm; else if (tid < 128) zn[tid - 64] = nrm;
    __syncthreads();
    const int m = m0 + wr * 32 + l31;
    if (m >= Q) return;
    const float qs = qn[wr * 32 + l31];
#pragma unroll
    for (int e = 0; e < 16; ++e) {
        const int nl = wc * 32 + 8 * (e >> 2) + 4 * hi + (e & 3);
        const int n = n0 + nl;
        if (n < N) {
            const float v = __fadd_rn(__fadd_rn(-2.f * acc[e], qs), zn[nl]);
            const float d = sqrtf(fmaxf(v, 0.f));
            out[(size_t)m * ldd + n] = __fmul_rn(d, d);
        }
    }
}

// ---- stage 2: softmax fusion, one wave per query row ----------------------------------------------
// Lane l owns classes n = i*64 + l (scalar mapping) — coalesced 256-byte row segments.
template <int NV>
__device__ __forceinline__ void load_row(const float* __restrict__ row, int N, int lane, float (&v)[NV]) {
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        int n = i * 64 + lane;
        v[i] = n < N ? row[n] : 0.f;
    }
}

// e[i] = exp(beta*(-d_i) - max_n beta*(-d_n)) for valid classes (0 for padding); returns the wave-wide
// sum.  Rounding is monotone, so the max is beta*(-dmin) for beta >= 0 and beta*(-dmax) otherwise.
template <int NV>
__device__ __forceinline__ float softmax_terms(const float (&d)[NV], float beta, float dmin, float dmax, int N,
                                               int lane, float (&e)[NV]) {
    const float mx = __fmul_rn(beta, beta >= 0.f ? -dmin : -dmax);
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < NV; ++i) {
        e[i] = (i * 64 + lane < N) ? expf(__fsub_rn(__fmul_rn(beta, -d[i]), mx)) : 0.f;
        s += e[i];
    }
    return wave_sum(s);
}

// (min, max) over the valid classes of a row
template <int NV>
__device__ __forceinline__ void row_minmax(const float (&d)[NV], int N, int lane, float& mn, float& mx) {
    mn = __builtin_inff(); mx = -__builtin_inff();
#pragma unroll
    for (int i = 0; i < NV; ++i)
        if (i * 64 + lane < N) { mn = fminf(mn, d[i]); mx = fmaxf(mx, d[i]); }
    mn = wave_min(mn); mx = wave_max(mx);
}

template <int NV>
__global__ __launch_bounds__(256) void fuse_probs_kernel(const float* __restrict__ d2i, const float* __restrict__ d2t,
                                                         int Q, int N, int ldd, float alpha, float oma, float beta,
                                                         float* __restrict__ p, int32_t* __restrict__ argmax,
                                                         float* __restrict__ topk_p, int32_t* __restrict__ topk_i,
                                                         int k) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < Q; row += gridDim.x * 4) {
        float di[NV], dt[NV], ei[NV], et[NV];
        float mn, mx;
        load_row<NV>(d2i + (size_t)row * ldd, N, lane, di);
        row_minmax<NV>(di, N, lane, mn, mx);
        const float li = softmax_terms<NV>(di, beta, mn, mx, N, lane, ei);
        float lt = 1.f;
        if (d2t) {
            load_row<NV>(d2t + (size_t)row * ldd, N, lane, dt);
            row_minmax<NV>(dt, N, lane, mn, mx);
            lt = softmax_terms<NV>(dt, beta, mn, mx, N, lane, et);
        }
        float best = -1.f;
        int besti = 0x7fffffff;
#pragma unroll
        for (int i = 0; i < NV; ++i) {
            const int n = i * 64 + lane;
            float pv = __fmul_rn(alpha, __fdiv_rn(ei[i], li));
            if (d2t) pv = __fadd_rn(pv, __fmul_rn(oma, __fdiv_rn(et[i], lt)));
            ei[i] = pv;   // reuse as p
            if (n < N) {
                if (p) p[(size_t)row * N + n] = pv;
                if (pv > best) { best = pv; besti = n; }   // ascending n: first max kept
            } else {
                ei[i] = -1.f;
            }
        }
        if (argmax) {
            float bv = best; int bi = besti;
            wave_argmax(bv, bi);
            if (lane == 0) argmax[row] = bi;
        }
        if (topk_p || topk_i) {
            for (int t = 0; t < k; ++t) {
                float bv = -2.f; int bi = 0x7fffffff;
#pragma unroll
                for (int i = 0; i < NV; ++i)
                    if (ei[i] > bv) { bv = ei[i]; bi = i * 64 + lane; }
                wave_argmax(bv, bi);
                if (lane == 0) {
                    if (topk_p) topk_p[(size_t)row * k + t] = bv;
                    if (topk_i) topk_i[(size_t)row * k + t] = bi;
                }
#pragma unroll
                for (int i = 0; i < NV; ++i)
                    if (i * 64 + lane == bi) ei[i] = -2.f;   // remove the winner
            }
        }
    }
}

// (alpha, beta) grid: distances are loaded once per row; per beta the two softmaxes are formed once and
// every alpha only costs a fused multiply/add + argmax.  Correct-counts are accumulated in LDS and
// flushed with one atomic per (pair, workgroup).
template <int NV>
__global__ __launch_bounds__(256) void hp_sweep_kernel(const float* __restrict__ d2i, const float* __restrict__ d2t,
                                                       const int32_t* __restrict__ labels, int Q, int N, int ldd,
                                                       const float* __restrict__ alphas, const float* __restrict__ omas,
                                                       int na, const float* __restrict__ betas, int nb,
                                                       int32_t* __restrict__ correct) {
    extern __shared__ int32_t cnt[];
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < na * nb; i += 256) cnt[i] = 0;
    __syncthreads();
    for (int row = blockIdx.x * 4 + wave; row < Q; row += gridDim.x * 4) {
        float di[NV], dt[NV], ei[NV], et[NV];
        load_row<NV>(d2i + (size_t)row * ldd, N, lane, di);
        load_row<NV>(d2t + (size_t)row * ldd, N, lane, dt);
        float mni, mxi, mnt, mxt;
        row_minmax<NV>(di, N, lane, mni, mxi);
        row_minmax<NV>(dt, N, lane, mnt, mxt);
        const int label = labels[row];
        for (int ib = 0; ib < nb; ++ib) {
            const float beta = betas[ib];
            const float li = softmax_terms<NV>(di, beta, mni, mxi, N, lane, ei);
            const float lt = softmax_terms<NV>(dt, beta, mnt, mxt, N, lane, et);
#pragma unroll
            for (int i = 0; i < NV; ++i) {
                ei[i] = __fdiv_rn(ei[i], li);
                et[i] = __fdiv_rn(et[i], lt);
            }
            for (int ia = 0; ia < na; ++ia) {
                const float a = alphas[ia], oma = omas[ia];
                float best = -1.f; int besti = 0x7fffffff;
#pragma unroll
                for (int i = 0; i < NV; ++i) {
                    const int n = i * 64 + lane;
                    const float pv = __fadd_rn(__fmul_rn(a, ei[i]), __fmul_rn(oma, et[i]));
                    if (n < N && pv > best) { best = pv; besti = n; }
                }
                wave_argmax(best, besti);
                if (lane == 0 && besti == label) atomicAdd(&cnt[ia * nb + ib], 1);
            }
        }
    }
    __syncthreads();
    for (int i = threadIdx.x; i < na * nb; i += 256)
        if (cnt[i]) atomicAdd(&correct[i], cnt[i]);
}

// ---- small class counts (N <= 32): pclip_classify_small.h -------------------------------------------------------
template <int NT, bool TWO>
__global__ __launch_bounds__(256) void classify_small_kernel(const half_t* __restrict__ q, const half_t* __restrict__ zi,
                                                             const half_t* __restrict__ zt, int Q, int N, int D, float alpha,
                                                             float oma, float beta, float* __restrict__ p,
                                                             int32_t* __restrict__ argmax, float* __restrict__ topk_p,
                                                             int32_t* __restrict__ topk_i, int k) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    classify_small_body<NT, TWO>(smem, blockIdx.x, gridDim.x, q, zi, zt, Q, N, D, alpha, oma, beta, p, argmax, topk_p, topk_i, k);
}

template <int NT, bool TWO>
int launch_classify_small(const void* q, const void* zi, const void* zt, int Q, int N, int D, float alpha, float oma, float beta,
                          float* p, int32_t* argmax, float* topk_p, int32_t* topk_i, int k, int cus, hipStream_t s) {
    constexpr int NWAVES = 4, GPW = TWO ? NWAVES / 2 : NWAVES;      // four waves: two (visual, textual) pairs, or four groups with one bank
    const int ngroups = ceil_div(Q, 16);
    int grid = ceil_div(ngroups, GPW);
    if (grid > 2 * cus) grid = 2 * cus;                             // ~200 registers per lane: two workgroups per CU; larger Q loops
    classify_small_kernel<NT, TWO><<<grid, NWAVES * 64, TWO ? classify_small_lds(NT, NWAVES) : 0, s>>>((const half_t*)q, (const half_t*)zi, (const half_t*)zt, Q, N, D,
                                                                                                        alpha, oma, beta, p, argmax, topk_p, topk_i, k);
    return pclip_check_launch("classify (small N)");
}

inline int row_grid(int R, int cap) { int g = ceil_div(R, 4); return g < 1 ? 1 : (g > cap ? cap : g); }

struct SqWs { float *q_sq, *zi_sq, *zt_sq; size_t bytes; };
inline SqWs carve_sq(void* ws, int Q, int N) {
    SqWs w;
    char* b = (char*)ws;
    size_t o = 0;
    w.q_sq = (float*)(b + o); o += align_up((size_t)Q * 4, 256);
    w.zi_sq = (float*)(b + o); o += align_up((size_t)N * 4, 256);
    w.zt_sq = (float*)(b + o); o += align_up((size_t)N * 4, 256);
    w.bytes = o;
    return w;
}
inline int padded_ld(int N) { return (N + 63) / 64 * 64; }

}  // namespace

#define DISPATCH_NV(N, CALL)                                   \
    do {                                                       \
        if ((N) <= 64) { constexpr int NV = 1; CALL; }         \
        else if ((N) <= 256) { constexpr int NV = 4; CALL; }   \
        else if ((N) <= 1024) { constexpr int NV = 16; CALL; } \
        else { constexpr int NV = 64; CALL; }                  \
    } while (0)

extern "C" size_t pclip_workspace_bytes(int op, int Q, int N, int D) {
    (void)D;
    if (Q < 0 || N < 0) return 0;
    size_t sq = carve_sq(nullptr, Q, N).bytes;
    switch (op) {
        case PCLIP_OP_SQDIST: return sq;
        case PCLIP_OP_CLASSIFY: return sq + 2 * align_up((size_t)Q * padded_ld(N) * 4, 256);
        case PCLIP_OP_ADAPTER_FC: {
            // h1 [Q, D/4] fp16, h1n [Q, D/4] fp16, h2 [Q, D] fp16  (N is the hidden width here)
            return 2 * align_up((size_t)Q * N * 2, 256) + align_up((size_t)Q * D * 2, 256);
        }
        default: return 0;
    }
}

extern "C" int pclip_sqdist_f16(const void* q, const void* zi, const void* zt, int Q, int N, int D,
                                const float* q_sq, const float* zi_sq, const float* zt_sq, float* d2i, float* d2t,
                                int ldd, void* ws, size_t ws_bytes, pclip_stream_t stream) {
    PCLIP_REQUIRE(q && zi && d2i, "pclip_sqdist_f16: null pointer");
    PCLIP_REQUIRE(!zt || d2t, "pclip_sqdist_f16: zt given without d2t");
    PCLIP_REQUIRE(Q >= 0 && N > 0 && ldd >= N && ldd % 4 == 0, "pclip_sqdist_f16: bad Q=%d N=%d ldd=%d (ldd %% 4 == 0)", Q, N, ldd);
    PCLIP_REQUIRE(D > 0 && D % 64 == 0 && D <= 4096, "pclip_sqdist_f16: D=%d must be a multiple of 64, <= 4096", D);
    if (Q == 0) return PCLIP_OK;
    hipStream_t s = (hipStream_t)stream;
    if (!q_sq || !zi_sq || (zt && !zt_sq)) {
        SqWs w = carve_sq(ws, Q, N);
        PCLIP_REQUIRE(ws != nullptr, "pclip_sqdist_f16: workspace required when norms are not supplied");
        if (ws_bytes < w.bytes) { pclip_set_error("pclip_sqdist_f16: workspace %zu < %zu", ws_bytes, w.bytes); return PCLIP_E_WORKSPACE; }
        int e;
        if (!q_sq) { if ((e = pclip_row_sqnorm_f16(q, Q, D, w.q_sq, stream))) return e; q_sq = w.q_sq; }
        if (!zi_sq) { if ((e = pclip_row_sqnorm_f16(zi, N, D, w.zi_sq, stream))) return e; zi_sq = w.zi_sq; }
        if (zt && !zt_sq) { if ((e = pclip_row_sqnorm_f16(zt, N, D, w.zt_sq, stream))) return e; zt_sq = w.zt_sq; }
    }
    {   // enough 256x256 tiles to fill the chip a few times over: persistent big-tile kernel
        using CB = pgemm::Cfg<256, 256, 2, 4>;
        const int tm = ceil_div(Q, CB::BM), tn = ceil_div(N, CB::BN), per_bank = tm * tn, ntiles = per_bank * (zt ? 2 : 1);
        static int cus = 0;
        static int big_ok = -1;
        if (!cus) {
            cus = pclip_device_cus();
            if (cus <= 0) cus = 256;
            const char* e = getenv("PCLIP_SQDIST_BIG");
            big_ok = e ? atoi(e) : 1;
        }
        if (big_ok && ntiles >= 3 * cus && D >= 128 && Q % 4 == 0 && N % 4 == 0 && Q >= 4 && N >= 4) {
            static DevOnce attr;
            if (!attr.done()) {
                if (hipFuncSetAttribute((const void*)sqdist_big_kernel, hipFuncAttributeMaxDynamicSharedMemorySize, CB::LDS_BYTES + 4096) != hipSuccess) {
                    pclip_set_error("pclip_sqdist_f16: cannot raise the dynamic LDS limit to %d", CB::LDS_BYTES + 4096);
                    return PCLIP_E_LAUNCH;
                }
                attr.set();
            }
            sqdist_big_kernel<<<ntiles < cus ? ntiles : cus, 512, CB::LDS_BYTES + 4096, s>>>((const half_t*)q, (const half_t*)zi, (const half_t*)zt, Q, N, D,
                                                                                    q_sq, zi_sq, zt_sq, d2i, d2t, ldd, tn, per_bank, ntiles);
            return pclip_check_launch("sqdist (big tile)");
        }
    }
    const int tiles_m = ceil_div(Q, pgemm::CfgSmall::BM), tiles_n = ceil_div(N, pgemm::CfgSmall::BN);
    dim3 grid(tiles_m * tiles_n, zt ? 2 : 1);
    sqdist_kernel<<<grid, 256, pgemm::CfgSmall::LDS_BYTES, s>>>((const half_t*)q, (const half_t*)zi, (const half_t*)zt, Q, N, D,
                                                      q_sq, zi_sq, zt_sq, d2i, d2t, ldd, tiles_n);
    return pclip_check_launch("sqdist");
}

extern "C" int pclip_sqdist_f32(const float* q, const float* zi, const float* zt, int Q, int N, int D, float* d2i,
                                float* d2t, int ldd, pclip_stream_t stream) {
    PCLIP_REQUIRE(q && zi && d2i, "pclip_sqdist_f32: null pointer");
    PCLIP_REQUIRE(!zt || d2t, "pclip_sqdist_f32: zt given without d2t");
    PCLIP_REQUIRE(Q >= 0 && N > 0 && D > 0 && ldd >= N, "pclip_sqdist_f32: bad Q=%d N=%d D=%d ldd=%d", Q, N, D, ldd);
    if (Q == 0) return PCLIP_OK;
    dim3 grid(ceil_div(N, 64), ceil_div(Q, 64));
    sqdist_f32_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(q, zi, Q, N, D, d2i, ldd);
    if (zt) sqdist_f32_kernel<<<grid, 256, 0, (hipStream_t)stream>>>(q, zt, Q, N, D, d2t, ldd);
    return pclip_check_launch("sqdist_f32");
}

extern "C" int pclip_fuse_probs(const float* d2i, const float* d2t, int Q, int N, int ldd, float alpha,
                                float one_minus_alpha, float beta, float* p, int32_t* argmax, float* topk_p,
                                int32_t* topk_i, int k, pclip_stream_t stream) {
    PCLIP_REQUIRE(d2i, "pclip_fuse_probs: null distances");
    PCLIP_REQUIRE(Q >= 0 && N > 0 && N <= 4096 && ldd >= N, "pclip_fuse_probs: bad Q=%d N=%d (<=4096) ldd=%d", Q, N, ldd);
    PCLIP_REQUIRE(k >= 0 && k <= 16 && k <= N, "pclip_fuse_probs: k=%d must be in [0, min(16, N)]", k);
    PCLIP_REQUIRE((!topk_p && !topk_i) || k > 0, "pclip_fuse_probs: top-k output without k");
    if (Q == 0) return PCLIP_OK;
    DISPATCH_NV(N, (fuse_probs_kernel<NV><<<row_grid(Q, 16384), 256, 0, (hipStream_t)stream>>>(
                       d2i, d2t, Q, N, ldd, alpha, one_minus_alpha, beta, p, argmax, topk_p, topk_i, k)));
    return pclip_check_launch("fuse_probs");
}

extern "C" int pclip_classify_f16(const void* q, const void* zi, const void* zt, int Q, int N, int D,
                                  const float* q_sq, const float* zi_sq, const float* zt_sq, float alpha,
                                  float one_minus_alpha, float beta, float* p, int32_t* argmax, float* topk_p,
                                  int32_t* topk_i, int k, void* ws, size_t ws_bytes, pclip_stream_t stream) {
    PCLIP_REQUIRE(ws != nullptr, "pclip_classify_f16: workspace required");
    const size_t need = pclip_workspace_bytes(PCLIP_OP_CLASSIFY, Q, N, D);
    if (ws_bytes < need) { pclip_set_error("pclip_classify_f16: workspace %zu < %zu", ws_bytes, need); return PCLIP_E_WORKSPACE; }
    if (Q == 0) return PCLIP_OK;
    // mid-sized class counts (32 < N <= 256; tests / probes: any N <= 256), p and / or argmax: one launch, norms in-kernel (pclip_classify_mid.hip).  PCLIP_CLASSIFY_MID=0: off.
    if (q && zi && (p || argmax) && pclip_classify_mid_applies(Q, N, D, zt != nullptr, topk_p || topk_i || k > 0))
        return pclip_classify_mid_launch(q, zi, zt, Q, N, D, alpha, one_minus_alpha, beta, p, argmax, (hipStream_t)stream);
    {   // small class counts: one launch (env PCLIP_CLASSIFY_SMALL=0 switches it off)
        static int mode = -1, cus = 0;
        if (mode < 0) {
            const char* e = getenv("PCLIP_CLASSIFY_SMALL");
            mode = e ? atoi(e) : 1;
            cus = pclip_device_cus();
            if (cus <= 0) cus = 256;
        }
        const int nt = N <= 16 ? 1 : 2;
        if (mode > 0 && N > 0 && N <= 32 && D > 0 && D % 32 == 0 && q && zi && k >= 0 && k <= N && k <= 16 &&
            ((!topk_p && !topk_i) || k > 0)) {
            hipStream_t s = (hipStream_t)stream;
#define PCLIP_SMALL(NT)                                                                                                                   \
    return zt ? launch_classify_small<NT, true>(q, zi, zt, Q, N, D, alpha, one_minus_alpha, beta, p, argmax, topk_p, topk_i, k, cus, s) \
              : launch_classify_small<NT, false>(q, zi, zt, Q, N, D, alpha, one_minus_alpha, beta, p, argmax, topk_p, topk_i, k, cus, s)
            if (nt == 1) { PCLIP_SMALL(1); }
            PCLIP_SMALL(2);
#undef PCLIP_SMALL
        }
    }
    SqWs w = carve_sq(ws, Q, N);
    // large class counts, argmax only: the fused row-panel kernel (pclip_classify_panel.hip) — no distance rows in HBM.  PCLIP_CLASSIFY_PANEL=0: two stages.
    if (zt && argmax && !p && !topk_p && !topk_i && q && zi && pclip_classify_panel_applies(Q, N, D, alpha, one_minus_alpha, beta) &&
        ws_bytes >= w.bytes + pclip_classify_panel_workspace(Q, N, D)) {
        // (norms that were not supplied are formed by the kernel's own preparation launch: the arithmetic of pclip_row_sqnorm_f16)
        return pclip_classify_panel_launch(q, zi, zt, Q, N, D, q_sq, zi_sq, zt_sq, alpha, one_minus_alpha, beta, argmax, nullptr, false, (char*)ws + w.bytes, (hipStream_t)stream);
    }
    const int ldd = padded_ld(N);
    float* d2i = (float*)((char*)ws + w.bytes);
    float* d2t = zt ? (float*)((char*)d2i + align_up((size_t)Q * ldd * 4, 256)) : nullptr;
    int e = pclip_sqdist_f16(q, zi, zt, Q, N, D, q_sq, zi_sq, zt_sq, d2i, d2t, ldd, ws, w.bytes, stream);
    if (e) return e;
    return pclip_fuse_probs(d2i, d2t, Q, N, ldd, alpha, one_minus_alpha, beta, p, argmax, topk_p, topk_i, k, stream);
}

// Which kernels pclip_classify_f16 takes for a call of this shape under the current settings (the routes differ in fp32 summation order — the same
// query can get a different argmax at a near-tie of p (margin < 1e-6) depending on the batch it travels in — so callers can ask): 0 two stages (pclip_sqdist_f16 +
// pclip_fuse_probs: torch.cdist's arithmetic operation for operation), 1 one launch for N <= 16 / top-k up to N = 32 (classify_small), 2 one launch for 16 < N <= 256
// (classify_mid), 3 fused row panels (argmax only, large Q N; distances without cdist's sqrt round trip unless PCLIP_CLASSIFY_PANEL_EXACT=1).
extern "C" int pclip_classify_route(int Q, int N, int D, float alpha, float one_minus_alpha, float beta, int has_zt, int want_p, int want_argmax, int topk,
                                    size_t ws_bytes) {
    if (Q <= 0 || N <= 0) return 0;
    if ((want_p || want_argmax) && pclip_classify_mid_applies(Q, N, D, has_zt != 0, topk > 0)) return 2;
    static const int small_mode = getenv("PCLIP_CLASSIFY_SMALL") ? atoi(getenv("PCLIP_CLASSIFY_SMALL")) : 1;
    if (small_mode > 0 && N <= 32 && D > 0 && D % 32 == 0 && topk >= 0 && topk <= N && topk <= 16) return 1;
    if (has_zt && want_argmax && !want_p && topk == 0 && pclip_classify_panel_applies(Q, N, D, alpha, one_minus_alpha, beta) &&
        ws_bytes >= carve_sq(nullptr, Q, N).bytes + pclip_classify_panel_workspace(Q, N, D))
        return 3;
    return 0;
}

// Test entry: the distances the fused row-panel kernel forms for its first tile (query rows 0 .. 255 x classes 0 .. 127 of both banks, [2][256][128] fp32): exact != 0
// — with torch.cdist's sqrt -> square round trip — they must be the bits pclip_sqdist_f16 writes, exact == 0 (the product's arithmetic) within one fp32 ulp of them.
extern "C" int pclip_classify_panel_dump_f16(const void* q, const void* zi, const void* zt, int Q, int N, int D, float* dump, int exact, void* ws, size_t ws_bytes,
                                             pclip_stream_t stream) {
    PCLIP_REQUIRE(q && zi && zt && dump && ws, "pclip_classify_panel_dump_f16: null pointer");
    PCLIP_REQUIRE(N > 32 && D >= 128 && D % 64 == 0 && D <= 4096 && Q >= 1, "pclip_classify_panel_dump_f16: shape outside the fused kernel");
    SqWs w = carve_sq(ws, Q, N);
    if (ws_bytes < w.bytes + pclip_classify_panel_workspace(Q, N, D)) { pclip_set_error("pclip_classify_panel_dump_f16: workspace too small"); return PCLIP_E_WORKSPACE; }
    int e;
    if ((e = pclip_row_sqnorm_f16(q, Q, D, w.q_sq, stream))) return e;
    if ((e = pclip_row_sqnorm_f16(zi, N, D, w.zi_sq, stream))) return e;
    if ((e = pclip_row_sqnorm_f16(zt, N, D, w.zt_sq, stream))) return e;
    return pclip_classify_panel_launch(q, zi, zt, Q, N, D, w.q_sq, w.zi_sq, w.zt_sq, 0.5f, 0.5f, 1.f, nullptr, dump, exact != 0, (char*)ws + w.bytes, (hipStream_t)stream);
}

extern "C" int pclip_hp_sweep(const float* d2i, const float* d2t, const int32_t* labels, int Q, int N, int ldd,
                              const float* alphas, const float* one_minus_alphas, int na, const float* betas, int nb,
                              int32_t* correct, pclip_stream_t stream) {
    PCLIP_REQUIRE(d2i && d2t && labels && alphas && one_minus_alphas && betas && correct, "pclip_hp_sweep: null pointer");
    PCLIP_REQUIRE(Q >= 0 && N > 0 && N <= 4096 && ldd >= N, "pclip_hp_sweep: bad Q=%d N=%d (<=4096) ldd=%d", Q, N, ldd);
    PCLIP_REQUIRE(na > 0 && nb > 0 && na * nb <= 8192, "pclip_hp_sweep: bad grid %d x %d", na, nb);
    if (Q == 0) return PCLIP_OK;
    DISPATCH_NV(N, (hp_sweep_kernel<NV><<<row_grid(Q, 4096), 256, (size_t)na * nb * 4, (hipStream_t)stream>>>(
                       d2i, d2t, labels, Q, N, ldd, alphas, one_minus_alphas, na, betas, nb, correct)));
    return pclip_check_launch("hp_sweep");
}
