// nn.Linear / convolution-as-GEMM of the CLIP towers (clip/model.py:176-190, 43-52): the eight-wave persistent MFMA kernels with their fused
// epilogues, the latency-oriented ring kernel + split-K, the implicit-GEMM 3x3 convolution, the tile-choice / row-split dispatch and the pclip_gemm_* entry points.
// (The four-wave asm-loop kernel that takes the 256 x 256 tiles lives in pclip_gemm4w.hip.)
#include "pclip_encoder_common.h"
#include <stdlib.h>
#include <type_traits>

unsigned long long* pclip_gemm_time_slot();       // the measurement hook's next slot, or null (defined below: pclip_gemm_timing)

namespace {
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
                                                                     int ntiles, const half_t* residual = nullptr, int band = 0,
                                                                     unsigned long long* tslot = nullptr) {
    // ACT 5: relu(r16(r16(r16(acc) * scale + shift) + residual)) — bn3 + `out += identity` + ReLU of a bottleneck (clip/model.py:49-52)
    // in the epilogue of its conv3 GEMM; the residual rows are read row-major in the coalesced store pass.
    // ACT 6: r16(residual + r16(acc + bias)) — `x = x + attn(..)` / `x = x + mlp(..)` of a transformer block (clip/model.py:188-189)
    // in the epilogue of out_proj / c_proj; Cout may BE residual (the residual stream is updated in place: every 16-byte chunk is
    // read and then written by the same thread)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr bool M16 = true;                                               // 16x16x32 MFMAs (accumulator layout of pgemm::mainloop_sr)
    half_t* bias_lds = reinterpret_cast<half_t*>(smem + C::LDS_BYTES);       // [2][BN] fp16
    float* affine_lds = reinterpret_cast<float*>(smem + C::LDS_BYTES);       // ACT >= 2: [2][ scale BN | shift BN ] fp32
    constexpr bool AFFINE = ACT == 2 || ACT == 3 || ACT == 5;
    const int G = gridDim.x;
    int tile = pgemm::xcd_remap(blockIdx.x, G);
    if (tile >= ntiles) return;
    pgemm::time_begin(tslot);                                                 // (measurement hook: null in the product)
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
    const pgemm::rsrc_t rs_bias = pgemm::make_rsrc(bias, 0x7fffffffu), rs_scale = pgemm::make_rsrc(scale, 0x7fffffffu), rs_shift = pgemm::make_rsrc(shift, 0x7fffffffu);
    (void)rs_bias; (void)rs_scale; (void)rs_shift;
    auto copy_bias = [&](int t, int par) {
        int tm_, tn;
        decomp(t, tm_, tn);
        (void)tn;
#if defined(__HIP_DEVICE_COMPILE__)
        if (lane < C::BN / 8)      // (buffer LDS-DMA like the tiles: a FLAT-encoded global_load_lds in flight turns every LDS wait of the first K-tile into lgkmcnt(0): pclip_gemm.h make_rsrc)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_bias, (pgemm::lds_ptr_t)(bias_lds + par * C::BN), 16, (tn * C::BN + lane * 8) * 2, 0, 0, 0);
#endif
    };
    // eval-mode BatchNorm (+ReLU) of the ResNet tower (clip/model.py:43-52) as the epilogue of the convolution's GEMM: the
    // per-column scale / shift strips travel like the bias strip, one tile ahead
    auto copy_affine = [&](int t, int par) {
        int tm_, tn;
        decomp(t, tm_, tn);
        (void)tn;
#if defined(__HIP_DEVICE_COMPILE__)
        if (lane < C::BN / 4) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_scale, (pgemm::lds_ptr_t)(affine_lds + par * 2 * C::BN), 16, (tn * C::BN + lane * 4) * 4, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_shift, (pgemm::lds_ptr_t)(affine_lds + par * 2 * C::BN + C::BN), 16, (tn * C::BN + lane * 4) * 4, 0, 0, 0);
        }
#endif
    };
    if (HAS_BIAS || AFFINE) {
        if (AFFINE) copy_affine(tile, 0); else copy_bias(tile, 0);
        pgemm::wait_vm<0>();
        pgemm::lds_barrier();
    }
    {
        int tm, tn;
        decomp(tile, tm, tn);
        tp.prepare(A, lda, B, ldb, M, N, tm * C::BM, tn * C::BN, wave, lane);
        tp.stage(0, smem + p * C::STAGE_BYTES, wave);
    }
    // vector-memory operations a wave issues between a tile's K-tile 0 pieces and the first wait of its K-loop: the previous tile's stores + the strip copies
    constexpr int YOUNGER = C::NH * C::NPASS + (AFFINE ? 2 : (HAS_BIAS ? 1 : 0));
    bool prev_full = false;
    int parity = 0;
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
        // (Measured and rejected, profiles/r03_ab_rejected.txt: pulling the residual tile's 1024 lines into L2 during the K-loop with one
        // 4-byte LDS-DMA per line — out_proj 301 -> 341 us, c_proj 854 -> 879 us: 1024 more requests per tile in the queue the operand
        // DMAs wait in.)
        const int next = tile + G;
        pgemm::mainloop_sr<C, YOUNGER, !HAS_BIAS, TP>(tp, K / pgemm::BK, smem, acc, p, prev_full, wave, lane);
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
        auto pre = [&](int i, int j, int coff, float4_t v, int rl, int g) {
            if (ACT == 1) return quick_gelu16x4(v);
            half4_t h;
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
        constexpr bool RES = ACT == 5 || ACT == 6;
        // (measured, profiles/r03_ab_epilogue_pipe.txt: c_fc + QuickGELU 1001 -> 972 us; the bias-only and residual epilogues do not profit — their phases
        // are bound by the LDS write rate / the stores' address path / the residual loads' latency one after the other either way — and keep epilogue_f16)
        constexpr bool PIPE = PCLIP_EPI_PIPE && (ACT == 1 || PCLIP_EPI_PIPE == 2) && C::BM == 256 && C::BN == 256 && C::WM == 2 && C::WN == 4;
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
        if constexpr (PIPE) {
            static_assert(C::NPASS == 8, "rr[pass % NPASS] pairs slab parity and pass");
            if (full)
                pgemm::epilogue_pipe<C>(acc, stg, ahead, pre, [&](int r, int c, int pass, half8_t h) {
                    const size_t o = (size_t)(m0 + r) * ldc + col;
                    if (RES) h = add_res(pass, h);
                    st_out(Cout + o, h);
                });
            else
                pgemm::epilogue_pipe<C>(acc, stg, ahead, pre, [&](int r, int c, int pass, half8_t h) {
                    const size_t o = (size_t)(m0 + r) * ldc + col;
                    if (RES) h = add_res(pass, h);
                    if (m0 + r < M) st_out(Cout + o, h);
                });
            prev_full = full;
            continue;
        }
        if (full)
            pgemm::epilogue_f16<C, M16>(acc, stg, slab, pre, [&](int r, int c, int pass, half8_t h) {
                const size_t o = (size_t)(m0 + r) * ldc + col;
                if (RES) h = add_res(pass, h);
                st_out(Cout + o, h);
            });
        else
            pgemm::epilogue_f16<C, M16>(acc, stg, slab, pre, [&](int r, int c, int pass, half8_t h) {
                const size_t o = (size_t)(m0 + r) * ldc + col;
                if (RES) h = add_res(pass, h);
                if (m0 + r < M) st_out(Cout + o, h);
            });
        prev_full = full;
    }
    pgemm::time_end(tslot);
}

// ---- 3x3 convolution (stride 1, pad 1, NHWC) + eval BatchNorm (+ReLU) as an implicit GEMM -----------------------------------
// Same persistent structure as linear_fast_kernel; the A operand is gathered by pgemm::ConvGather instead of read from an
// im2col matrix (clip/model.py:20-22, 45-46: conv2 / bn2 / relu of every bottleneck).  w is [Cout, ky, kx, Cin].
// K-loop: the software-pipelined loop of the linears (pgemm::mainloop_sr: fragments a group ahead, K-tile t + 2 requested in two halves during iteration t) over
// pgemm::ConvPair (round 6; before: mainloop_g's read-everything-then-multiply loop — the same k order, the same bits, RN50 +2 % in a same-box A/B).
template <class C, int ACT>
__global__ __launch_bounds__(C::NTHREADS, 2) void conv3x3_fast_kernel(const half_t* __restrict__ x,
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
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6), wn = wave % C::WN;
    pgemm::ConvPair<C> cp(x, H, W, Cin, M);
    pgemm::ConvGather<C>& ga = cp.a.g;
    const pgemm::rsrc_t rs_scale = pgemm::make_rsrc(scale, 0x7fffffffu), rs_shift = pgemm::make_rsrc(shift, 0x7fffffffu);
    (void)rs_scale; (void)rs_shift;
    auto copy_affine = [&](int t, int par) {
        const int tn = t - (t / tiles_n) * tiles_n;
        (void)tn;
#if defined(__HIP_DEVICE_COMPILE__)
        if (lane < C::BN / 4) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_scale, (pgemm::lds_ptr_t)(affine_lds + par * 2 * C::BN), 16, (tn * C::BN + lane * 4) * 4, 0, 0, 0);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs_shift, (pgemm::lds_ptr_t)(affine_lds + par * 2 * C::BN + C::BN), 16, (tn * C::BN + lane * 4) * 4, 0, 0, 0);
        }
#endif
    };
    auto stage0 = [&](int t, int pbuf) {                       // K-tile 0 of tile t into buffer pbuf (ga prepared for t)
        const int tm = t / tiles_n, tn = t - tm * tiles_n;
        char* a = smem + pbuf * C::STAGE_BYTES;
        cp.b.prepare(w, ldb, tn * C::BN, N, wave, lane);
        cp.stage(0, a, wave);
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
        pgemm::mainloop_sr<C, YOUNGER, true, pgemm::ConvPair<C>>(cp, nt, smem, acc, p, prev_full, wave, lane);
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
struct TileOrder { int band, rev, band_n; };      // band: launches with >= 8 column tiles (c_fc); band_n: narrower ones (in_proj: 9 -> 8 counts as wide; out_proj / c_proj: 3)
static TileOrder read_tile_order() {
    const char* b = getenv("PCLIP_GEMM_BAND");
    const char* r = getenv("PCLIP_GEMM_REV");
    const char* bn = getenv("PCLIP_GEMM_BAND_N");
    return TileOrder{b ? atoi(b) : 0, r ? atoi(r) : 2, bn ? atoi(bn) : 0};
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
    constexpr int LDS = C::LDS_BYTES + ((ACT == 2 || ACT == 3 || ACT == 5) ? 2 * 2 * C::BN * 4 : 2 * C::BN * 2) + 256;   // K-tile ring + double-buffered bias / affine strips + prefetch scrap
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
    const bool rev = rev_mode == 1 || (rev_mode == 2 && K <= 1024);
    linear_fast_kernel<C, HAS_BIAS, ACT><<<grid, C::NTHREADS, LDS, s>>>(
        (const half_t*)A, lda, (const half_t*)B, ldb, M, N, K, epi.bias, epi.scale, epi.shift, epi.C, epi.ldc, tiles_n, ntiles, epi.residual,
        rev ? -1 : (tiles_n >= 8 ? band : 0), pclip_gemm_time_slot());
    return pclip_check_launch("gemm_f16");
}

template <class C>
static int launch_fast(const void* A, int lda, const void* B, int ldb, int M, int N, int K, const LinearEpi& epi,
                       int slots, hipStream_t s) {
    if (epi.act == 2) return launch_fast2<C, false, 2>(A, lda, B, ldb, M, N, K, epi, slots, s);
    if (epi.act == 3) return launch_fast2<C, false, 3>(A, lda, B, ldb, M, N, K, epi, slots, s);
    if (epi.act == 5) return launch_fast2<C, false, 5>(A, lda, B, ldb, M, N, K, epi, slots, s);
    if (epi.act == 6) return launch_fast2<C, true, 6>(A, lda, B, ldb, M, N, K, epi, slots, s);
    if (epi.bias) {
        if (epi.act == 1) return launch_fast2<C, true, 1>(A, lda, B, ldb, M, N, K, epi, slots, s);
        return launch_fast2<C, true, 0>(A, lda, B, ldb, M, N, K, epi, slots, s);
    }
    if (epi.act == 1) return launch_fast2<C, false, 1>(A, lda, B, ldb, M, N, K, epi, slots, s);
    return launch_fast2<C, false, 0>(A, lda, B, ldb, M, N, K, epi, slots, s);
}

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
// Timing buffer of the measurement hook (pgemm::time_begin / time_end): [nslots][2] unsigned 64-bit of device memory, begin slots pre-set to ~0, end slots to 0 by
// the caller; every instrumented GEMM launch takes the next slot.  Null = off (the product).
static unsigned long long* g_time_buf = nullptr;
static int g_time_cap = 0, g_time_next = 0;
unsigned long long* pclip_gemm_time_slot() {
    if (!g_time_buf || g_time_next >= g_time_cap) return nullptr;
    return g_time_buf + 2 * (size_t)g_time_next++;
}
extern "C" int pclip_gemm_timing(void* buf, int nslots) {
    g_time_buf = (unsigned long long*)buf;
    g_time_cap = buf ? nslots : 0;
    g_time_next = 0;
    return PCLIP_OK;
}
extern "C" int pclip_gemm_timing_count(void) { return g_time_next; }
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

inline double tile_cost(const TileCfg& c, long M, int N, int cus) {
    if (N % c.bn) return 1e30;
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
                        const half_t* residual, int slots, int rev, int band, hipStream_t s);
namespace {
int gemm_dispatch(const half_t* A, int lda, const half_t* B, int ldb, int M, int N, int K, LinearEpi epi, int cus, int forced,
                  bool may_split, hipStream_t s) {
    const bool aligned = (!epi.residual || ((epi.act == 5 || epi.act == 6) && ((uintptr_t)epi.residual & 15) == 0)) && epi.ldc % 8 == 0 && ((uintptr_t)epi.C & 15) == 0 &&
                         (!epi.bias || ((uintptr_t)epi.bias & 15) == 0);
    static const bool small_on = !(getenv("PCLIP_GEMM_SMALL") && getenv("PCLIP_GEMM_SMALL")[0] == '0');
    if (aligned && forced == -1 && small_on && (epi.act <= 1 || epi.act == 6 || (((uintptr_t)epi.scale | (uintptr_t)epi.shift) & 15) == 0) && small_applies(M, N, cus))
        return launch_small_one(A, lda, B, ldb, M, N, K, epi, s);
    double cost = 1e30;
    int pick = aligned ? best_cfg(M, N, cus, &cost) : -1;
    if (forced == -2) { may_split = false; pick = -1; }        // generic kernel
    if (epi.act == 5 && pick < 0) { pclip_set_error("pclip_gemm_bn_res_f16: N=%d / alignment not supported by the fused epilogue", N); return PCLIP_E_INVALID; }
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
            return gemm_dispatch(A + (size_t)split_rows * lda, lda, B, ldb, M - (int)split_rows, N, K, tail, cus, -1, true, s);
        }
    }
    ++g_gemm_launches;
    if (pick < 0 && epi.act == 6) epi.act = 0;                  // generic kernel: bias + residual operands, same roundings
    if (pick == 2) {
        // the same tile on four waves with the hand-scheduled K-loop (bit-identical): PCLIP_GEMM_4W=1 (default: see DESIGN §3)
        if (g_use4w < 0) { const char* e = getenv("PCLIP_GEMM_4W"); g_use4w = e ? (atoi(e) != 0) : PCLIP_GEMM_4W_DEFAULT; }
        if (g_use4w && pclip_gemm4w_supports(M, N, K, lda, ldb, epi.ldc, epi.C, epi.bias, epi.residual, epi.act)) {
            const TileOrder& order = tile_order();
            const bool rev = order.rev == 1 || (order.rev == 2 && K <= 1024);
            return pclip_gemm4w_launch(A, lda, B, ldb, M, N, K, epi.bias, epi.C, epi.ldc, epi.act, epi.residual, cus, rev ? 1 : 0, N / 256 >= 8 ? order.band : order.band_n, s);
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
                                                                           const half_t* residual = nullptr, unsigned long long* tslot = nullptr) {
    using C = CfgSplit;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    pgemm::time_begin(tslot);                                                 // (measurement hook: null in the product)
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
            if ((ACT == 5 || ACT == 6) && valid) {
                const half8_t rr = ld_half8(residual + o);
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    const float y = r16((float)rr[j] + (float)h[j]);
                    h[j] = (half_t)(ACT == 5 ? fmaxf(y, 0.f) : y);
                }
            }
            if (valid) st_half8(Cout + o, h);
        });
        pgemm::time_end(tslot);
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
__global__ __launch_bounds__(CfgSplit::NTHREADS, 1) void conv3x3_small_kernel(const half_t* __restrict__ x,
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
    pgemm::ConvGather<C> ga(x, H, W, Cin, M);
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
    unsigned long long* tslot = pclip_gemm_time_slot();
#define PCLIP_SMALL_LAUNCH(ACT)                                                                                                          \
    linear_small_kernel<ACT><<<grid, CfgSplit::NTHREADS, kSmallLds, s>>>(A, lda, B, ldb, M, N, K, tiles_n, 1, steps, nullptr, epi.bias, epi.C, \
                                                                        epi.ldc, epi.scale, epi.shift, epi.residual, tslot)
    if (epi.act == 5) PCLIP_SMALL_LAUNCH(5);
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

// tuning override of the BatchNorm-epilogue GEMMs (A/B runs): PCLIP_GEMM_BN_CFG = 0 small (128 x 128, two workgroups per CU), 1 wide, 2 big, 4 narrow; default: the cost model
static int bn_forced_cfg() {
    static const int f = getenv("PCLIP_GEMM_BN_CFG") ? atoi(getenv("PCLIP_GEMM_BN_CFG")) : -1;
    return f;
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
    return gemm_dispatch((const half_t*)A, lda, (const half_t*)B, ldb, M, N, K, epi, cus, bn_forced_cfg(), true, (hipStream_t)stream);
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
    return gemm_dispatch((const half_t*)A, lda, (const half_t*)B, ldb, M, N, K, epi, cus, strips_ok ? bn_forced_cfg() : -2, true, (hipStream_t)stream);
}

namespace {
template <class C, int ACT>
int launch_conv2(const void* x, const void* w, int B, int H, int W, int Cin, int Cout, const float* scale,
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
        (const half_t*)x, (const half_t*)w, H, W, Cin, M, Cout, scale, shift, (half_t*)y, tiles_n, ntiles);
    return pclip_check_launch("conv3x3_bn");
}
template <class C>
int launch_conv(const void* x, const void* w, int B, int H, int W, int Cin, int Cout, const float* scale,
                const float* shift, int relu, void* y, int slots, hipStream_t s) {
    return relu ? launch_conv2<C, 3>(x, w, B, H, W, Cin, Cout, scale, shift, y, slots, s)
                : launch_conv2<C, 2>(x, w, B, H, W, Cin, Cout, scale, shift, y, slots, s);
}
}  // namespace

// pclip_conv_strip.hip
extern "C" int pclip_conv3x3_strip_applies(int B, int H, int W, int Cin, int Cout);
int pclip_conv3x3_strip_launch(const void* x, const void* w, int B, int H, int W, int Cin, int Cout, const float* scale, const float* shift, int relu,
                               void* y, int cus, hipStream_t s, int pool);

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
    // the gather addresses the activations through a buffer descriptor with 32-bit offsets: a larger batch goes in slices of whole images (same kernels, same bits)
    const long per_image = (long)H * W * Cin * 2;
    PCLIP_REQUIRE(per_image < (1L << 31), "pclip_conv3x3_bn_f16: one image of %ld bytes is too large", per_image);
    if ((long)B * per_image >= (1L << 31)) {
        const int slice = (int)((1L << 31) / per_image);
        for (int b0 = 0; b0 < B; b0 += slice) {
            const int nb = B - b0 < slice ? B - b0 : slice;
            if (int e = pclip_conv3x3_bn_f16((const char*)x + (size_t)b0 * per_image, w, zero_line, nb, H, W, Cin, Cout, scale, shift, relu,
                                             (char*)y + (size_t)b0 * H * W * Cout * 2, stream))
                return e;
        }
        return PCLIP_OK;
    }
    if (pclip_conv3x3_strip_applies(B, H, W, Cin, Cout))                        // narrow layers at 56 x 56 / 112 x 112: weights in registers, halo blocks in LDS
        return pclip_conv3x3_strip_launch(x, w, B, H, W, Cin, Cout, scale, shift, relu, y, cus, s, 1);
    if (Cout == 32)                                                             // the stem's 32 -> 32 convolution: 256 x 32 tiles
        return launch_conv<CfgThin>(x, w, B, H, W, Cin, Cout, scale, shift, relu, y, 2 * cus, s);
    static const bool small_on = !(getenv("PCLIP_GEMM_SMALL") && getenv("PCLIP_GEMM_SMALL")[0] == '0');
    if (small_on && small_applies(B * H * W, Cout, cus)) {                     // a request of a few images: the ring kernel
        if (int e = small_attr()) return e;
        const int tiles_n = Cout / CfgSplit::BN, grid = ceil_div(B * H * W, CfgSplit::BM) * tiles_n;
        if (relu)
            conv3x3_small_kernel<3><<<grid, CfgSplit::NTHREADS, kSmallLds, s>>>((const half_t*)x, (const half_t*)w, H, W, Cin,
                                                                           B * H * W, Cout, scale, shift, (half_t*)y, tiles_n);
        else
            conv3x3_small_kernel<2><<<grid, CfgSplit::NTHREADS, kSmallLds, s>>>((const half_t*)x, (const half_t*)w, H, W, Cin,
                                                                           B * H * W, Cout, scale, shift, (half_t*)y, tiles_n);
        return pclip_check_launch("conv3x3_bn (small M)");
    }
    double cost;
    int pick = best_cfg((long)B * H * W, Cout, cus, &cost);
    if (pick == 4 || pick < 0) pick = 3;                        // the 4-wave thin tile has no gather variant; Cout % 64 == 0 always fits 256x64
    if (pick == 2) return launch_conv<CfgBig>(x, w, B, H, W, Cin, Cout, scale, shift, relu, y, cus, s);
    if (pick == 1) return launch_conv<CfgWide>(x, w, B, H, W, Cin, Cout, scale, shift, relu, y, cus, s);
    if (pick == 0) return launch_conv<CfgSmall>(x, w, B, H, W, Cin, Cout, scale, shift, relu, y, 2 * cus, s);
    return launch_conv<CfgNarrow>(x, w, B, H, W, Cin, Cout, scale, shift, relu, y, cus, s);
}