// Fused classification for LARGE class counts, argmax only (utils.py:225-244 `P` + main.py:190 `.max(1)[1]`).
// The two-stage path (pclip_sqdist_f16 -> fp32 [Q, ldd] x 2 in HBM -> pclip_fuse_probs) moves ~800 MB of distance rows for an ImageNet test split whose
// algorithmic traffic is 53 MB.  Here a workgroup owns a PANEL of 256 query rows and walks every class tile of both prototype banks TWICE on the matrix pipe:
//   pass 1: per query and bank the running minimum distance and the sum of exp(-beta (d2 - min)) (online softmax, in registers)
//   pass 2: the same distances again -> p = alpha e_i / S_i + (1 - alpha) e_t / S_t -> running (best p, class) per query
// No distance leaves the chip; the 2 MB of prototypes stay L2-resident.  The second contraction costs as many FLOPs again (2 x 102 GFLOP for ImageNet); what it
// buys is the 800 MB round trip and the second launch.
// ONE pass where the result can be proven (default since round 5's second half; pclip_classify_panel_passes): pass 1 also leaves, per (row, lane-slot, tile) group of
// 16 classes, a 32-byte record in the workspace — per bank the nearest class with BOTH its distances, and the group's second smallest distance.  The nearest classes
// are the candidates; every other class of a group is bounded through the second smallest distances (p is monotone in both).  When the largest bound of a row is
// below its best candidate's p, the candidates' argmax is the row's argmax — the bits pass 2 would produce — and a panel all of whose rows satisfy that skips pass 2.
// Class-structured data (the few-shot setting) proves every panel: ImageNet split 245 -> ~175 us; structureless rows (every distance alike, flat p) mostly take the
// second pass and pay the records on top (+13 %).
// Layout trick: both banks travel as ONE operand of 2 N rows, row 2 n = visual prototype n, row 2 n + 1 = textual prototype n (`interleave_kernel`, 2 MB, into the
// caller's workspace): in the (operand-swapped) accumulator layout a lane then holds BOTH banks' distances of a class in adjacent registers, so pass 2 mixes them
// without exchanging anything.  d2 is the expression of sqdist_kernel — (sqrt(max(||q||^2 + ||z||^2 - 2 q.z, 0)))^2 with the same fp32 norms and the same MFMA k
// order — i.e. bit-identical distances (tests/test_gpu_parity.py: a sampled tile); the softmax arithmetic is the online form (exp2 with beta log2 e folded in), so p
// agrees with pclip_fuse_probs to fp32 rounding and the argmax wherever the top-2 margin exceeds that.
#include "pclip_gemm.h"
#include "pclip_proto_dev.h"
#include <stdlib.h>
#include <type_traits>

typedef int int2_t __attribute__((ext_vector_type(2)));
typedef float float2_t __attribute__((ext_vector_type(2)));

namespace {
using CP = pgemm::Cfg<256, 256, 4, 2>;                   // 4 x 2 waves: a lane owns 4 query rows x 32 interleaved columns (16 classes x 2 banks) of a tile
constexpr float BIG_D2 = 1e30f;                           // squared norm of the padding prototypes: never the minimum, exp() == 0

// Operand preparation in ONE launch, a wave per row: rows [0, rows2) build zz[2 n + b] = z_b[n] (zero rows beyond N) with zz_sq (BIG_D2 beyond N), rows
// [rows2, rows2 + Qp) the padded query norms (whole 256-row panels: the strips are fetched by LDS-DMA without bounds).  Norms the caller did not supply are
// computed here with pclip_row_sqnorm_f16's arithmetic (load_row_sq of pclip_proto_dev.h: the same bits) — the path was five launches before the panels started.
template <int NCH>
__global__ __launch_bounds__(256) void panel_prep_kernel(const half_t* __restrict__ q, const half_t* __restrict__ zi, const half_t* __restrict__ zt,
                                                         const float* __restrict__ q_sq, const float* __restrict__ zi_sq, const float* __restrict__ zt_sq, int Q, int Qp,
                                                         int N, int D, int rows2, half_t* __restrict__ zz, float* __restrict__ zz_sq, float* __restrict__ q_sqp) {
    const int lane = threadIdx.x & 63;
    for (int r = blockIdx.x * 4 + (threadIdx.x >> 6); r < rows2 + Qp; r += gridDim.x * 4) {
        if (r < rows2) {
            const int n = r >> 1, b = r & 1;
            RowRegs<NCH> rr;
            float ss = BIG_D2;
            if (n < N) {
                const half_t* src = (b ? zt : zi) + (size_t)n * D;
                const float* sq = b ? zt_sq : zi_sq;
                ss = load_row_sq<NCH>(src, D, lane, rr);
                if (sq) ss = sq[n];
            } else {
#pragma unroll
                for (int c = 0; c < NCH; ++c) rr.v[c] = half8_t{};
            }
#pragma unroll
            for (int c = 0; c < NCH; ++c)
                if (c * 512 + lane * 8 < D) st_half8(zz + (size_t)r * D + c * 512 + lane * 8, rr.v[c]);
            if (lane == 0) zz_sq[r] = ss;
        } else {
            const int i = r - rows2;
            float ss = 0.f;
            if (i < Q) {
                if (q_sq) ss = q_sq[i];
                else { RowRegs<NCH> rr; ss = load_row_sq<NCH>(q + (size_t)i * D, D, lane, rr); }
            }
            if (lane == 0) q_sqp[i] = ss;
        }
    }
}

template <bool EXACT>
__device__ __forceinline__ float d2_of(float acc, float qs, float zs) {
    const float v = __fadd_rn(__fadd_rn(-2.f * acc, qs), zs);          // sqdist_kernel's expression, operation for operation
    if (!EXACT) return fmaxf(v, 0.f);
    const float d = sqrtf(fmaxf(v, 0.f));
    // the ROUNDED product, opaque to the compiler: __fmul_rn is a plain `*` to hipcc, and under -ffp-contract=fast the second pass's `min - d * d` became one fma (the
    // unrounded square) while the candidate records hold the rounded value — p of the same class differed by an ulp between the proof and the second pass, and the
    // merge of the two broke exact ties towards the wrong class (tests/test_gpu_parity.py::test_classify_fused_exact_ties_take_the_lowest_class under this mode)
    float r = d * d;
    asm("" : "+v"(r));
    return r;
}

// DUMP (tests): instead of classifying, the distances of panel 0 / tile 0 are written as sqdist_kernel would ([256][128] per bank) — the bit-identity check
template <bool EXACT, bool DUMP, bool CAND>
__global__ __launch_bounds__(512, 2) void classify_panel_kernel(const half_t* __restrict__ q, const half_t* __restrict__ zz, int Q, int rows2, int D,
                                                                const float* __restrict__ q_sqp, const float* __restrict__ zz_sq, float alpha, float oma,
                                                                float w, int32_t* __restrict__ argmax, float* __restrict__ dump, int npanels,
                                                                float* __restrict__ rec, int* __restrict__ stats, int mode) {
    using C = CP;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    float* strips = reinterpret_cast<float*>(smem + C::LDS_BYTES);        // [2][256] prototype norms of a tile | [2][256] query norms of a panel
    float* zstrip = strips;
    float* qstrip = strips + 512;
    float* rowc = strips + 1024;                                          // [256][4] per-row constants of pass 2
    const int G = gridDim.x;
    int panel = pgemm::xcd_remap(blockIdx.x, G);
    if (panel >= npanels) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int wm = wave / C::WN, wn = wave % C::WN;
    const int tiles_n = rows2 / C::BN, nt = D / pgemm::BK;
    using TP = pgemm::TilePairR<C>;
    TP tp;
    int p = 0, zpar = 0, qpar = 0;
    // request K-tile 0 of tile (m0, tn) into buffer p, with the tile's norm strip (every wave copies the strip: uniform vmcnt bookkeeping)
    auto issue = [&](int m0, int tn, int zp) {
        __builtin_amdgcn_global_load_lds((pgemm::gbl_ptr_t)(zz_sq + tn * C::BN + 4 * lane), (pgemm::lds_ptr_t)(zstrip + zp * 256), 16, 0, 0);
        tp.prepare(q, D, zz, D, Q, rows2, m0, tn * C::BN, wave, lane);
        tp.stage(0, smem + p * C::STAGE_BYTES, wave);
    };
    auto issue_q = [&](int m0, int qp) {
        __builtin_amdgcn_global_load_lds((pgemm::gbl_ptr_t)(q_sqp + m0 + 4 * lane), (pgemm::lds_ptr_t)(qstrip + qp * 256), 16, 0, 0);
    };
    issue_q(panel * C::BM, qpar);
    issue(panel * C::BM, 0, zpar);
    for (; panel < npanels; panel += G) {
        const int m0 = panel * C::BM;
        const int next_panel = panel + G;
        // One pass over the panel's class tiles; PASS 0: statistics, PASS 1: argmax.  Separate instantiations (and scopes) per pass: the per-row state of one pass is
        // not alive during the other's K-loops (all of it at once spilled 29 registers).  `tile_fn(tn, k, d, rowconst)` consumes the lane's distances of row k.
        // PASS 0 with `then_next`: the pass is expected to be the panel's only one (candidate records, below), so its last tile prefetches the NEXT panel
        // `tmask`: the class tiles to walk (bit tn; pass 1: all of them; the second pass of the candidate form: the tiles whose bounds the proof could not beat)
        auto walk = [&](auto pass_tag, const bool then_next, unsigned tmask, auto&& tile_fn) {
            constexpr int PASS = decltype(pass_tag)::value;
#pragma unroll 1
            for (unsigned left = tmask; left != 0;) {
                const int tn = __builtin_ctz(left);
                left &= left - 1;
                pgemm::Acc<C> acc;
                pgemm::mainloop_sr<C, 0, true, TP>(tp, nt, smem, acc, p, false, wave, lane);
                const int zp = zpar;
                // the next tile's K-tile 0 (this panel's next tile, the first tile of its second pass, or the next panel's first) under this tile's arithmetic
                {
                    const bool last = left == 0;
                    zpar ^= 1;
                    if (!last) issue(m0, __builtin_ctz(left), zpar);
                    else if (PASS == 0 && !then_next) issue(m0, 0, zpar);
                    else if (next_panel < npanels) { issue_q(next_panel * C::BM, qpar ^ 1); issue(next_panel * C::BM, 0, zpar); }
                }
                const float* zn = zstrip + zp * 256 + wn * 128;
                const float* qn = qstrip + qpar * 256 + wm * 64;
#pragma unroll
                for (int k = 0; k < 4; ++k) {
                    const int i = k >> 1, a = k & 1;
                    const float qs = qn[i * 32 + a * 16 + (lane & 15)];
                    float d[2][16];                        // [bank][class]: class index inside the lane = j * 4 + b * 2 + (e >> 1)
#pragma unroll
                    for (int j = 0; j < 4; ++j)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            const float4_t zs = *reinterpret_cast<const float4_t*>(zn + j * 32 + b * 16 + 4 * (lane >> 4));
#pragma unroll
                            for (int e = 0; e < 4; ++e) d[e & 1][j * 4 + b * 2 + (e >> 1)] = d2_of<EXACT>(acc.v[i][j][(a * 2 + b) * 4 + e], qs, zs[e]);
                        }
                    tile_fn(tn, k, d);
                }
            }
        };
        auto row_of = [&](int k) { return wm * 64 + (k >> 1) * 32 + (k & 1) * 16 + (lane & 15); };     // the lane's four query rows
        const int slot = wn * 4 + (lane >> 4);                                                         // 0 .. 7: the lanes x waves that share a row
        constexpr bool cand = CAND && !DUMP;                                                           // one pass + candidate records (two passes only where the proof fails)
        float* rec_wg = rec + (size_t)blockIdx.x * tiles_n * (8 * 256 * 8);
        bool second_pass = !cand;
        const unsigned all_tiles = tiles_n >= 32 ? 0xffffffffu : ((1u << tiles_n) - 1u);
        unsigned tmask2 = all_tiles;                                                                   // tiles of the second pass
        if (DUMP) {
            walk(std::integral_constant<int, 1>{}, false, all_tiles, [&](int tn, int k, const float (&d)[2][16]) {
                if (panel != 0 || tn != 0) return;
#pragma unroll
                for (int bank = 0; bank < 2; ++bank)
#pragma unroll
                    for (int c = 0; c < 16; ++c) {
                        const int j = c >> 2, b = (c >> 1) & 1, e2 = c & 1;
                        const int cls = (wn * 128 + j * 32 + b * 16 + 4 * (lane >> 4)) / 2 + e2;
                        dump[((size_t)bank * 256 + row_of(k)) * 128 + cls] = d[bank][c];
                    }
            });
            qpar ^= 1;
            continue;
        }
        // ---- pass 1: running minimum / sum of exp2((min - d2) w) per row and bank ----
        {
            float mn[4][2], sm[4][2];
#pragma unroll
            for (int k = 0; k < 4; ++k) { mn[k][0] = mn[k][1] = 3e38f; sm[k][0] = sm[k][1] = 0.f; }
            walk(std::integral_constant<int, 0>{}, cand, all_tiles, [&](int tn, int k, const float (&d)[2][16]) {
                float gmin[2] = {0.f, 0.f};
                if (cand) {
                    // candidate record of this (row, lane-slot, tile) group of 16 classes: per bank the smallest distance with its class and that class's distance in
                    // the OTHER bank, and the second smallest distance (a bound for everybody else of the group); 5 VALU per element and bank, no state across tiles
                    float rm1[2], rpr[2], rm2[2];
                    int rcls[2];
                    const int cls0 = tn * (C::BN / 2) + (wn * 128 + 4 * (lane >> 4)) / 2;
#pragma unroll
                    for (int bank = 0; bank < 2; ++bank) {
                        float m1 = d[bank][0], m2 = 3e38f, pr = d[bank ^ 1][0];
                        int i1 = 0;
#pragma unroll
                        for (int c = 1; c < 16; ++c) {
                            const bool lt = d[bank][c] < m1;
                            m2 = __builtin_amdgcn_fmed3f(d[bank][c], m1, m2);
                            pr = lt ? d[bank ^ 1][c] : pr;
                            i1 = lt ? c : i1;
                            m1 = lt ? d[bank][c] : m1;
                        }
                        rcls[bank] = cls0 + (i1 >> 2) * 16 + ((i1 >> 1) & 1) * 8 + (i1 & 1);
                        rm1[bank] = gmin[bank] = m1; rpr[bank] = pr; rm2[bank] = m2;
                    }
                    // record: { d_i min, its d_t, second d_i, d_t min | its d_i, second d_t, class of d_i min, class of d_t min } — floats and ints in words of their own type
                    float* dst = rec_wg + (((size_t)tn * 8 + slot) * 256 + row_of(k)) * 8;
                    *reinterpret_cast<float4_t*>(dst) = float4_t{rm1[0], rpr[0], rm2[0], rm1[1]};
                    *reinterpret_cast<float2_t*>(dst + 4) = float2_t{rpr[1], rm2[1]};
                    *reinterpret_cast<int2_t*>(dst + 6) = int2_t{rcls[0], rcls[1]};
                }
#pragma unroll
                for (int bank = 0; bank < 2; ++bank) {
                    float t = d[bank][0];
                    if (cand) {
                        t = gmin[bank];                                        // the group's minimum is in the record already
                    } else {
#pragma unroll
                        for (int c = 1; c < 16; ++c) t = fminf(t, d[bank][c]);
                    }
                    const float m2 = fminf(mn[k][bank], t);
                    float s = sm[k][bank] * __builtin_amdgcn_exp2f((m2 - mn[k][bank]) * w);
                    // exponent argument as ONE fma per element (m2 w - d w; round 6: was a subtraction and a multiplication — 2 of the ~14 VALU slots an element
                    // of this pass costs): the statistics S change in their last bits only, and every later expression (proof, second pass) reads S through rowc
                    // m2w = m2 w rounded DOWN (w >= 0, m2 >= 0): exact(-d w) + m2w <= 0 for every d >= m2, so no argument comes out positive — a lane group of padding
                    // classes (d = m2 = 1e30) has a rounding gap of ~1e24 between the two products, and exp2(+1e24) is inf
                    float m2w = m2 * w;
                    if (__builtin_fmaf(m2, w, -m2w) < 0.f) m2w = __builtin_bit_cast(float, __builtin_bit_cast(unsigned, m2w) - 1u);
                    const float nw = -w;
#pragma unroll
                    for (int c = 0; c < 16; ++c) s += __builtin_amdgcn_exp2f(__builtin_fmaf(d[bank][c], nw, m2w));
                    mn[k][bank] = m2;
                    sm[k][bank] = s;
                }
            });
            // combine the partials of the 8 lanes x waves that share a query row through LDS (the buffer of the last K-tile; the prefetch above went to buffer p)
            float* scr = reinterpret_cast<float*>(smem + (p ^ 1) * C::STAGE_BYTES);
            pgemm::lds_barrier();
#pragma unroll
            for (int k = 0; k < 4; ++k) *reinterpret_cast<float4_t*>(scr + (row_of(k) * 8 + slot) * 4) = float4_t{mn[k][0], sm[k][0], mn[k][1], sm[k][1]};
            pgemm::lds_barrier();
            if (tid < 256) {
                float m[2] = {3e38f, 3e38f}, s[2] = {0.f, 0.f};
                float4_t v[8];
#pragma unroll
                for (int x = 0; x < 8; ++x) { v[x] = *reinterpret_cast<const float4_t*>(scr + (tid * 8 + x) * 4); m[0] = fminf(m[0], v[x][0]); m[1] = fminf(m[1], v[x][2]); }
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    s[0] += v[x][1] * __builtin_amdgcn_exp2f((m[0] - v[x][0]) * w);
                    s[1] += v[x][3] * __builtin_amdgcn_exp2f((m[1] - v[x][2]) * w);
                }
                // what pass 2 needs per row: alpha / S_i, min_i, (1 - alpha) / S_t, min_t — in its own LDS strip (the K-tile buffers are about to be refilled)
                *reinterpret_cast<float4_t*>(rowc + tid * 4) = float4_t{alpha / s[0], m[0], oma / s[1], m[1]};
            }
            pgemm::lds_barrier();
        }
        // ---- candidates: the argmax from the records of pass 1, with a proof that nobody else can win ----
        // Every class that is not a group's nearest in one of the banks has d_i >= the group's second smallest d_i and d_t >= its second smallest d_t, hence (the
        // expression below is monotone in both) p <= U_group.  If max_groups U < the best candidate's p, the candidates' argmax (lowest class among equal maxima) IS
        // the argmax of the row: same bits as pass 2 would give.  Otherwise (flat distributions: small beta, near-ties) the panel takes the second pass.
        if (cand) {
            // PER TILE (round 6): a row's proof fails only against the groups whose bound reaches its best candidate; the second pass walks just the class tiles that
            // hold such a group for SOME row of the panel (`tmask2`), and its result is merged with the candidates' best — every class of a tile that is not walked
            // is either a candidate (its exact p is in the merge) or bounded below the best, so the merge is the argmax the full second pass returns, bit for bit.
            // (Before: one row short of its proof sent the whole panel through all tiles again — 23 of 196 panels on the class-structured ImageNet-sized split of
            // tests/test_gpu_parity.py::test_full_size_default_routing_C3, i.e. the launch as slow as its slowest panel: 277 instead of 172 us.)
            float* scr = reinterpret_cast<float*>(smem + (p ^ 1) * C::STAGE_BYTES);
            float* umt = scr + 256 * 4;                                        // [half][row][16]: the largest bound of each tile, by half of the groups
            int* fail = reinterpret_cast<int*>(rowc + 1024);                   // the panel's tile mask
            float* candb = rowc + 1024 + 16;                                   // [256][2]: the candidates' best (p, class) of a row, for the merge behind the second pass
            const int row = tid & 255, half = tid >> 8, ng = tiles_n * 8;
            const bool per_tile = tiles_n <= 16;
            const float4_t c4 = *reinterpret_cast<const float4_t*>(rowc + row * 4);
            float bv = -1.f, umax = 0.f, um = 0.f;
            int bi = 0x7fffffff;
            if (tid == 0) *fail = 0;
            __syncthreads();                                                   // the records of every lane are complete (vmcnt(0)) and visible to the workgroup
            const int g_lo = half * (ng >> 1), g_hi = (half + 1) * (ng >> 1);
#pragma unroll 2
            for (int g = g_lo; g < g_hi; ++g) {
                // (plain loads: the records were written through this CU's L1 by this workgroup and the barrier above waited for them)
                const float* rp = rec_wg + ((size_t)g * 256 + row) * 8;
                const float4_t a = *reinterpret_cast<const float4_t*>(rp);
                const float2_t b = *reinterpret_cast<const float2_t*>(rp + 4);
                const int2_t ci = *reinterpret_cast<const int2_t*>(rp + 6);
                const float p1 = __builtin_fmaf(c4[0], __builtin_amdgcn_exp2f((c4[1] - a[0]) * w), c4[2] * __builtin_amdgcn_exp2f((c4[3] - a[1]) * w));
                const float p2 = __builtin_fmaf(c4[0], __builtin_amdgcn_exp2f((c4[1] - b[0]) * w), c4[2] * __builtin_amdgcn_exp2f((c4[3] - a[3]) * w));
                const float u = __builtin_fmaf(c4[0], __builtin_amdgcn_exp2f((c4[1] - a[2]) * w), c4[2] * __builtin_amdgcn_exp2f((c4[3] - b[1]) * w));
                const int i1 = ci[0], i2 = ci[1];
                if (p1 > bv || (p1 == bv && i1 < bi)) { bv = p1; bi = i1; }
                if (p2 > bv || (p2 == bv && i2 < bi)) { bv = p2; bi = i2; }
                umax = fmaxf(umax, u);
                um = fmaxf(um, u);
                if (per_tile && ((g & 7) == 7 || g + 1 == g_hi)) { umt[(half * 256 + row) * 16 + (g >> 3)] = um; um = 0.f; }     // (a tile's groups are consecutive)
            }
            if (half) { scr[row * 4] = bv; reinterpret_cast<int*>(scr)[row * 4 + 1] = bi; scr[row * 4 + 2] = umax; }
            __syncthreads();
            if (!half) {
                const float ov = scr[row * 4], ou = scr[row * 4 + 2];
                const int oi = reinterpret_cast<const int*>(scr)[row * 4 + 1];
                if (ov > bv || (ov == bv && oi < bi)) { bv = ov; bi = oi; }
                umax = fmaxf(umax, ou);
                scr[row * 4 + 3] = bv;                                         // the row's best candidate, for the other half's tile test
                candb[row * 2] = bv;
                reinterpret_cast<int*>(candb)[row * 2 + 1] = bi;
                if (m0 + row < Q && !per_tile && (!(umax < bv) || mode == 2)) atomicOr(fail, (int)all_tiles);
            }
            __syncthreads();
            if (per_tile && m0 + row < Q) {
                const float bvf = scr[row * 4 + 3];
                unsigned mk = 0;
                for (int t = g_lo >> 3; t <= (g_hi - 1) >> 3; ++t)
                    if (!(umt[(half * 256 + row) * 16 + t] < bvf)) mk |= 1u << t;
                if (mode == 2) mk = all_tiles;                                  // (mode 2, tests: every panel takes the whole second pass)
                if (mk) atomicOr(fail, (int)mk);
            }
            __syncthreads();
            tmask2 = (unsigned)__builtin_amdgcn_readfirstlane(*fail);
            second_pass = tmask2 != 0;
            if (!second_pass && !half && m0 + row < Q) argmax[m0 + row] = bi;
            if (stats && tid == 0) { atomicAdd(stats, 1); if (second_pass) { atomicAdd(stats + 1, 1); atomicAdd(stats + 2, __builtin_popcount(tmask2)); } }
            __syncthreads();                                                   // the scratch is a K-tile buffer again; `fail` may be rewritten by the next panel
            if (second_pass) issue(m0, __builtin_ctz(tmask2), zpar);           // the prefetched tile was the next panel's: this panel's first second-pass tile (same buffer, same strip: in order)
        }
        // ---- pass 2: p = alpha e_i / S_i + (1 - alpha) e_t / S_t, running (best, class) per row ----
        if (second_pass) {
            float best[4];
            int besti[4];
#pragma unroll
            for (int k = 0; k < 4; ++k) { best[k] = -1.f; besti[k] = 0x7fffffff; }
            walk(std::integral_constant<int, 1>{}, false, tmask2, [&](int tn, int k, const float (&d)[2][16]) {
                const float4_t c4 = *reinterpret_cast<const float4_t*>(rowc + row_of(k) * 4);
                const int cls0 = tn * (C::BN / 2) + (wn * 128 + 4 * (lane >> 4)) / 2;
#pragma unroll
                for (int c = 0; c < 16; ++c) {
                    const int j = c >> 2, b = (c >> 1) & 1, e2 = c & 1;
                    const float pv = __builtin_fmaf(c4[0], __builtin_amdgcn_exp2f((c4[1] - d[0][c]) * w), c4[2] * __builtin_amdgcn_exp2f((c4[3] - d[1][c]) * w));
                    if (pv > best[k]) { best[k] = pv; besti[k] = cls0 + j * 16 + b * 8 + e2; }      // ascending class order inside the lane: first maximum kept
                }
            });
            float* scr = reinterpret_cast<float*>(smem + (p ^ 1) * C::STAGE_BYTES);
            pgemm::lds_barrier();
#pragma unroll
            for (int k = 0; k < 4; ++k) {
                scr[(row_of(k) * 8 + slot) * 2] = best[k];
                reinterpret_cast<int*>(scr)[(row_of(k) * 8 + slot) * 2 + 1] = besti[k];
            }
            pgemm::lds_barrier();
            if (tid < 256 && m0 + tid < Q) {
                float bv = -1.f;
                int bi = 0x7fffffff;
#pragma unroll
                for (int x = 0; x < 8; ++x) {
                    const float v = scr[(tid * 8 + x) * 2];
                    const int ix = reinterpret_cast<const int*>(scr)[(tid * 8 + x) * 2 + 1];
                    if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }           // lowest class among equal maxima (main.py:190 on the CPU)
                }
                if (cand) {                                                             // the candidates' best: the classes of the tiles this pass did not walk
                    const float* candb = rowc + 1024 + 16;
                    const float cv = candb[tid * 2];
                    const int cx = reinterpret_cast<const int*>(candb)[tid * 2 + 1];
                    if (cv > bv || (cv == bv && cx < bi)) { bv = cv; bi = cx; }
                }
                argmax[m0 + tid] = bi;
            }
            pgemm::lds_barrier();                          // the scratch is a K-tile buffer again
        }
        qpar ^= 1;
    }
}
}  // namespace

size_t pclip_classify_panel_workspace(int Q, int N, int D) {
    const size_t rows2 = (size_t)2 * ((N + 127) / 128 * 128), Qp = (size_t)(Q + 255) / 256 * 256;
    int cus = pclip_device_cus();
    if (cus <= 0) cus = 256;
    const size_t npanels = Qp / 256, grid = npanels < (size_t)cus ? npanels : (size_t)cus;
    // + the candidate records: per resident panel and class tile 8 x 256 groups x 32 B (a quarter of the fp32 distance rows the two stages would write)
    return align_up(rows2 * D * 2, 256) + align_up(rows2 * 4, 256) + align_up(Qp * 4, 256) + grid * (rows2 / 256) * (8 * 256 * 32) + 256;
}

static int g_panel_passes = -1;                // -1: PCLIP_CLASSIFY_PANEL_PASSES / default 0; 0 one pass + candidates (second pass where the proof fails), 1 always two passes, 2 tests
extern "C" int pclip_classify_panel_passes(int mode) {
    const int before = g_panel_passes;
    if (mode >= 0) g_panel_passes = mode > 2 ? 0 : mode;
    return before;
}
__device__ int g_panel_stats[3];               // panels classified | panels that took a second pass | class tiles those second passes walked (since the last reset)
extern "C" int pclip_classify_panel_stats(int* out3, int reset) {
    int z[3] = {0, 0, 0};
    if (out3 && hipMemcpyFromSymbol(out3, HIP_SYMBOL(g_panel_stats), sizeof(z)) != hipSuccess) return PCLIP_E_LAUNCH;
    if (reset && hipMemcpyToSymbol(HIP_SYMBOL(g_panel_stats), z, sizeof(z)) != hipSuccess) return PCLIP_E_LAUNCH;
    return PCLIP_OK;
}

static int g_panel_mode = -1;                  // -1: PCLIP_CLASSIFY_PANEL / default (on), decided at the first call
extern "C" int pclip_classify_panel_config(int mode) {
    const int before = g_panel_mode;
    if (mode >= 0) g_panel_mode = mode > 2 ? 1 : mode;
    return before;
}

bool pclip_classify_panel_applies(int Q, int N, int D, float alpha, float one_minus_alpha, float beta) {
    if (g_panel_mode < 0) { const char* e = getenv("PCLIP_CLASSIFY_PANEL"); g_panel_mode = e ? atoi(e) : 1; if (g_panel_mode < 0 || g_panel_mode > 2) g_panel_mode = 1; }
    // Routing by measurement (tools/fused_routing_probe.py, profiles/r05_fused_routing.txt): up to one panel per CU the fused kernel's time is ~12 + 17 us per class tile
    // whatever Q is (a CU walks its panel's tiles alone), the two stages cost ~22 us + 8e-6 us per (query, class) — fused from Q N >= 2e6 tiles - 1e6 (ImageNet: Q >= 15 k;
    // Food-101's 30 k queries: 32 vs 58 us; SUN397: 79 vs 107; below that the chip is mostly idle and the two stages win: FewSOL-198, Q = 666: 26 vs 50 us);
    // mode 2 forces the fused kernel for every shape it can run (tests)
    const double tiles = (double)(2 * ((N + 127) / 128 * 128) / 256);
    const bool enough = g_panel_mode == 2 || (double)Q * (double)N >= 2.0e6 * tiles - 1.0e6;
    // the candidate proof bounds a class through p's monotonicity in both distances: both mixing weights and beta must be non-negative (a user's --alpha outside
    // [0, 1] takes the two stages)
    const bool monotone = alpha >= 0.f && one_minus_alpha >= 0.f && beta >= 0.f;
    return g_panel_mode > 0 && enough && monotone && N > 32 && D >= 128 && D % 64 == 0 && D <= 4096 && Q >= 1 && (long)256 * D * 2 < 0x7fffffffL;
}

// q_sq / zi_sq / zt_sq: device arrays or null (computed by the preparation launch with pclip_row_sqnorm_f16's arithmetic).  dump != nullptr: test mode (distances of panel 0 / tile 0, no argmax; dump_exact:
// with / without the sqrt round trip).
int pclip_classify_panel_launch(const void* q, const void* zi, const void* zt, int Q, int N, int D, const float* q_sq, const float* zi_sq, const float* zt_sq,
                                float alpha, float oma, float beta, int32_t* argmax, float* dump, bool dump_exact, void* ws, hipStream_t s) {
    const int rows2 = 2 * ((N + 127) / 128 * 128), Qp = (Q + 255) / 256 * 256;
    char* b = (char*)ws;
    half_t* zz = (half_t*)b; b += align_up((size_t)rows2 * D * 2, 256);
    float* zz_sq = (float*)b; b += align_up((size_t)rows2 * 4, 256);
    float* q_sqp = (float*)b; b += align_up((size_t)Qp * 4, 256);
    float* rec = (float*)b;
    if (g_panel_passes < 0) { const char* e = getenv("PCLIP_CLASSIFY_PANEL_PASSES"); g_panel_passes = e ? atoi(e) : 0; if (g_panel_passes < 0 || g_panel_passes > 2) g_panel_passes = 0; }
    static int* stats_dev[64];                                      // the counters' device address, looked up once per device
    int dev = 0;
    int* stats = nullptr;
    if (hipGetDevice(&dev) == hipSuccess && dev >= 0 && dev < 64) {
        if (!stats_dev[dev] && hipGetSymbolAddress((void**)&stats_dev[dev], HIP_SYMBOL(g_panel_stats)) != hipSuccess) stats_dev[dev] = nullptr;
        stats = stats_dev[dev];
    }
    {
        const int jobs = rows2 + Qp, pg = ceil_div(jobs, 4) < 4096 ? ceil_div(jobs, 4) : 4096;
#define PCLIP_PREP(NCH) panel_prep_kernel<NCH><<<pg, 256, 0, s>>>((const half_t*)q, (const half_t*)zi, (const half_t*)zt, q_sq, zi_sq, zt_sq, Q, Qp, N, D, rows2, zz, zz_sq, q_sqp)
        if (D <= 512) PCLIP_PREP(1); else if (D <= 1024) PCLIP_PREP(2); else if (D <= 2048) PCLIP_PREP(4); else PCLIP_PREP(8);
#undef PCLIP_PREP
    }
    int cus = pclip_device_cus();
    if (cus <= 0) cus = 256;
    const int npanels = Qp / 256, grid = npanels < cus ? npanels : cus;
    constexpr int LDS = CP::LDS_BYTES + 8192 + 64 + 2048;              // K-tile ring | norm strips + per-row constants | tile mask | the candidates' best per row
    // d2 = max(||q||^2 + ||z||^2 - 2 q.z, 0) by default: without torch.cdist's sqrt -> square round trip (<= 1 fp32 ulp from the two-stage path's distances, whose
    // correctly rounded sqrtf costs twelve VALU instructions per element: 371 vs 250 us on the ImageNet split — the arithmetic this kernel is bound by);
    // PCLIP_CLASSIFY_PANEL_EXACT=1 keeps the round trip (bit-identical distances)
    static const bool exact_env = getenv("PCLIP_CLASSIFY_PANEL_EXACT") && getenv("PCLIP_CLASSIFY_PANEL_EXACT")[0] == '1';
    const bool exact = dump ? dump_exact : exact_env;
    const float w = beta * 1.4426950408889634f;
#define PCLIP_PANEL(EX, DU, CA)                                                                                                                  \
    do {                                                                                                                                         \
        static DevOnce attr;                                                                                                                     \
        if (!attr.done()) {                                                                                                                      \
            if (hipFuncSetAttribute((const void*)classify_panel_kernel<EX, DU, CA>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) { \
                pclip_set_error("pclip_classify_f16: cannot raise the dynamic LDS limit to %d", LDS);                                            \
                return PCLIP_E_LAUNCH;                                                                                                           \
            }                                                                                                                                    \
            attr.set();                                                                                                                          \
        }                                                                                                                                        \
        classify_panel_kernel<EX, DU, CA><<<DU ? 1 : grid, 512, LDS, s>>>((const half_t*)q, zz, Q, rows2, D, q_sqp, zz_sq, alpha, oma, w, argmax, dump, DU ? 1 : npanels, rec, stats, g_panel_passes); \
    } while (0)
    const bool cand = g_panel_passes != 1;
    if (dump && exact) PCLIP_PANEL(true, true, false);
    else if (dump) PCLIP_PANEL(false, true, false);
    else if (exact && cand) PCLIP_PANEL(true, false, true);
    else if (exact) PCLIP_PANEL(true, false, false);
    else if (cand) PCLIP_PANEL(false, false, true);
    else PCLIP_PANEL(false, false, false);
#undef PCLIP_PANEL
    return pclip_check_launch("classify (row panels)");
}
