// Classification for MID-SIZED class counts (16 < N <= 256: Caltech-101, FewSOL-198, OxfordPets, DTD ...) in ONE launch (round 6; reference
// utils.py:225-244 `P` + main.py:190 `.max(1)[1]`; toolkit proto_clip_classifier.py:146-147).
// The two-stage path costs five launches there (three norm passes, the distance GEMM, the softmax pass): 24 - 31 us whatever the size, for 2 - 20 MB of traffic.
// Here a workgroup owns a group of 16 queries and walks every class of both banks once; its 2 SL waves are (bank, class-tile slot) pairs (SL = 4 up to four
// 16-class tiles, 8 beyond: sixteen waves halve the softmax arithmetic per wave):
//   * the 16 query rows go to LDS once (16-byte chunks, chunk index XOR row: D % 128 == 0 makes a row whole 256-byte LDS lines, and the XOR is then conflict-free
//     under ds_read_b128's lane groups in the MFMA operand layout, lane = row l & 15, k-chunk l >> 4);
//   * wave (bank, slot) owns the class tiles slot, slot + SL, ...  Its bank rows arrive by LDS-DMA as FULL-LINE pieces (8 rows x 128 B per instruction) in a
//     wave-private ring of 2 KB blocks (16 classes x 64 k), swizzled on the source side, waited for with the wave's own counted vmcnt — no barrier in the block loop
//     (D = 512 / 768 / 1024: unrolled; other D % 128 == 0: a plain loop that loads straight into the MFMA operand layout);
//   * the arithmetic is classify_small's (pclip_classify_small.h), operation for operation: v_mfma_f32_16x16x32_f16 with the classes as the first operand (a lane
//     ends with 4 consecutive classes of ONE query per tile), fp32 norms from the fragments the MFMAs consume (v_dot2 chains),
//     d2 = (sqrt(max(qq + zz - 2 q.z, 0)))^2;
//   * the softmax of a bank spans SL waves: per query the (min, max) and then the sum of the exponentials cross the waves through LDS (two barriers), the textual
//     waves hand their alpha-weighted terms to the visual waves (third barrier), which add (visual + textual, the reference's order), write p and reduce the argmax
//     (fourth barrier; lowest class among equal maxima).  These exchange buffers alias the rings (dead by then).
// What bounds it: a workgroup streams both banks (2 N D 2 bytes: 410 KB for Caltech-101 / RN50, 608 KB for FewSOL-198 / ViT-L/14) through ONE CU's vector-memory
// path whatever the number of queries — 30 B/clk with the full-line LDS-DMA pieces (the MFMA operand layout's 16 half-lines per load: 14 B/clk; full lines through
// registers + ds_write: 25): ~6 / 8 us of the kernel's 11 / 14.5 us; the rest is the launch (~2 us) and the correctly rounded sqrt / exp / division chains of the
// softmax (tools/mid_probe.py, profiles/r06_mid_probe.txt).
#include "pclip_gemm.h"
#include "pclip_classify_small.h"
#include <stdlib.h>
#include <type_traits>

namespace {

constexpr int MID_U = 4;                                                            // U: k-steps (32 wide) per register buffer = 128 of D (generic-width path)

// slots of a wave's private ring of [16 rows][128 B] blocks = blocks in flight + 1 — eight waves: 3 + 1 (64 KB: two workgroups per CU at D = 512, which
// is what a call with more query groups than CUs runs on), sixteen waves: 2 + 1 (96 KB); deeper rings (7 + 1 / 3 + 1 = 128 KB) measured the same single-group latency
// (14.5 vs 14.6 us at FewSOL-198's size) and cost the second resident workgroup (N = 64, Q = 20 000: 25.2 vs 18.9 us).  The softmax's exchange buffers alias the ring,
// which is dead by then.  (The same blocks through registers + ds_write_b128 — the form before — streamed 25 instead of 30 B/clk and held 64 registers of prefetch.)
__host__ __device__ constexpr int mid_ring_slots(int sl) { return sl == 8 ? 3 : 4; }
// LDS: the query group [16][D] fp16 | then EITHER the waves' rings (main loop) OR red [2 banks][SL][16 queries][2] fp32 | xch [SL][TPW * 4][64] fp32 |
// best [SL][16][2] (softmax: behind a barrier)  (the rings exist for the unrolled widths only: D = 512 / 768 / 1024)
__host__ __device__ constexpr size_t classify_mid_lds(int D, int tpw, int sl, bool ring) {
    const size_t aux = 2 * sl * 16 * 2 * 4 + (size_t)sl * tpw * 4 * 64 * 4 + sl * 16 * 2 * 4, rings = ring ? (size_t)2 * sl * mid_ring_slots(sl) * 2048 : 0;
    return (size_t)16 * D * 2 + (aux > rings ? aux : rings);
}

// TPW: class tiles per wave (ceil(ceil(N / 16) / SL): 1 or 2).  NCH: D / 128 when it is a compile-time constant (4 / 6 / 8: D = 512 / 768 / 1024 — the loop over the
// bank is then fully unrolled straight-line code: every wait is counted), 0: any D % 128 == 0 (run-time trip count, two register buffers).
// Why it matters: a workgroup's run time is its chain of L2 round trips; with branches between the loads and their uses hipcc waits with vmcnt(0) at every block
// boundary (first build: 25.8 us at FewSOL-198's size, the two stages' 26).
// SL: waves (slots) per bank — 4 (up to four class tiles: eight waves) or 8 (sixteen waves).
template <int TPW, int NCH, int SL>
__global__ __launch_bounds__(2 * SL * 64) void classify_mid_kernel(const half_t* __restrict__ q, const half_t* __restrict__ zi, const half_t* __restrict__ zt, int Q,
                                                                      int N, int D, float alpha, float oma, float beta, float* __restrict__ p,
                                                                      int32_t* __restrict__ argmax) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int U = MID_U, MID_SLOTS = SL, MID_WAVES = 2 * SL;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int qr = lane & 15, kg = lane >> 4;
    const int bank = wave & 1, slot = wave >> 1;
    const int nch = NCH ? NCH : D >> 7, ngroups = (Q + 15) >> 4, ntiles = (N + 15) >> 4;
    const int qstride = D * 2;                                                         // bytes per staged query row: whole 256-byte LDS lines, chunks XOR-swizzled by the row
    float* red = reinterpret_cast<float*>(smem + 16 * qstride);                        // [bank][slot][query][2]
    float* xch = red + 2 * MID_SLOTS * 16 * 2;                                         // [slot][TPW * 4][64]
    float* bst = xch + MID_SLOTS * TPW * 4 * 64;                                       // [slot][query][2]
    char* tbuf = smem + 16 * qstride + wave * mid_ring_slots(SL) * 2048;               // this wave's private ring of transposition blocks (aliases red / xch / bst)
    const half_t* z = bank ? zt : zi;
    const int cls0 = 4 * kg;
    const int cpr = D >> 3;                                                            // 16-byte chunks per query row; 16 cpr = 2 D chunks per group, D / 256 per thread
    for (int g = blockIdx.x; g < ngroups; g += gridDim.x) {
        if (g != (int)blockIdx.x) __syncthreads();                                      // the previous group's readers are done
        // the lane's class row of each of the wave's tiles (clamped: rows >= N are read, never used).  Formed per group from an OPAQUE lane id: the bank loads do not
        // depend on the group, and left visible hipcc hoists all of them (and their norm chains) out of this loop — every fragment live across it, hundreds spilled
        int oqr = qr;
#if defined(__HIP_DEVICE_COMPILE__)
        asm volatile("" : "+v"(oqr));
#endif
        const half_t* zrow[TPW];
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            const int c = (slot + MID_SLOTS * i) * 16 + oqr;
            zrow[i] = z + (size_t)(c < N ? c : N - 1) * D + kg * 8;
        }
        auto load = [&](half8_t (&buf)[U][TPW], int c) {
#pragma unroll
            for (int u = 0; u < U; ++u)
#pragma unroll
                for (int i = 0; i < TPW; ++i) buf[u][i] = ld_half8(zrow[i] + (c * U + u) * 32);
        };
        float4_t acc[TPW];
        float znp[TPW], qs = 0.f;
#pragma unroll
        for (int i = 0; i < TPW; ++i) { acc[i] = float4_t{0.f, 0.f, 0.f, 0.f}; znp[i] = 0.f; }
        const char* qrowl = smem + qr * qstride;
        auto compute = [&](const half8_t (&buf)[U][TPW], int c) {
#pragma unroll
            for (int u = 0; u < U; ++u) {
                const int ch = (c * U + u) * 4 + kg;
                const half8_t qf = *reinterpret_cast<const half8_t*>(qrowl + ((ch ^ qr) << 4));
                qs = sq8(qf, qs);
#pragma unroll
                for (int i = 0; i < TPW; ++i) {
                    znp[i] = sq8(buf[u][i], znp[i]);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(buf[u][i], qf, acc[i], 0, 0, 0);
                }
            }
        };
        // ---- the group's 16 query rows -> LDS: every thread requests its chunks first (one round trip), rows beyond Q repeat the last row (dropped at the stores)
        auto stage_q = [&](auto nq_tag) {
            constexpr int NQ = decltype(nq_tag)::value;                                 // chunks per thread, rounded up
            half8_t t[NQ];
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const int c = tid + j * MID_WAVES * 64, r = c / cpr, cc = c - r * cpr, m = g * 16 + (r < 16 ? r : 15);
                t[j] = ld_half8(q + (size_t)(m < Q ? m : Q - 1) * D + (r < 16 ? cc : 0) * 8);
            }
#pragma unroll
            for (int j = 0; j < NQ; ++j) {
                const int c = tid + j * MID_WAVES * 64, r = c / cpr, cc = c - r * cpr;
                if (r < 16) *reinterpret_cast<half8_t*>(smem + r * qstride + ((cc ^ r) << 4)) = t[j];
            }
        };
        if constexpr (NCH > 0) {
            // Bank rows by FULL-LINE pieces: an instruction covers 8 class rows x 128 B (lane = row l >> 3, 16-byte chunk l & 7: 8 whole cache lines) — the MFMA
            // operand layout (lane = row l & 15, chunk l >> 4: 16 rows x 64 B = 16 half lines per instruction) streamed at 14 B/clk per CU whatever the row stride
            // (tools/mid_probe.py: one workgroup, 608 KB, 17.8 us; the vector-memory path pays per line touched).  Each wave passes its blocks — 16 classes x 64 k:
            // two pieces — through its private ring (pgemm's swizzle: chunk ^ ((row >> 1) & 7), conflict-free for the fragment reads).  Blocks b = kb * TPW + i
            // (k-block kb of the wave's tile i).
            constexpr int NKB = NCH * 2, NBLK = NKB * TPW;
            const int lrow = lane >> 3, lch = lane & 7;
            int roff[2];                                                               // read offsets of the lane's two fragments (k-steps 0 / 1 of a block)
#pragma unroll
            for (int j = 0; j < 2; ++j) roff[j] = qr * 128 + (((j * 4 + kg) ^ ((qr >> 1) & 7)) << 4);
            auto block_math = [&](const char* tb, int kb, int i) {
#pragma unroll
                for (int s2 = 0; s2 < 2; ++s2) {
                    const half8_t zf = *reinterpret_cast<const half8_t*>(tb + roff[s2]);
                    const int ch = (kb * 2 + s2) * 4 + kg;
                    const half8_t qf = *reinterpret_cast<const half8_t*>(qrowl + ((ch ^ qr) << 4));
                    if (i == 0) qs = sq8(qf, qs);
                    znp[i] = sq8(zf, znp[i]);
                    acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(zf, qf, acc[i], 0, 0, 0);
                }
            };
            // A block's two pieces (8 rows x 128 B each) go straight from L2 into the wave's ring — no registers, no ds_write (13 LDS cycles per
            // kilobyte on the store path).  The swizzle sits on the SOURCE side (lane = LDS row l >> 3, LDS chunk l & 7 fetches source chunk (l & 7) ^ key(row):
            // pgemm::stage_tile's addressing), the hardware places lane l at base + 16 l.  PFD blocks in flight; block b + PFD lands in the slot block b - 1 was read
            // from (its fragments are in registers: the MFMAs that consumed them precede the request).  Every wait is the wave's own counted vmcnt — the ring is private.
            constexpr int PFD = mid_ring_slots(SL) - 1, NSLOT = PFD + 1;
            // (BUFFER LDS-DMA, the k-block in the scalar offset: behind a FLAT-encoded global_load_lds hipcc answers every LDS wait with lgkmcnt(0) while the piece
            // is in flight — here always — pclip_gemm.h: make_rsrc)
            const pgemm::rsrc_t rs = pgemm::make_rsrc(z, (unsigned)N * (unsigned)D * 2u);
            (void)rs;
            int zoff[TPW][2];
#pragma unroll
            for (int i = 0; i < TPW; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j) {
                    int r = lrow + 8 * j;
#if defined(__HIP_DEVICE_COMPILE__)
                    asm volatile("" : "+v"(r));                                         // opaque: see zrow
#endif
                    const int c = (slot + MID_SLOTS * i) * 16 + r;
                    zoff[i][j] = ((c < N ? c : N - 1) * D + ((lch ^ ((r >> 1) & 7)) << 3)) * 2;
                }
            auto request = [&](int b) {
                const int kb = b / TPW, i = b % TPW;
                char* dst = tbuf + (b % NSLOT) * 2048;
                (void)kb; (void)dst;
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (pgemm::lds_ptr_t)(dst + j * 1024), 16, zoff[i][j], kb * 128, 0, 0);
#endif
            };
#pragma unroll
            for (int b = 0; b < PFD && b < NBLK; ++b) request(b);
            // (plain loads behind the first blocks' requests: hipcc waits for them with vmcnt(0), i.e. for those blocks too — which block 0 needs anyway: ONE round trip)
            stage_q(std::integral_constant<int, (NCH * 256 + MID_WAVES * 64 - 1) / (MID_WAVES * 64)>{});      // 2 D chunks over the workgroup's threads
            pgemm::lds_barrier();                                                      // the query rows are staged (LDS-only: the first blocks stay in flight)
#pragma unroll
            for (int b = 0; b < NBLK; ++b) {
                const int younger = (NBLK - 1 - b < PFD - 1 ? NBLK - 1 - b : PFD - 1) * 2;     // pieces requested behind block b's
                static_assert(PFD <= 3, "the counted waits below cover up to two younger blocks");
                if (younger == 4) pgemm::wait_vm<4>(); else if (younger == 2) pgemm::wait_vm<2>(); else pgemm::wait_vm<0>();
                asm volatile("" ::: "memory");
                block_math(tbuf + (b % NSLOT) * 2048, b / TPW, b % TPW);
                __builtin_amdgcn_sched_barrier(0);
                if (b + PFD < NBLK) request(b + PFD);
                __builtin_amdgcn_sched_barrier(0);
            }
        } else {
            half8_t za[U][TPW], zb[U][TPW];
            load(za, 0);
            for (int c = tid; c < 16 * cpr; c += MID_WAVES * 64) {                      // (uncommon widths: the plain loop, a round trip per chunk)
                const int r = c / cpr, cc = c - r * cpr, m = g * 16 + r;
                *reinterpret_cast<half8_t*>(smem + r * qstride + ((cc ^ r) << 4)) = ld_half8(q + (size_t)(m < Q ? m : Q - 1) * D + cc * 8);
            }
            __syncthreads();
            for (int c = 0; c < nch; c += 2) {
                if (c + 1 < nch) load(zb, c + 1);
                __builtin_amdgcn_sched_barrier(0);
                compute(za, c);
                __builtin_amdgcn_sched_barrier(0);
                if (c + 2 < nch) load(za, c + 2);
                __builtin_amdgcn_sched_barrier(0);
                if (c + 1 < nch) compute(zb, c + 1);
                __builtin_amdgcn_sched_barrier(0);
            }
        }
        if (NCH > 0) pgemm::lds_barrier();                                              // every wave is through with its ring: the exchange buffers below alias it
        // ---- norms into the accumulator layout, cdist epilogue (classify_small's expressions)
        qs += lane_xor<16>(qs);
        qs += lane_xor<32>(qs);
        float d2[TPW][4], mn = __builtin_inff(), mx = -__builtin_inff();
#pragma unroll
        for (int i = 0; i < TPW; ++i) {
            float zsq = znp[i];
            zsq += lane_xor<16>(zsq);
            zsq += lane_xor<32>(zsq);
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                const float zs = __shfl(zsq, cls0 + e, WAVE);
                const float v = __fadd_rn(__fadd_rn(-2.f * acc[i][e], qs), zs);
                const float d = sqrtf(fmaxf(v, 0.f));
                d2[i][e] = __fmul_rn(d, d);
                if ((slot + MID_SLOTS * i) * 16 + cls0 + e < N) { mn = fminf(mn, d2[i][e]); mx = fmaxf(mx, d2[i][e]); }
            }
        }
        mn = fminf(mn, lane_xor<16>(mn)); mn = fminf(mn, lane_xor<32>(mn));
        mx = fmaxf(mx, lane_xor<16>(mx)); mx = fmaxf(mx, lane_xor<32>(mx));
        float* myred = red + ((bank * MID_SLOTS + slot) * 16 + qr) * 2;
        if (kg == 0) { myred[0] = mn; myred[1] = mx; }
        __syncthreads();
#pragma unroll
        for (int s = 0; s < MID_SLOTS; ++s) {
            const float* r = red + ((bank * MID_SLOTS + s) * 16 + qr) * 2;
            mn = fminf(mn, r[0]);
            mx = fmaxf(mx, r[1]);
        }
        const float top = __fmul_rn(beta, beta >= 0.f ? -mn : -mx);
        float sum = 0.f;
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) {
                d2[i][e] = ((slot + MID_SLOTS * i) * 16 + cls0 + e < N) ? expf(__fsub_rn(__fmul_rn(beta, -d2[i][e]), top)) : 0.f;
                sum += d2[i][e];
            }
        sum += lane_xor<16>(sum);
        sum += lane_xor<32>(sum);
        __syncthreads();                                                               // every wave has read the (min, max) pairs: the slots are free again
        if (kg == 0) myred[0] = sum;
        __syncthreads();
        {
            const float* r = red + (bank * MID_SLOTS * 16 + qr) * 2;
            sum = r[0];                                                               // fixed order: every wave of the bank forms the same sum
#pragma unroll
            for (int sidx = 1; sidx < MID_SLOTS; ++sidx) sum += r[sidx * 16 * 2];
        }
        const float w = bank ? oma : alpha;
        float term[TPW][4];
#pragma unroll
        for (int i = 0; i < TPW; ++i)
#pragma unroll
            for (int e = 0; e < 4; ++e) term[i][e] = __fmul_rn(w, __fdiv_rn(d2[i][e], sum));
        float* buf = xch + slot * TPW * 4 * 64;
        if (bank) {
#pragma unroll
            for (int i = 0; i < TPW; ++i)
#pragma unroll
                for (int e = 0; e < 4; ++e) buf[(i * 4 + e) * 64 + lane] = term[i][e];
        }
        __syncthreads();
        const int m = g * 16 + qr;
        const bool mv = m < Q;
        float best = -1.f;
        int besti = 0x7fffffff;
        if (!bank) {
#pragma unroll
            for (int i = 0; i < TPW; ++i) {
                const int c0 = (slot + MID_SLOTS * i) * 16 + cls0;
                float4_t pr;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    pr[e] = __fadd_rn(term[i][e], buf[(i * 4 + e) * 64 + lane]);
                    if (c0 + e < N && pr[e] > best) { best = pr[e]; besti = c0 + e; }   // ascending classes inside the lane: first maximum kept
                }
                if (p && mv && (slot + MID_SLOTS * i) < ntiles) {
                    float* dst = p + (size_t)m * N + c0;
                    if (c0 + 3 < N && (N & 3) == 0) *reinterpret_cast<float4_t*>(dst) = pr;
                    else {
#pragma unroll
                        for (int e = 0; e < 4; ++e)
                            if (c0 + e < N) dst[e] = pr[e];
                    }
                }
            }
            if (argmax) {
                { const float ov = lane_xor<16>(best); const int oi = lane_xor_i<16>(besti); if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; } }
                { const float ov = lane_xor<32>(best); const int oi = lane_xor_i<32>(besti); if (ov > best || (ov == best && oi < besti)) { best = ov; besti = oi; } }
                if (kg == 0) { bst[(slot * 16 + qr) * 2] = best; reinterpret_cast<int*>(bst)[(slot * 16 + qr) * 2 + 1] = besti; }
            }
        }
        if (argmax) {
            __syncthreads();
            if (tid < 16 && g * 16 + tid < Q) {
                float bv = -1.f;
                int bi = 0x7fffffff;
#pragma unroll
                for (int s = 0; s < MID_SLOTS; ++s) {
                    const float v = bst[(s * 16 + tid) * 2];
                    const int ix = reinterpret_cast<const int*>(bst)[(s * 16 + tid) * 2 + 1];
                    if (v > bv || (v == bv && ix < bi)) { bv = v; bi = ix; }               // lowest class among equal maxima (main.py:190 on the CPU)
                }
                argmax[g * 16 + tid] = bi;
            }
        }
    }
}

template <int TPW, int NCH, int SL>
int launch_mid2(const void* q, const void* zi, const void* zt, int Q, int N, int D, float alpha, float oma, float beta, float* p, int32_t* argmax, int cus,
                hipStream_t s) {
    const size_t lds = classify_mid_lds(D, TPW, SL, NCH > 0);
    static DevOnce attr;
    if (!attr.done()) {
        if (hipFuncSetAttribute((const void*)classify_mid_kernel<TPW, NCH, SL>, hipFuncAttributeMaxDynamicSharedMemorySize, (int)classify_mid_lds(NCH > 0 ? NCH * 128 : 2048, TPW, SL, NCH > 0)) != hipSuccess) {
            pclip_set_error("pclip_classify_f16: cannot raise the dynamic LDS limit (mid-N kernel)");
            return PCLIP_E_LAUNCH;
        }
        attr.set();
    }
    const int ngroups = ceil_div(Q, 16);
    const int cap = (SL == 8 ? 1 : 2) * cus;
    const int grid = ngroups < cap ? ngroups : cap;
    classify_mid_kernel<TPW, NCH, SL><<<grid, 2 * SL * 64, lds, s>>>((const half_t*)q, (const half_t*)zi, (const half_t*)zt, Q, N, D, alpha, oma, beta, p, argmax);
    return pclip_check_launch("classify (mid N)");
}

template <int TPW, int SL>
int launch_mid(const void* q, const void* zi, const void* zt, int Q, int N, int D, float alpha, float oma, float beta, float* p, int32_t* argmax, int cus,
               hipStream_t s) {
    switch (D) {
        case 512: return launch_mid2<TPW, 4, SL>(q, zi, zt, Q, N, D, alpha, oma, beta, p, argmax, cus, s);
        case 768: return launch_mid2<TPW, 6, SL>(q, zi, zt, Q, N, D, alpha, oma, beta, p, argmax, cus, s);
        case 1024: return launch_mid2<TPW, 8, SL>(q, zi, zt, Q, N, D, alpha, oma, beta, p, argmax, cus, s);
        default: return launch_mid2<TPW, 0, SL>(q, zi, zt, Q, N, D, alpha, oma, beta, p, argmax, cus, s);
    }
}

}  // namespace

static int g_mid_mode = -1;                     // -1: PCLIP_CLASSIFY_MID / default (1), 0 off, 1 routed by size, 2 every shape the kernel can run
extern "C" int pclip_classify_mid_config(int mode) {
    const int before = g_mid_mode;
    if (mode >= 0) g_mid_mode = mode > 2 ? 1 : mode;
    return before;
}

// Shapes the one-launch mid-N kernel takes: both banks, p and / or argmax (top-k goes to the other routes), N <= 256, D a multiple of 128 up to 2048.  Routed
// by size: a workgroup streams both banks per 16 queries, so the kernel's time grows with Q / (16 x 2 CUs) bank passes where the two stages amortise the banks
// over 128 x 128 tiles — beyond Q N ~ 2e6 the two stages (or, from 2e6 x tiles - 1e6, the fused row panels) take over (tools/small_bench.py).
bool pclip_classify_mid_applies(int Q, int N, int D, bool has_zt, bool topk) {
    if (g_mid_mode < 0) { const char* e = getenv("PCLIP_CLASSIFY_MID"); g_mid_mode = e ? atoi(e) : 1; if (g_mid_mode < 0 || g_mid_mode > 2) g_mid_mode = 1; }
    if (!g_mid_mode || !has_zt || topk) return false;
    if (!(N >= 1 && N <= 256 && D >= 128 && D % 128 == 0 && D <= 2048 && Q >= 1)) return false;
    if (g_mid_mode == 2) return true;
    // N <= 16 stays with classify_small (ONE tile of bank fragments per wave: EuroSAT 6.4 vs 8.6 us, N = 10 / D = 1024 / Q = 8100: 9.6 vs 14.2); from two tiles on this
    // kernel measures faster (N = 17, D = 512, Q = 300: 5.5 vs 8.5 us; N = 32, Q = 4000: 5.9 vs 9.8; N = 24, D = 1024, Q = 20 000: 31.3 vs 37.2) — tools/mid_probe.py
    return N > 16 && (double)Q * (double)N <= 2.0e6;
}

int pclip_classify_mid_launch(const void* q, const void* zi, const void* zt, int Q, int N, int D, float alpha, float oma, float beta, float* p, int32_t* argmax,
                              hipStream_t s) {
    int cus = pclip_device_cus();
    if (cus <= 0) cus = 256;
    const int ntiles = ceil_div(N, 16);
    if (ntiles <= 4) return launch_mid<1, 4>(q, zi, zt, Q, N, D, alpha, oma, beta, p, argmax, cus, s);
    if (ntiles <= 8) return launch_mid<1, 8>(q, zi, zt, Q, N, D, alpha, oma, beta, p, argmax, cus, s);
    return launch_mid<2, 8>(q, zi, zt, Q, N, D, alpha, oma, beta, p, argmax, cus, s);
}
