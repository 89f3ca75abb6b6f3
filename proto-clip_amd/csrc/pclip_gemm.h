// fp16 x fp16 -> fp32 MFMA GEMM core for gfx950:  C[M,N] = A[M,K] . B[N,K]^T  (both operands
// K-contiguous — the layout of every contraction on the Proto-CLIP path: queries x prototypes
// (utils.py:230-233) and activations x nn.Linear weights (clip/model.py:176-178)).
//
// Tile configurations (template Cfg): BM x BN x 64 per workgroup of WM x WN waves, each wave owning a
// (BM/WM) x (BN/WN) block as TM x TN 32x32 accumulator tiles, each computed by 2x2 v_mfma_f32_16x16x32_f16 (default, M16)
// or one v_mfma_f32_32x32x16_f16.
//   Cfg<256,256,2,4>  512 threads, 128 acc VGPRs/lane, 128 KiB LDS, 1 workgroup/CU — large GEMMs
//   Cfg<256,128,4,2>  512 threads,  64 acc VGPRs/lane,  96 KiB LDS, 1 workgroup/CU — N = 768-wide layers
//   Cfg<128,128,2,2>  256 threads,  64 acc VGPRs/lane,  64 KiB LDS, 2 workgroups/CU — small problems
// Staging: global_load_lds_dwordx4 straight into a double-buffered LDS image (no VGPR round trip); because
// the LDS destination of that instruction is lane-linear, the bank-conflict swizzle is applied to the
// per-lane SOURCE address and undone on the ds_read_b128 side (guide §5.4 rule 21): LDS slot (row, s) holds
// global 16-byte chunk s ^ ((row >> 1) & 7) of that row, which makes every 16-lane group of a ds_read_b128
// fragment read hit 16 distinct 16-byte slots of the 256-byte bank row (conflict-free).  One barrier per
// K-tile; the next tile's loads are in flight while the current one feeds the matrix cores.
//
// The MFMA operands are SWAPPED (D = Btile . Atile^T), so a lane owns ONE output row m and, per
// accumulator register quad, FOUR CONSECUTIVE output columns: epilogues get float4 / half4 vectors.
#pragma once
#include "pclip_common.h"

namespace pgemm {

constexpr int BK = 64;
constexpr int ROW_BYTES = BK * 2;                  // 128 B of K per tile row

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int BM_, int BN_, int WM_, int WN_>
struct Cfg {
    static constexpr int BM = BM_, BN = BN_, WM = WM_, WN = WN_;
    static constexpr int NWAVES = WM * WN, NTHREADS = NWAVES * 64;
    static constexpr int TM = BM / WM / 32, TN = BN / WN / 32;      // 32x32 accumulators per wave
    static constexpr int A_BYTES = BM * ROW_BYTES, B_BYTES = BN * ROW_BYTES;
    static constexpr int STAGE_BYTES = A_BYTES + B_BYTES;
    static constexpr int LDS_BYTES = 2 * STAGE_BYTES;
    // epilogue: the output tile passes through LDS in NH row-slabs of HR rows (HR * BN * 2 <= STAGE_BYTES)
    static constexpr int HR = 128, NH = BM / HR;
    static constexpr int CPR = BN / 8;                               // 16-byte chunks per output row
    static constexpr int ROWS_PER_PASS = NTHREADS / CPR, NPASS = HR / ROWS_PER_PASS;
    static_assert(HR * BN * 2 <= STAGE_BYTES, "epilogue slab must fit one stage buffer");
    static_assert(BM % (8 * NWAVES) == 0 && BN % (8 * NWAVES) == 0, "tile rows must split over the waves");
};
using CfgSmall = Cfg<128, 128, 2, 2>;

__device__ __forceinline__ int swz_key(int row) { return (row >> 1) & 7; }

// ---- race-stress build (-DPCLIP_RACE_STRESS -> libpclip_stress.so) -------------------------------------------------------------------------
// Every hand-counted s_waitcnt vmcnt(N) / LDS-only barrier of the library goes through wait_vm / lds_barrier below, and every LDS-DMA piece of the persistent kernels
// through TileSrc / TilePairR::stage.  In the stress build each of those points first pauses the wave for 0 / 256 / 1024 cycles, pseudo-randomly per wave and call
// (clock bits ^ hardware wave id ^ call-site salt): a wave that arrives late at a barrier, a piece that is requested late, a wait that returns while the partner wave
// is far ahead.  A wait that is one piece too weak, or a barrier that does not cover a refill, then reads stale LDS in some launches; tests/test_gpu_stress.py
// demands the normal library's bits from GEMM (every tile configuration), sqdist_big, the fused classification, attention (every piece-count class) and gemm_res_ln.
#ifdef PCLIP_RACE_STRESS
__device__ __forceinline__ void stress_jitter(int salt) {
#if defined(__HIP_DEVICE_COMPILE__)
    const unsigned t = (unsigned)__builtin_readcyclecounter();
    unsigned id;
    asm volatile("s_getreg_b32 %0, hwreg(HW_REG_HW_ID, 0, 12)" : "=s"(id));
    const unsigned h = ((t >> 3) ^ (id * 0x9E3779B1u) ^ ((unsigned)salt * 0x85EBCA6Bu)) >> 7;
    if ((h & 7) == 0) __builtin_amdgcn_s_sleep(16);
    else if ((h & 7) == 1) __builtin_amdgcn_s_sleep(4);
#endif
}
#else
__device__ __forceinline__ void stress_jitter(int) {}
#endif

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain the vector-memory
// counter, so global stores and LDS-DMA prefetches stay in flight across it (guide §5 "Pipelining across
// barriers").  Every LDS read/write issued before it has completed (lgkmcnt(0)) when the wave arrives.
__device__ __forceinline__ void lds_barrier() {
    stress_jitter(1);
    asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    __builtin_amdgcn_s_barrier();
    asm volatile("" ::: "memory");
}
// Wait until at most N of this wave's vector-memory operations (in issue order) are outstanding.
// Uses the builtin (not inline asm) so that the compiler's own wait-count bookkeeping sees it.
template <int N>
__device__ __forceinline__ void wait_vm() {
    // gfx9 s_waitcnt immediate: vmcnt[3:0] | expcnt[6:4] | lgkmcnt[11:8] | vmcnt_hi[15:14]; exp/lgkm left at max
    __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14));
    stress_jitter(2 + N);
}

#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#else
struct rsrc_t { int w[4]; };                     // the host pass only parses the kernels; the descriptor type is device-only
#endif

// A buffer descriptor over [base, base + bytes): LDS-DMA through buffer_load ... lds instead of global_load_lds.  Two reasons (round 6, csrc/pclip_conv_strip.hip):
// (i) hipcc models a FLAT-encoded LDS load as "may touch LDS and memory": while one is in flight EVERY LDS wait it emits is lgkmcnt(0) — a K-loop that prefetches its
// next tile by global_load_lds cannot overlap its fragment reads with its MFMAs; behind a buffer load the waits are counted (lgkmcnt(N)); (ii) offsets beyond `bytes`
// return zeros: padding without a zero line.  The inputs go through readfirstlane: hipcc must be able to PROVE the descriptor wave-uniform, otherwise every buffer op
// is wrapped in a waterfall loop (guide T20).
__device__ __forceinline__ rsrc_t make_rsrc(const void* base, unsigned bytes) {
#if defined(__HIP_DEVICE_COMPILE__)
    const uint64_t addr = (uint64_t)base;
    const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)addr), hi = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
    return __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, (int)bytes, 0x00020000);
#else
    return rsrc_t{};
#endif
}

// All waves of the workgroup stage a ROWS x 64 tile: each LDS-DMA piece is 8 rows x 128 B (64 lanes x 16 B).  The descriptor starts at the tile's first row, so
// the lane offsets stay small whatever the matrix (rows beyond nrows re-read the last row: never stored).
template <int ROWS, int NWAVES>
__device__ __forceinline__ void stage_tile(const half_t* __restrict__ g, int ld, int row0, int nrows, int k0,
                                           char* lds_tile, int wave, int lane) {
    constexpr int RPW = ROWS / NWAVES;                               // rows per wave
    const rsrc_t rs = make_rsrc(g + (size_t)row0 * ld, 0x7fffffffu);
    const int last = nrows - 1 - row0;
    (void)rs; (void)last;
#pragma unroll
    for (int i = 0; i < RPW / 8; ++i) {
        const int r = wave * RPW + i * 8 + (lane >> 3);              // tile row this lane fills
        const int c = (lane & 7) ^ swz_key(r);                       // source chunk for LDS slot (lane&7)
        const int voff = ((r < last ? r : last) * ld + k0 + c * 8) * 2;
        char* dst = lds_tile + (wave * RPW + i * 8) * ROW_BYTES;     // wave-uniform base; HW adds lane*16
        (void)voff; (void)dst;
#if defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)dst, 16, voff, 0, 0, 0);
#endif
    }
}

__device__ __forceinline__ half8_t lds_frag(const char* lds_tile, int row, int kc) {
    return *reinterpret_cast<const half8_t*>(lds_tile + row * ROW_BYTES + ((kc ^ swz_key(row)) << 4));
}

// Measurement hook of the GEMM kernels (bench.py's roofline; pclip_gemm_timing): with a slot pointer, every workgroup folds the device's constant 100 MHz counter
// into slot[0] (minimum: the launch's first instruction) and slot[1] (maximum: its last) — the span rocprofv3 reports as the kernel's duration, taken INSIDE the
// step, without a launch or an event between the kernels (either puts ~33 us between consecutive GEMMs).  A null pointer (the product) costs one scalar branch.
__device__ __forceinline__ void time_begin(unsigned long long* slot) {
    if (slot && threadIdx.x == 0) atomicMin(slot, (unsigned long long)wall_clock64());
}
__device__ __forceinline__ void time_end(unsigned long long* slot) {
    if (slot && threadIdx.x == 0) atomicMax(slot + 1, (unsigned long long)wall_clock64());
}

// XCD-aware, bijective remap of a linear workgroup id (guide §5.5 T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Accumulator layout after mainloop(): acc.v[i][j][e] is C[m][n] with (wm, wn = wave / WN, wave % WN)
//   m = m0 + wm*(BM/WM) + i*32 + (lane & 31)
//   n = n0 + wn*(BN/WN) + j*32 + 8*(e >> 2) + 4*(lane >> 5) + (e & 3)
template <class C>
struct Acc {
    float16_t v[C::TM][C::TN];
};

// Four consecutive-column accumulators (quad g of 32x32 tile (i, j)) of either accumulator container: the eight-wave kernels keep 32x32 tiles of 16 registers
// (Acc), the four-wave kernel (pclip_gemm4w.hip) one float4 per 16x16 MFMA tile (Acc16: tile (2 i + (g >> 1), 2 j + (g & 1)) — the operands of its asm K-loop).
struct Acc16 {
    float4_t q[8][8];
};
template <class C>
__device__ __forceinline__ float4_t acc_quad(const Acc<C>& acc, int i, int j, int g) {
    return float4_t{acc.v[i][j][4 * g], acc.v[i][j][4 * g + 1], acc.v[i][j][4 * g + 2], acc.v[i][j][4 * g + 3]};
}
__device__ __forceinline__ float4_t acc_quad(const Acc16& acc, int i, int j, int g) { return acc.q[2 * i + (g >> 1)][2 * j + (g & 1)]; }

// Stage K-tile 0 of output tile (m0, n0) into buffer p.
template <class C>
__device__ __forceinline__ void stage_first(const half_t* __restrict__ A, int lda, const half_t* __restrict__ B, int ldb,
                                            int M, int N, int m0, int n0, char* smem, int p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* a = smem + p * C::STAGE_BYTES;
    stage_tile<C::BM, C::NWAVES>(A, lda, m0, M, 0, a, wave, lane);
    stage_tile<C::BN, C::NWAVES>(B, ldb, n0, N, 0, a + C::A_BYTES, wave, lane);
}

// K-loop over a tile whose K-tile 0 has ALREADY been staged into buffer `p` (0/1) by stage_first().
// On return `p` names the FREE buffer (the one that held K-tile nt-2): a persistent caller stages the next
// output tile's K-tile 0 there before running its epilogue out of the other buffer.
// YOUNGER = number of vector-memory operations this wave issued AFTER the K-tile-0 staging and that may stay
// in flight across the first barrier (epilogue stores of the previous tile, bias loads): the first wait is
// vmcnt(YOUNGER) instead of a full drain.
// `stage_a(t, dst)` stages K-tile t of the tile's A operand (BM rows x 64 halves, swizzled as stage_tile does) into dst.
// M16: the 32x32 accumulator tile is computed as 2x2 v_mfma_f32_16x16x32_f16 (K = 32 per instruction) instead of one
// v_mfma_f32_32x32x16_f16: same FLOPs per cycle and the same LDS traffic, a quarter of the accumulator-register traffic per
// FLOP.  Element e of acc.v[i][j] is then sub-tile (a, b) = (e >> 3, (e >> 2) & 1), row a*16 + (lane & 15), columns
// b*16 + 4*(lane >> 4) + (e & 3).
template <class C, int YOUNGER = 0, bool ZERO_ACC = true, bool M16 = false, class StageA>
__device__ __forceinline__ void mainloop_g(const StageA& stage_a, const half_t* __restrict__ B, int ldb, int N, int nt, int n0,
                                           char* smem, Acc<C>& acc, int& p, bool counted_first = false) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN;
    if (ZERO_ACC) {                                        // otherwise the caller pre-loaded the accumulators (e.g. with the bias)
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
            for (int j = 0; j < C::TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc.v[i][j][e] = 0.f;
    }

    for (int t = 0; t < nt; ++t) {
        // this wave's share of K-tile t has landed; the barrier then publishes every wave's share
        if (t == 0 && counted_first) wait_vm<YOUNGER>(); else wait_vm<0>();
        lds_barrier();
        const char* la = smem + p * C::STAGE_BYTES;
        const char* lb = la + C::A_BYTES;
        // The two waves that share a SIMD (wave w and w + NWAVES/2) issue the next tile's LDS-DMA at different
        // points of the iteration, so that one of them feeds the matrix pipe while the other spends its ~400
        // cycles of address arithmetic / M0 set-up.
        const bool late = C::NWAVES == 8 && __builtin_amdgcn_readfirstlane(wave) >= 4;
        auto stage_next = [&]() {
            if (t + 1 < nt) {
                char* na = smem + (p ^ 1) * C::STAGE_BYTES;
                stage_a(t + 1, na);
                stage_tile<C::BN, C::NWAVES>(B, ldb, n0, N, (t + 1) * BK, na + C::A_BYTES, wave, lane);
            }
        };
        if (!late) stage_next();
        if (M16) {
#pragma unroll
            for (int ks = 0; ks < BK / 32; ++ks) {
                if (ks == BK / 64 && late) stage_next();
                const int kc = ks * 4 + (lane >> 4);                          // 16-byte chunk: k = ks*32 + 8*(lane >> 4) ..
                half8_t af[C::TM][2], bf[C::TN][2];
#pragma unroll
                for (int i = 0; i < C::TM; ++i)
#pragma unroll
                    for (int a = 0; a < 2; ++a) af[i][a] = lds_frag(la, wm * (C::BM / C::WM) + i * 32 + a * 16 + (lane & 15), kc);
#pragma unroll
                for (int j = 0; j < C::TN; ++j)
#pragma unroll
                    for (int b = 0; b < 2; ++b) bf[j][b] = lds_frag(lb, wn * (C::BN / C::WN) + j * 32 + b * 16 + (lane & 15), kc);
#pragma unroll
                for (int i = 0; i < C::TM; ++i)
#pragma unroll
                    for (int j = 0; j < C::TN; ++j)
#pragma unroll
                        for (int a = 0; a < 2; ++a)
#pragma unroll
                            for (int b = 0; b < 2; ++b) {
                                float4_t c = {acc.v[i][j][(a * 2 + b) * 4], acc.v[i][j][(a * 2 + b) * 4 + 1], acc.v[i][j][(a * 2 + b) * 4 + 2],
                                              acc.v[i][j][(a * 2 + b) * 4 + 3]};
                                c = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j][b], af[i][a], c, 0, 0, 0);
#pragma unroll
                                for (int r = 0; r < 4; ++r) acc.v[i][j][(a * 2 + b) * 4 + r] = c[r];
                            }
            }
            p ^= 1;
            continue;
        }
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            if (ks == BK / 32 && late) stage_next();
            const int kc = ks * 2 + (lane >> 5);
            half8_t af[C::TM], bf[C::TN];
#pragma unroll
            for (int i = 0; i < C::TM; ++i) af[i] = lds_frag(la, wm * (C::BM / C::WM) + i * 32 + (lane & 31), kc);
#pragma unroll
            for (int j = 0; j < C::TN; ++j) bf[j] = lds_frag(lb, wn * (C::BN / C::WN) + j * 32 + (lane & 31), kc);
#pragma unroll
            for (int i = 0; i < C::TM; ++i)
#pragma unroll
                for (int j = 0; j < C::TN; ++j)
                    acc.v[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc.v[i][j], 0, 0, 0);
        }
        p ^= 1;
    }
}

template <class C, int YOUNGER = 0, bool ZERO_ACC = true, bool M16 = false>
__device__ __forceinline__ void mainloop(const half_t* __restrict__ A, int lda, const half_t* __restrict__ B, int ldb,
                                         int M, int N, int K, int m0, int n0, char* smem, Acc<C>& acc, int& p,
                                         bool counted_first = false) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    mainloop_g<C, YOUNGER, ZERO_ACC, M16>([&](int t, char* dst) { stage_tile<C::BM, C::NWAVES>(A, lda, m0, M, t * BK, dst, wave, lane); },
                                     B, ldb, N, K / BK, n0, smem, acc, p, counted_first);
}

// ---- buffer-descriptor staging + register-pipelined K-loop (the persistent linear kernels) ---------------------------------
// The K-loop above leaves two things on the table (read off its ISA): (1) every LDS-DMA piece recomputes a 64-bit
// per-lane global address (3 v_lshl_add_u64 + readfirstlane + s_mov m0 per piece, ~90 VGPRs of address state for the 16
// pieces of a 256x256 tile), which is why the kernel sat at the 256-register cap with scratch spills; (2) hipcc reads two
// fragments, waits lgkmcnt(0), issues 8 MFMAs, reads the next two ... — the LDS latency is exposed eight times per K-tile.
// Here a tile operand is ONE buffer descriptor (SGPRs, base = first row of the tile) + one 32-bit byte offset per piece and
// lane, fixed for the whole tile; the K-tile advances through the instruction's SCALAR offset (no VALU work per piece:
// s_mov m0 + buffer_load ... lds).  The fragments are software-pipelined in GROUPS of (half of the wave's A rows) x (all
// of its B columns) x one 32-wide k-step: while the 2*TM*TN MFMAs of a group issue, the next group's fragments are already
// on their way from LDS (A next half; B of the next k-step during the second half).  Only the first group of a K-tile waits
// for LDS with nothing to cover it.  Same k-order per accumulator as mainloop_g's M16 branch: bit-identical results.

#ifndef PCLIP_NT_A
#define PCLIP_NT_A 0             // cache-policy bits of the A-operand LDS-DMA (2 = nt measured 7 % slower: every A line has 3 - 12 readers)
#endif
template <int ROWS, int NWAVES>
struct TileSrc {
    static constexpr int RPW = ROWS / NWAVES, NL = RPW / 8;          // rows per wave, LDS-DMA pieces per wave and K-tile
    rsrc_t rs;
    int voff[NL];
    // rows >= nrows are clamped to the last row (never stored); row0 <= nrows - 1
    __device__ __forceinline__ void prepare(const half_t* __restrict__ g, int ld, int row0, int nrows, int wave, int lane) {
#if defined(__HIP_DEVICE_COMPILE__)
        // descriptor inputs through readfirstlane: hipcc must be able to PROVE the descriptor wave-uniform, otherwise every
        // buffer op is wrapped in a waterfall loop (guide T20)
        const uint64_t addr = (uint64_t)(g + (size_t)row0 * ld);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)addr), hi = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, 0x7fffffff, 0x00020000);
#endif
        const int last = nrows - 1 - row0;
#pragma unroll
        for (int i = 0; i < NL; ++i) {
            const int r = wave * RPW + i * 8 + (lane >> 3);
            const int c = (lane & 7) ^ swz_key(r);
            voff[i] = ((r < last ? r : last) * ld + c * 8) * 2;
        }
    }
    // AUX: cache-policy bits of the buffer load (0 default, 2 = nt: streamed operand)
    template <int AUX = 0>
    __device__ __forceinline__ void stage(int k_bytes, char* lds_tile, int wave) const {
        stress_jitter(100);
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < NL; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(lds_tile + (wave * RPW + i * 8) * ROW_BYTES), 16, voff[i], k_bytes, 0, AUX);
#endif
    }
};

template <class C>
struct TilePair {
    static constexpr bool ROLES = false;
    static constexpr int NA = TileSrc<C::BM, C::NWAVES>::NL, NB = TileSrc<C::BN, C::NWAVES>::NL;
    TileSrc<C::BM, C::NWAVES> a;
    TileSrc<C::BN, C::NWAVES> b;
    __device__ __forceinline__ void prepare(const half_t* __restrict__ A, int lda, const half_t* __restrict__ B, int ldb, int M, int N,
                                            int m0, int n0, int wave, int lane) {
        a.prepare(A, lda, m0, M, wave, lane);
        b.prepare(B, ldb, n0, N, wave, lane);
    }
    __device__ __forceinline__ void stage(int t, char* stage_buf, int wave) const {
        a.template stage<PCLIP_NT_A>(t * (BK * 2), stage_buf, wave);
        b.template stage<0>(t * (BK * 2), stage_buf + C::A_BYTES, wave);
    }
};

// ---- role split of the LDS-DMA issue (eight-wave tiles) ---------------------------------------------------------------------
// An LDS-DMA instruction costs its wave 60 - 185 issue cycles (guide, microarch table) during which the wave issues no MFMA, and in
// TilePair every wave issues four pieces at each refill point — i.e. BOTH waves of every SIMD (w and w + 4) stall on DMA issue at the
// same moment, right behind the same barrier, and the matrix pipe of that SIMD idles.  Here waves 0 - 3 stage the whole B tile (at the B
// refill point) and waves 4 - 7 the whole A tile (at the A refill point): each wave still issues (BM or BN) / 32 pieces per K-tile, but in
// one burst while its SIMD partner feeds the matrix pipe.  Same LDS image, same fragment order: bit-identical results.
#ifndef PCLIP_DMA_ROLES
#define PCLIP_DMA_ROLES 1
#endif
template <class C>
struct TilePairR {
    static constexpr bool ROLES = true;
    static_assert(C::NWAVES == 8, "role split: waves w and w + 4 share a SIMD");
    static constexpr int NA = C::BM / 32, NB = C::BN / 32;             // pieces per A wave / per B wave and K-tile
    static_assert(C::BM % 64 == 0 && C::BN % 64 == 0, "a wave stages ROWS / 4 rows, a multiple of 16");
    // Per-lane state: TWO byte offsets.  Piece i of a wave covers rows w4 * ROWS/4 + 8 i + (lane >> 3); the swizzle key (row >> 1) & 7
    // only depends on the parity of i, so piece i = piece (i & 1) + (i >> 1) * 16 rows, and those 16 rows travel in the instruction's
    // SCALAR offset together with the K-tile.  Rows beyond the operand are not clamped but cut off by the descriptor's size (a buffer
    // load past num_records returns zeros; such rows are never stored).
    rsrc_t rs;
    int voff[2];
    int row16;                                                        // bytes of 16 operand rows (wave-uniform)
    bool is_b;                                                        // wave-uniform
    __device__ __forceinline__ void prepare(const half_t* __restrict__ A, int lda, const half_t* __restrict__ B, int ldb, int M, int N,
                                            int m0, int n0, int wave, int lane) {
        is_b = wave < 4;
        const half_t* g = is_b ? B : A;
        const int ld = is_b ? ldb : lda, row0 = is_b ? n0 : m0, nrows = is_b ? N : M, rpw = (is_b ? C::BN : C::BM) / 4, w4 = wave & 3;
#if defined(__HIP_DEVICE_COMPILE__)
        const uint64_t addr = (uint64_t)(g + (size_t)row0 * ld);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)addr), hi = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
        const long left = (long)(nrows - row0) * ld * 2;               // bytes from the tile's first row to the end of the operand
        const uint32_t size = __builtin_amdgcn_readfirstlane((uint32_t)(left < 0x7fffffffL ? left : 0x7fffffffL));
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, size, 0x00020000);
#endif
        row16 = 16 * ld * 2;
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = w4 * rpw + i * 8 + (lane >> 3);
            const int c = (lane & 7) ^ swz_key(r);
            voff[i] = (r * ld + c * 8) * 2;
        }
    }
    // this wave's share of K-tile t into stage buffer `stage_buf` (A image at +0, B image at +A_BYTES)
    __device__ __forceinline__ void stage(int t, char* stage_buf, int wave) const {
        stress_jitter(101);
#if defined(__HIP_DEVICE_COMPILE__)
        const int w4 = wave & 3, k = t * (BK * 2);
        if (is_b) {
#pragma unroll
            for (int i = 0; i < NB; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(stage_buf + C::A_BYTES + (w4 * (C::BN / 4) + i * 8) * ROW_BYTES), 16, voff[i & 1],
                                                         k + (i >> 1) * row16, 0, 0);
        } else {
#pragma unroll
            for (int i = 0; i < NA; ++i)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(stage_buf + (w4 * (C::BM / 4) + i * 8) * ROW_BYTES), 16, voff[i & 1],
                                                         k + (i >> 1) * row16, 0, PCLIP_NT_A);
        }
#endif
    }
};

// ---- staggered refill: the same K-loop with the LDS-DMA of K-tile t + 2 issued DURING iteration t --------------------------
// tools/probe/dma_probe.hip: what bounds the operand delivery of a CU is the number of bytes it keeps in flight — a 64 KB
// burst that is awaited before the next one is issued (mainloop_bl: buffer p ^ 1 is refilled at the top of iteration t) delivers
// 1.2 - 1.4 us per K-tile at the bench's shapes, two 32 KB stages continuously in flight 0.9 - 1.2 us; touching the lines early
// (L2 prefetch) does nothing.  There is no LDS for a third buffer, but the buffer being CONSUMED frees up in two steps: its B
// half once every wave holds the k-step-1 B fragments in registers (before the third group of MFMAs), its A half once the last
// A fragments are in (before the fourth).  Two extra LDS-only barriers mark those points and K-tile t + 2 is requested right
// there — B first, then A — into the half that just became free, i.e. 1.3 - 1.6 iterations ahead of its use instead of 1.0, and
// the requests of a K-tile are spread over the iteration instead of bursting at its top.  Per wave the iteration issues NB then
// NA pieces, so the wait at the top of iteration t + 1 is the counted vmcnt(NA + NB): K-tile t + 1 (requested during t - 1) has
// landed, K-tile t + 2 stays in flight.  Same fragment order as mainloop_bl: bit-identical results.
// Measured alternatives (same-process A/B, tools/ab_lib.py): ONE mid-iteration barrier with both halves requested at 75 % is
// 4 - 8 % slower (the spread of the requests matters more than the barrier); reading every fragment earlier so that the halves
// free up at 25 % / 50 % spills (72 B of scratch) and is 1 - 6 % slower; __builtin_amdgcn_sched_barrier(0) after every block of
// fragment reads (so that hipcc cannot sink them behind the MFMAs they are meant to overlap) gives the intended read-ahead
// stream but costs registers: equal where it does not spill (bias + QuickGELU), 30 % slower where it does (88 B, bias only).
#ifndef PCLIP_IGLP
#define PCLIP_IGLP 1             // __builtin_amdgcn_iglp_opt strategy of the K-loop's first scheduling region (-1: none; 0 / 2 / 3 measured: no gain)
#endif
// K-loop over a tile whose K-tile 0 is already on its way into buffer `p` (TP::stage(0, ..)); semantics of `p`, YOUNGER, counted_first as
// mainloop_g.  `wave` must be wave-uniform (readfirstlane).
template <class C, int YOUNGER = 0, bool ZERO_ACC = true, class TP = TilePair<C>>
__device__ __forceinline__ void mainloop_sr(const TP& tp, int nt, char* smem, Acc<C>& acc, int& p, bool counted_first, int wave, int lane) {
    static_assert(C::TM >= 2 && C::TM % 2 == 0, "group pipeline splits the wave's A rows in two halves");
    constexpr int HM = C::TM / 2;
    constexpr int NA = TP::NA, NB = TP::NB;
    // requests of ONE K-tile this wave leaves in flight across the top barrier: every wave NA + NB pieces (TilePair), or the pieces
    // of its own operand (TilePairR)
    auto wait_ahead = [&]() {
        if constexpr (TP::ROLES) { if (wave < 4) wait_vm<NB>(); else wait_vm<NA>(); }
        else wait_vm<NA + NB>();
    };
    const int wm = wave / C::WN, wn = wave % C::WN;
    if (ZERO_ACC) {
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
            for (int j = 0; j < C::TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc.v[i][j][e] = 0.f;
    }
    const int key = (lane & 15) >> 1, q = lane >> 4;
    const int col0 = (q ^ key) << 4;
    const int offa = (wm * (C::BM / C::WM) + (lane & 15)) * ROW_BYTES + col0;
    const int offb = C::A_BYTES + (wn * (C::BN / C::WN) + (lane & 15)) * ROW_BYTES + col0;
    // (s_setprio measured on this loop, tools/ab_multi.py gemm, profiles/r03_ab_gemm_prio.txt: priority 1 around every MFMA group
    // is neutral at N = 3072 and 5 - 32 % SLOWER at N = 768 / 2304; a static priority for the younger half of the waves is +-0.5 %.)

    for (int t = 0; t < nt; ++t) {
        if (t == 0) { if (counted_first) wait_vm<YOUNGER>(); else wait_vm<0>(); }
        else if (t + 1 < nt) wait_ahead();
        else wait_vm<0>();
        lds_barrier();
        char* cur = smem + p * C::STAGE_BYTES;
        const char* base = cur;
        auto fa = [&](int ks, int i, int a) { return *reinterpret_cast<const half8_t*>(base + (offa ^ (ks << 6)) + (i * 32 + a * 16) * ROW_BYTES); };
        auto fb = [&](int ks, int j, int b) { return *reinterpret_cast<const half8_t*>(base + (offb ^ (ks << 6)) + (j * 32 + b * 16) * ROW_BYTES); };
        half8_t bcur[C::TN][2], acur[HM][2], anext[HM][2], bnext[C::TN][2];
        auto load_a = [&](half8_t (&dst)[HM][2], int ks, int half) {
#pragma unroll
            for (int i = 0; i < HM; ++i)
#pragma unroll
                for (int a = 0; a < 2; ++a) dst[i][a] = fa(ks, half * HM + i, a);
        };
        auto load_b = [&](half8_t (&dst)[C::TN][2], int ks) {
#pragma unroll
            for (int j = 0; j < C::TN; ++j)
#pragma unroll
                for (int b = 0; b < 2; ++b) dst[j][b] = fb(ks, j, b);
        };
        auto group = [&](const half8_t (&af)[HM][2], const half8_t (&bf)[C::TN][2], int half) {
#pragma unroll
            for (int i = 0; i < HM; ++i)
#pragma unroll
                for (int j = 0; j < C::TN; ++j)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            float16_t& dst = acc.v[half * HM + i][j];
                            float4_t c = {dst[(a * 2 + b) * 4], dst[(a * 2 + b) * 4 + 1], dst[(a * 2 + b) * 4 + 2], dst[(a * 2 + b) * 4 + 3]};
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j][b], af[i][a], c, 0, 0, 0);
#pragma unroll
                            for (int r = 0; r < 4; ++r) dst[(a * 2 + b) * 4 + r] = c[r];
                        }
        };
        const bool refill = t + 2 < nt;                               // workgroup-uniform
#if PCLIP_IGLP >= 0 && defined(__HIP_DEVICE_COMPILE__)
        __builtin_amdgcn_iglp_opt(PCLIP_IGLP);   // LLVM's DS-read / MFMA interleaving for the region behind the first barrier: +1.4 - 2.2 % on every bench shape (r03_ab_gemm_sched.txt), same bits
#endif
        load_b(bcur, 0);
        load_a(acur, 0, 0);
        if (t == 0 && nt > 1) tp.stage(1, smem + (p ^ 1) * C::STAGE_BYTES, wave);   // the other buffer was the previous tile's epilogue staging area until the barrier above
        load_a(anext, 0, 1);
        group(acur, bcur, 0);                    // ks 0, rows half 0
        load_b(bnext, 1);
        load_a(acur, 1, 0);
        group(anext, bcur, 1);                   // ks 0, rows half 1
        if (refill) {
            lds_barrier();                       // every wave holds its last B fragments of this K-tile: the B half of `cur` is free
            if constexpr (TP::ROLES) { if (wave < 4) tp.stage(t + 2, cur, wave); }
            else tp.b.template stage<0>((t + 2) * (BK * 2), cur + C::A_BYTES, wave);
        }
        load_a(anext, 1, 1);
        group(acur, bnext, 0);                   // ks 1, rows half 0
        if (refill) {
            lds_barrier();                       // ... and its last A fragments: the A half is free
            if constexpr (TP::ROLES) { if (wave >= 4) tp.stage(t + 2, cur, wave); }
            else tp.a.template stage<PCLIP_NT_A>((t + 2) * (BK * 2), cur, wave);
        }
        group(anext, bnext, 1);                  // ks 1, rows half 1
        p ^= 1;
    }
}

// (Measured and removed, round 4, profiles/r04_ab_drain.txt item 5: EVERY fragment of the K-tile read behind the top barrier (96 registers), ONE barrier that frees
// the whole buffer, waves 0 - 3 issuing their eight pieces right behind it and waves 4 - 7 a group and a half later — each burst beside >= 32 MFMAs of
// the SIMD partner with no barrier in between.  A timeline model with independent MFMA / vector-memory issue predicts -20 % per K-tile; measured,
// bit-identical: in_proj +1 %, c_fc +3 %, the residual instantiations +6 ... +10 % (108 B of scratch): the 24 up-front fragment reads (768 LDS cycles for the
// workgroup) and the late first burst cost what the freed barrier saves.  tools/probe/issue_overlap_probe.hip (profiles/r04_issue_overlap_probe.txt) shows the chip itself
// overlaps the pieces with MFMAs at + 6 % when nothing couples them; a barrier between the roles' bursts costs + 18 %.)
// (Measured and removed, profiles/r03_ab_gemm_sched.txt: the same loop software-pipelined ACROSS the K-tile boundary — the barrier that
// publishes K-tile t + 1 moved in front of the last MFMA group of iteration t and the first fragments of t + 1 requested under that
// group, so that an iteration starts issuing MFMAs at once.  Bit-identical and 4 - 6 % SLOWER: waiting for K-tile t + 1 a quarter of an
// iteration earlier exposes the operand delivery itself — the LDS-DMA round trip of ~1.25 iterations is what this loop waits for, not
// the fragment latency behind the first barrier.)


// ---- implicit-GEMM gather for a 3x3 / stride 1 / pad 1 convolution on NHWC fp16 activations ------------------------------
// Row m of the GEMM is output pixel (b, y, x); K-tile t covers tap = (t*64) / Cin and input channels c0 = (t*64) % Cin
// (Cin % 64 == 0; Cin = 8 / 16 / 32 take the per-chunk path of stage()), i.e. the 128 contiguous bytes x[b, y+dy-1, x+dx-1, c0 .. c0+63] — or 128 zero bytes outside the image:
// a buffer load beyond the descriptor's range (an LDS-DMA cannot be predicated per lane without leaving stale LDS; round 6: was a caller-provided zero line).
// The im2col matrix (9x the activation) is never materialised.
template <class C>
struct ConvGather {
    static constexpr int NR = C::BM / C::NWAVES / 8;       // A rows per lane
    const half_t* __restrict__ x;
    int H, W, Cin, M;
    rsrc_t rs;                                             // over the M x Cin activations: a tap outside the image takes an offset beyond it (zeros)
    int pix[NR], yx[NR];
    int off[NR], taps[NR];                                 // Cin % 64 == 0: byte offset of the lane's chunk of pixel i (swizzled), 9-bit mask of the taps inside the image
    __device__ __forceinline__ ConvGather(const half_t* x_, int H_, int W_, int Cin_, int M_)
        : x(x_), H(H_), W(W_), Cin(Cin_), M(M_), rs(make_rsrc(x_, (unsigned)M_ * (unsigned)Cin_ * 2u)) {}
    __device__ __forceinline__ void prepare(int m0) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = wave * (C::BM / C::NWAVES) + i * 8 + (lane >> 3);
            int m = m0 + r;
            m = m < M ? m : M - 1;                          // rows beyond M are never stored
            const int b = m / (H * W), rem = m - b * H * W, y = rem / W, xx = rem - y * W;
            pix[i] = m;                                     // (b*H + y)*W + x == m for stride 1 / pad 1
            yx[i] = (y << 16) | xx;
            off[i] = (m * Cin + (((lane & 7) ^ swz_key(r)) << 3)) * 2;
            const int rows = (y > 0 ? 1 : 0) | 2 | (y + 1 < H ? 4 : 0), cols = (xx > 0 ? 1 : 0) | 2 | (xx + 1 < W ? 4 : 0);
            taps[i] = ((rows & 1) ? cols : 0) | ((rows & 2) ? cols << 3 : 0) | ((rows & 4) ? cols << 6 : 0);
        }
    }
    // K-tile t into `dst` (the A image of a stage buffer); `wave` wave-uniform (readfirstlane): the LDS destination travels in M0
    __device__ __forceinline__ void stage(int t, char* dst, int wave) const {
        const int lane = threadIdx.x & 63;
        (void)lane;
        if (Cin < BK) {
            // Cin = 8 / 16 / 32 (the ResNet stem): a K-tile spans 64 / Cin taps, so the tap belongs to the 16-byte chunk, not to the
            // row; taps >= 9 (K padded to the K-tile, zero weights there) read zeros
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int r = wave * (C::BM / C::NWAVES) + i * 8 + (lane >> 3);
                const int c = (lane & 7) ^ swz_key(r);
                const int k = t * BK + c * 8, tap = k / Cin, ci = k - tap * Cin, dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                const int y = (yx[i] >> 16) + dy, xx = (yx[i] & 0xffff) + dx;
                const bool in = tap < 9 && (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
                const unsigned voff = in ? (unsigned)((pix[i] + dy * W + dx) * Cin + ci) * 2u : 0xfffffff0u;
                (void)voff;
#if defined(__HIP_DEVICE_COMPILE__)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + (wave * (C::BM / C::NWAVES) + i * 8) * ROW_BYTES), 16, (int)voff, 0, 0, 0);
#endif
            }
            return;
        }
        // Cin % 64 == 0: the tap and the channel block are wave-uniform — one scalar offset per K-tile, per piece a mask test and an add
        const int k0 = t * BK, tap = k0 / Cin, c0 = k0 - tap * Cin, dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
        const int tapoff = ((dy * W + dx) * Cin + c0) * 2;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const unsigned voff = (taps[i] >> tap) & 1 ? (unsigned)(off[i] + tapoff) : 0xfffffff0u;
            (void)voff;
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + (wave * (C::BM / C::NWAVES) + i * 8) * ROW_BYTES), 16, (int)voff, 0, 0, 0);
#endif
        }
    }
    __device__ __forceinline__ void stage(int t, char* dst) const { stage(t, dst, (int)(threadIdx.x >> 6)); }
};

// The convolution's operand pair for mainloop_sr (the software-pipelined K-loop with the staggered refill): A = the gather above, B = the weights [Cout, 9 Cin].
template <class C>
struct ConvPair {
    static constexpr bool ROLES = false;
    static constexpr int NA = ConvGather<C>::NR, NB = TileSrc<C::BN, C::NWAVES>::NL;
    struct AOp {
        ConvGather<C> g;
        template <int AUX = 0>
        __device__ __forceinline__ void stage(int k_bytes, char* dst, int wave) const { g.stage(k_bytes / (BK * 2), dst, wave); }
    } a;
    TileSrc<C::BN, C::NWAVES> b;
    __device__ __forceinline__ ConvPair(const half_t* x, int H, int W, int Cin, int M) : a{ConvGather<C>(x, H, W, Cin, M)} {}
    __device__ __forceinline__ void stage(int t, char* stage_buf, int wave) const {
        a.g.stage(t, stage_buf, wave);
        b.template stage<0>(t * (BK * 2), stage_buf + C::A_BYTES, wave);
    }
};

// ---- latency-oriented K-loop: an NS-slot ring of K-tiles, NS-1 LDS-DMA stages in flight -------------------------------
// For launches with fewer tiles than CUs (a serving request) the double-buffered loop above is a chain of dependent round
// trips (~0.7 us per K-tile).  Here K-tiles t+1 .. t+NS-1 are in flight while tile t is multiplied (Little: a K-tile every
// latency / (NS-1)); every wave issues the same number of LDS-DMA instructions per stage (PER), so the wait before tile t
// is a counted vmcnt(stages still allowed in flight x PER).  16x16x32 MFMAs, same fragment order as mainloop_g's M16 branch
// (bit-identical accumulation).  `init()` runs after the first NS-1 stages are on their way and sets the accumulators (zero,
// or the bias as in linear_fast_kernel: its loads overlap the stages' latency).
// `stage_a(t, dst)` stages K-tile t of the A operand (stage_tile, or ConvGather::stage for the implicit convolution).
template <class C, int NS, class StageA, class Init>
__device__ __forceinline__ void mainloop_ring_g(const StageA& stage_a, const half_t* __restrict__ B, int ldb, int N, int nt, int n0,
                                                char* smem, Acc<C>& acc, const Init& init) {
    constexpr int PER = (C::BM + C::BN) / (8 * C::NWAVES);          // LDS-DMA instructions per wave per stage
    static_assert(NS >= 3 && NS <= 6 && (NS - 2) * PER < 64, "ring depth / vmcnt range");
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN;
    auto stage = [&](int t, int slot) {
        char* a = smem + slot * C::STAGE_BYTES;
        stage_a(t, a);
        stage_tile<C::BN, C::NWAVES>(B, ldb, n0, N, t * BK, a + C::A_BYTES, wave, lane);
    };
    for (int t = 0; t < NS - 1 && t < nt; ++t) stage(t, t);
    init();
    int slot = 0, fill = NS - 1;                                    // slot of tile t; slot of tile t + NS - 1
    for (int t = 0; t < nt; ++t) {
        const int ahead = nt - 1 - t;                               // younger stages already issued: min(ahead, NS - 2)
        if (NS >= 6 && ahead >= 4) wait_vm<4 * PER>();
        else if (NS >= 5 && ahead >= 3) wait_vm<3 * PER>();
        else if (NS >= 4 && ahead >= 2) wait_vm<2 * PER>();
        else if (ahead >= 1) wait_vm<PER>();
        else wait_vm<0>();
        lds_barrier();                                              // tile t visible; the slot of tile t-1 is free (everyone is past it)
        if (t + NS - 1 < nt) stage(t + NS - 1, fill);
        fill = slot;                                                // next iteration refills the slot being consumed now
        const char* la = smem + slot * C::STAGE_BYTES;
        slot = slot + 1 == NS ? 0 : slot + 1;
        const char* lb = la + C::A_BYTES;
        // every fragment of the K-tile first (ONE exposed LDS latency per tile), then the MFMAs in mainloop_g's order
        half8_t af[BK / 32][C::TM][2], bf[BK / 32][C::TN][2];
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks) {
            const int kc = ks * 4 + (lane >> 4);
#pragma unroll
            for (int i = 0; i < C::TM; ++i)
#pragma unroll
                for (int a = 0; a < 2; ++a) af[ks][i][a] = lds_frag(la, wm * (C::BM / C::WM) + i * 32 + a * 16 + (lane & 15), kc);
#pragma unroll
            for (int j = 0; j < C::TN; ++j)
#pragma unroll
                for (int b = 0; b < 2; ++b) bf[ks][j][b] = lds_frag(lb, wn * (C::BN / C::WN) + j * 32 + b * 16 + (lane & 15), kc);
        }
#pragma unroll
        for (int ks = 0; ks < BK / 32; ++ks)
#pragma unroll
            for (int i = 0; i < C::TM; ++i)
#pragma unroll
                for (int j = 0; j < C::TN; ++j)
#pragma unroll
                    for (int a = 0; a < 2; ++a)
#pragma unroll
                        for (int b = 0; b < 2; ++b) {
                            float4_t c = {acc.v[i][j][(a * 2 + b) * 4], acc.v[i][j][(a * 2 + b) * 4 + 1], acc.v[i][j][(a * 2 + b) * 4 + 2],
                                          acc.v[i][j][(a * 2 + b) * 4 + 3]};
                            c = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[ks][j][b], af[ks][i][a], c, 0, 0, 0);
#pragma unroll
                            for (int r = 0; r < 4; ++r) acc.v[i][j][(a * 2 + b) * 4 + r] = c[r];
                        }
    }
}

template <class C, int NS, class Init>
__device__ __forceinline__ void mainloop_ring(const half_t* __restrict__ A, int lda, const half_t* __restrict__ B, int ldb, int M,
                                              int N, int nt, int m0, int n0, char* smem, Acc<C>& acc, const Init& init) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    mainloop_ring_g<C, NS>([&](int t, char* dst) { stage_tile<C::BM, C::NWAVES>(A, lda, m0, M, t * BK, dst, wave, lane); }, B, ldb, N, nt,
                           n0, smem, acc, init);
}

// ---- fp16 output through an LDS-staged, fully coalesced epilogue -----------------------------------------
// The BM x BN tile leaves in NH slabs of HR = 128 rows.  For slab h:
//  step 0 `slab(h)`: caller hook before the slab is staged;
//  step 1 (MFMA layout, waves owning rows of the slab): `pre(i, j, coff, v4, rl, g)` (rl = the lane's row inside the 32-row block i, g = the accumulator quad: compile-time after unrolling) turns four consecutive-column
//          accumulators (columns coff .. coff+3 of the wave's 32-column tile j) into fp16 values (activation); they are written as 8-byte units into a
//          [HR][BN] fp16 LDS image with unit' = unit ^ (row & 15)  (conflict-free ds_write_b64);
//  step 2 (row-major, all waves): `post(row_in_tile, chunk, pass, half8)` receives 8 consecutive columns of a
//          row (one ds_read_b128; the XOR may swap the two 8-byte halves) — a wave instruction covers whole
//          512-byte / 256-byte row segments: 16-byte fully coalesced global stores.
template <class C, bool M16 = false, class AccT, class Slab, class Pre, class Post>
__device__ __forceinline__ void epilogue_f16(const AccT& acc, char* stg, const Slab& slab, const Pre& pre, const Post& post) {
    // The lane-constant addressing of the epilogue is recomputed per tile from an OPAQUE copy of the thread id: hipcc otherwise hoists
    // it out of the persistent tile loop and keeps ~10 registers live across the K-loop — spilled around it in the residual kernels
    // (scratch reloads beside LDS-DMA drain the vector-memory counter, guide: "recompute per block").
    int tid = threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(tid));
#endif
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN, hi = lane >> 5;
    constexpr int RB = C::BN * 2;                                    // bytes per staged row
    constexpr int WROWS = C::BM / C::WM;                             // rows per wave
    constexpr int SWZ = C::BN / 4 >= 16 ? 15 : C::BN / 4 - 1;         // XOR mask of the 8-byte units (a row has BN / 4 of them)
#pragma unroll
    for (int h = 0; h < C::NH; ++h) {
        slab(h);           // caller hook: e.g. issue this slab's residual loads so they fly during the staging
        lds_barrier();     // slab buffer free: K-loop reads (h = 0) / previous slab's row-major reads are done
        if ((wm * WROWS) / C::HR == h) {
#pragma unroll
            for (int i = 0; i < C::TM; ++i) {
#pragma unroll
                for (int j = 0; j < C::TN; ++j)
#pragma unroll
                    for (int g = 0; g < 4; ++g) {
                        // (row, first of four columns) inside the wave's 32x32 tile (i, j) of accumulator elements 4g .. 4g+3
                        const int rl = M16 ? (g >> 1) * 16 + (lane & 15) : (lane & 31);
                        const int coff = M16 ? (g & 1) * 16 + 4 * (lane >> 4) : 8 * g + 4 * hi;
                        const int ml = (wm * WROWS) % C::HR + i * 32 + rl;
                        const int nl = wn * (C::BN / C::WN) + j * 32 + coff;
                        const float4_t v = acc_quad(acc, i, j, g);
                        const half4_t hv = pre(i, j, coff, v, rl, g);
                        const int unit = (nl >> 2) ^ (ml & SWZ);
                        *reinterpret_cast<half4_t*>(stg + ml * RB + unit * 8) = hv;
                    }
            }
        }
        lds_barrier();
        const int c = tid % C::CPR;
#pragma unroll
        for (int ps = 0; ps < C::NPASS; ++ps) {
            const int r = tid / C::CPR + ps * C::ROWS_PER_PASS;       // row inside the slab
            const int pair = c ^ ((r & SWZ) >> 1);
            half8_t hv = *reinterpret_cast<const half8_t*>(stg + r * RB + pair * 16);
            if (r & 1) hv = half8_t{hv[4], hv[5], hv[6], hv[7], hv[0], hv[1], hv[2], hv[3]};
            post(h * C::HR + r, c, h * C::NPASS + ps, hv);
        }
    }
}

// ---- the same LDS-staged epilogue as a four-slab pipeline (256-row tiles, 16x16x32 accumulator layout) --------------------------------
// tools/trace_tile.py: epilogue_f16 is a chain of phases — barrier, staging writes (half of the waves), barrier, row-major reads + stores (bound by
// the CU's one address path: 19 cycles per 1 KB store) — twice per tile, 15 % of a K = 768 tile with bias only and 30 % with QuickGELU (whose
// arithmetic is VALU-bound and ran on four waves at a time).  Here slab k (k = 0 .. 3) holds 32-row block k of EVERY wave (64 rows; every wave converts /
// activates 8 quads per slab), the 64 KB staging buffer is two 32 KB halves, and in interval k every wave first stages its share of slab k + 1 into the
// other half and then reads + stores slab k: the VALU / LDS-write work of one slab overlaps the stores of the previous one, five barriers per tile.
// `ahead(k)`: caller hook at the top of interval k - 1 (k = 0: before the first barrier) — e.g. request slab k's residual chunks one interval early.
// `post(row_in_tile, chunk, pass = 4 k + ps, half8)` as in epilogue_f16; 16 stores per thread and tile as there.
#ifndef PCLIP_EPI_PIPE_ALT
#define PCLIP_EPI_PIPE_ALT 1
#endif
template <class C, class Ahead, class Pre, class Post>
__device__ __forceinline__ void epilogue_pipe(const Acc<C>& acc, char* stg, const Ahead& ahead, const Pre& pre, const Post& post) {
    static_assert(C::TM == 4 && C::BM == 256 && C::BN % 64 == 0, "four 32-row blocks per wave");
    int tid = threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(tid));
#endif
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN;
    constexpr int RB = C::BN * 2, SLAB = 64 * RB;                    // bytes per staged row / per slab (BN = 256: 32 KB)
    constexpr int SWZ = 15;
    constexpr int SR = 256 / C::WM / 4 * C::WM;                       // rows of a slab = 32-row block k of each of the WM wave rows ... (WM = 2: 64)
    static_assert(SR == 32 * C::WM && 2 * SR * RB <= C::STAGE_BYTES, "two slabs must fit one stage buffer");
    constexpr int RPP = C::NTHREADS / C::CPR, NP4 = SR / RPP;          // rows per pass, passes per slab
    auto stage = [&](int k) {
        char* buf = stg + (k & 1) * SLAB;
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int rl = (g >> 1) * 16 + (lane & 15), coff = (g & 1) * 16 + 4 * (lane >> 4);
                const int ml = wm * 32 + rl, nl = wn * (C::BN / C::WN) + j * 32 + coff;
                float4_t v;
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    // acc.v[k] with a compile-time k after unrolling (the caller's loop over k is unrolled)
                    v[e] = acc.v[k][j][4 * g + e];
                }
                const half4_t hv = pre(k, j, coff, v, rl, g);
                const int unit = (nl >> 2) ^ (ml & SWZ);
                *reinterpret_cast<half4_t*>(buf + ml * RB + unit * 8) = hv;
            }
    };
    ahead(0);
    lds_barrier();                                                     // staging buffer free (the K-loop's reads of the last K-tile are done)
    stage(0);
#pragma unroll
    for (int k = 0; k < 4; ++k) {
        if (k + 1 < 4) ahead(k + 1);
        lds_barrier();                                                 // slab k staged; the other half's readers (slab k - 1) are done
        // half of the waves stage slab k + 1 first and store slab k afterwards, the other half the other way round: the LDS-write path and the
        // address path of the stores are busy at the same time instead of one after the other
        const bool stage_first = PCLIP_EPI_PIPE_ALT ? (wave & 1) == 0 : true;
        if (k + 1 < 4 && stage_first) stage(k + 1);
        const char* buf = stg + (k & 1) * SLAB;
        const int c = tid % C::CPR;
#pragma unroll
        for (int ps = 0; ps < NP4; ++ps) {
            const int r = tid / C::CPR + ps * RPP;                     // row inside the slab: wave-row block r >> 5, row r & 31 of its block k
            const int pair = c ^ ((r & SWZ) >> 1);
            half8_t hv = *reinterpret_cast<const half8_t*>(buf + r * RB + pair * 16);
            if (r & 1) hv = half8_t{hv[4], hv[5], hv[6], hv[7], hv[0], hv[1], hv[2], hv[3]};
            post((r >> 5) * (C::BM / C::WM) + k * 32 + (r & 31), c, k * NP4 + ps, hv);
        }
        if (k + 1 < 4 && !stage_first) stage(k + 1);
    }
}

}  // namespace pgemm
