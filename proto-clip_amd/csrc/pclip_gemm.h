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

#ifndef PCLIP_ABL
#define PCLIP_ABL 0          // compile-time ablation builds: 1 no LDS-DMA in the K-loop, 2 no MFMAs, 4 no epilogue
#endif
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

// Workgroup barrier that orders LDS traffic only: unlike __syncthreads() it does not drain the vector-memory
// counter, so global stores and LDS-DMA prefetches stay in flight across it (guide §5 "Pipelining across
// barriers").  Every LDS read/write issued before it has completed (lgkmcnt(0)) when the wave arrives.
__device__ __forceinline__ void lds_barrier() {
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
}

// All waves of the workgroup stage a ROWS x 64 tile: each glds piece is 8 rows x 128 B (64 lanes x 16 B).
template <int ROWS, int NWAVES>
__device__ __forceinline__ void stage_tile(const half_t* __restrict__ g, int ld, int row0, int nrows, int k0,
                                           char* lds_tile, int wave, int lane) {
    constexpr int RPW = ROWS / NWAVES;                               // rows per wave
#pragma unroll
    for (int i = 0; i < RPW / 8; ++i) {
        const int r = wave * RPW + i * 8 + (lane >> 3);              // tile row this lane fills
        const int c = (lane & 7) ^ swz_key(r);                       // source chunk for LDS slot (lane&7)
        int gr = row0 + r;
        gr = gr < nrows ? gr : nrows - 1;                            // clamp: out-of-range rows are never stored
        const half_t* src = g + (size_t)gr * ld + k0 + c * 8;
        char* dst = lds_tile + (wave * RPW + i * 8) * ROW_BYTES;     // wave-uniform base; HW adds lane*16
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
    }
}

__device__ __forceinline__ half8_t lds_frag(const char* lds_tile, int row, int kc) {
    return *reinterpret_cast<const half8_t*>(lds_tile + row * ROW_BYTES + ((kc ^ swz_key(row)) << 4));
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
#if defined(__HIP_DEVICE_COMPILE__)
typedef __amdgpu_buffer_rsrc_t rsrc_t;
#else
struct rsrc_t { int w[4]; };                     // the host pass only parses the kernels; the descriptor type is device-only
#endif

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
    int pf_voff;                  // L2 prefetch (below): byte offset of this lane's row in the operand wave `wave` touches
    bool pf_a;
    __device__ __forceinline__ void prepare(const half_t* __restrict__ A, int lda, const half_t* __restrict__ B, int ldb, int M, int N,
                                            int m0, int n0, int wave, int lane) {
        a.prepare(A, lda, m0, M, wave, lane);
        b.prepare(B, ldb, n0, N, wave, lane);
        // wave u touches rows 64u .. 64u+63 of A, the waves after the A rows touch B's rows (leftover waves re-touch A)
        constexpr int UA = C::BM / 64;
        const int ub = wave - UA;
        pf_a = !(ub >= 0 && ub * 64 < C::BN);
        const int r = (pf_a ? (wave < UA ? wave : 0) : ub) * 64 + lane;
        const int last = (pf_a ? M - 1 - m0 : N - 1 - n0);
        pf_voff = (r < last ? r : last) * (pf_a ? lda : ldb) * 2;
    }
    // One 4-byte LDS-DMA per lane = one 128-byte line of K-tile t per row, dropped into a 256-byte LDS scrap area: its only purpose
    // is to pull the lines of a LATER K-tile from HBM into L2 while the current one is multiplied, so that the real LDS-DMA of
    // that K-tile (issued one K-tile ahead, all the LDS there is room for) finds them in L2.  Ablation (tools/ablate_gemm.py):
    // the K-loop without MFMAs takes 80 % of the full kernel's time — every 64 KB batch contains activation rows nobody has
    // touched before, and its slowest line is an HBM round trip.
    __device__ __forceinline__ void prefetch(int t, char* scrap, int wave) const {
#if defined(__HIP_DEVICE_COMPILE__)
        if (pf_a) __builtin_amdgcn_raw_ptr_buffer_load_lds(a.rs, (lds_ptr_t)scrap, 4, pf_voff, t * (BK * 2), 0, 0);
        else __builtin_amdgcn_raw_ptr_buffer_load_lds(b.rs, (lds_ptr_t)scrap, 4, pf_voff, t * (BK * 2), 0, 0);
#endif
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
// PERM (column-permuted B tile, for pgemm::epilogue_lane): the B fragment of MFMA block blk = 2 j + b takes, for MFMA column c = lane & 15, the tile row
// wn * 64 + 16 (c >> 2) + 4 blk + (c & 3) instead of wn * 64 + 16 blk + c — a lane's accumulator quads of the four blocks then are 16 CONSECUTIVE output
// columns.  Those 16 rows of a fragment read differ in row bits 0, 1, 4, 5, so the B image's swizzle key becomes ((row >> 1) & 1) | (((row >> 4) & 3) << 1)
// (conflict-free under ds_read_b128's lane groups like the default key); on the staging side it depends on the 16-row group of a piece: four offsets.
template <class C, bool PERM = false>
struct TilePairR {
    static constexpr bool ROLES = true;
    static_assert(C::NWAVES == 8, "role split: waves w and w + 4 share a SIMD");
    static constexpr int NA = C::BM / 32, NB = C::BN / 32;             // pieces per A wave / per B wave and K-tile
    static_assert(C::BM % 64 == 0 && C::BN % 64 == 0, "a wave stages ROWS / 4 rows, a multiple of 16");
    // Per-lane state: TWO byte offsets.  Piece i of a wave covers rows w4 * ROWS/4 + 8 i + (lane >> 3); the swizzle key (row >> 1) & 7
    // only depends on the parity of i, so piece i = piece (i & 1) + (i >> 1) * 16 rows, and those 16 rows travel in the instruction's
    // SCALAR offset together with the K-tile.  Rows beyond the operand are not clamped but cut off by the descriptor's size (a buffer
    // load past num_records returns zeros; such rows are never stored).
    static_assert(!PERM || C::BN == 256, "permuted B tile: four 16-row groups per staging wave");
    rsrc_t rs;
    int voff[PERM ? 4 : 2];
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
        if (PERM && is_b) {
#pragma unroll
            for (int v = 0; v < (PERM ? 4 : 0); ++v) {             // 16-row group v of the wave's 64 rows: rows 16 v + (lane >> 3) (+ 8 in the scalar offset)
                const int r = w4 * rpw + 16 * v + (lane >> 3);
                const int c = (lane & 7) ^ (((lane >> 4) & 1) | (v << 1));
                voff[v] = (r * ld + c * 8) * 2;
            }
            return;
        }
#pragma unroll
        for (int i = 0; i < 2; ++i) {
            const int r = w4 * rpw + i * 8 + (lane >> 3);
            const int c = (lane & 7) ^ swz_key(r);
            voff[i] = (r * ld + c * 8) * 2;
        }
    }
    // this wave's share of K-tile t into stage buffer `stage_buf` (A image at +0, B image at +A_BYTES)
    __device__ __forceinline__ void stage(int t, char* stage_buf, int wave) const {
#if defined(__HIP_DEVICE_COMPILE__)
        const int w4 = wave & 3, k = t * (BK * 2);
        if (is_b && PERM) {
#pragma unroll
            for (int i = 0; i < NB; ++i)                               // piece i = rows 16 (i >> 1) + 8 (i & 1) + (lane >> 3) of the wave's share
                __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(stage_buf + C::A_BYTES + (w4 * (C::BN / 4) + i * 8) * ROW_BYTES), 16, voff[PERM ? i >> 1 : 0],
                                                         k + (i & 1) * (row16 >> 1), 0, 0);
        } else if (is_b) {
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

// K-loop over a tile whose K-tile 0 is already on its way into buffer `p` (TilePair::stage(0, ..)); semantics of `p`, YOUNGER,
// counted_first as mainloop_g.  `wave` must be wave-uniform (readfirstlane).
// PF: every iteration additionally issues TilePair::prefetch of K-tile t + 2 (clamped) into `scrap` right after the LDS-DMA of
// K-tile t + 1; the wait at the top of the next iteration then leaves that one youngest operation in flight (vmcnt(1)).
template <class C, int YOUNGER = 0, bool ZERO_ACC = true, bool PF = true>
__device__ __forceinline__ void mainloop_bl(const TilePair<C>& tp, int nt, char* smem, Acc<C>& acc, int& p, bool counted_first, int wave,
                                            int lane, char* scrap = nullptr) {
    static_assert(C::TM >= 2 && C::TM % 2 == 0, "group pipeline splits the wave's A rows in two halves");
    constexpr int HM = C::TM / 2;                                    // 32-row blocks per half
    const int wm = wave / C::WN, wn = wave % C::WN;
    if (ZERO_ACC) {
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
            for (int j = 0; j < C::TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc.v[i][j][e] = 0.f;
    }
    // fragment addressing: row = 32-aligned base + 16*x + (lane & 15) -> swz_key(row) = (lane & 15) >> 1 for every fragment;
    // 16-byte chunk kc = 4*ks + (lane >> 4)  ->  byte offset in the row ((kc ^ key) << 4); ks = 1 is ks = 0 with bit 6 flipped
    const int key = (lane & 15) >> 1, q = lane >> 4;
    const int col0 = (q ^ key) << 4;
    const int offa = (wm * (C::BM / C::WM) + (lane & 15)) * ROW_BYTES + col0;
    const int offb = C::A_BYTES + (wn * (C::BN / C::WN) + (lane & 15)) * ROW_BYTES + col0;
    const bool late = C::NWAVES == 8 && wave >= 4;

    for (int t = 0; t < nt; ++t) {
        if (t == 0) { if (counted_first) wait_vm<YOUNGER>(); else wait_vm<0>(); }
        else if (PF) wait_vm<1>();
        else wait_vm<0>();
        lds_barrier();
        const char* base = smem + p * C::STAGE_BYTES;
        auto fa = [&](int ks, int i, int a) { return *reinterpret_cast<const half8_t*>(base + (offa ^ (ks << 6)) + (i * 32 + a * 16) * ROW_BYTES); };
        auto fb = [&](int ks, int j, int b) { return *reinterpret_cast<const half8_t*>(base + (offb ^ (ks << 6)) + (j * 32 + b * 16) * ROW_BYTES); };
#ifndef PCLIP_ABL
#define PCLIP_ABL 0          // compile-time ablation builds (tools/gpu_ablate.sh): 1 no LDS-DMA in the K-loop, 2 no MFMAs, 4 no epilogue
#endif
        auto stage_next = [&]() {
            if (t + 1 < nt && !(PCLIP_ABL & 1)) {
                tp.stage(t + 1, smem + (p ^ 1) * C::STAGE_BYTES, wave);
                if (PF) tp.prefetch(t + 2 < nt ? t + 2 : nt - 1, scrap, wave);
            }
        };
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
#if (PCLIP_ABL & 2) && defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int i = 0; i < HM; ++i) { asm volatile("" ::"v"(af[i][0]), "v"(af[i][1])); }
#pragma unroll
            for (int j = 0; j < C::TN; ++j) { asm volatile("" ::"v"(bf[j][0]), "v"(bf[j][1])); }
            return;
#endif
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
        load_b(bcur, 0);
        load_a(acur, 0, 0);
        if (!late) stage_next();
        load_a(anext, 0, 1);
        group(acur, bcur, 0);                    // ks 0, rows half 0   (covers anext)
        load_b(bnext, 1);
        load_a(acur, 1, 0);
        group(anext, bcur, 1);                   // ks 0, rows half 1   (covers bnext, acur)
        if (late) stage_next();
        load_a(anext, 1, 1);
        group(acur, bnext, 0);                   // ks 1, rows half 0   (covers anext)
        group(anext, bnext, 1);                  // ks 1, rows half 1
        p ^= 1;
    }
}

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
#ifndef PCLIP_TRACE
#define PCLIP_TRACE 0            // debug build: s_memtime stamps around every wait of the K-loop (tools/trace_gemm.py)
#endif
#ifndef PCLIP_IGLP
#define PCLIP_IGLP 1             // __builtin_amdgcn_iglp_opt strategy of the K-loop's first scheduling region (-1: none; 0 / 2 / 3 measured: no gain)
#endif
// TWO: K-tiles 0 AND 1 were requested by the caller (direct-store epilogue: no LDS staging area between tiles); `counted_first` then
// also covers the wait for K-tile 1, whose pieces sit in front of the previous tile's YOUNGER - NA|NB stores and strip copies.
// PERM: the B image is column-permuted (TilePairR<C, true>): accumulator element (a * 2 + b) * 4 + e of acc.v[i][j] is then output column
// wn * (BN / WN) + 16 (lane >> 4) + 4 (2 j + b) + e (row i * 32 + a * 16 + (lane & 15) as before).
// PFN / touch(k): in iteration nt - 5 every wave issues PFN extra vector-memory operations `touch(0 .. PFN - 1)` right behind its own pieces of that iteration
// (the residual epilogues pull their tile of the residual stream into L2 this way, ~3 iterations before the epilogue asks for it).  They are YOUNGER than the
// pieces the next two top-of-iteration waits are for, so those two waits leave PFN more operations outstanding — an in-order wait that did not would sit out
// the touches' whole HBM round trip in the middle of the K-loop.
struct NoTouch { __device__ __forceinline__ void operator()(int) const {} };
template <class C, int YOUNGER = 0, bool ZERO_ACC = true, class TP = TilePair<C>, bool TWO = false, bool PERM = false, int PFN = 0, class Touch = NoTouch>
__device__ __forceinline__ void mainloop_sr(const TP& tp, int nt, char* smem, Acc<C>& acc, int& p, bool counted_first, int wave,
                                            int lane, unsigned long long* g_tr = nullptr, const Touch& touch = Touch()) {
    static_assert(C::TM >= 2 && C::TM % 2 == 0, "group pipeline splits the wave's A rows in two halves");
    constexpr int HM = C::TM / 2;
    constexpr int NA = TP::NA, NB = TP::NB;
    // requests of ONE K-tile this wave leaves in flight across the top barrier: every wave NA + NB pieces (TilePair), or the pieces
    // of its own operand (TilePairR)
    auto wait_ahead = [&]() {
        if constexpr (TP::ROLES) { if (wave < 4) wait_vm<NB>(); else wait_vm<NA>(); }
        else wait_vm<NA + NB>();
    };
    auto wait_ahead_pf = [&]() {
        if constexpr (TP::ROLES) { if (wave < 4) wait_vm<NB + PFN>(); else wait_vm<NA + PFN>(); }
        else wait_vm<NA + NB + PFN>();
    };
    const int t_pf = (PFN > 0 && nt >= 6) ? nt - 5 : -1;                // iteration that issues the touches (it refills: t_pf + 2 < nt)
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
    const int cb = lane & 15;
    const int offb = PERM ? C::A_BYTES + (wn * (C::BN / C::WN) + 16 * (cb >> 2) + (cb & 3)) * ROW_BYTES + ((q ^ (((cb >> 1) & 1) | ((cb >> 2) << 1))) << 4)
                          : C::A_BYTES + (wn * (C::BN / C::WN) + (lane & 15)) * ROW_BYTES + col0;
    // (s_setprio measured on this loop, tools/ab_multi.py gemm, profiles/r03_ab_gemm_prio.txt: priority 1 around every MFMA group
    // is neutral at N = 3072 and 5 - 32 % SLOWER at N = 768 / 2304; a static priority for the younger half of the waves is +-0.5 %.)

#if PCLIP_TRACE && defined(__HIP_DEVICE_COMPILE__)
#define PCLIP_STAMP(var) unsigned long long var; asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(var)::"memory")
    PCLIP_STAMP(tr_begin);
#else
#define PCLIP_STAMP(var)
#endif
    for (int t = 0; t < nt; ++t) {
        PCLIP_STAMP(tr0);
        if (t == 0) { if (counted_first) wait_vm<YOUNGER>(); else wait_vm<0>(); }
        else if (TWO && t == 1 && counted_first && nt > 2) wait_vm<YOUNGER>();   // behind K-tile 1: the stores / strips and K-tile 2's pieces = as many as behind K-tile 0
        else if (PFN > 0 && (t == t_pf + 1 || t == t_pf + 2) && t_pf >= 0) wait_ahead_pf();
        else if (t + 1 < nt) wait_ahead();
        else wait_vm<0>();
        PCLIP_STAMP(tr1);
        lds_barrier();
        PCLIP_STAMP(tr2);
#if PCLIP_TRACE && defined(__HIP_DEVICE_COMPILE__)
        g_tr[0] += tr1 - tr0; g_tr[1] += tr2 - tr1;
#endif
        char* cur = smem + p * C::STAGE_BYTES;
        const char* base = cur;
        auto fa = [&](int ks, int i, int a) { return *reinterpret_cast<const half8_t*>(base + (offa ^ (ks << 6)) + (i * 32 + a * 16) * ROW_BYTES); };
        auto fb = [&](int ks, int j, int b) { return *reinterpret_cast<const half8_t*>(base + (offb ^ (ks << 6)) + (PERM ? (2 * j + b) * 4 : j * 32 + b * 16) * ROW_BYTES); };
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
#if (PCLIP_ABL & 2) && defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
            for (int i = 0; i < HM; ++i) { asm volatile("" ::"v"(af[i][0]), "v"(af[i][1])); }
#pragma unroll
            for (int j = 0; j < C::TN; ++j) { asm volatile("" ::"v"(bf[j][0]), "v"(bf[j][1])); }
            return;
#endif
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
        if (!TWO && t == 0 && nt > 1 && !(PCLIP_ABL & 1)) tp.stage(1, smem + (p ^ 1) * C::STAGE_BYTES, wave);   // the other buffer was the previous tile's epilogue staging area until the barrier above
        load_a(anext, 0, 1);
        group(acur, bcur, 0);                    // ks 0, rows half 0
        load_b(bnext, 1);
        load_a(acur, 1, 0);
        group(anext, bcur, 1);                   // ks 0, rows half 1
        if (refill) {
            PCLIP_STAMP(tr3);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PCLIP_STAMP(tr4);
            lds_barrier();                       // every wave holds its last B fragments of this K-tile: the B half of `cur` is free
            PCLIP_STAMP(tr5);
#if PCLIP_TRACE && defined(__HIP_DEVICE_COMPILE__)
            g_tr[2] += tr4 - tr3; g_tr[3] += tr5 - tr4;
#endif
            if constexpr (TP::ROLES) { if (wave < 4) tp.stage(t + 2, cur, wave); }
            else if (!(PCLIP_ABL & 1)) tp.b.template stage<0>((t + 2) * (BK * 2), cur + C::A_BYTES, wave);
            if constexpr (PFN > 0 && TP::ROLES) {
                if (t == t_pf && wave < 4) {
#pragma unroll
                    for (int k = 0; k < PFN; ++k) touch(k);
                }
            }
        }
        load_a(anext, 1, 1);
        group(acur, bnext, 0);                   // ks 1, rows half 0
        if (refill) {
            PCLIP_STAMP(tr6);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            PCLIP_STAMP(tr7);
            lds_barrier();                       // ... and its last A fragments: the A half is free
            PCLIP_STAMP(tr8);
#if PCLIP_TRACE && defined(__HIP_DEVICE_COMPILE__)
            g_tr[4] += tr7 - tr6; g_tr[5] += tr8 - tr7;
#endif
            if constexpr (TP::ROLES) { if (wave >= 4) tp.stage(t + 2, cur, wave); }
            else if (!(PCLIP_ABL & 1)) tp.a.template stage<PCLIP_NT_A>((t + 2) * (BK * 2), cur, wave);
            if constexpr (PFN > 0) {
                if (t == t_pf && (!TP::ROLES || wave >= 4)) {
#pragma unroll
                    for (int k = 0; k < PFN; ++k) touch(k);
                }
            }
        }
        group(anext, bnext, 1);                  // ks 1, rows half 1
        p ^= 1;
    }
#if PCLIP_TRACE && defined(__HIP_DEVICE_COMPILE__)
    { PCLIP_STAMP(tr_end); g_tr[6] += tr_end - tr_begin; g_tr[7] += nt; }
#endif
}

// (Measured and removed, profiles/r03_ab_gemm_sched.txt: the same loop software-pipelined ACROSS the K-tile boundary — the barrier that
// publishes K-tile t + 1 moved in front of the last MFMA group of iteration t and the first fragments of t + 1 requested under that
// group, so that an iteration starts issuing MFMAs at once.  Bit-identical and 4 - 6 % SLOWER: waiting for K-tile t + 1 a quarter of an
// iteration earlier exposes the operand delivery itself — the LDS-DMA round trip of ~1.25 iterations is what this loop waits for, not
// the fragment latency behind the first barrier.)

// ---- ping-pong K-loop: 32-wide K-steps in a ring of four 32 KB slots, the two waves of a SIMD in opposite phases ---------------
// mainloop_sr keeps all eight waves in lock-step: they meet at three barriers per K-tile, ask the LDS for their first fragments
// together and stall on DMA issue together — the ablation builds say MFMA + LDS + barriers alone (no DMA) already take 1.5 x the
// matrix pipe's time.  Here waves 0 - 3 (group 0, one per SIMD) and 4 - 7 (group 1) ALTERNATE between a memory phase M(s)
// (the wave's 12 fragment reads of K-step s, its 4 LDS-DMA pieces of K-step s + 3) and a compute phase C(s) (its 32 MFMAs of
// K-step s, operands in registers), one barrier per phase, group 1 one phase behind group 0:
//     phase 2s     group 0: M(s)      group 1: C(s-1)
//     phase 2s+1   group 0: C(s)      group 1: M(s)
// so the matrix pipe of every SIMD always has exactly one wave feeding it while its partner talks to the LDS and the DMA queue.
// K-step s lives in slot (slot0 + s) & 3; it is read in phases 2s and 2s+1, so the slot is free behind the barrier that ends
// phase 2s+1 and is refilled with K-step s + 4 in phases 2s+2 (group 0: the B rows) and 2s+3 (group 1: the A rows) — five to six
// phases (>= 2.5 K-steps of MFMA time) ahead of its first use, with three K-steps (96 KB) in flight per CU (tools/probe/dma_probe:
// rings of three or more 32 KB pieces deliver 25 % faster than the two 64 KB buffers of mainloop_sr).
// A wave can only wait for its OWN pieces: K-step s is published by the barrier that ends phase 2s-1, in front of which every
// wave waits until at most the pieces it issued later are outstanding (counted vmcnt; the previous tile's epilogue stores and the
// strip copies — YOUNGER of them when `counted_first` — sit between K-steps 1 and 2 in issue order).
// LDS image of a K-step: [A: 256 rows x 64 B | B: 256 rows x 64 B]; the 16-byte slot c' of row r holds global chunk
// c' ^ g[(r >> 2) & 3], g = {0, 3, 2, 1}: under ds_read_b128's lane groups ({0-3, 12-15, 20-27}, {4-11, 16-19, 28-31}, ..: microarch
// guide) the 16 lanes of a group then hit 16 distinct 16-byte columns of the 256-byte bank row.  One LDS-DMA piece = 16 rows x 64 B
// (lane l: row l >> 2, slot l & 3), so the swizzle key of a lane is g[l >> 4] for every piece: ONE byte offset per lane, the piece and
// the K-step travel in the scalar offset.  Same fragments in the same k order per accumulator as mainloop_sr: bit-identical results.
template <class C>
struct TilePairP {
    static_assert(C::BM == 256 && C::BN == 256 && C::WM == 2 && C::WN == 4, "ping-pong loop: 256 x 256 tile, 2 x 4 waves");
    static constexpr int KS = 32, ROWB = KS * 2, A_IMG = C::BM * ROWB, SLOT = (C::BM + C::BN) * ROWB, NSLOT = 4;
    static constexpr int NP = 4;                                       // LDS-DMA pieces per wave and K-step (64 operand rows)
    static_assert(NSLOT * SLOT == C::LDS_BYTES, "the ring occupies the two stage buffers");
    rsrc_t rs;
    int voff, row16, lds_off;
    __device__ __forceinline__ void prepare(const half_t* __restrict__ A, int lda, const half_t* __restrict__ B, int ldb, int M, int N,
                                            int m0, int n0, int wave, int lane) {
        const bool is_b = wave < 4;                                    // group 0 stages the B rows, group 1 the A rows
        const half_t* g = is_b ? B : A;
        const int ld = is_b ? ldb : lda, row0 = is_b ? n0 : m0, nrows = is_b ? N : M, w4 = wave & 3;
#if defined(__HIP_DEVICE_COMPILE__)
        const uint64_t addr = (uint64_t)(g + (size_t)row0 * ld);
        const uint32_t lo = __builtin_amdgcn_readfirstlane((uint32_t)addr), hi = __builtin_amdgcn_readfirstlane((uint32_t)(addr >> 32));
        const long left = (long)(nrows - row0) * ld * 2;               // rows beyond the operand: cut off by the descriptor (zeros, never stored)
        const uint32_t size = __builtin_amdgcn_readfirstlane((uint32_t)(left < 0x7fffffffL ? left : 0x7fffffffL));
        rs = __builtin_amdgcn_make_buffer_rsrc((void*)(((uint64_t)hi << 32) | lo), 0, size, 0x00020000);
#endif
        row16 = 16 * ld * 2;
        voff = ((w4 * 64 + (lane >> 2)) * ld + (((lane & 3) ^ ((0 - (lane >> 4)) & 3)) << 3)) * 2;
        lds_off = (is_b ? A_IMG : 0) + w4 * 64 * ROWB;
    }
    // piece i (16 operand rows) of this wave's share of K-step s
    __device__ __forceinline__ void stage_piece(int s, int slot, char* smem, int i) const {
#if defined(__HIP_DEVICE_COMPILE__)
        if ((PCLIP_ABL & 1) && s >= 2) return;
        __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(smem + slot * SLOT + lds_off + i * 16 * ROWB), 16, voff, s * ROWB + i * row16, 0, 0);
#endif
    }
    // this wave's pieces of K-step s into ring slot `slot`
    __device__ __forceinline__ void stage(int s, int slot, char* smem) const {
#if defined(__HIP_DEVICE_COMPILE__)
        if ((PCLIP_ABL & 1) && s >= 2) return;
        char* dst = smem + slot * SLOT + lds_off;
#pragma unroll
        for (int i = 0; i < NP; ++i)
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (lds_ptr_t)(dst + i * 16 * ROWB), 16, voff, s * ROWB + i * row16, 0, 0);
#endif
    }
};

__device__ __forceinline__ void pp_barrier() {
#if defined(__HIP_DEVICE_COMPILE__)
    __builtin_amdgcn_sched_barrier(0);
    __builtin_amdgcn_s_barrier();
    __builtin_amdgcn_sched_barrier(0);
#endif
}

// `p` in: the slot pair (0 / 1) that holds K-steps 0 and 1 of this tile (already requested); out: the pair the NEXT tile's K-steps
// 0 and 1 were requested into by `next_tile(p)` — the epilogue stages out of the other pair.  S = K / 32 (even, >= 2).
template <class C, int YOUNGER = 0, bool ZERO_ACC = true, class Next>
__device__ __forceinline__ void mainloop_pp(TilePairP<C>& tp, int S, char* smem, Acc<C>& acc, int& p, bool counted_first, int wave,
                                            int lane, const Next& next_tile, unsigned long long* g_tr = nullptr) {
    using TP = TilePairP<C>;
    constexpr int NP = TP::NP;
    static_assert(YOUNGER + 2 * NP < 64, "vmcnt range");
    if (ZERO_ACC) {
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
            for (int j = 0; j < C::TN; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc.v[i][j][e] = 0.f;
    }
    const int r = lane & 15, q = lane >> 4;
    const int lane_part = r * TP::ROWB + ((q ^ ((0 - (r >> 2)) & 3)) << 4);
    const int offa = (wave >> 2) * (C::BM / C::WM) * TP::ROWB + lane_part;
    const int offb = TP::A_IMG + (wave & 3) * (C::BN / C::WN) * TP::ROWB + lane_part;
    const int slot0 = 2 * p;
    half8_t af[C::TM][2], bf[C::TN][2];
    auto read = [&](int sl) {
        const char* base = smem + sl * TP::SLOT;
#if (PCLIP_ABL & 8) && defined(__HIP_DEVICE_COMPILE__)
        if (sl >= 0) {       // timing ablation: fragments keep whatever they hold
#pragma unroll
            for (int i = 0; i < C::TM; ++i) { asm volatile("" : "+v"(af[i][0]), "+v"(af[i][1])); }
#pragma unroll
            for (int j = 0; j < C::TN; ++j) { asm volatile("" : "+v"(bf[j][0]), "+v"(bf[j][1])); }
            return;
        }
#endif
#pragma unroll
        for (int j = 0; j < C::TN; ++j)
#pragma unroll
            for (int b = 0; b < 2; ++b) bf[j][b] = *reinterpret_cast<const half8_t*>(base + offb + (j * 32 + b * 16) * TP::ROWB);
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
            for (int a = 0; a < 2; ++a) af[i][a] = *reinterpret_cast<const half8_t*>(base + offa + (i * 32 + a * 16) * TP::ROWB);
    };
    // the 32 MFMAs of a K-step; `between(i)` runs behind the MFMAs of 32-row block i (8 of them)
    auto mfmas_x = [&](auto&& between) {
#pragma unroll
        for (int i = 0; i < C::TM; ++i) {
#pragma unroll
            for (int j = 0; j < C::TN; ++j)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        float16_t& dst = acc.v[i][j];
                        float4_t c = {dst[(a * 2 + b) * 4], dst[(a * 2 + b) * 4 + 1], dst[(a * 2 + b) * 4 + 2], dst[(a * 2 + b) * 4 + 3]};
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j][b], af[i][a], c, 0, 0, 0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) dst[(a * 2 + b) * 4 + e] = c[e];
                    }
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
            between(i);
#if defined(__HIP_DEVICE_COMPILE__)
            __builtin_amdgcn_sched_barrier(0);
#endif
        }
    };
    auto mfmas = [&]() {
#if (PCLIP_ABL & 2) && defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int i = 0; i < C::TM; ++i) { asm volatile("" ::"v"(af[i][0]), "v"(af[i][1])); }
#pragma unroll
        for (int j = 0; j < C::TN; ++j) { asm volatile("" ::"v"(bf[j][0]), "v"(bf[j][1])); }
        return;
#endif
#pragma unroll
        for (int i = 0; i < C::TM; ++i)
#pragma unroll
            for (int j = 0; j < C::TN; ++j)
#pragma unroll
                for (int a = 0; a < 2; ++a)
#pragma unroll
                    for (int b = 0; b < 2; ++b) {
                        float16_t& dst = acc.v[i][j];
                        float4_t c = {dst[(a * 2 + b) * 4], dst[(a * 2 + b) * 4 + 1], dst[(a * 2 + b) * 4 + 2], dst[(a * 2 + b) * 4 + 3]};
                        c = __builtin_amdgcn_mfma_f32_16x16x32_f16(bf[j][b], af[i][a], c, 0, 0, 0);
#pragma unroll
                        for (int e = 0; e < 4; ++e) dst[(a * 2 + b) * 4 + e] = c[e];
                    }
    };
    // this wave's pieces of K-step s have landed: at most the pieces it issued after them (K-steps s+1 .. last) are still outstanding
    auto wait_pub = [&](int s, int last) {
        const int after = (last < S - 1 ? last : S - 1) - s;           // 2, 1 or 0 K-steps behind s
        if (s == 0 && !counted_first) { wait_vm<0>(); return; }
        if ((PCLIP_ABL & 16) && s > 0) { if (s == S - 1) wait_vm<0>(); return; }   // timing ablation: nobody waits for the K-steps in the loop
        if (s <= 1 && counted_first) {
            if (after == 2) wait_vm<YOUNGER + 2 * NP>(); else if (after == 1) wait_vm<YOUNGER + NP>(); else wait_vm<YOUNGER>();
        } else {
            if (after == 2) wait_vm<2 * NP>(); else if (after == 1) wait_vm<NP>(); else wait_vm<0>();
        }
    };
#ifndef PCLIP_PP_DMA_C
#define PCLIP_PP_DMA_C 0     // where a wave requests its pieces of K-step s + 3: 0 in its memory phase (fastest of the three), 1 behind its 32 MFMAs, 2 one piece behind every 8 MFMAs
#endif
    constexpr int DC = PCLIP_PP_DMA_C;
    wait_pub(0, 1);
    lds_barrier();                               // K-step 0 visible; the previous tile's epilogue is done with its slot pair
    if (wave < 4) {
        for (int s = 0; s < S; ++s) {
            const int sl = (slot0 + s) & 3;
            PCLIP_STAMP(tq0);
            read(sl);
            if (s == 0) {
                if (2 < S) tp.stage(2, (sl + 2) & 3, smem);
                if (!DC && 3 < S) tp.stage(3, (sl + 3) & 3, smem);
            } else if (!DC && s + 3 < S)
                tp.stage(s + 3, (sl + 3) & 3, smem);
            PCLIP_STAMP(tq1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            pp_barrier();                        // end of phase 2s
            PCLIP_STAMP(tq2);
            if (DC == 2) { const bool go = s + 3 < S; mfmas_x([&](int i) { if (go) tp.stage_piece(s + 3, (sl + 3) & 3, smem, i); }); }
            else { mfmas(); if (DC && s + 3 < S) tp.stage(s + 3, (sl + 3) & 3, smem); }
            PCLIP_STAMP(tq3);
            if (s + 1 < S) wait_pub(s + 1, s + 3);
            PCLIP_STAMP(tq4);
            pp_barrier();                        // end of phase 2s+1: K-step s+1 published, slot of K-step s free
            PCLIP_STAMP(tq5);
#if PCLIP_TRACE && defined(__HIP_DEVICE_COMPILE__)
            g_tr[0] += tq1 - tq0; g_tr[1] += tq2 - tq1; g_tr[2] += tq3 - tq2; g_tr[3] += tq4 - tq3; g_tr[4] += tq5 - tq4; g_tr[6] += tq5 - tq0; g_tr[7] += 1;
#endif
        }
        p = ((slot0 + S) & 3) >> 1;
        next_tile(p);
        pp_barrier();                            // end of phase 2S (group 1's last compute phase)
    } else {
        if (2 < S) tp.stage(2, (slot0 + 2) & 3, smem);
        if (!DC && 3 < S) tp.stage(3, (slot0 + 3) & 3, smem);
        pp_barrier();                            // end of phase 0
        for (int s = 0; s < S; ++s) {
            const int sl = (slot0 + s) & 3;
            PCLIP_STAMP(tq0);
            read(sl);
            if (!DC && s >= 1 && s + 3 < S) tp.stage(s + 3, (sl + 3) & 3, smem);
            PCLIP_STAMP(tq1);
            asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
            if (s + 1 < S) wait_pub(s + 1, DC ? s + 2 : s + 3);
            PCLIP_STAMP(tq2);
            pp_barrier();                        // end of phase 2s+1
            PCLIP_STAMP(tq3);
            if (DC == 2) { const bool go = s + 3 < S; mfmas_x([&](int i) { if (go) tp.stage_piece(s + 3, (sl + 3) & 3, smem, i); }); }
            else { mfmas(); if (DC && s + 3 < S) tp.stage(s + 3, (sl + 3) & 3, smem); }
            PCLIP_STAMP(tq4);
            pp_barrier();                        // end of phase 2s+2
            PCLIP_STAMP(tq5);
#if PCLIP_TRACE && defined(__HIP_DEVICE_COMPILE__)
            g_tr[0] += tq1 - tq0; g_tr[1] += tq2 - tq1; g_tr[2] += tq3 - tq2; g_tr[3] += tq4 - tq3; g_tr[4] += tq5 - tq4; g_tr[6] += tq5 - tq0; g_tr[7] += 1;
#endif
        }
        p = ((slot0 + S) & 3) >> 1;
        next_tile(p);
    }
}

// ---- implicit-GEMM gather for a 3x3 / stride 1 / pad 1 convolution on NHWC fp16 activations ------------------------------
// Row m of the GEMM is output pixel (b, y, x); K-tile t covers tap = (t*64) / Cin and input channels c0 = (t*64) % Cin
// (Cin % 64 == 0; Cin = 8 / 16 / 32 take the per-chunk path of stage()), i.e. the 128 contiguous bytes x[b, y+dy-1, x+dx-1, c0 .. c0+63] — or 128 zero bytes outside the image,
// fetched from a caller-provided zero line, because an LDS-DMA cannot be predicated per lane without leaving stale LDS.
// The im2col matrix (9x the activation) is never materialised.
template <class C>
struct ConvGather {
    static constexpr int NR = C::BM / C::NWAVES / 8;       // A rows per lane
    const half_t* __restrict__ x;
    const half_t* __restrict__ zero;
    int H, W, Cin, M;
    int pix[NR], yx[NR];
    __device__ __forceinline__ void prepare(int m0) {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            int m = m0 + wave * (C::BM / C::NWAVES) + i * 8 + (lane >> 3);
            m = m < M ? m : M - 1;                          // rows beyond M are never stored
            const int b = m / (H * W), rem = m - b * H * W, y = rem / W, xx = rem - y * W;
            pix[i] = m;                                     // (b*H + y)*W + x == m for stride 1 / pad 1
            yx[i] = (y << 16) | xx;
        }
    }
    __device__ __forceinline__ void stage(int t, char* dst) const {
        const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
        if (Cin < BK) {
            // Cin = 8 / 16 / 32 (the ResNet stem): a K-tile spans 64 / Cin taps, so the tap belongs to the 16-byte chunk, not to the
            // row; taps >= 9 (K padded to the K-tile, zero weights there) read the zero line
#pragma unroll
            for (int i = 0; i < NR; ++i) {
                const int r = wave * (C::BM / C::NWAVES) + i * 8 + (lane >> 3);
                const int c = (lane & 7) ^ swz_key(r);
                const int k = t * BK + c * 8, tap = k / Cin, ci = k - tap * Cin, dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
                const int y = (yx[i] >> 16) + dy, xx = (yx[i] & 0xffff) + dx;
                const bool in = tap < 9 && (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
                const half_t* src = in ? x + ((size_t)(pix[i] + dy * W + dx) * Cin + ci) : zero + c * 8;
                __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(dst + (wave * (C::BM / C::NWAVES) + i * 8) * ROW_BYTES), 16, 0, 0);
            }
            return;
        }
        const int k0 = t * BK, tap = k0 / Cin, c0 = k0 - tap * Cin, dy = tap / 3 - 1, dx = tap - (tap / 3) * 3 - 1;
#pragma unroll
        for (int i = 0; i < NR; ++i) {
            const int r = wave * (C::BM / C::NWAVES) + i * 8 + (lane >> 3);
            const int c = (lane & 7) ^ swz_key(r);
            const int y = (yx[i] >> 16) + dy, xx = (yx[i] & 0xffff) + dx;
            const bool in = (unsigned)y < (unsigned)H && (unsigned)xx < (unsigned)W;
            const half_t* src = in ? x + ((size_t)(pix[i] + dy * W + dx) * Cin + c0 + c * 8) : zero + c * 8;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)(dst + (wave * (C::BM / C::NWAVES) + i * 8) * ROW_BYTES), 16, 0, 0);
        }
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
template <class C, bool M16 = false, class Slab, class Pre, class Post>
__device__ __forceinline__ void epilogue_f16(const Acc<C>& acc, char* stg, const Slab& slab, const Pre& pre, const Post& post, unsigned long long* e_tr = nullptr) {
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
        PCLIP_STAMP(te0);
        slab(h);           // caller hook: e.g. issue this slab's residual loads so they fly during the staging
        lds_barrier();     // slab buffer free: K-loop reads (h = 0) / previous slab's row-major reads are done
        PCLIP_STAMP(te1);
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
                        float4_t v = {acc.v[i][j][4 * g], acc.v[i][j][4 * g + 1], acc.v[i][j][4 * g + 2], acc.v[i][j][4 * g + 3]};
                        const half4_t hv = pre(i, j, coff, v, rl, g);
                        const int unit = (nl >> 2) ^ (ml & SWZ);
                        *reinterpret_cast<half4_t*>(stg + ml * RB + unit * 8) = hv;
                    }
            }
        }
        PCLIP_STAMP(te2);
        lds_barrier();
        PCLIP_STAMP(te3);
        const int c = tid % C::CPR;
#pragma unroll
        for (int ps = 0; ps < C::NPASS; ++ps) {
            const int r = tid / C::CPR + ps * C::ROWS_PER_PASS;       // row inside the slab
            const int pair = c ^ ((r & SWZ) >> 1);
            half8_t hv = *reinterpret_cast<const half8_t*>(stg + r * RB + pair * 16);
            if (r & 1) hv = half8_t{hv[4], hv[5], hv[6], hv[7], hv[0], hv[1], hv[2], hv[3]};
            post(h * C::HR + r, c, h * C::NPASS + ps, hv);
        }
        PCLIP_STAMP(te4);
#if PCLIP_TRACE && defined(__HIP_DEVICE_COMPILE__)
        if (e_tr) { e_tr[0] += te1 - te0; e_tr[1] += te2 - te1; e_tr[2] += te3 - te2; e_tr[3] += te4 - te3; }
#endif
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

// ---- fp16 output straight from the accumulator layout (no LDS staging, no barrier) -----------------------------------------------
// A lane owns, per accumulator quad, four consecutive columns of one row: one 8-byte store.  A wave instruction then covers 16 rows x
// 32 bytes — a quarter of the coalescing of the LDS-staged pass, but the epilogue needs no LDS (the next tile's K-tiles 0 AND 1 fly
// during it) and no barrier.  `pre` as in epilogue_f16; `row_ok(m)` predicates the rows of a partial tile.  TM * TN * 4 stores per wave.
template <class C, class Pre, class RowOk>
__device__ __forceinline__ void epilogue_direct(const Acc<C>& acc, half_t* __restrict__ Cout, int ldc, int m0, int n0, const Pre& pre, const RowOk& row_ok) {
    int tid = threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(tid));
#endif
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN;
    half_t* base = Cout + (size_t)(m0 + wm * (C::BM / C::WM) + (lane & 15)) * ldc + n0 + wn * (C::BN / C::WN) + 4 * (lane >> 4);
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            const int rl = (g >> 1) * 16 + (lane & 15);
            const bool ok = row_ok(m0 + wm * (C::BM / C::WM) + i * 32 + rl);
#pragma unroll
            for (int j = 0; j < C::TN; ++j) {
                const int coff = (g & 1) * 16 + 4 * (lane >> 4);
                const float4_t v = {acc.v[i][j][4 * g], acc.v[i][j][4 * g + 1], acc.v[i][j][4 * g + 2], acc.v[i][j][4 * g + 3]};
                const half4_t hv = pre(i, j, coff, v, rl, g);
                if (ok) *reinterpret_cast<half4_t*>(base + (size_t)(i * 32 + (g >> 1) * 16) * ldc + j * 32 + (g & 1) * 16) = hv;
            }
        }
}

// ---- fp16 output in 16-byte pieces straight from the (column-permuted) accumulator layout -------------------------------------------
// With the PERM layout a lane holds, for every 16-row half (i, a) of its rows, the 16 consecutive columns 16 q .. 16 q + 15 of row a * 16 + r
// (r = lane & 15, q = lane >> 4): two 16-byte pieces P0 | P1.  Lanes r and r ^ 8 exchange one piece each (DPP row_ror:8), after which lane r < 8 holds
// P0 of rows r and r + 8 and lane r >= 8 holds P1 of rows r - 8 and r: a wave store instruction then covers 8 rows x 128 contiguous bytes — the same
// 1 KB / 8 lines per instruction as the LDS-staged pass, with no LDS, no barrier, and every wave on its own.  TM * 2 * 2 = 16 stores per wave and tile.
// `fin(piece, row_in_tile, col_in_tile, k)`: last touch of a 16-byte piece before its store (residual add), k = 2 (2 i + a) + {0, 1} its index.
typedef unsigned uint4v_t __attribute__((ext_vector_type(4)));
#ifndef PCLIP_LANE_BPERM
#define PCLIP_LANE_BPERM 1
#endif
template <class C, class Pre, class Fin, class RowOk>
__device__ __forceinline__ void epilogue_lane(const Acc<C>& acc, half_t* __restrict__ Cout, int ldc, int m0, int n0, const Pre& pre, const Fin& fin, const RowOk& row_ok) {
    static_assert(C::TN == 2 && C::BN / C::WN == 64, "a wave owns 64 output columns");
    int tid = threadIdx.x;
#if defined(__HIP_DEVICE_COMPILE__)
    asm volatile("" : "+v"(tid));
#endif
    const int lane = tid & 63, wave = tid >> 6;
    const int wm = wave / C::WN, wn = wave % C::WN, r = lane & 15, q = lane >> 4;
    const bool lo = r < 8;
    const int row_l = wm * (C::BM / C::WM) + (r & 7), col_l = wn * 64 + 16 * q + (lo ? 0 : 8);      // piece X0 of (i, a): row row_l + i * 32 + a * 16, X1: + 8
    const int src_lane4 = (16 * ((lane >> 1) & 3) + (lane >> 3) + 8 * (lane & 1)) * 4;              // byte address of the ds_bpermute source lane (see below)
#pragma unroll
    for (int i = 0; i < C::TM; ++i)
#pragma unroll
        for (int a = 0; a < 2; ++a) {
            unsigned pk[8];                                                      // the lane's 16 halves of row i * 32 + a * 16 + r: P0 = pk[0..3], P1 = pk[4..7]
#pragma unroll
            for (int blk = 0; blk < 4; ++blk) {
                const int j = blk >> 1, b = blk & 1, g = a * 2 + b;
                const float4_t v = {acc.v[i][j][4 * g], acc.v[i][j][4 * g + 1], acc.v[i][j][4 * g + 2], acc.v[i][j][4 * g + 3]};
                const half4_t hv = pre(i, j, 16 * q + 4 * blk, v, a * 16 + r, g);
                const half2_t h01 = {hv[0], hv[1]}, h23 = {hv[2], hv[3]};
                pk[2 * blk] = __builtin_bit_cast(unsigned, h01);
                pk[2 * blk + 1] = __builtin_bit_cast(unsigned, h23);
            }
            uint4v_t x0, x1;
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                const unsigned send = lo ? pk[4 + d] : pk[d];
#if defined(__HIP_DEVICE_COMPILE__)
                const unsigned recv = (unsigned)__builtin_amdgcn_update_dpp((int)send, (int)send, 0x128, 0xF, 0xF, false);   // row_ror:8 = lane r ^ 8 of the 16-lane row
#else
                const unsigned recv = send;
#endif
                x0[d] = lo ? pk[d] : recv;
                x1[d] = lo ? recv : pk[4 + d];
            }
#if PCLIP_LANE_BPERM && defined(__HIP_DEVICE_COMPILE__)
            // The address path coalesces CONSECUTIVE lanes: with lane = (q, r) consecutive lanes are consecutive ROWS and a store instruction touches 64 lines a
            // quad at a time (measured: 95 cycles per store instead of 19).  One ds_bpermute per dword (the LDS crossbar: no LDS memory, no barrier) brings the
            // pieces into store order: lane l' takes row l' >> 3, 16-byte chunk l' & 7 of the wave's 128-byte row segment = the piece of lane (r = (l' >> 3) + 8 (l' & 1), q = (l' >> 1) & 3).
#pragma unroll
            for (int d = 0; d < 4; ++d) {
                x0[d] = (unsigned)__builtin_amdgcn_ds_bpermute(src_lane4, (int)x0[d]);
                x1[d] = (unsigned)__builtin_amdgcn_ds_bpermute(src_lane4, (int)x1[d]);
            }
            const int row0 = wm * (C::BM / C::WM) + i * 32 + a * 16 + (lane >> 3), col_s = wn * 64 + 8 * (lane & 7);
            x0 = fin(x0, row0, col_s, 2 * (2 * i + a));
            x1 = fin(x1, row0 + 8, col_s, 2 * (2 * i + a) + 1);
            if (row_ok(m0 + row0)) *reinterpret_cast<uint4v_t*>(Cout + (size_t)(m0 + row0) * ldc + n0 + col_s) = x0;
            if (row_ok(m0 + row0 + 8)) *reinterpret_cast<uint4v_t*>(Cout + (size_t)(m0 + row0 + 8) * ldc + n0 + col_s) = x1;
#else
            const int row0 = row_l + i * 32 + a * 16;
            x0 = fin(x0, row0, col_l, 2 * (2 * i + a));
            x1 = fin(x1, row0 + 8, col_l, 2 * (2 * i + a) + 1);
            if (row_ok(m0 + row0)) *reinterpret_cast<uint4v_t*>(Cout + (size_t)(m0 + row0) * ldc + n0 + col_l) = x0;
            if (row_ok(m0 + row0 + 8)) *reinterpret_cast<uint4v_t*>(Cout + (size_t)(m0 + row0 + 8) * ldc + n0 + col_l) = x1;
#endif
        }
}

}  // namespace pgemm
