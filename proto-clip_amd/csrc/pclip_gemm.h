// fp16 x fp16 -> fp32 MFMA GEMM core for gfx950:  C[M,N] = A[M,K] . B[N,K]^T  (both operands
// K-contiguous — the layout of every contraction on the Proto-CLIP path: queries x prototypes
// (utils.py:230-233) and activations x nn.Linear weights (clip/model.py:176-178)).
//
// Tile: 128 x 128 x 64 per 256-thread workgroup (4 waves in a 2x2 grid, 64x64 per wave as 2x2
// v_mfma_f32_32x32x16_f16 accumulators = 64 accumulator registers).  Staging: global_load_lds_dwordx4
// straight into a double-buffered 64 KiB LDS image (no VGPR round trip); because the LDS destination of
// that instruction is lane-linear, the bank-conflict swizzle is applied to the per-lane SOURCE address
// and undone on the ds_read_b128 side (guide §5.4 rule 21): LDS slot (row, s) holds global 16-byte chunk
// s ^ ((row >> 1) & 7) of that row, which makes every 16-lane group of a ds_read_b128 fragment read hit
// 16 distinct 16-byte slots of the 256-byte bank row (conflict-free).  One barrier per K-tile; the next
// tile's loads are in flight while the current one feeds the matrix cores; two workgroups per CU overlap
// each other's barrier stalls.  Workgroup ids are remapped so each XCD (private L2) walks a contiguous run
// of tiles.
//
// The MFMA operands are SWAPPED (D = Btile . Atile^T), so a lane owns ONE output row m and, per
// accumulator register quad, FOUR CONSECUTIVE output columns: epilogues get float4 / half4 vectors.
#pragma once
#include "pclip_common.h"

namespace pgemm {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;            // 16 KiB per operand tile
constexpr int LDS_BYTES = 4 * TILE_BYTES;          // A0 B0 A1 B1

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

__device__ __forceinline__ int swz_key(int row) { return (row >> 1) & 7; }

// One wave stages rows [wave*32, wave*32+32) of a 128-row tile: 4 x (8 rows x 128 B) glds pieces.
__device__ __forceinline__ void stage_tile(const half_t* __restrict__ g, int ld, int row0, int nrows, int k0,
                                           char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wave * 32 + i * 8 + (lane >> 3);           // tile row this lane fills
        const int c = (lane & 7) ^ swz_key(r);                   // source chunk for LDS slot (lane&7)
        int gr = row0 + r;
        gr = gr < nrows ? gr : nrows - 1;                        // clamp: out-of-range rows are never stored
        const half_t* src = g + (size_t)gr * ld + k0 + c * 8;
        char* dst = lds_tile + (wave * 32 + i * 8) * (BK * 2);   // wave-uniform base; HW adds lane*16
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
    }
}

__device__ __forceinline__ half8_t lds_frag(const char* lds_tile, int row, int kc) {
    return *reinterpret_cast<const half8_t*>(lds_tile + row * (BK * 2) + ((kc ^ swz_key(row)) << 4));
}

// XCD-aware, bijective remap of a linear workgroup id (guide §5.5 T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Accumulator layout after mainloop(): acc[i][j][e] is C[m][n] with
//   m = m0 + wr*64 + i*32 + (lane & 31)
//   n = n0 + wc*64 + j*32 + 8*(e >> 2) + 4*(lane >> 5) + (e & 3)
struct Acc {
    float16_t v[2][2];
};

// K-loop over a tile whose K-tile 0 has ALREADY been staged into buffer `p` (0/1) by stage_first().
// On return `p` names the FREE buffer (the one that held K-tile nt-2): a persistent caller stages the next
// output tile's K-tile 0 there before running its epilogue out of the other buffer.
__device__ __forceinline__ void stage_first(const half_t* __restrict__ A, int lda, const half_t* __restrict__ B, int ldb,
                                            int M, int N, int m0, int n0, char* smem, int p) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    char* a = smem + p * 2 * TILE_BYTES;
    stage_tile(A, lda, m0, M, 0, a, wave, lane);
    stage_tile(B, ldb, n0, N, 0, a + TILE_BYTES, wave, lane);
}

__device__ __forceinline__ void mainloop(const half_t* __restrict__ A, int lda, const half_t* __restrict__ B, int ldb,
                                         int M, int N, int K, int m0, int n0, char* smem, Acc& acc, int& p) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int nt = K / BK;
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc.v[i][j][e] = 0.f;

    for (int t = 0; t < nt; ++t) {
        __syncthreads();   // drains this wave's glds (vmcnt(0)) and orders all waves: tile t is in LDS
        const char* la = smem + p * 2 * TILE_BYTES;
        const char* lb = la + TILE_BYTES;
        if (t + 1 < nt) {
            char* na = smem + (p ^ 1) * 2 * TILE_BYTES;
            stage_tile(A, lda, m0, M, (t + 1) * BK, na, wave, lane);
            stage_tile(B, ldb, n0, N, (t + 1) * BK, na + TILE_BYTES, wave, lane);
        }
#pragma unroll
        for (int ks = 0; ks < BK / 16; ++ks) {
            const int kc = ks * 2 + (lane >> 5);
            half8_t af[2], bf[2];
#pragma unroll
            for (int i = 0; i < 2; ++i) af[i] = lds_frag(la, wr * 64 + i * 32 + (lane & 31), kc);
#pragma unroll
            for (int j = 0; j < 2; ++j) bf[j] = lds_frag(lb, wc * 64 + j * 32 + (lane & 31), kc);
#pragma unroll
            for (int i = 0; i < 2; ++i)
#pragma unroll
                for (int j = 0; j < 2; ++j)
                    acc.v[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(bf[j], af[i], acc.v[i][j], 0, 0, 0);
        }
        p ^= 1;
    }
}

// ---- fp16 output through an LDS-staged, fully coalesced epilogue -----------------------------------------
// Step 1 (MFMA layout): `pre(j, g, v4)` turns four consecutive-column accumulators (columns
// n0 + wc*64 + j*32 + 8g + 4*(lane>>5) + 0..3) into the fp16-rounded
// values (bias / activation), written as 8-byte units into a [128][128] fp16 LDS image whose units are
// XOR-swizzled by 2*(row & 15) (conflict-free ds_write_b64, pairs of units stay adjacent).
// Step 2 (row-major): each thread owns 8 consecutive columns of a row: one ds_read_b128, optional 16-byte
// residual load, one 16-byte global store — a wave instruction covers 4 rows x 256 contiguous bytes.
template <class Pre>
__device__ __forceinline__ void stage_out_f16(const Acc& acc, char* smem /* 32 KiB region */, const Pre& pre) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    const int wr = wave >> 1, wc = wave & 1, hi = lane >> 5;
    __syncthreads();   // every wave is done reading the last K-tile
#pragma unroll
    for (int i = 0; i < 2; ++i) {
        const int ml = wr * 64 + i * 32 + (lane & 31);
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int g = 0; g < 4; ++g) {
                const int nl = wc * 64 + j * 32 + 8 * g + 4 * hi;
                float4_t v = {acc.v[i][j][4 * g], acc.v[i][j][4 * g + 1], acc.v[i][j][4 * g + 2], acc.v[i][j][4 * g + 3]};
                const half4_t h = pre(j, g, v);
                const int unit = (nl >> 2) ^ ((ml & 15) << 1);
                *reinterpret_cast<half4_t*>(smem + ml * 256 + unit * 8) = h;
            }
    }
    __syncthreads();
}

__device__ __forceinline__ half8_t staged_row_chunk(const char* smem, int r, int c) {
    const int unit = (2 * c) ^ ((r & 15) << 1);
    return *reinterpret_cast<const half8_t*>(smem + r * 256 + unit * 8);
}

}  // namespace pgemm
