// fp16 x fp16 -> fp32 MFMA GEMM core for gfx950:  C[M,N] = A[M,K] . B[N,K]^T  (both operands
// K-contiguous — the layout of every contraction on the Proto-CLIP path: queries x prototypes
// (utils.py:230-233) and activations x nn.Linear weights (clip/model.py:176-178)).
//
// Tile: 128 x 128 x 64 per 256-thread workgroup (4 waves in a 2x2 grid, 64x64 per wave as 2x2
// v_mfma_f32_32x32x16_f16 accumulators = 64 AGPR/VGPRs).  Staging: global_load_lds_dwordx4 straight
// into a double-buffered 64 KiB LDS image (no VGPR round trip); because the LDS destination of that
// instruction is lane-linear, the bank-conflict swizzle is applied to the per-lane SOURCE address and
// undone on the ds_read_b128 side (guide §5.4 rule 21): LDS slot (row, c) holds global 16-byte chunk
// c ^ (row & 7) of that row.  One barrier per K-tile; the next tile's loads are in flight while the
// current one feeds the matrix cores; two workgroups per CU overlap each other's barrier stalls.
// Workgroup ids are remapped so that each XCD (private L2) walks a contiguous run of tiles.
#pragma once
#include "pclip_common.h"

namespace pgemm {

constexpr int BM = 128, BN = 128, BK = 64;
constexpr int TILE_BYTES = BM * BK * 2;            // 16 KiB per operand tile
constexpr int LDS_BYTES = 4 * TILE_BYTES;          // A0 B0 A1 B1

typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

// One wave stages rows [wave*32, wave*32+32) of a 128-row tile: 4 x (8 rows x 128 B) glds pieces.
__device__ __forceinline__ void stage_tile(const half_t* __restrict__ g, int ld, int row0, int nrows, int k0,
                                           char* lds_tile, int wave, int lane) {
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int r = wave * 32 + i * 8 + (lane >> 3);           // tile row this lane fills
        const int c = (lane & 7) ^ (r & 7);                      // source chunk for LDS slot (lane&7)
        int gr = row0 + r;
        gr = gr < nrows ? gr : nrows - 1;                        // clamp: out-of-range rows are never stored
        const half_t* src = g + (size_t)gr * ld + k0 + c * 8;
        char* dst = lds_tile + (wave * 32 + i * 8) * (BK * 2);   // wave-uniform base; HW adds lane*16
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)src, (lds_ptr_t)dst, 16, 0, 0);
    }
}

__device__ __forceinline__ half8_t lds_frag(const char* lds_tile, int row, int kc) {
    return *reinterpret_cast<const half8_t*>(lds_tile + row * (BK * 2) + ((kc ^ (row & 7)) << 4));
}

// XCD-aware, bijective remap of a linear workgroup id (guide §5.5 T1).
__device__ __forceinline__ int xcd_remap(int bid, int nwg) {
    const int q = nwg >> 3, r = nwg & 7, xcd = bid & 7, idx = bid >> 3;
    const int base = xcd < r ? xcd * (q + 1) : r * (q + 1) + (xcd - r) * q;
    return base + idx;
}

// Epi: struct with  __device__ void operator()(int row, int col, float acc) const
// Computes the tile (tile_m, tile_n); Epi is invoked for in-range elements only.
template <class Epi>
__device__ __forceinline__ void gemm_tile(const half_t* __restrict__ A, int lda, const half_t* __restrict__ B, int ldb,
                                          int M, int N, int K, int tile_m, int tile_n, char* smem, const Epi& epi) {
    const int tid = threadIdx.x, lane = tid & 63, wave = tid >> 6;
    const int wr = wave >> 1, wc = wave & 1;
    const int m0 = tile_m * BM, n0 = tile_n * BN;
    const int nt = K / BK;

    float16_t acc[2][2];
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j)
#pragma unroll
            for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;

    stage_tile(A, lda, m0, M, 0, smem, wave, lane);
    stage_tile(B, ldb, n0, N, 0, smem + TILE_BYTES, wave, lane);

    for (int t = 0; t < nt; ++t) {
        __syncthreads();   // drains this wave's glds (vmcnt(0)) and orders all waves: tile t is in LDS
        const char* la = smem + (t & 1) * 2 * TILE_BYTES;
        const char* lb = la + TILE_BYTES;
        if (t + 1 < nt) {
            char* na = smem + ((t + 1) & 1) * 2 * TILE_BYTES;
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
                    acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(af[i], bf[j], acc[i][j], 0, 0, 0);
        }
    }

    // C/D layout of 32x32 MFMA: col = lane & 31, row = (reg & 3) + 8 * (reg >> 2) + 4 * (lane >> 5)
#pragma unroll
    for (int i = 0; i < 2; ++i)
#pragma unroll
        for (int j = 0; j < 2; ++j) {
            const int col = n0 + wc * 64 + j * 32 + (lane & 31);
#pragma unroll
            for (int e = 0; e < 16; ++e) {
                const int row = m0 + wr * 64 + i * 32 + (e & 3) + 8 * (e >> 2) + 4 * (lane >> 5);
                if (row < M && col < N) epi(row, col, acc[i][j][e]);
            }
        }
}

}  // namespace pgemm
