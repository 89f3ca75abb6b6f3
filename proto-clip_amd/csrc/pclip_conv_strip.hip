// 3x3 convolution (stride 1, pad 1, NHWC fp16) + eval BatchNorm (+ ReLU) for the NARROW layers of the ModifiedResNet image tower (clip/model.py:20-22, 45-46, 100-108 of
// the reference: the stem's conv2 / conv3 at 112 x 112 and layer1's conv2 at 56 x 56; Cin, Cout in {32, 64}) — the launches where the implicit-GEMM kernel of
// pclip_linear.hip is slowest (profiles/r06_rn50_profile.md: 0.10 - 0.14 MFMA busy, every input pixel gathered NINE times from L2, 44 % of its waves parked).
//
// Design (MI355X: 160 KB of LDS and 512 registers per lane at one wave per SIMD):
//  * a workgroup (4 waves) owns output tiles of 8 rows x 56 pixels and walks them persistently; a tile's input block WITH ITS HALO (10 x 58 pixels, all channels)
//    comes into LDS ONCE, by LDS-DMA, as contiguous row segments of the NHWC image (full 128-byte lines); two such blocks (2 x 76 KB) ping-pong, so the next tile's
//    block lands under this tile's arithmetic.  Out-of-image pixels are buffer loads beyond the descriptor's range (zeros): the padding costs no branch in the arithmetic.
//  * the WEIGHTS LIVE IN REGISTERS: a wave computes 32 output channels, i.e. its share of w is 32 x 9 Cin halves = 144 registers per lane at Cin = 64, loaded once
//    per workgroup.  Nothing but pixels is ever read from LDS.
//  * pixels are the MFMA's N side (v_mfma_f32_16x16x32_f16 with A = weights, B = pixels): a lane ends up with 8 consecutive output channels of one pixel — one
//    16-byte store per (pixel, lane) after BatchNorm / ReLU, no staging through LDS.
//  * a 16-pixel block is 8 consecutive pixels of row j and of row j + 4: the fragment a tap (dy, dx) needs for block j is the fragment of block j + dy at column
//    shift dx, so the six fragments of an (8 pixels x 8 rows) sub-tile at one dx serve 4 blocks x 3 dy x 2 channel groups = 24 MFMAs: 0.26 LDS reads per MFMA (the
//    four-wave GEMM's ratio; a tap-by-tap loop would need 0.5 and saturate LDS).
//  * LDS layout: pixel-major, the 16-byte channel slots of a pixel XOR-swizzled by (column, row) bits so that the 16 pixels a fragment read touches per quarter-wave
//    fall into 16 different bank groups; the swizzle is applied on the SOURCE side of the DMA (LDS-DMA writes lanes contiguously).
// Accumulation order per output: (dx, channel chunk, dy) instead of the implicit GEMM's (dy, dx, chunk): fp32 sums differ in their last bits, the fp16 results in
// ~1e-4 of the elements by one fp16 ulp (tests/test_gpu_encoder.py compares both kernels and the fp32 reference).
#include "pclip_common.h"
#include "pclip_gemm.h"
#include <stdlib.h>
#include <type_traits>

namespace {

typedef int int4_t __attribute__((ext_vector_type(4)));
constexpr int ST_ROWS = 8, ST_COLS = 56, ST_HR = ST_ROWS + 2, ST_HC = ST_COLS + 2;

template <int CIN>
struct StripGeo {
    static constexpr int PXB = CIN * 2;                               // bytes per pixel
    static constexpr int SPP = PXB / 16;                              // 16-byte slots per pixel
    static constexpr int NCH = CIN / 32;                              // 32-channel chunks (one MFMA K each)
    static constexpr int PIECES = ST_HR * ST_HC * SPP;                // 16-byte pieces of a halo block
    static constexpr int ROUNDS = (PIECES + 255) / 256;               // DMA rounds of the workgroup (256 pieces each)
    static constexpr int BUF = ROUNDS * 4096;                         // bytes of one block buffer (the last round's tail lands in slack)
};

// XOR key of the 16-byte slots of LDS pixel (row, col): see the header (bank groups of a fragment read)
template <int CIN>
__device__ __forceinline__ int strip_key(int row, int col) {
    if (CIN == 64) return ((col >> 1) & 3) | (((row >> 2) & 1) << 2);
    return ((col >> 2) & 1) | (((row >> 2) & 1) << 1);
}

template <int I, int N, class F>
__device__ __forceinline__ void static_for(F&& f) {
    if constexpr (I < N) {
        f(std::integral_constant<int, I>{});
        static_for<I + 1, N>(f);
    }
}

// The arithmetic of blocks B0 .. B0 + NB - 1 of a tile (block b = 4 s + j: sub-tile s = 8 pixel columns, rows j and j + 4), all taps, for the wave's 32 channels.
// A GROUP = (column shift dx, channel chunk, sub-tile): up to six fragment reads feeding up to 24 MFMAs.  The reads of group g + 1 are issued BEFORE the MFMAs of
// group g (two fragment sets; sched_barrier keeps hipcc from sinking them behind the MFMAs, where their latency would be exposed once per group).
template <int CIN, int B0, int NB>
struct StripGroups {
    static constexpr int PXB = CIN * 2, NCH = CIN / 32, B1 = B0 + NB - 1, S0 = B0 / 4, S1 = B1 / 4, NS = S1 - S0 + 1, NG = 3 * NCH * NS;
    static constexpr int dx(int g) { return g / (NCH * NS); }
    static constexpr int ch(int g) { return (g / NS) % NCH; }
    static constexpr int s(int g) { return S0 + g % NS; }
    static constexpr int jlo(int g) { return (B0 > 4 * s(g) ? B0 : 4 * s(g)) - 4 * s(g); }
    static constexpr int jhi(int g) { return (B1 < 4 * s(g) + 3 ? B1 : 4 * s(g) + 3) - 4 * s(g); }
};

template <int CIN, int B0, int NB>
__device__ __forceinline__ void strip_compute(const char* __restrict__ buf, const int (&base)[3][2], const half8_t (&wr)[9][CIN / 32][2], float4_t (&acc)[NB][2]) {
    using T = StripGroups<CIN, B0, NB>;
    half8_t f[2][6];                                                   // [set][LDS row jlo .. jhi + 2] (lanes of the upper half: row + 4)
    auto load = [&](auto gc) {
        constexpr int g = decltype(gc)::value, dx = T::dx(g), ch = T::ch(g), s = T::s(g);
        static_for<T::jlo(g), T::jhi(g) + 3>([&](auto rc) {
            constexpr int R0 = decltype(rc)::value;
            constexpr int X = (CIN == 64 ? ch : 0) ^ ((R0 >> 2) & 1);
            constexpr int imm = (R0 * ST_HC + 8 * s + dx) * T::PXB;
            f[g & 1][R0] = *reinterpret_cast<const half8_t*>(buf + base[dx][X] + imm);
        });
    };
    load(std::integral_constant<int, 0>{});
    static_for<0, T::NG>([&](auto gc) {
        constexpr int g = decltype(gc)::value, dx = T::dx(g), ch = T::ch(g), s = T::s(g);
        if constexpr (g + 1 < T::NG) load(std::integral_constant<int, g + 1>{});
        __builtin_amdgcn_sched_barrier(0);
        static_for<0, 3>([&](auto dyc) {
            constexpr int dy = decltype(dyc)::value;
            static_for<T::jlo(g), T::jhi(g) + 1>([&](auto jc) {
                constexpr int j = decltype(jc)::value, b = 4 * s + j - B0;
                acc[b][0] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[dy * 3 + dx][ch][0], f[g & 1][j + dy], acc[b][0], 0, 0, 0);
                acc[b][1] = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[dy * 3 + dx][ch][1], f[g & 1][j + dy], acc[b][1], 0, 0, 0);
            });
        });
        __builtin_amdgcn_sched_barrier(0);
    });
}

// POOL: the 2 x 2 average pool behind the convolution (the stem's conv3 -> bn3 -> relu -> avgpool, clip/model.py:104-105, 142-143) in the epilogue — y is the POOLED
// image [B, H / 2, W / 2, COUT]: the fp16 results of a window (rows j, j + 1: two blocks of the lane; columns: the neighbouring lane, one DPP exchange) are summed in
// pclip_avgpool_nhwc_f16's order and rounded once — the same bits without writing and re-reading the 4x larger activation.
template <int CIN, int COUT, int ACT, bool POOL = false>
__global__ __launch_bounds__(256, 1) void conv3x3_strip_kernel(const half_t* __restrict__ x, const half_t* __restrict__ w, int ldw,
                                                               int H, int W, int ntiles, const float* __restrict__ scale, const float* __restrict__ shift,
                                                               half_t* __restrict__ y) {
    using Geo = StripGeo<CIN>;
    constexpr int PXB = Geo::PXB, SPP = Geo::SPP, NCH = Geo::NCH, ROUNDS = Geo::ROUNDS;
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int G = gridDim.x;
    int tile = pgemm::xcd_remap(blockIdx.x, G);
    if (tile >= ntiles) return;
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = W / ST_COLS, tiles_y = H / ST_ROWS, tpi = tiles_x * tiles_y;

    // ---- the lane's DMA table: source offset of LDS piece (round, tid) relative to the tile's first pixel, border flags in the four low bits ----
    int rel[ROUNDS];
#pragma unroll
    for (int r = 0; r < ROUNDS; ++r) {
        const int q = r * 256 + tid;
        int v = 0;                                                    // (the tail of the last round copies the tile's first piece into slack)
        if (q < Geo::PIECES) {
            const int pix = q / SPP, slot = (q - pix * SPP) ^ strip_key<CIN>(pix / ST_HC, pix % ST_HC);
            const int row = pix / ST_HC, col = pix - row * ST_HC;
            v = ((row - 1) * W + (col - 1)) * PXB + slot * 16;
            v |= (row == 0 ? 1 : 0) | (row == ST_HR - 1 ? 2 : 0) | (col == 0 ? 4 : 0) | (col == ST_HC - 1 ? 8 : 0);
        }
        rel[r] = v;
    }
    // ---- the lane's fragment bases: pixel (4 (c >> 3), c & 7) of a block, channel slot kq, for the three column shifts and the two row-bit parities ----
    const int c = lane & 15, kq = lane >> 4, ci = c & 7, hb = c >> 3;
    int base[3][2];
#pragma unroll
    for (int dx = 0; dx < 3; ++dx)
#pragma unroll
        for (int X = 0; X < 2; ++X) {
            const int pixoff = (4 * hb * ST_HC + ci) * PXB;
            int slot;
            if (CIN == 64) slot = ((X ^ hb) << 2) | (kq ^ (((dx + ci) >> 1) & 3));
            else slot = kq ^ ((((dx + ci) >> 2) & 1) | ((X ^ hb) << 1));
            base[dx][X] = pixoff + slot * 16;
        }
    // ---- the wave's weights, BatchNorm scale / shift of the lane's 8 output channels ----
    const int nh = COUT == 64 ? wave >> 1 : 0, mpart = COUT == 64 ? wave & 1 : wave;
    half8_t wr[9][NCH][2];
#pragma unroll
    for (int tap = 0; tap < 9; ++tap)
#pragma unroll
        for (int ch = 0; ch < NCH; ++ch)
#pragma unroll
            for (int nb = 0; nb < 2; ++nb) {
                const int co = nh * 32 + (c >> 2) * 8 + nb * 4 + (c & 3);
                wr[tap][ch][nb] = *reinterpret_cast<const half8_t*>(w + (size_t)co * ldw + tap * CIN + ch * 32 + kq * 8);
            }
    const int co0 = nh * 32 + kq * 8;                                 // the lane's output channels co0 .. co0 + 7 (index nb * 4 + e)
    float sc[8], sh[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { sc[i] = scale[co0 + i]; sh[i] = shift[co0 + i]; }

    // LDS-DMA through a BUFFER descriptor over the whole input (not global_load_lds): out-of-image pieces take an offset beyond the buffer and the hardware writes
    // zeros — the padding needs no zero line — and hipcc keeps counting its LDS waits (lgkmcnt(N)) while the DMA is in flight: with a FLAT-encoded LDS load pending it
    // falls back to lgkmcnt(0) before every group of MFMAs, i.e. it exposes the latency of the fragment reads it had just issued ahead.
    const pgemm::rsrc_t rs = pgemm::make_rsrc(x, (unsigned)ntiles * (unsigned)(ST_ROWS * ST_COLS * PXB));
    auto issue = [&](int t, int pbuf) {
        const int img = t / tpi, rem = t - img * tpi, ty = rem / tiles_x, tx = rem - ty * tiles_x;
        const int tflags = (ty == 0 ? 1 : 0) | (ty == tiles_y - 1 ? 2 : 0) | (tx == 0 ? 4 : 0) | (tx == tiles_x - 1 ? 8 : 0);
        const unsigned org = (unsigned)((img * H + ty * ST_ROWS) * W + tx * ST_COLS) * (unsigned)PXB;
        pgemm::stress_jitter(110);
        (void)tflags; (void)org; (void)rs;                            // (the host pass parses the lambda without the device-only builtins)
#if defined(__HIP_DEVICE_COMPILE__)
#pragma unroll
        for (int r = 0; r < ROUNDS; ++r) {
            const unsigned voff = (rel[r] & tflags) ? 0xfffffff0u : org + (unsigned)(rel[r] & ~15);
            __builtin_amdgcn_raw_ptr_buffer_load_lds(rs, (pgemm::lds_ptr_t)(smem + pbuf * Geo::BUF + (r * 256 + wave * 64) * 16), 16, (int)voff, 0, 0, 0);
        }
#endif
    };
    auto run = [&](auto b0c, auto nbc, const char* buf, half_t* yorg) {
        constexpr int B0 = decltype(b0c)::value, NB = decltype(nbc)::value;
        float4_t acc[NB][2];
#pragma unroll
        for (int b = 0; b < NB; ++b) { acc[b][0] = float4_t{0.f, 0.f, 0.f, 0.f}; acc[b][1] = float4_t{0.f, 0.f, 0.f, 0.f}; }
        strip_compute<CIN, B0, NB>(buf, base, wr, acc);
        // BatchNorm (+ ReLU) with the implicit-GEMM kernel's rounding points (r16(acc): the convolution's fp16 output; r16 of the affine), 16-byte stores
        auto finish = [&](int b) {
            half8_t h;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = r16(r16(acc[b][i >> 2][i & 3]) * sc[i] + sh[i]);
                if (ACT == 3) v = fmaxf(v, 0.f);
                h[i] = (half_t)v;
            }
            return h;
        };
        if constexpr (POOL) {
            static_assert(!POOL || (B0 % 2 == 0 && NB % 2 == 0), "a window's two rows are consecutive blocks of one wave");
#pragma unroll
            for (int b = 0; b < NB; b += 2) {
                const int s = (B0 + b) >> 2, j = (B0 + b) & 3;                // j = 0 / 2: rows (j, j + 1) of both halves
                const half8_t h0 = finish(b), h1 = finish(b + 1);
                const int4_t i0 = __builtin_bit_cast(int4_t, h0), i1 = __builtin_bit_cast(int4_t, h1);
                int4_t n0, n1;                                                // the neighbouring column's values (lane ^ 1)
#pragma unroll
                for (int e = 0; e < 4; ++e) {
                    n0[e] = __builtin_amdgcn_update_dpp(0, i0[e], 0xB1, 0xF, 0xF, false);      // quad_perm [1, 0, 3, 2]
                    n1[e] = __builtin_amdgcn_update_dpp(0, i1[e], 0xB1, 0xF, 0xF, false);
                }
                const half8_t g0 = __builtin_bit_cast(half8_t, n0), g1 = __builtin_bit_cast(half8_t, n1);
                half8_t o;
#pragma unroll
                for (int i = 0; i < 8; ++i) {                                  // (dy 0, dx 0), (0, 1), (1, 0), (1, 1): pclip_avgpool_nhwc_f16's order, for the even column
                    float a = 0.f;
                    a += (float)h0[i]; a += (float)g0[i]; a += (float)h1[i]; a += (float)g1[i];
                    o[i] = (half_t)(a * 0.25f);
                }
                if (!(ci & 1))
                    *reinterpret_cast<half8_t*>(yorg + ((size_t)((j >> 1) + 2 * hb) * (W >> 1) + 4 * s + (ci >> 1)) * COUT + co0) = o;
            }
        } else {
#pragma unroll
            for (int b = 0; b < NB; ++b) {
                const int s = (B0 + b) >> 2, j = (B0 + b) & 3;
                *reinterpret_cast<half8_t*>(yorg + ((size_t)(j + 4 * hb) * W + 8 * s + ci) * COUT + co0) = finish(b);
            }
        }
    };

    issue(tile, 0);
    int p = 0;
    for (; tile < ntiles; tile += G, p ^= 1) {
        pgemm::wait_vm<0>();                                          // this wave's pieces of the block (and its stores of the previous tile)
        pgemm::lds_barrier();                                         // everybody's pieces; everybody is done reading the other buffer
        if (tile + G < ntiles) issue(tile + G, p ^ 1);
        const int img = tile / tpi, rem = tile - img * tpi, ty = rem / tiles_x, tx = rem - ty * tiles_x;
        half_t* yorg = POOL ? y + ((size_t)(img * (H >> 1) + ty * (ST_ROWS / 2)) * (W >> 1) + tx * (ST_COLS / 2)) * COUT
                            : y + ((size_t)(img * H + ty * ST_ROWS) * W + tx * ST_COLS) * COUT;
        const char* buf = smem + p * Geo::BUF;
        if constexpr (COUT == 64) {
            if (mpart == 0) run(std::integral_constant<int, 0>{}, std::integral_constant<int, 14>{}, buf, yorg);
            else run(std::integral_constant<int, 14>{}, std::integral_constant<int, 14>{}, buf, yorg);
        } else {
            if (mpart == 0) run(std::integral_constant<int, 0>{}, std::integral_constant<int, 7>{}, buf, yorg);
            else if (mpart == 1) run(std::integral_constant<int, 7>{}, std::integral_constant<int, 7>{}, buf, yorg);
            else if (mpart == 2) run(std::integral_constant<int, 14>{}, std::integral_constant<int, 7>{}, buf, yorg);
            else run(std::integral_constant<int, 21>{}, std::integral_constant<int, 7>{}, buf, yorg);
        }
    }
}

// ---- the stem's first convolution: 3 -> COUT channels, 3x3, stride 2, pad 1, straight from the NCHW image (clip/model.py:100-102, 138 of the reference) ----
// Before: pclip_im2col3x3_f16 wrote a [B * 112 * 112, 64] matrix (27 of 64 columns used: 411 MB at 256 images) for a GEMM with K = 64 — 0.43 ms of RN50's 5.4 ms per
// 256 images, plus the fp32 -> fp16 cast of the images as a pass of its own.  Here a workgroup loads the 17 x 114-pixel input patch of an 8 x 56 output tile into LDS
// (fp32 images are rounded to fp16 on the way, as the cast did), every lane assembles the 27 taps of its pixel into one MFMA operand (K = 32, zero beyond 27), and the
// result leaves through the strip kernel's epilogue: BatchNorm + ReLU, 16-byte stores of 8 channels per (pixel, lane).  Memory-bound: 77 MB in, 205 MB out per 256 images.
constexpr int SM_PR = 2 * ST_ROWS + 1, SM_PP = ST_COLS + 2, SM_PITCH = 2 * SM_PP + 2;     // patch rows, column PAIRS, pitch in halves (118: rows 8 apart fall 24 banks apart)

template <bool F32IN, int COUT, int ACT>
__global__ __launch_bounds__(256) void stem_conv_kernel(const void* __restrict__ img, int R, int Ho, int Wo, const half_t* __restrict__ w, int ldw,
                                                        const float* __restrict__ scale, const float* __restrict__ shift, half_t* __restrict__ y) {
    __shared__ __attribute__((aligned(16))) half_t patch[3 * SM_PR * SM_PITCH];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    const int tiles_x = Wo / ST_COLS, tiles_y = Ho / ST_ROWS, tpi = tiles_x * tiles_y;
    const int tile = blockIdx.x, b = tile / tpi, rem = tile - b * tpi, ty = rem / tiles_x, tx = rem - ty * tiles_x;
    const int oy0 = ty * ST_ROWS, ox0 = tx * ST_COLS;
    // ---- the patch: input rows 2 oy0 - 1 .. 2 oy0 + 15, columns 2 ox0 - 2 .. 2 ox0 + 113 as aligned pairs; zeros outside the image ----
    for (int q = tid; q < 3 * SM_PR * SM_PP; q += 256) {
        const int ch = q / (SM_PR * SM_PP), r2 = q - ch * (SM_PR * SM_PP), pr = r2 / SM_PP, pp = r2 - pr * SM_PP;
        const int yin = 2 * oy0 - 1 + pr, xin = 2 * ox0 - 2 + 2 * pp;
        half2_t v = {(half_t)0.f, (half_t)0.f};
        if (yin >= 0 && yin < R && xin >= 0 && xin + 1 < R) {
            const size_t o = ((size_t)(b * 3 + ch) * R + yin) * R + xin;
            if (F32IN) {
                const float2 f = *reinterpret_cast<const float2*>(reinterpret_cast<const float*>(img) + o);
                v = half2_t{(half_t)f.x, (half_t)f.y};
            } else {
                v = *reinterpret_cast<const half2_t*>(reinterpret_cast<const half_t*>(img) + o);
            }
        }
        *reinterpret_cast<half2_t*>(patch + (ch * SM_PR + pr) * SM_PITCH + 2 * pp) = v;
    }
    // ---- weights (A operand: 16 output channels x K = 32 per fragment), BatchNorm constants of the lane's channels ----
    const int c = lane & 15, kq = lane >> 4, ci = c & 7, hb = c >> 3;
    constexpr int NH = COUT / 32;
    half8_t wr[NH][2];
    float sc[NH][8], sh[NH][8];
#pragma unroll
    for (int nh = 0; nh < NH; ++nh) {
#pragma unroll
        for (int nb = 0; nb < 2; ++nb)
            wr[nh][nb] = *reinterpret_cast<const half8_t*>(w + (size_t)(nh * 32 + (c >> 2) * 8 + nb * 4 + (c & 3)) * ldw + kq * 8);
#pragma unroll
        for (int i = 0; i < 8; ++i) { sc[nh][i] = scale[nh * 32 + kq * 8 + i]; sh[nh][i] = shift[nh * 32 + kq * 8 + i]; }
    }
    // the lane's taps: k = kq * 8 + i = (ky * 3 + kx) * 3 + channel (the im2col column order of the weights), patch offset of tap k relative to the pixel's origin
    int koff[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) {
        const int k = kq * 8 + i, tap = k / 3, ch = k - tap * 3, ky = tap / 3, kx = tap - ky * 3;
        koff[i] = k < 27 ? (ch * SM_PR + ky) * SM_PITCH + kx + 1 : -1;
    }
    __syncthreads();
    half_t* yorg = y + ((size_t)(b * Ho + oy0) * Wo + ox0) * COUT;
#pragma unroll 1
    for (int bi = 0; bi < 7; ++bi) {
        const int blk = wave * 7 + bi, s = blk >> 2, j = blk & 3;
        const int ly = j + 4 * hb, lx = 8 * s + ci;
        const int po = 2 * ly * SM_PITCH + 2 * lx;
        half8_t f;
#pragma unroll
        for (int i = 0; i < 8; ++i) f[i] = koff[i] >= 0 ? patch[po + koff[i]] : (half_t)0.f;
#pragma unroll
        for (int nh = 0; nh < NH; ++nh) {
            const float4_t z = {0.f, 0.f, 0.f, 0.f};
            const float4_t a0 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[nh][0], f, z, 0, 0, 0);
            const float4_t a1 = __builtin_amdgcn_mfma_f32_16x16x32_f16(wr[nh][1], f, z, 0, 0, 0);
            half8_t h;
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                float v = r16(r16(i < 4 ? a0[i & 3] : a1[i & 3]) * sc[nh][i] + sh[nh][i]);
                if (ACT == 3) v = fmaxf(v, 0.f);
                h[i] = (half_t)v;
            }
            *reinterpret_cast<half8_t*>(yorg + ((size_t)ly * Wo + lx) * COUT + nh * 32 + kq * 8) = h;
        }
    }
}

int g_strip_mode = -1;                                                // -1: environment (PCLIP_CONV_STRIP, default on), 0 off, 1 on

template <int CIN, int COUT, int ACT, bool POOL = false>
int launch_strip(const void* x, const void* w, int B, int H, int W, const float* scale, const float* shift, void* y, int cus, hipStream_t s) {
    static DevOnce attr;
    constexpr int LDS = 2 * StripGeo<CIN>::BUF;
    if (!attr.done()) {
        if (hipFuncSetAttribute((const void*)conv3x3_strip_kernel<CIN, COUT, ACT, POOL>, hipFuncAttributeMaxDynamicSharedMemorySize, LDS) != hipSuccess) {
            pclip_set_error("pclip_conv3x3_bn_f16: cannot raise the dynamic LDS limit to %d", LDS);
            return PCLIP_E_LAUNCH;
        }
        attr.set();
    }
    const int ntiles = B * (H / ST_ROWS) * (W / ST_COLS), ldw = (9 * CIN + 63) / 64 * 64;
    conv3x3_strip_kernel<CIN, COUT, ACT, POOL><<<ntiles < cus ? ntiles : cus, 256, LDS, s>>>((const half_t*)x, (const half_t*)w, ldw, H, W, ntiles, scale,
                                                                                      shift, (half_t*)y);
    return pclip_check_launch("conv3x3_bn (strip)");
}

}  // namespace

// Does the strip kernel take this convolution?  (pclip_conv3x3_bn_f16's routing; exported for the tests' route check.)
extern "C" int pclip_conv3x3_strip_applies(int B, int H, int W, int Cin, int Cout) {
    static const bool env_on = !(getenv("PCLIP_CONV_STRIP") && getenv("PCLIP_CONV_STRIP")[0] == '0');
    const bool on = g_strip_mode < 0 ? env_on : g_strip_mode != 0;
    // (by the layer's shape alone, never by the batch: an image's features do not depend on how many images are encoded with it — tests/test_gpu_encoder.py)
    return on && (Cin == 32 || Cin == 64) && (Cout == 32 || Cout == 64) && H % ST_ROWS == 0 && W % ST_COLS == 0 && B > 0 && (long)H * W * Cin * 2 < (1L << 31);
}

// -1: the environment's choice (PCLIP_CONV_STRIP, default on); 0 / 1: off / on.  Returns the previous mode.  (Tests and A/B runs.)
extern "C" int pclip_conv3x3_strip_config(int mode) {
    const int prev = g_strip_mode;
    g_strip_mode = mode < 0 ? -1 : (mode != 0);
    return prev;
}

int pclip_conv3x3_strip_launch(const void* x, const void* w, int B, int H, int W, int Cin, int Cout, const float* scale, const float* shift, int relu,
                               void* y, int cus, hipStream_t s, int pool) {
    // the buffer descriptor addresses 2^31 bytes: larger batches go in slices of whole images (the same kernel, the same bits)
    const long per_image = (long)H * W * Cin * 2;
    const int slice = (int)((1L << 31) / per_image);
    for (int b0 = 0; b0 < B; b0 += slice) {
        const int nb = B - b0 < slice ? B - b0 : slice;
        const void* xs = (const char*)x + (size_t)b0 * per_image;
        void* ys = (char*)y + (size_t)b0 * (H / pool) * (W / pool) * Cout * 2;
        int rc = PCLIP_E_INVALID;
        if (pool == 2) {
            if (Cin == 32 && Cout == 64 && relu) rc = launch_strip<32, 64, 3, true>(xs, w, nb, H, W, scale, shift, ys, cus, s);
            if (rc != PCLIP_OK) return rc;
            continue;
        }
#define PCLIP_STRIP_CASE(CI, CO)                                                                                             \
    if (Cin == CI && Cout == CO)                                                                                             \
        rc = relu ? launch_strip<CI, CO, 3>(xs, w, nb, H, W, scale, shift, ys, cus, s) : launch_strip<CI, CO, 2>(xs, w, nb, H, W, scale, shift, ys, cus, s);
        PCLIP_STRIP_CASE(64, 64)
        PCLIP_STRIP_CASE(32, 64)
        PCLIP_STRIP_CASE(32, 32)
        PCLIP_STRIP_CASE(64, 32)
#undef PCLIP_STRIP_CASE
        if (rc != PCLIP_OK) return rc;
    }
    return PCLIP_OK;
}

// The stem's first convolution + BatchNorm + ReLU straight from the NCHW images (fp32 or fp16): see stem_conv_kernel.  R even, ((R - 1) / 2 + 1) a multiple of 56 (and of
// 8), Cout 32 or 64, w [Cout][64] in im2col column order (ky, kx, channel; zero beyond 27).
extern "C" int pclip_stem_conv_applies(int R, int Cout) {
    static const bool env_on = !(getenv("PCLIP_CONV_STEM") && getenv("PCLIP_CONV_STEM")[0] == '0');
    const int Ho = (R - 1) / 2 + 1;
    return env_on && R > 0 && R % 2 == 0 && Ho % ST_ROWS == 0 && Ho % ST_COLS == 0 && (Cout == 32 || Cout == 64);
}

extern "C" int pclip_stem_conv_bn_f16(const void* img, int img_is_f32, int B, int R, const void* w, int Cout, const float* scale, const float* shift, int relu, void* y,
                                      pclip_stream_t stream) {
    PCLIP_REQUIRE(img && w && scale && shift && y, "pclip_stem_conv_bn_f16: null pointer");
    PCLIP_REQUIRE(B >= 0 && R > 0 && R % 2 == 0 && ((R - 1) / 2 + 1) % ST_COLS == 0 && ((R - 1) / 2 + 1) % ST_ROWS == 0,
                  "pclip_stem_conv_bn_f16: image side %d unsupported (even, half of it a multiple of 56; use pclip_im2col3x3_f16 + pclip_gemm_bn_f16)", R);
    PCLIP_REQUIRE(Cout == 32 || Cout == 64, "pclip_stem_conv_bn_f16: Cout=%d unsupported (32 or 64)", Cout);
    PCLIP_REQUIRE(((uintptr_t)img & 7) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)y & 15) == 0, "pclip_stem_conv_bn_f16: pointers must be aligned (image 8, w / y 16 bytes)");
    if (B == 0) return PCLIP_OK;
    const int Ho = (R - 1) / 2 + 1, ntiles = B * (Ho / ST_ROWS) * (Ho / ST_COLS);
    hipStream_t s = (hipStream_t)stream;
#define PCLIP_STEM_CASE(F32, CO, ACT) stem_conv_kernel<F32, CO, ACT><<<ntiles, 256, 0, s>>>(img, R, Ho, Ho, (const half_t*)w, 64, scale, shift, (half_t*)y)
    if (Cout == 32) {
        if (img_is_f32) { if (relu) PCLIP_STEM_CASE(true, 32, 3); else PCLIP_STEM_CASE(true, 32, 2); }
        else { if (relu) PCLIP_STEM_CASE(false, 32, 3); else PCLIP_STEM_CASE(false, 32, 2); }
    } else {
        if (img_is_f32) { if (relu) PCLIP_STEM_CASE(true, 64, 3); else PCLIP_STEM_CASE(true, 64, 2); }
        else { if (relu) PCLIP_STEM_CASE(false, 64, 3); else PCLIP_STEM_CASE(false, 64, 2); }
    }
#undef PCLIP_STEM_CASE
    return pclip_check_launch("stem_conv_bn");
}

// relu(bn(conv3x3(x))) followed by nn.AvgPool2d(2), in one launch: y [B * (H / 2) * (W / 2), Cout] — the stem's conv3 / bn3 / relu / avgpool (clip/model.py:104-105,
// 142-143).  Same bits as pclip_conv3x3_bn_f16 (strip kernel) + pclip_avgpool_nhwc_f16.  Shapes: pclip_conv3x3_pool_applies (Cin 32, Cout 64, H % 8 == 0, W % 56 == 0).
extern "C" int pclip_conv3x3_pool_applies(int H, int W, int Cin, int Cout) {
    static const bool env_on = !(getenv("PCLIP_CONV_POOL") && getenv("PCLIP_CONV_POOL")[0] == '0');
    return env_on && pclip_conv3x3_strip_applies(1, H, W, Cin, Cout) && Cin == 32 && Cout == 64;
}

extern "C" int pclip_conv3x3_bn_pool_f16(const void* x, const void* w, int B, int H, int W, int Cin, int Cout, const float* scale, const float* shift, void* y,
                                         pclip_stream_t stream) {
    PCLIP_REQUIRE(x && w && scale && shift && y, "pclip_conv3x3_bn_pool_f16: null pointer");
    PCLIP_REQUIRE(B >= 0 && H > 0 && W > 0 && pclip_conv3x3_pool_applies(H, W, Cin, Cout),
                  "pclip_conv3x3_bn_pool_f16: shape H=%d W=%d Cin=%d Cout=%d not supported (pclip_conv3x3_pool_applies; use pclip_conv3x3_bn_f16 + pclip_avgpool_nhwc_f16)", H, W, Cin, Cout);
    PCLIP_REQUIRE(((uintptr_t)x & 15) == 0 && ((uintptr_t)w & 15) == 0 && ((uintptr_t)y & 15) == 0, "pclip_conv3x3_bn_pool_f16: pointers must be 16-byte aligned");
    if (B == 0) return PCLIP_OK;
    int cus = pclip_device_cus();
    if (cus <= 0) cus = 256;
    return pclip_conv3x3_strip_launch(x, w, B, H, W, Cin, Cout, scale, shift, 1, y, cus, (hipStream_t)stream, 2);
}
