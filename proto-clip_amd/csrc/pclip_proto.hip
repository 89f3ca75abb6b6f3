// Memory-bank / prototype reductions (SURVEY §8 rows a4, a5, a7 and the multi-GPU shard of §8e).
// All kernels are HBM-streaming: one wave64 owns one fp16 row, 16-byte loads per lane (8 halves),
// wave-shuffle reductions for the per-vector norms, fp32 accumulation with fp16 rounding at exactly
// the points the reference's fp16 tensors round.
#include "pclip_proto_dev.h"

namespace {


// ---- row normalise: y = r16(x / r16(||x||)) ----------------------------------------------------
template <int NCH>
__global__ __launch_bounds__(256) void l2norm_rows_kernel(const half_t* __restrict__ x, half_t* y, int R, int D,
                                                          float* __restrict__ sq_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < R; row += gridDim.x * 4) {
        RowRegs<NCH> r;
        float ss = load_row_sq<NCH>(x + (size_t)row * D, D, lane, r);
        const RowDiv dn(r16(sqrtf(ss)));
        float ss2 = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            int d = c * 512 + lane * 8;
            if (d < D) {
                half8_t o;
                if (dn.fast) {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (half_t)dn.div_fast((float)r.v[c][j]);
                } else {
#pragma unroll
                    for (int j = 0; j < 8; ++j) o[j] = (half_t)((float)r.v[c][j] / dn.d);
                }
#pragma unroll
                for (int j = 0; j < 8; ++j) {
                    float f = (float)o[j];
                    ss2 += f * f;
                }
                st_half8(y + (size_t)row * D + d, o);
            }
        }
        if (sq_out) {
            ss2 = wave_sum(ss2);
            if (lane == 0) sq_out[row] = ss2;
        }
    }
}

template <int NCH>
__global__ __launch_bounds__(256) void row_sqnorm_kernel(const half_t* __restrict__ x, int R, int D,
                                                         float* __restrict__ sq_out) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int row = blockIdx.x * 4 + wave; row < R; row += gridDim.x * 4) {
        RowRegs<NCH> r;
        float ss = load_row_sq<NCH>(x + (size_t)row * D, D, lane, r);
        if (lane == 0) sq_out[row] = ss;
    }
}


template <int NCH>
__global__ __launch_bounds__(256) void proto_build_kernel(const half_t* __restrict__ mem, int K, int D,
                                                          int per_shot_norm, half_t* proto_f16, float* proto_f32,
                                                          float* proto_sq) {
    __shared__ float red[4 * NCH * 512];
    const int n = blockIdx.x, lane = threadIdx.x & 63;
    float acc[NCH][8];
    class_sum<NCH>(mem, n * K, n * K + K, D, per_shot_norm, acc, red);
    if ((threadIdx.x >> 6) == 0) finish_prototype<NCH>(acc, (float)K, n, D, lane, proto_f16, proto_f32, proto_sq);
}

// labels are non-decreasing: [lo,hi) = equal_range(labels, n)
__device__ __forceinline__ int lower_bound_i32(const int32_t* a, int n, int key) {
    int lo = 0, hi = n;
    while (lo < hi) {
        int mid = (lo + hi) >> 1;
        if (a[mid] < key) lo = mid + 1; else hi = mid;
    }
    return lo;
}

template <int NCH>
__global__ __launch_bounds__(256) void partial_sums_kernel(const half_t* __restrict__ mem,
                                                           const int32_t* __restrict__ labels, int R, int D,
                                                           int per_shot_norm, float* __restrict__ sums,
                                                           int32_t* __restrict__ counts) {
    __shared__ float red[4 * NCH * 512];
    const int n = blockIdx.x, lane = threadIdx.x & 63;
    const int lo = lower_bound_i32(labels, R, n), hi = lower_bound_i32(labels, R, n + 1);
    float acc[NCH][8];
    class_sum<NCH>(mem, lo, hi, D, per_shot_norm, acc, red);
    if ((threadIdx.x >> 6) == 0) {
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            int d = c * 512 + lane * 8;
            if (d < D) {
                float4_t o0 = {acc[c][0], acc[c][1], acc[c][2], acc[c][3]};
                float4_t o1 = {acc[c][4], acc[c][5], acc[c][6], acc[c][7]};
                *reinterpret_cast<float4_t*>(sums + (size_t)n * D + d) = o0;
                *reinterpret_cast<float4_t*>(sums + (size_t)n * D + d + 4) = o1;
            }
        }
        if (lane == 0) counts[n] = hi - lo;
    }
}

template <int NCH>
__global__ __launch_bounds__(64) void proto_finalize_kernel(const float* __restrict__ sums,
                                                            const int32_t* __restrict__ counts, int W, int N, int D,
                                                            half_t* proto_f16, float* proto_f32, float* proto_sq) {
    const int n = blockIdx.x, lane = threadIdx.x;
    float acc[NCH][8];
#pragma unroll
    for (int c = 0; c < NCH; ++c)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[c][j] = 0.f;
    int cnt = 0;
    for (int w = 0; w < W; ++w) {   // rank order: identical on every rank
        cnt += counts[(size_t)w * N + n];
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            int d = c * 512 + lane * 8;
            if (d < D) {
                const float* s = sums + ((size_t)w * N + n) * D + d;
                float4_t a = *reinterpret_cast<const float4_t*>(s), b = *reinterpret_cast<const float4_t*>(s + 4);
#pragma unroll
                for (int j = 0; j < 4; ++j) { acc[c][j] += a[j]; acc[c][j + 4] += b[j]; }
            }
        }
    }
    finish_prototype<NCH>(acc, (float)(cnt > 0 ? cnt : 1), n, D, lane, proto_f16, proto_f32, proto_sq);
}

// ---- visual bank: mean over augment epochs, normalise, optional row gather (sort by label) -------
template <int NCH>
__global__ __launch_bounds__(256) void bank_reduce_kernel(const half_t* __restrict__ feats, int A, int R, int D,
                                                          const int32_t* __restrict__ perm, half_t* __restrict__ keys) {
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int j = blockIdx.x * 4 + wave; j < R; j += gridDim.x * 4) {
        const int src = perm ? perm[j] : j;
        float acc[NCH][8];
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) acc[c][e] = 0.f;
        for (int a = 0; a < A; ++a) {
            const half_t* xr = feats + ((size_t)a * R + src) * D;
#pragma unroll
            for (int c = 0; c < NCH; ++c) {
                int d = c * 512 + lane * 8;
                if (d < D) {
                    half8_t v = ld_half8(xr + d);
#pragma unroll
                    for (int e = 0; e < 8; ++e) acc[c][e] += (float)v[e];
                }
            }
        }
        float ss = 0.f;
#pragma unroll
        for (int c = 0; c < NCH; ++c)
#pragma unroll
            for (int e = 0; e < 8; ++e) {
                acc[c][e] = r16(acc[c][e] / (float)A);
                ss += acc[c][e] * acc[c][e];
            }
        ss = wave_sum(ss);
        const float n = r16(sqrtf(ss));
#pragma unroll
        for (int c = 0; c < NCH; ++c) {
            int d = c * 512 + lane * 8;
            if (d < D) {
                half8_t o;
#pragma unroll
                for (int e = 0; e < 8; ++e) o[e] = (half_t)(acc[c][e] / n);
                st_half8(keys + (size_t)j * D + d, o);
            }
        }
    }
}

__global__ __launch_bounds__(256) void transpose_kernel(const half_t* __restrict__ x, int R, int C,
                                                        half_t* __restrict__ y) {
    __shared__ half_t tile[64][66];
    const int tx = threadIdx.x & 63, ty = threadIdx.x >> 6;
    const int r0 = blockIdx.y * 64, c0 = blockIdx.x * 64;
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int r = ty + 4 * i;
        if (r0 + r < R && c0 + tx < C) tile[r][tx] = x[(size_t)(r0 + r) * C + c0 + tx];
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < 16; ++i) {
        int c = ty + 4 * i;
        if (c0 + c < C && r0 + tx < R) y[(size_t)(c0 + c) * R + r0 + tx] = tile[tx][c];
    }
}

__global__ __launch_bounds__(256) void cast_f32_f16_kernel(const float* __restrict__ x, half_t* __restrict__ y,
                                                           size_t n) {
    size_t i = ((size_t)blockIdx.x * 256 + threadIdx.x) * 8;
    const size_t stride = (size_t)gridDim.x * 256 * 8;
    for (; i + 8 <= n; i += stride) {
        float4_t a = *reinterpret_cast<const float4_t*>(x + i), b = *reinterpret_cast<const float4_t*>(x + i + 4);
        half8_t o = {(half_t)a[0], (half_t)a[1], (half_t)a[2], (half_t)a[3],
                     (half_t)b[0], (half_t)b[1], (half_t)b[2], (half_t)b[3]};
        st_half8(y + i, o);
    }
    // tail (n % 8) handled by the last few threads of block 0
    if (blockIdx.x == 0 && threadIdx.x < (n & 7)) {
        size_t t = (n & ~(size_t)7) + threadIdx.x;
        y[t] = (half_t)x[t];
    }
}

inline int row_grid(int R) { int g = ceil_div(R, 4); return g < 1 ? 1 : (g > 8192 ? 8192 : g); }

}  // namespace

#define DISPATCH_NCH(D, CALL)                                            \
    do {                                                                 \
        if ((D) <= 512) { constexpr int NCH = 1; CALL; }                 \
        else if ((D) <= 1024) { constexpr int NCH = 2; CALL; }           \
        else if ((D) <= 2048) { constexpr int NCH = 4; CALL; }           \
        else { constexpr int NCH = 8; CALL; }                            \
    } while (0)

static int check_rows(const char* fn, const void* x, int R, int D) {
    PCLIP_REQUIRE(x != nullptr || R == 0, "%s: null pointer", fn);
    PCLIP_REQUIRE(R >= 0, "%s: negative row count %d", fn, R);
    PCLIP_REQUIRE(D > 0 && D % 8 == 0 && D <= 4096, "%s: D=%d must be a positive multiple of 8, <= 4096", fn, D);
    return 0;
}

extern "C" int pclip_l2norm_rows_f16(const void* x, void* y, int R, int D, float* sq_out, pclip_stream_t stream) {
    if (int e = check_rows("pclip_l2norm_rows_f16", x, R, D)) return e;
    PCLIP_REQUIRE(y != nullptr || R == 0, "pclip_l2norm_rows_f16: null output");
    if (R == 0) return PCLIP_OK;
    DISPATCH_NCH(D, (l2norm_rows_kernel<NCH><<<row_grid(R), 256, 0, (hipStream_t)stream>>>(
                        (const half_t*)x, (half_t*)y, R, D, sq_out)));
    return pclip_check_launch("l2norm_rows");
}

extern "C" int pclip_row_sqnorm_f16(const void* x, int R, int D, float* sq_out, pclip_stream_t stream) {
    if (int e = check_rows("pclip_row_sqnorm_f16", x, R, D)) return e;
    PCLIP_REQUIRE(sq_out != nullptr, "pclip_row_sqnorm_f16: null output");
    if (R == 0) return PCLIP_OK;
    DISPATCH_NCH(D, (row_sqnorm_kernel<NCH><<<row_grid(R), 256, 0, (hipStream_t)stream>>>((const half_t*)x, R, D, sq_out)));
    return pclip_check_launch("row_sqnorm");
}

extern "C" int pclip_proto_build_f16(const void* mem, int N, int K, int D, int per_shot_norm, void* proto_f16,
                                     float* proto_f32, float* proto_sq, pclip_stream_t stream) {
    if (int e = check_rows("pclip_proto_build_f16", mem, N, D)) return e;
    PCLIP_REQUIRE(K > 0, "pclip_proto_build_f16: K=%d must be positive", K);
    PCLIP_REQUIRE(proto_f16 || proto_f32, "pclip_proto_build_f16: no output requested");
    if (N == 0) return PCLIP_OK;
    DISPATCH_NCH(D, (proto_build_kernel<NCH><<<N, 256, 0, (hipStream_t)stream>>>(
                        (const half_t*)mem, K, D, per_shot_norm, (half_t*)proto_f16, proto_f32, proto_sq)));
    return pclip_check_launch("proto_build");
}

extern "C" int pclip_partial_sums_f16(const void* mem, const int32_t* labels, int R, int N, int D,
                                      int per_shot_norm, float* sums, int32_t* counts, pclip_stream_t stream) {
    PCLIP_REQUIRE(D > 0 && D % 8 == 0 && D <= 4096, "pclip_partial_sums_f16: bad D=%d", D);
    PCLIP_REQUIRE(N > 0 && R >= 0, "pclip_partial_sums_f16: bad N=%d R=%d", N, R);
    PCLIP_REQUIRE(sums && counts && (R == 0 || (mem && labels)), "pclip_partial_sums_f16: null pointer");
    DISPATCH_NCH(D, (partial_sums_kernel<NCH><<<N, 256, 0, (hipStream_t)stream>>>(
                        (const half_t*)mem, labels, R, D, per_shot_norm, sums, counts)));
    return pclip_check_launch("partial_sums");
}

extern "C" int pclip_proto_finalize(const float* sums, const int32_t* counts, int W, int N, int D, void* proto_f16,
                                    float* proto_f32, float* proto_sq, pclip_stream_t stream) {
    PCLIP_REQUIRE(D > 0 && D % 8 == 0 && D <= 4096, "pclip_proto_finalize: bad D=%d", D);
    PCLIP_REQUIRE(N > 0 && W > 0, "pclip_proto_finalize: bad N=%d W=%d", N, W);
    PCLIP_REQUIRE(sums && counts && (proto_f16 || proto_f32), "pclip_proto_finalize: null pointer");
    DISPATCH_NCH(D, (proto_finalize_kernel<NCH><<<N, 64, 0, (hipStream_t)stream>>>(
                        sums, counts, W, N, D, (half_t*)proto_f16, proto_f32, proto_sq)));
    return pclip_check_launch("proto_finalize");
}

extern "C" int pclip_bank_reduce_f16(const void* feats, int A, int R, int D, const int32_t* perm, void* keys,
                                     pclip_stream_t stream) {
    if (int e = check_rows("pclip_bank_reduce_f16", feats, R, D)) return e;
    PCLIP_REQUIRE(A > 0 && keys, "pclip_bank_reduce_f16: bad A=%d or null output", A);
    if (R == 0) return PCLIP_OK;
    DISPATCH_NCH(D, (bank_reduce_kernel<NCH><<<row_grid(R), 256, 0, (hipStream_t)stream>>>(
                        (const half_t*)feats, A, R, D, perm, (half_t*)keys)));
    return pclip_check_launch("bank_reduce");
}

extern "C" int pclip_transpose_f16(const void* x, int R, int C, void* y, pclip_stream_t stream) {
    PCLIP_REQUIRE(x && y && R >= 0 && C >= 0, "pclip_transpose_f16: bad arguments");
    if (R == 0 || C == 0) return PCLIP_OK;
    dim3 grid(ceil_div(C, 64), ceil_div(R, 64));
    transpose_kernel<<<grid, 256, 0, (hipStream_t)stream>>>((const half_t*)x, R, C, (half_t*)y);
    return pclip_check_launch("transpose");
}

extern "C" int pclip_cast_f32_f16(const float* x, void* y, size_t n, pclip_stream_t stream) {
    PCLIP_REQUIRE(x && y, "pclip_cast_f32_f16: null pointer");
    if (n == 0) return PCLIP_OK;
    size_t blocks = (n / 8 + 255) / 256;
    if (blocks < 1) blocks = 1;
    if (blocks > 16384) blocks = 16384;
    cast_f32_f16_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(x, (half_t*)y, n);
    return pclip_check_launch("cast_f32_f16");
}
