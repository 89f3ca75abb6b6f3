// ModifiedResNet tower pieces: reference clip/model.py:10-152.  Activations are kept
// NHWC fp16 ([B*H*W, C] rows), so every 1x1 convolution IS the MFMA GEMM of pclip_gemm.h and every 3x3
// convolution is an im2col gather followed by that GEMM (weights re-ordered once on the host to
// [Cout, ky, kx, Cin]).  BatchNorm (eval) + ReLU + the bottleneck's residual add are one streaming pass with the
// reference's rounding points; the anti-aliasing average pools and the attention-pool token assembly are small
// HBM-bound kernels.
#include "pclip_common.h"

namespace {

// x element (b, y, x, c) lives at b*sb + y*sh + x*sw + c*sc (halves): NHWC activations or the NCHW input image.
// cols[((b*Ho + oy)*Wo + ox) * ld + (ky*3 + kx)*C + c] = x(b, oy*stride + ky - 1, ox*stride + kx - 1, c), 0 outside.
template <bool VEC, bool GATHER = false>
__global__ __launch_bounds__(256) void im2col3x3_kernel(const half_t* __restrict__ x, long sb, long sh, long sw, long sc,
                                                        int B, int H, int W, int C, int stride, int Ho, int Wo, int ld,
                                                        half_t* __restrict__ cols) {
    constexpr int V = VEC ? 8 : 1;
    const int ldv = ld / V;
    const size_t total = (size_t)B * Ho * Wo * ldv;
    const int K = 9 * C;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int k = (int)(i % ldv) * V;
        const size_t row = i / ldv;
        half_t* dst = cols + row * ld + k;
        const int ox = (int)(row % Wo), oy = (int)((row / Wo) % Ho), b = (int)(row / ((size_t)Wo * Ho));
        const int tap = k / C, c = k - tap * C;
        const int iy = oy * stride + tap / 3 - 1, ix = ox * stride + tap % 3 - 1;
        const bool in = k < K && iy >= 0 && iy < H && ix >= 0 && ix < W;
        if (VEC && GATHER) {
            // 3-channel NCHW image (the stem's first convolution): eight (tap, channel) elements gathered one by one, ONE 16-byte
            // store — the scalar path below writes 2 bytes per thread (0.7 TB/s on the 411 MB matrix of a 256-image pass)
            half8_t v;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                const int kj = k + j, tj = kj / C, cj = kj - tj * C;
                const int yj = oy * stride + tj / 3 - 1, xj = ox * stride + tj % 3 - 1;
                v[j] = (kj < K && yj >= 0 && yj < H && xj >= 0 && xj < W) ? x[b * sb + yj * sh + xj * sw + (long)cj * sc] : (half_t)0.f;
            }
            st_half8(dst, v);
        } else if (VEC) {
            half8_t v;
            if (in) v = ld_half8(x + b * sb + iy * sh + ix * sw + (long)c * sc);      // sc == 1 on the vector path
            else {
#pragma unroll
                for (int j = 0; j < 8; ++j) v[j] = (half_t)0.f;
            }
            st_half8(dst, v);
        } else {
            *dst = in ? x[b * sb + iy * sh + ix * sw + (long)c * sc] : (half_t)0.f;
        }
    }
}

// y = relu?( r16( r16(x*scale + shift) + residual ) ): BatchNorm2d in eval mode on an fp16 tensor (fp32 statistics
// folded into scale/shift on the host), the bottleneck's `out += identity`, and ReLU (clip/model.py:43-52).
__global__ __launch_bounds__(256) void bn_act_kernel(const half_t* __restrict__ x, const float* __restrict__ scale,
                                                     const float* __restrict__ shift, const half_t* __restrict__ residual,
                                                     int relu, half_t* __restrict__ y, size_t rows, int C) {
    const int CV = C / 8;
    const size_t total = rows * CV;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % CV) * 8;
        const size_t o = (i / CV) * C + c;
        const half8_t v = ld_half8(x + o);
        half8_t r, out;
        if (residual) r = ld_half8(residual + o);
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            float f = r16((float)v[j] * scale[c + j] + shift[c + j]);
            if (residual) f = r16(f + (float)r[j]);
            if (relu) f = fmaxf(f, 0.f);
            out[j] = (half_t)f;
        }
        st_half8(y + o, out);
    }
}

// nn.AvgPool2d(k) on NHWC fp16 (fp32 accumulate, one rounding), H and W multiples of k.
__global__ __launch_bounds__(256) void avgpool_kernel(const half_t* __restrict__ x, int B, int H, int W, int C, int k,
                                                      half_t* __restrict__ y) {
    const int Ho = H / k, Wo = W / k, CV = C / 8;
    const size_t total = (size_t)B * Ho * Wo * CV;
    const float inv = 1.f / (float)(k * k);
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % CV) * 8;
        const size_t p = i / CV;
        const int ox = (int)(p % Wo), oy = (int)((p / Wo) % Ho), b = (int)(p / ((size_t)Wo * Ho));
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        for (int dy = 0; dy < k; ++dy)
            for (int dx = 0; dx < k; ++dx) {
                const half8_t v = ld_half8(x + (((size_t)b * H + oy * k + dy) * W + ox * k + dx) * C + c);
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[j] += (float)v[j];
            }
        half8_t out;
#pragma unroll
        for (int j = 0; j < 8; ++j) out[j] = (half_t)(acc[j] * inv);
        st_half8(y + p * C + c, out);
    }
}

// AttentionPool2d token assembly (clip/model.py:68-70): tokens[b,0] = r16(r16(mean_hw x[b]) + pos[0]),
// tokens[b,1+p] = r16(x[b,p] + pos[1+p]); x NHWC rows [B*HW, C], pos fp16 [HW+1, C].
__global__ __launch_bounds__(256) void attnpool_tokens_kernel(const half_t* __restrict__ x, const half_t* __restrict__ pos,
                                                              int B, int HW, int C, half_t* __restrict__ tokens) {
    const int CV = C / 8;
    const size_t total = (size_t)B * CV;
    for (size_t i = (size_t)blockIdx.x * 256 + threadIdx.x; i < total; i += (size_t)gridDim.x * 256) {
        const int c = (int)(i % CV) * 8;
        const size_t b = i / CV;
        float acc[8] = {0, 0, 0, 0, 0, 0, 0, 0};
        half_t* tb = tokens + b * (HW + 1) * C;
        for (int p = 0; p < HW; ++p) {
            const half8_t v = ld_half8(x + (b * HW + p) * C + c), pp = ld_half8(pos + (size_t)(p + 1) * C + c);
            half8_t o;
#pragma unroll
            for (int j = 0; j < 8; ++j) {
                acc[j] += (float)v[j];
                o[j] = (half_t)((float)v[j] + (float)pp[j]);
            }
            st_half8(tb + (size_t)(p + 1) * C + c, o);
        }
        const half8_t p0 = ld_half8(pos + c);
        half8_t o;
#pragma unroll
        for (int j = 0; j < 8; ++j) o[j] = (half_t)(r16(acc[j] / (float)HW) + (float)p0[j]);
        st_half8(tb + c, o);
    }
}

inline int flat_grid(size_t n) { size_t g = (n + 255) / 256; return (int)(g < 1 ? 1 : (g > 32768 ? 32768 : g)); }

}  // namespace

extern "C" int pclip_im2col3x3_f16(const void* x, long sb, long sh, long sw, long sc, int B, int H, int W, int C, int stride,
                                   void* cols, int ld, pclip_stream_t stream) {
    PCLIP_REQUIRE(x && cols, "pclip_im2col3x3_f16: null pointer");
    PCLIP_REQUIRE(B >= 0 && H > 0 && W > 0 && C > 0 && (stride == 1 || stride == 2) && ld >= 9 * C,
                  "pclip_im2col3x3_f16: bad shape B=%d H=%d W=%d C=%d stride=%d ld=%d", B, H, W, C, stride, ld);
    if (B == 0) return PCLIP_OK;
    const int Ho = (H - 1) / stride + 1, Wo = (W - 1) / stride + 1;           // kernel 3, padding 1
    const bool vec = C % 8 == 0 && sc == 1 && ld % 8 == 0 && sb % 8 == 0 && sh % 8 == 0 && sw % 8 == 0;
    if (!vec && ld % 8 == 0) {                              // rows of 16-byte units, elements gathered singly
        const size_t tot8 = (size_t)B * Ho * Wo * (ld / 8);
        im2col3x3_kernel<true, true><<<flat_grid(tot8), 256, 0, (hipStream_t)stream>>>((const half_t*)x, sb, sh, sw, sc, B, H, W, C, stride, Ho, Wo, ld, (half_t*)cols);
        return pclip_check_launch("im2col3x3 (gather)");
    }
    const size_t total = (size_t)B * Ho * Wo * (vec ? ld / 8 : ld);
    if (vec) im2col3x3_kernel<true><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>((const half_t*)x, sb, sh, sw, sc, B, H, W, C, stride, Ho, Wo, ld, (half_t*)cols);
    else im2col3x3_kernel<false><<<flat_grid(total), 256, 0, (hipStream_t)stream>>>((const half_t*)x, sb, sh, sw, sc, B, H, W, C, stride, Ho, Wo, ld, (half_t*)cols);
    return pclip_check_launch("im2col3x3");
}

extern "C" int pclip_bn_act_f16(const void* x, const float* scale, const float* shift, const void* residual, int relu,
                                void* y, size_t rows, int C, pclip_stream_t stream) {
    PCLIP_REQUIRE(x && scale && shift && y, "pclip_bn_act_f16: null pointer");
    PCLIP_REQUIRE(C > 0 && C % 8 == 0, "pclip_bn_act_f16: C=%d must be a positive multiple of 8", C);
    if (rows == 0) return PCLIP_OK;
    bn_act_kernel<<<flat_grid(rows * (C / 8)), 256, 0, (hipStream_t)stream>>>((const half_t*)x, scale, shift, (const half_t*)residual,
                                                                             relu, (half_t*)y, rows, C);
    return pclip_check_launch("bn_act");
}

extern "C" int pclip_avgpool_nhwc_f16(const void* x, int B, int H, int W, int C, int k, void* y, pclip_stream_t stream) {
    PCLIP_REQUIRE(x && y, "pclip_avgpool_nhwc_f16: null pointer");
    PCLIP_REQUIRE(B >= 0 && k > 0 && H % k == 0 && W % k == 0 && C % 8 == 0, "pclip_avgpool_nhwc_f16: bad shape H=%d W=%d C=%d k=%d", H, W, C, k);
    if (B == 0) return PCLIP_OK;
    avgpool_kernel<<<flat_grid((size_t)B * (H / k) * (W / k) * (C / 8)), 256, 0, (hipStream_t)stream>>>((const half_t*)x, B, H, W, C, k, (half_t*)y);
    return pclip_check_launch("avgpool");
}

extern "C" int pclip_attnpool_tokens_f16(const void* x, const void* pos, int B, int HW, int C, void* tokens, pclip_stream_t stream) {
    PCLIP_REQUIRE(x && pos && tokens, "pclip_attnpool_tokens_f16: null pointer");
    PCLIP_REQUIRE(B >= 0 && HW > 0 && C % 8 == 0, "pclip_attnpool_tokens_f16: bad shape HW=%d C=%d", HW, C);
    if (B == 0) return PCLIP_OK;
    attnpool_tokens_kernel<<<flat_grid((size_t)B * (C / 8)), 256, 0, (hipStream_t)stream>>>((const half_t*)x, (const half_t*)pos, B, HW, C, (half_t*)tokens);
    return pclip_check_launch("attnpool_tokens");
}
