// Image pre-processing on the device: the reference runs PIL + torchvision on the host for every image
// (clip/clip.py:77-84 Resize(BICUBIC) -> CenterCrop -> ToTensor -> Normalize; datasets/imagenet.py:8-23 RandomResizedCrop ->
// RandomHorizontalFlip -> ToTensor -> Normalize).  Here a whole batch of decoded uint8 HWC images goes through three launches.
// The arithmetic is Pillow's (libImaging/Resample.c), restated so the result is bit-identical:
//   1. coefficients: bicubic filter (a = -0.5) with antialiasing support 2*max(scale, 1), in fp64 exactly as precompute_coeffs
//      does (this file is compiled with -ffp-contract=off: no fused multiply-add may change a double), normalised and
//      converted to 22-bit fixed point as normalize_coeffs_8bpc does — only for the n_px columns / rows that survive the crop;
//   2. horizontal 8-bit pass over the rows of the source box into a uint8 scratch image (rounding to uint8 between the passes
//      is part of Pillow's result);
//   3. vertical 8-bit pass fused with the crop window, the optional horizontal flip, ToTensor (/255) and Normalize
//      ((x - mean) / std), every step a correctly rounded fp32 operation as in torch, written as CHW fp32 or fp16.
// HBM-bound byte work: reads are coalesced along the row, one thread per output pixel (3 bands), no LDS needed.
#include "pclip_common.h"

namespace {

constexpr int PBITS = 32 - 8 - 2;                 // Resample.c PRECISION_BITS
enum { D_H, D_W, D_BOX_TOP, D_BOX_LEFT, D_BOX_H, D_BOX_W, D_RS_H, D_RS_W, D_WIN_TOP, D_WIN_LEFT, D_FLIP, D_COEF_OFF, D_TMP_OFF,
       D_KS_H, D_KS_V, D_RESERVED, D_FIELDS };

__device__ __forceinline__ double bicubic_filter(double x) {
    const double a = -0.5;
    if (x < 0.0) x = -x;
    if (x < 1.0) return ((a + 2.0) * x - (a + 3.0)) * x * x + 1;
    if (x < 2.0) return (((x - 5) * x + 8) * x - 4) * a;
    return 0.0;
}

__device__ __forceinline__ int clip8(int v) {
    v >>= PBITS;                                   // arithmetic shift, like clip8_lookups[in >> PRECISION_BITS]
    return v < 0 ? 0 : (v > 255 ? 255 : v);
}

// Coefficient rows of the n_px window columns (blockIdx.z == 0) and rows (== 1) of image blockIdx.y.
// Row i of a table: [xmin, count, kk[0..ksize)] as int32.
__global__ __launch_bounds__(64) void coeffs_kernel(const int32_t* __restrict__ desc, int n_px, int32_t* __restrict__ ws) {
    const int32_t* d = desc + (size_t)blockIdx.y * D_FIELDS;
    const int dir = blockIdx.z;
    const int i = blockIdx.x * 64 + threadIdx.x;
    if (i >= n_px) return;
    const int in_size = dir ? d[D_BOX_H] : d[D_BOX_W], out_size = dir ? d[D_RS_H] : d[D_RS_W];
    const int ksize = dir ? d[D_KS_V] : d[D_KS_H];
    int32_t* tab = ws + d[D_COEF_OFF] + (dir ? n_px * (2 + d[D_KS_H]) : 0) + (size_t)i * (2 + ksize);
    const int xx = (dir ? d[D_WIN_TOP] : d[D_WIN_LEFT]) + i;              // index in the resized box
    if (in_size == out_size) {                                             // Pillow skips the pass: identity
        tab[0] = xx;
        tab[1] = 1;
        tab[2] = 1 << PBITS;
        for (int x = 1; x < ksize; ++x) tab[2 + x] = 0;
        return;
    }
    const double scale = (double)(float)in_size / out_size;               // (double)(in1 - in0) / outSize, in0 = 0 after crop()
    const double filterscale = scale < 1.0 ? 1.0 : scale;
    const double support = 2.0 * filterscale;
    const double center = 0.0 + (xx + 0.5) * scale;
    const double ss = 1.0 / filterscale;
    int xmin = (int)(center - support + 0.5);
    if (xmin < 0) xmin = 0;
    int xmax = (int)(center + support + 0.5);
    if (xmax > in_size) xmax = in_size;
    xmax -= xmin;
    double ww = 0.0;
    for (int x = 0; x < xmax; ++x) ww += bicubic_filter((x + xmin - center + 0.5) * ss);
    for (int x = 0; x < ksize; ++x) {
        int k = 0;
        if (x < xmax) {
            double w = bicubic_filter((x + xmin - center + 0.5) * ss);
            if (ww != 0.0) w /= ww;
            k = w < 0 ? (int)(-0.5 + w * (1 << PBITS)) : (int)(0.5 + w * (1 << PBITS));
        }
        tab[2 + x] = k;
    }
    tab[0] = xmin;
    tab[1] = xmax;
}

// tmp[r][i][band] for every row r of the source box and the n_px window columns i.
__global__ __launch_bounds__(256) void horizontal_kernel(const uint8_t* const* __restrict__ srcs, const int32_t* __restrict__ desc,
                                                         int n_px, const int32_t* __restrict__ wsi, uint8_t* __restrict__ wsb) {
    const int32_t* d = desc + (size_t)blockIdx.y * D_FIELDS;
    const int box_h = d[D_BOX_H];
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= box_h * n_px) return;
    const int r = idx / n_px, i = idx - r * n_px;
    const int ks = d[D_KS_H];
    const int32_t* tab = wsi + d[D_COEF_OFF] + (size_t)i * (2 + ks);
    const int xmin = tab[0], cnt = tab[1];
    const uint8_t* row = srcs[blockIdx.y] + ((size_t)(d[D_BOX_TOP] + r) * d[D_W] + d[D_BOX_LEFT] + xmin) * 3;
    int s0 = 1 << (PBITS - 1), s1 = s0, s2 = s0;
    for (int x = 0; x < cnt; ++x) {
        const int k = tab[2 + x];
        s0 += row[3 * x] * k;
        s1 += row[3 * x + 1] * k;
        s2 += row[3 * x + 2] * k;
    }
    uint8_t* t = wsb + d[D_TMP_OFF] + (size_t)idx * 3;
    t[0] = (uint8_t)clip8(s0);
    t[1] = (uint8_t)clip8(s1);
    t[2] = (uint8_t)clip8(s2);
}

template <typename OT>
__global__ __launch_bounds__(256) void vertical_finish_kernel(const int32_t* __restrict__ desc, int n_px, const int32_t* __restrict__ wsi,
                                                              const uint8_t* __restrict__ wsb, float m0, float m1, float m2, float d0,
                                                              float d1, float d2, OT* __restrict__ out) {
    const int32_t* d = desc + (size_t)blockIdx.y * D_FIELDS;
    const int idx = blockIdx.x * 256 + threadIdx.x;
    if (idx >= n_px * n_px) return;
    const int oy = idx / n_px, ox = idx - oy * n_px;
    const int col = d[D_FLIP] ? n_px - 1 - ox : ox;
    const int ks = d[D_KS_V];
    const int32_t* tab = wsi + d[D_COEF_OFF] + n_px * (2 + d[D_KS_H]) + (size_t)oy * (2 + ks);
    const int ymin = tab[0], cnt = tab[1];
    const uint8_t* t = wsb + d[D_TMP_OFF] + ((size_t)ymin * n_px + col) * 3;
    int s0 = 1 << (PBITS - 1), s1 = s0, s2 = s0;
    for (int y = 0; y < cnt; ++y) {
        const int k = tab[2 + y];
        const uint8_t* p = t + (size_t)y * n_px * 3;
        s0 += p[0] * k;
        s1 += p[1] * k;
        s2 += p[2] * k;
    }
    const float v0 = (float)clip8(s0) / 255.f, v1 = (float)clip8(s1) / 255.f, v2 = (float)clip8(s2) / 255.f;   // ToTensor
    const size_t plane = (size_t)n_px * n_px;
    OT* o = out + (size_t)blockIdx.y * 3 * plane + idx;
    o[0] = (OT)((v0 - m0) / d0);                                                                               // Normalize
    o[plane] = (OT)((v1 - m1) / d1);
    o[2 * plane] = (OT)((v2 - m2) / d2);
}

}  // namespace

extern "C" int pclip_preprocess_u8(const void* const* srcs, const int32_t* desc, int B, int n_px, int max_box_h, float mean0,
                                   float mean1, float mean2, float std0, float std1, float std2, void* out, int out_f16, void* ws,
                                   pclip_stream_t stream) {
    PCLIP_REQUIRE((srcs && desc && out && ws) || B == 0, "pclip_preprocess_u8: null pointer");
    PCLIP_REQUIRE(B >= 0 && n_px > 0 && n_px <= 4096 && max_box_h > 0, "pclip_preprocess_u8: bad B=%d n_px=%d max_box_h=%d", B, n_px, max_box_h);
    PCLIP_REQUIRE(std0 != 0.f && std1 != 0.f && std2 != 0.f, "pclip_preprocess_u8: zero std");
    if (B == 0) return PCLIP_OK;
    hipStream_t s = (hipStream_t)stream;
    coeffs_kernel<<<dim3(ceil_div(n_px, 64), B, 2), 64, 0, s>>>(desc, n_px, (int32_t*)ws);
    horizontal_kernel<<<dim3(ceil_div(max_box_h * n_px, 256), B), 256, 0, s>>>((const uint8_t* const*)srcs, desc, n_px,
                                                                               (const int32_t*)ws, (uint8_t*)ws);
    const dim3 g3(ceil_div(n_px * n_px, 256), B);
    if (out_f16)
        vertical_finish_kernel<half_t><<<g3, 256, 0, s>>>(desc, n_px, (const int32_t*)ws, (const uint8_t*)ws, mean0, mean1, mean2, std0,
                                                          std1, std2, (half_t*)out);
    else
        vertical_finish_kernel<float><<<g3, 256, 0, s>>>(desc, n_px, (const int32_t*)ws, (const uint8_t*)ws, mean0, mean1, mean2, std0,
                                                         std1, std2, (float*)out);
    return pclip_check_launch("preprocess_u8");
}
