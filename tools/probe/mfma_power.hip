// Energy per flop of the two fp16 MFMA shapes under the socket's power cap: a register-only loop (no LDS, no memory) of v_mfma_f32_16x16x32_f16 vs
// v_mfma_f32_32x32x16_f16 on N(0,1) operands, one wave per SIMD with 256 accumulator registers (the four-wave GEMM's occupancy), long enough to settle at the cap.
// The chip is power-bound in the GEMMs: the shape that sustains more TFLOP/s here is the one a K-loop should be built from.
//   hipcc -O3 --offload-arch=gfx950 tools/probe/mfma_power.hip -o tools/probe/mfma_power && tools/probe/mfma_power
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cmath>
#include <vector>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));

template <int SHAPE>   // 0: 16x16x32 (64 accumulator tiles), 1: 32x32x16 (16 tiles)
__global__ __launch_bounds__(256) void mfma_loop(const half8_t* __restrict__ src, float* __restrict__ out, int iters) {
    const int lane = threadIdx.x & 63;
    half8_t a[8], b[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) { a[i] = src[(i * 64 + lane)]; b[i] = src[((8 + i) * 64 + lane)]; }
    float s = 0.f;
    if (SHAPE == 0) {
        float4_t acc[8][8];
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int i = 0; i < 8; ++i)
#pragma unroll
                for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
    } else {
        float16_t acc[4][4];
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j)
#pragma unroll
                for (int e = 0; e < 16; ++e) acc[i][j][e] = 0.f;
        for (int it = 0; it < iters; ++it) {
            // the same flops per trip as the other shape: 4 x 4 tiles x 2 k-steps of 16
#pragma unroll
            for (int ks = 0; ks < 2; ++ks)
#pragma unroll
                for (int i = 0; i < 4; ++i)
#pragma unroll
                    for (int j = 0; j < 4; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_32x32x16_f16(a[i + 4 * ks], b[j + 4 * ks], acc[i][j], 0, 0, 0);
        }
#pragma unroll
        for (int i = 0; i < 4; ++i)
#pragma unroll
            for (int j = 0; j < 4; ++j) s += acc[i][j][0] + acc[i][j][15];
    }
    if (s == 12345.678f) out[0] = s;       // keep the loop alive
}

// SHAPE 2 / 3: the 16x16x32 loop with its fragments RE-READ from LDS every trip at the four-wave GEMM's ratio (16 ds_read_b128 per 64 MFMAs; 3: twice that, the
// eight-wave kernel's ratio) — what the LDS read traffic costs in sustained rate under the cap.  LDS holds 64 KB of the same random halves, conflict-free rows.
template <int READS>
__global__ __launch_bounds__(256) void mfma_lds_loop(const half8_t* __restrict__ src, float* __restrict__ out, int iters) {
    __shared__ half8_t lds[4096];                      // 64 KB
    const int lane = threadIdx.x & 63, wave = threadIdx.x >> 6;
    for (int i = threadIdx.x; i < 4096; i += 256) lds[i] = src[i & 1023];
    __syncthreads();
    float4_t acc[8][8];
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) acc[i][j] = float4_t{0.f, 0.f, 0.f, 0.f};
    int base = wave * 1024 + lane;
    for (int it = 0; it < iters; ++it) {
        half8_t a[8], b[8];
#pragma unroll
        for (int i = 0; i < 8; ++i) { a[i] = lds[(base + i * 64) & 4095]; b[i] = lds[(base + 512 + i * 64) & 4095]; }
        if (READS == 2) {
            half8_t a2[8];
#pragma unroll
            for (int i = 0; i < 8; ++i) a2[i] = lds[(base + 2048 + i * 64) & 4095];
#pragma unroll
            for (int i = 0; i < 8; ++i) { half8_t t = lds[(base + 2560 + i * 64) & 4095]; b[i] = b[i] + a2[i] * (_Float16)0 + t * (_Float16)0; }
        }
#pragma unroll
        for (int i = 0; i < 8; ++i)
#pragma unroll
            for (int j = 0; j < 8; ++j) acc[i][j] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a[i], b[j], acc[i][j], 0, 0, 0);
        base += 64;
    }
    float s = 0.f;
#pragma unroll
    for (int i = 0; i < 8; ++i)
#pragma unroll
        for (int j = 0; j < 8; ++j) s += acc[i][j][0] + acc[i][j][3];
    if (s == 12345.678f) out[0] = s;
}

int main(int argc, char** argv) {
    const double secs = argc > 1 ? atof(argv[1]) : 1.5;
    const float scale = argc > 2 ? atof(argv[2]) : 1.0f;     // operand scale (0: zeros)
    int cus = 0;
    hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0);
    std::vector<_Float16> h(16 * 64 * 8);
    unsigned long long st = 88172645463325252ull;
    auto rnd = [&]() { st ^= st << 13; st ^= st >> 7; st ^= st << 17; return (double)(st >> 11) / 9007199254740992.0; };
    for (auto& v : h) { double u1 = rnd() + 1e-12, u2 = rnd(); v = (_Float16)(scale * 0.05 * sqrt(-2.0 * log(u1)) * cos(6.283185307179586 * u2)); }
    half8_t* src; float* out;
    hipMalloc(&src, h.size() * 2); hipMalloc(&out, 64);
    hipMemcpy(src, h.data(), h.size() * 2, hipMemcpyHostToDevice);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const double flop_per_trip_wave = 64.0 * 2 * 16 * 16 * 32;      // both shapes
    for (int rep = 0; rep < 3; ++rep)
        for (int shape = 0; shape < 4; ++shape) {
            int iters = 2000;
            for (int pass = 0; pass < 2; ++pass) {                   // calibrate, then the timed run of ~secs
                hipEventRecord(e0);
                if (shape == 0) mfma_loop<0><<<cus, 256>>>(src, out, iters);
                else if (shape == 1) mfma_loop<1><<<cus, 256>>>(src, out, iters);
                else if (shape == 2) mfma_lds_loop<1><<<cus, 256>>>(src, out, iters);
                else mfma_lds_loop<2><<<cus, 256>>>(src, out, iters);
                hipEventRecord(e1); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (pass == 0) { iters = (int)(iters * secs * 1e3 / ms); continue; }
                const double tf = flop_per_trip_wave * iters * 4.0 * cus / (ms * 1e-3) / 1e12;
                printf("%s  %8.1f ms  %8.1f TFLOP/s (%d CUs, one wave per SIMD, operand scale %.2f)\n", shape == 0 ? "v_mfma_f32_16x16x32_f16" : shape == 1 ? "v_mfma_f32_32x32x16_f16" : shape == 2 ? "16x16x32 + 16 ds_read_b128 / 64" : "16x16x32 + 32 ds_read_b128 / 64", ms, tf, cus, scale);
                fflush(stdout);
            }
        }
    return 0;
}
