// VALU / transcendental / MFMA issue-rate probe (gfx950): how many cycles does a wave64 instruction of each kind cost on one SIMD,
// alone and with a second wave on the same SIMD?  Build: hipcc -O3 --offload-arch=gfx950 -o valu_probe valu_probe.hip
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float16_t __attribute__((ext_vector_type(16)));
template <int KIND>
__global__ void probe(float* out, int iters) {
    float a[8];
    for (int i = 0; i < 8; ++i) a[i] = threadIdx.x * 1e-3f + i;
    float16_t acc = {};
    half8_t x = {1, 2, 3, 4, 5, 6, 7, 8};
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 4; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) {
                if (KIND == 0) a[i] = fmaf(a[i], 1.0001f, 0.5f);
                if (KIND == 1) a[i] = __builtin_amdgcn_exp2f(a[i]);
                if (KIND == 2) a[i] = fmaxf(a[i], a[(i + 1) & 7]);
                if (KIND == 3) { a[i] = fmaf(a[i], 1.0001f, 0.5f); if (i == 7) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, acc, 0, 0, 0); }
                if (KIND == 4) { if (i == 7) acc = __builtin_amdgcn_mfma_f32_32x32x16_f16(x, x, acc, 0, 0, 0); }
                if (KIND == 5) a[i] = __builtin_amdgcn_rcpf(a[i]);
            }
    }
    float s = 0;
    for (int i = 0; i < 8; ++i) s += a[i];
    for (int i = 0; i < 16; ++i) s += acc[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s;
}
template <int KIND>
void run(const char* name, int threads, int per_iter) {
    float* out; hipMalloc(&out, 256 * 1024 * 4);
    const int iters = 20000;
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    probe<KIND><<<256, threads>>>(out, 100);
    hipEventRecord(e0);
    probe<KIND><<<256, threads>>>(out, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double waves_per_simd = threads / 256.0;
    const double ns_per_inst = ms * 1e6 / ((double)iters * per_iter * waves_per_simd);
    printf("%-28s %4d threads/CU: %.3f ns per wave-instruction per SIMD (= %.2f cycles at 2.4 GHz, %.2f at 2.0)\n", name, threads, ns_per_inst,
           ns_per_inst * 2.4, ns_per_inst * 2.0);
    hipFree(out);
}
int main() {
    for (int threads : {256, 512, 1024}) {
        if (threads == 256) { run<0>("v_fma_f32", 256, 32); run<1>("v_exp_f32", 256, 32); run<2>("v_max_f32", 256, 32); run<5>("v_rcp_f32", 256, 32); run<4>("mfma 32x32x16 f16 (chain)", 256, 4); run<3>("8 fma + 1 mfma (per fma)", 256, 32); }
        if (threads == 512) { run<0>("v_fma_f32", 512, 32); run<1>("v_exp_f32", 512, 32); run<4>("mfma 32x32x16 f16 (chain)", 512, 4); run<3>("8 fma + 1 mfma (per fma)", 512, 32); }
        if (threads == 1024) { run<0>("v_fma_f32", 1024, 32); run<1>("v_exp_f32", 1024, 32); run<4>("mfma 32x32x16 f16 (chain)", 1024, 4); run<3>("8 fma + 1 mfma (per fma)", 1024, 32); }
    }
    return 0;
}
