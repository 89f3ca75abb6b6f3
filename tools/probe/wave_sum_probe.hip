#include <hip/hip_runtime.h>
template <int CTRL> __device__ __forceinline__ float dppf(float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(0, __builtin_bit_cast(int, v), CTRL, 0xF, 0xF, false));
}
__device__ __forceinline__ float pair16(float v) {
    const unsigned a = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane16_swap(a, a, false, false);
    const unsigned x = r[0], y = r[1];
    return __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
}
__device__ __forceinline__ float pair32(float v) {
    const unsigned a = __builtin_bit_cast(unsigned, v);
    const auto r = __builtin_amdgcn_permlane32_swap(a, a, false, false);
    const unsigned x = r[0], y = r[1];
    return __builtin_bit_cast(float, x) + __builtin_bit_cast(float, y);
}
template <int CTRL, int BANKS> __device__ __forceinline__ float dppf_old(float old, float v) {
    return __builtin_bit_cast(float, __builtin_amdgcn_update_dpp(__builtin_bit_cast(int, old), __builtin_bit_cast(int, v), CTRL, 0xF, BANKS, false));
}
__device__ __forceinline__ float xor4(float v) {          // lane i <- lane i ^ 4: banks 0, 2 read four lanes up, banks 1, 3 four lanes down
    float t = dppf_old<0x104, 0x5>(v, v);                  // row_shl:4 into banks 0 and 2
    return dppf_old<0x114, 0xA>(t, v);                     // row_shr:4 into banks 1 and 3
}
__device__ __forceinline__ float wave_sum_dpp(float v) {  // the additions of the xor butterfly 32, 16, 8, 4, 2, 1 — without the LDS crossbar
    v = pair32(v); v = pair16(v);
    v += dppf<0x128>(v);                                   // row_ror:8 == lane ^ 8
    v += xor4(v);
    v += dppf<0x4E>(v); v += dppf<0xB1>(v);
    return v;
}
__device__ __forceinline__ float wave_sum_ref(float v) {
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}
__global__ void k(const float* in, float* o1, float* o2) {
    float v = in[blockIdx.x * 64 + threadIdx.x];
    o1[blockIdx.x * 64 + threadIdx.x] = wave_sum_dpp(v);
    o2[blockIdx.x * 64 + threadIdx.x] = wave_sum_ref(v);
}
int main() {
    const int n = 64 * 4096; float *in, *o1, *o2; hipMalloc(&in, n * 4); hipMalloc(&o1, n * 4); hipMalloc(&o2, n * 4);
    float* h = new float[n]; srand(1); for (int i = 0; i < n; ++i) h[i] = (rand() / (float)RAND_MAX - 0.5f) * (1 + (i % 7) * 100.f);
    hipMemcpy(in, h, n * 4, hipMemcpyHostToDevice); k<<<4096, 64>>>(in, o1, o2); 
    float* a = new float[n]; float* b = new float[n]; hipMemcpy(a, o1, n * 4, hipMemcpyDeviceToHost); hipMemcpy(b, o2, n * 4, hipMemcpyDeviceToHost);
    int bad = 0; for (int i = 0; i < n; ++i) bad += __builtin_memcmp(&a[i], &b[i], 4) != 0; printf("mismatches %d of %d (sample %g %g)\n", bad, n, a[5], b[5]); return bad != 0;
}
