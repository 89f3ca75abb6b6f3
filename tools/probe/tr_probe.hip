// Probe of ds_read_b64_tr_b16 (gfx950): which LDS halfs end up in which lane/element.  LDS holds its own index as a value.
#include <hip/hip_runtime.h>
#include <stdio.h>
typedef __fp16 fp16x4 __attribute__((__vector_size__(4 * sizeof(__fp16))));
__global__ void k(int stride_halfs, float* out) {
    __shared__ __fp16 lds[8192];
    for (int i = threadIdx.x; i < 8192; i += 64) lds[i] = (__fp16)(float)(i & 2047);
    __syncthreads();
    fp16x4 v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) fp16x4*)(lds + threadIdx.x * stride_halfs));
    for (int j = 0; j < 4; ++j) out[threadIdx.x * 4 + j] = (float)v[j];
}
int main() {
    float* d; hipMalloc(&d, 256 * 4); float h[256];
    for (int stride : {4, 64}) {
        k<<<1, 64>>>(stride, d); hipMemcpy(h, d, sizeof(h), hipMemcpyDeviceToHost);
        printf("lane address = lds + lane*%d halfs\n", stride);
        for (int l = 0; l < 64; ++l) printf("lane %2d: %5.0f %5.0f %5.0f %5.0f\n", l, h[l*4], h[l*4+1], h[l*4+2], h[l*4+3]);
    }
    return 0;
}
