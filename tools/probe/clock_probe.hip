// Which clock does s_memtime count on gfx950?  (GRBM_GUI_ACTIVE / wall time says 1.98 GHz under the c_fc GEMM,
// an s_memtime-instrumented build suggested 1.4 - 1.5 GHz.)
// One wave runs a dependent v_fma_f32 chain of a fixed instruction count and stamps s_memtime (clock64) and s_memrealtime
// (wall_clock64: the constant 100 MHz reference) before and after.  The chain's cost in SHADER cycles is a constant of the
// hardware, so:  - if s_memtime ticks per instruction stay the same when the chip is throttled (the probe shares the chip with a
// power-hungry GEMM train on another stream) while the real time per instruction grows, s_memtime counts shader cycles and
// ticks / real time IS the shader clock under that load;  - if the ticks per instruction grow with the real time, s_memtime is a
// constant-rate counter and says nothing about the shader clock.
// Build: hipcc -O3 --offload-arch=gfx950 -shared -fPIC -o libclock_probe.so clock_probe.hip     Driver: tools/clock_probe.py
#include <hip/hip_runtime.h>

__global__ void clock_probe_kernel(unsigned long long* out, long iters) {
    float a = threadIdx.x * 1e-3f;
    asm volatile("" : "+v"(a));
    const unsigned long long t0 = clock64(), r0 = wall_clock64();
    asm volatile("" : "+v"(a));
    for (long i = 0; i < iters; ++i) {
#pragma unroll
        for (int j = 0; j < 64; ++j) a = fmaf(a, 1.0001f, 0.5f);
    }
    asm volatile("" : "+v"(a));
    const unsigned long long t1 = clock64(), r1 = wall_clock64();
    if (threadIdx.x == 0) {
        out[0] = t1 - t0;                   // s_memtime ticks
        out[1] = r1 - r0;                   // 100 MHz reference ticks
        out[2] = (unsigned long long)iters * 64;   // dependent v_fma_f32 instructions
        out[3] = __float_as_uint(a);
    }
}

extern "C" int clock_probe_launch(unsigned long long* out, long iters, void* stream) {
    clock_probe_kernel<<<1, 64, 0, (hipStream_t)stream>>>(out, iters);
    return (int)hipGetLastError();
}
