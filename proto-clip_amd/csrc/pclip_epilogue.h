// Epilogue arithmetic shared by the persistent linear kernels (pclip_encoder.hip: eight waves; pclip_gemm4w.hip: four waves with the asm K-loop):
// the reference's fp16 rounding points of nn.Linear + QuickGELU (clip/model.py:164-166, 176-178) and the output store policy.
#pragma once
#include "pclip_common.h"

typedef float float2_t __attribute__((ext_vector_type(2)));

// QuickGELU with the reference's three fp16 roundings.  exp / reciprocal use the hardware approximations
// (v_exp_f32, v_rcp_f32: ~1-2 ulp in fp32), far inside the fp16 rounding that follows each step.
__device__ __forceinline__ float quick_gelu16(float v) {
    const float t = r16(1.702f * v);
    const float s = r16(__builtin_amdgcn_rcpf(1.f + __expf(-t)));
    return r16(v * s);
}

// Four at a time with packed fp16 instructions where the arithmetic IS fp16: the two conversions are v_cvt_pk_f16_f32 and the
// final product h * s of two fp16 values is one correctly rounded v_pk_mul_f16 (= r16 of the exact fp32 product).
__device__ __forceinline__ half4_t quick_gelu16x4(float4_t v) {
    const half2_t h01 = __builtin_convertvector(float2_t{v[0], v[1]}, half2_t), h23 = __builtin_convertvector(float2_t{v[2], v[3]}, half2_t);
    const half_t h[4] = {h01[0], h01[1], h23[0], h23[1]};
    // The activation arithmetic is what the c_fc epilogue is bound by (VALU: tools/trace_tile.py), so every instruction counts: the exponent's argument
    // as ONE v_fma_mix_f32 on the fp16 value t (fma(t, -log2 e, 0) == the fp32 product t * -log2 e that __expf(-t) forms, without the separate
    // v_cvt_f32_f16), the "1 +" of two elements as one v_pk_add_f32: 48 instead of 54 issue slots per four elements, same bits.
    float ex[4];
#pragma unroll
    for (int e = 0; e < 4; ++e) {
        const half_t t = (half_t)(1.702f * (float)h[e]);
        ex[e] = __builtin_amdgcn_exp2f(__builtin_fmaf((float)t, -1.4426950408889634f, 0.f));
    }
    const float2_t one = {1.f, 1.f};
    const float2_t d01 = float2_t{ex[0], ex[1]} + one, d23 = float2_t{ex[2], ex[3]} + one;
    const float s[4] = {__builtin_amdgcn_rcpf(d01[0]), __builtin_amdgcn_rcpf(d01[1]), __builtin_amdgcn_rcpf(d23[0]), __builtin_amdgcn_rcpf(d23[1])};
    const half2_t s01 = __builtin_convertvector(float2_t{s[0], s[1]}, half2_t), s23 = __builtin_convertvector(float2_t{s[2], s[3]}, half2_t);
    const half2_t y01 = h01 * s01, y23 = h23 * s23;
    return half4_t{y01[0], y01[1], y23[0], y23[1]};
}

// Output rows of the persistent linear kernels: non-temporal 16-byte stores (A/B switch PCLIP_NT_STORE) — a c_fc launch writes
// 1.2 GB that nobody re-reads before it has left the 4 MiB L2 anyway; keeping it out leaves the L2 to the operand panels.
#ifndef PCLIP_NT_STORE
#define PCLIP_NT_STORE 1
#endif
typedef float f32x4_t __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void st_out(half_t* p, half8_t v) {
#if PCLIP_NT_STORE
    __builtin_nontemporal_store(__builtin_bit_cast(f32x4_t, v), reinterpret_cast<f32x4_t*>(p));
#else
    st_half8(p, v);
#endif
}

