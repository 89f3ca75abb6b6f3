// Shared device/host helpers for libpclip (gfx950 only: wave64, MFMA, 160 KiB LDS).
#pragma once
#include <atomic>
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <stddef.h>
#include "../../include/pclip.h"

typedef _Float16 half_t;
typedef _Float16 half2_t __attribute__((ext_vector_type(2)));
typedef _Float16 half4_t __attribute__((ext_vector_type(4)));
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef float float16_t __attribute__((ext_vector_type(16)));

#define WAVE 64

// ---- error plumbing (host) -------------------------------------------------------------------
void pclip_set_error(const char* fmt, ...);
int pclip_check_launch(const char* what);
#define PCLIP_REQUIRE(cond, ...)              \
    do {                                      \
        if (!(cond)) {                        \
            pclip_set_error(__VA_ARGS__);     \
            return PCLIP_E_INVALID;           \
        }                                     \
    } while (0)

// ---- device helpers ----------------------------------------------------------------------------
// round fp32 -> fp16 -> fp32 (the r16() of SURVEY Appendix A; RNE like torch's .half())
__device__ __forceinline__ float r16(float x) { return (float)(half_t)x; }

// Value of lane (id ^ OFF), OFF in {1, 2, 4, 8, 16, 32}, on the VALU: DPP (quad_perm, row_ror:8, row_shl:4 / row_shr:4 under bank masks) and the gfx950 row / half swaps
// (v_permlane16_swap, v_permlane32_swap) — __shfl_xor compiles to ds_bpermute_b32, an LDS-pipe round trip of ~100 cycles per butterfly level, which IS the run time of
// the latency-bound kernels (one wave per row: prototype build, small-N classification, row norms).  Same partner lanes, so the reductions below keep their bits.
template <int OFF>
__device__ __forceinline__ int lane_xor_i(int v) {
    static_assert(OFF == 1 || OFF == 2 || OFF == 4 || OFF == 8 || OFF == 16 || OFF == 32, "lane_xor: power-of-two distance inside a wave");
#if defined(__HIP_DEVICE_COMPILE__)
    if constexpr (OFF == 32) {
        const auto r = __builtin_amdgcn_permlane32_swap((unsigned)v, (unsigned)v, false, false);     // {own, partner} below lane 32, {partner, own} above
        return (int)(r[0] ^ r[1] ^ (unsigned)v);
    } else if constexpr (OFF == 16) {
        const auto r = __builtin_amdgcn_permlane16_swap((unsigned)v, (unsigned)v, false, false);     // the same per pair of 16-lane rows
        return (int)(r[0] ^ r[1] ^ (unsigned)v);
    } else if constexpr (OFF == 8) {
        return __builtin_amdgcn_update_dpp(v, v, 0x128, 0xF, 0xF, false);                             // row_ror:8
    } else if constexpr (OFF == 4) {
        const int t = __builtin_amdgcn_update_dpp(v, v, 0x104, 0xF, 0x5, false);                      // row_shl:4 into the banks (4-lane groups) 0 and 2: lane i <- i + 4
        return __builtin_amdgcn_update_dpp(t, v, 0x114, 0xF, 0xA, false);                             // row_shr:4 into the banks 1 and 3:                lane i <- i - 4
    } else if constexpr (OFF == 2) {
        return __builtin_amdgcn_update_dpp(v, v, 0x4E, 0xF, 0xF, false);                              // quad_perm [2, 3, 0, 1]
    } else {
        return __builtin_amdgcn_update_dpp(v, v, 0xB1, 0xF, 0xF, false);                              // quad_perm [1, 0, 3, 2]
    }
#else
    return v;
#endif
}
template <int OFF>
__device__ __forceinline__ float lane_xor(float v) { return __builtin_bit_cast(float, lane_xor_i<OFF>(__builtin_bit_cast(int, v))); }

#define PCLIP_BUTTERFLY(STEP) do { STEP(32); STEP(16); STEP(8); STEP(4); STEP(2); STEP(1); } while (0)
__device__ __forceinline__ float wave_sum(float v) {
#define PCLIP_STEP(OFF) v += lane_xor<OFF>(v)
    PCLIP_BUTTERFLY(PCLIP_STEP);
#undef PCLIP_STEP
    return v;
}
__device__ __forceinline__ float wave_max(float v) {
#define PCLIP_STEP(OFF) v = fmaxf(v, lane_xor<OFF>(v))
    PCLIP_BUTTERFLY(PCLIP_STEP);
#undef PCLIP_STEP
    return v;
}
__device__ __forceinline__ float wave_min(float v) {
#define PCLIP_STEP(OFF) v = fminf(v, lane_xor<OFF>(v))
    PCLIP_BUTTERFLY(PCLIP_STEP);
#undef PCLIP_STEP
    return v;
}
// (value, index) argmax with lowest-index tie rule (torch CPU max, main.py:190)
__device__ __forceinline__ void wave_argmax(float& v, int& i) {
#define PCLIP_STEP(OFF) do { const float ov = lane_xor<OFF>(v); const int oi = lane_xor_i<OFF>(i); if (ov > v || (ov == v && oi < i)) { v = ov; i = oi; } } while (0)
    PCLIP_BUTTERFLY(PCLIP_STEP);
#undef PCLIP_STEP
}

// x / d for MANY x and ONE wave-uniform divisor d (a row's norm, a shot count): the correctly rounded fp32 quotient in 6 instead of 11 VALU instructions per element.
// hipcc expands an IEEE fp32 division into v_div_scale x2, v_rcp, two reciprocal refinements, a multiply, three quotient / residual steps, v_div_fmas, v_div_fixup;
// the scaling only acts at the ends of the exponent range and the reciprocal part depends on d alone.  prepare() does the reciprocal part once; div() repeats the
// SAME fma chain on the unscaled operands (v_div_scale returns them unchanged, v_div_fmas is a plain fma and v_div_fixup a pass-through for d in [2^-60, 2^60] and
// |x| in {0} + [2^-30, 2^30] — every use here: fp16-valued rows, sums of <= 2^10 of them), so the quotient has the same bits; the sign of a zero quotient is x's
// (v_div_fixup's rule, d > 0); any other divisor (zero, huge, NaN: degenerate rows) takes the compiler's division.  The latency-bound kernels (prototype build:
// eight divisions per row and lane) are instruction-issue-bound, one wave per SIMD (profiles/r05_c2_phases.txt).
struct RowDiv {
    float d, nd, y;
    bool fast;
    __device__ __forceinline__ explicit RowDiv(float div) : d(div), nd(-div), y(0.f), fast(div >= 0x1p-60f && div <= 0x1p60f) {
        const float y0 = __builtin_amdgcn_rcpf(div);
        y = __builtin_fmaf(__builtin_fmaf(nd, y0, 1.f), y0, y0);
    }
    __device__ __forceinline__ float div(float x) const { return fast ? div_fast(x) : x / d; }      // (wave-uniform choice; loops over many x test `fast` once themselves)
    __device__ __forceinline__ float div_fast(float x) const {
        const float q0 = x * y;
        const float q1 = __builtin_fmaf(__builtin_fmaf(nd, q0, x), y, q0);
        const float q2 = __builtin_fmaf(__builtin_fmaf(nd, q1, x), y, q1);
        return __builtin_copysignf(q2, x);
    }
};

__device__ __forceinline__ half8_t ld_half8(const half_t* p) { return *reinterpret_cast<const half8_t*>(p); }
__device__ __forceinline__ void st_half8(half_t* p, half8_t v) { *reinterpret_cast<half8_t*>(p) = v; }

// ds_read_b64_tr_b16: 64 bits per lane, 16-bit elements transposed inside each 16-lane group — lane l receives element (l & 3) of the 8-byte words
// addressed by lanes 4 jj + ((l & 15) >> 2), jj = 0 .. 3 (probed on gfx950: tools/probe/tr_probe.hip).
typedef __fp16 pclip_fp16x4_t __attribute__((__vector_size__(4 * sizeof(__fp16))));
__device__ __forceinline__ half4_t lds_tr_read4(const char* lds_addr) {
#if defined(__HIP_DEVICE_COMPILE__)
    const pclip_fp16x4_t v = __builtin_amdgcn_ds_read_tr16_b64_v4f16((__attribute__((address_space(3))) pclip_fp16x4_t*)(lds_addr));
    return __builtin_bit_cast(half4_t, v);
#else
    return half4_t{};
#endif
}

// One-shot flags for per-DEVICE state (hipFuncSetAttribute applies to the current device only): a bit per device id, so a process
// that drives several GPUs raises the LDS limit on each of them.
struct DevOnce {
    std::atomic<unsigned long long> mask{0};             // a host thread per GPU may race here: fetch_or, not a plain |=
    static unsigned long long bit() { int d = 0; (void)hipGetDevice(&d); return 1ull << (d & 63); }
    bool done() const { return (mask.load(std::memory_order_acquire) & bit()) != 0; }
    void set() { mask.fetch_or(bit(), std::memory_order_release); }
};

static inline int ceil_div(int a, int b) { return (a + b - 1) / b; }
static inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }
