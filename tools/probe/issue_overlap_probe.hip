// Does a wave's vector-memory issue (LDS-DMA pieces, global stores) slow down the MFMAs of the OTHER wave on its SIMD?  (gfx950)
// The persistent GEMM's tile time is the SUM of its matrix-pipe time and its vector-memory issue time in every loop structure built so far
// (DESIGN section 5, round 4); this probe takes the barriers and the data dependences away: one 512-thread workgroup per CU, waves 0 - 3 (one per
// SIMD) issue nothing but independent v_mfma_f32_16x16x32_f16 and time themselves with s_memtime, waves 4 - 7 (their SIMD partners) do one of
//   0 nothing   1 LDS-DMA pieces (buffer-less global_load_lds, 1 KB each) from an L2-resident region   2 the same from a 512 MB region (HBM / MALL)
//   3 16-byte global stores (1 KB per wave instruction)   4 ds_read_b128 fragment reads   5 MFMAs too (the pipe is shared: the 2x reference)
// in the GEMM's ratio (8 pieces / 16 stores / 24 reads per 64 MFMAs of the partner), and mode 6 puts the pieces into the MFMA waves' OWN stream
// (one piece per 8 MFMAs, no partner).  Output: cycles per MFMA of the timing waves.   Build: tools/probe/build.sh;  run: ./issue_overlap_probe
#include <hip/hip_runtime.h>
#include <stdio.h>
#include <stdlib.h>
typedef _Float16 half8_t __attribute__((ext_vector_type(8)));
typedef float float4_t __attribute__((ext_vector_type(4)));
typedef unsigned uint4_t __attribute__((ext_vector_type(4)));
typedef __attribute__((address_space(3))) void* lds_ptr_t;
typedef const __attribute__((address_space(1))) void* gbl_ptr_t;

template <int MODE>
__global__ __launch_bounds__(512, 2) void probe(const char* __restrict__ src, size_t region, char* __restrict__ dst, unsigned long long* __restrict__ cyc, float* __restrict__ sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    half8_t a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    for (int i = 0; i < 8; ++i) { a[i] += (_Float16)(lane & 3); b[i] -= (_Float16)(lane & 1); }
    float4_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = float4_t{0.f, 0.f, 0.f, 0.f};
    const unsigned mask = (unsigned)(region - 1), cu_off = (unsigned)(((size_t)blockIdx.x * 4 + (wave & 3)) * (region / 1024)) & mask;
    float keep = 0.f;
    if (wave < 4) {
        unsigned long long t0, t1;
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t0)::"memory");
        for (int it = 0; it < iters; ++it) {
#pragma unroll
            for (int g = 0; g < 8; ++g) {                              // 64 MFMAs: eight independent accumulators, eight rounds
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
                if (MODE == 6) {
                    const unsigned off = (cu_off + ((unsigned)(it * 8 + g) * 64 + lane) * 16) & mask;
                    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + off), (lds_ptr_t)(smem + (wave * 8 + g) * 1024), 16, 0, 0);
                }
            }
            if (MODE == 6) __builtin_amdgcn_s_waitcnt((8 & 15) | (7 << 4) | (15 << 8));          // vmcnt(8): one K-tile's worth in flight
        }
        asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(t1)::"memory");
        if (lane == 0) cyc[blockIdx.x * 4 + wave] = t1 - t0;
    } else {
        const int w4 = wave - 4;
        for (int it = 0; it < iters; ++it) {
            if (MODE == 1 || MODE == 2) {
#pragma unroll
                for (int g = 0; g < 8; ++g) {
                    const unsigned off = (cu_off + ((unsigned)(it * 8 + g) * 64 + lane) * 16) & mask;
                    __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + off), (lds_ptr_t)(smem + (w4 * 8 + g) * 1024), 16, 0, 0);
                }
                __builtin_amdgcn_s_waitcnt((8 & 15) | (7 << 4) | (15 << 8));
            } else if (MODE == 3) {
#pragma unroll
                for (int g = 0; g < 16; ++g) {
                    const unsigned off = (cu_off + ((unsigned)(it * 16 + g) * 64 + lane) * 16) & mask;
                    __builtin_nontemporal_store(uint4_t{(unsigned)it, (unsigned)g, (unsigned)lane, 0u}, reinterpret_cast<uint4_t*>(dst + off));
                }
            } else if (MODE == 4) {
#pragma unroll
                for (int g = 0; g < 24; ++g) {
                    const half8_t v = *reinterpret_cast<const half8_t*>(smem + ((w4 * 24 + g) * 1024 + lane * 16) % (96 * 1024));
                    keep += (float)v[0];
                }
            } else if (MODE == 5) {
#pragma unroll
                for (int g = 0; g < 8; ++g)
#pragma unroll
                    for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
            }
        }
    }
    float s = keep;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    sink[blockIdx.x * 512 + tid] = s;
}

template <int MODE>
void run(const char* name, const char* src, size_t region, char* dst, unsigned long long* cyc, float* sink) {
    const int iters = 2000, grid = 256;
    hipFuncSetAttribute((const void*)probe<MODE>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    probe<MODE><<<grid, 512, 128 * 1024>>>(src, region, dst, cyc, sink, 50);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    probe<MODE><<<grid, 512, 128 * 1024>>>(src, region, dst, cyc, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double sum = 0, mx = 0;
    for (int i = 0; i < 1024; ++i) { sum += (double)h[i]; if ((double)h[i] > mx) mx = (double)h[i]; }
    const double per = sum / 1024 / ((double)iters * 64);
    printf("%-58s %7.2f cycles per MFMA of the timing waves (slowest wave %7.2f); kernel %.3f ms = %.2f ns per MFMA\n", name, per, mx / ((double)iters * 64), ms,
           ms * 1e6 / ((double)iters * 64));
}


// ---- symmetric form: ALL eight waves run the GEMM's per-K-tile mix (64 MFMAs + 8 one-kilobyte LDS-DMA pieces per wave) with no data dependence between
// the two; what changes is WHERE the pieces are issued and how many workgroup barriers an iteration has.  Kernel time per iteration against the
// MFMA-only loop (2 waves per SIMD x 64 MFMAs x 16 cycles = 2048 pipe cycles) says whether the hardware overlaps the two when nothing forces a wait.
//   PLACE 0: no pieces   1: burst of 8 at the top of the iteration   2: one piece per 8 MFMAs   3: roles — waves 0 - 3 burst at the top, waves 4 - 7 after 32 MFMAs
//   NBAR: s_barrier per iteration (0, 1 at the top, 3 = top + after 32 + after 48 MFMAs, the staggered-refill loop's places)
template <int PLACE, int NBAR, int INFLIGHT>
__global__ __launch_bounds__(512, 2) void sym_probe(const char* __restrict__ src, size_t region, float* __restrict__ sink, int iters, unsigned long long* __restrict__ cyc) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    half8_t a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    for (int i = 0; i < 8; ++i) { a[i] += (_Float16)(lane & 3); b[i] -= (_Float16)(lane & 1); }
    float4_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = float4_t{0.f, 0.f, 0.f, 0.f};
    const unsigned mask = (unsigned)(region - 1), cu_off = (unsigned)(((size_t)blockIdx.x * 8 + wave) * (region / 2048)) & mask;
    auto piece = [&](int it, int g) {
        const unsigned off = (cu_off + ((unsigned)(it * 8 + g) * 64 + lane) * 16) & mask;
        __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + off), (lds_ptr_t)(smem + (wave * 8 + g) * 1024), 16, 0, 0);
    };
    unsigned long long ts0, ts1;
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts0)::"memory");
    for (int it = 0; it < iters; ++it) {
        if (PLACE != 0) __builtin_amdgcn_s_waitcnt((INFLIGHT & 15) | (7 << 4) | (15 << 8) | ((INFLIGHT >> 4) << 14));
        if (NBAR >= 1) __builtin_amdgcn_s_barrier();
        if (PLACE == 1 || (PLACE == 3 && wave < 4)) {
#pragma unroll
            for (int g = 0; g < 8; ++g) piece(it, g);
        }
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            if (NBAR == 3 && (g == 4 || g == 6)) __builtin_amdgcn_s_barrier();
            if (PLACE == 3 && wave >= 4 && g == 4) {
#pragma unroll
                for (int k = 0; k < 8; ++k) piece(it, k);
            }
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
            if (PLACE == 2) piece(it, g);
        }
    }
    asm volatile("s_memtime %0\n\ts_waitcnt lgkmcnt(0)" : "=s"(ts1)::"memory");
    if (lane == 0 && blockIdx.x < 128) cyc[blockIdx.x * 8 + wave] = ts1 - ts0;
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    sink[blockIdx.x * 512 + tid] = s;
}
template <int PLACE, int NBAR, int INFLIGHT>
void run_sym(const char* name, const char* src, size_t region, float* sink, int grid = 256) {
    const int iters = 2000;
    hipFuncSetAttribute((const void*)sym_probe<PLACE, NBAR, INFLIGHT>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    static unsigned long long* cyc = nullptr;
    if (!cyc) hipMalloc(&cyc, 1024 * 8);
    sym_probe<PLACE, NBAR, INFLIGHT><<<grid, 512, 128 * 1024>>>(src, region, sink, 50, cyc);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    sym_probe<PLACE, NBAR, INFLIGHT><<<grid, 512, 128 * 1024>>>(src, region, sink, iters, cyc);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    unsigned long long h[1024];
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    const int nw = (grid < 128 ? grid : 128) * 8;
    double sum = 0;
    for (int i = 0; i < nw; ++i) sum += (double)h[i];
    const double cpi = sum / nw / iters;
    printf("%-96s %8.1f ns = %7.0f s_memtime cycles per iteration (=> %.2f GHz; MFMA alone: 2048 pipe cycles)\n", name, ms * 1e6 / iters, cpi, cpi / (ms * 1e6 / iters));
}

// ---- role-asymmetric placement: three barriers per iteration as in the staggered-refill loop (top | B half free | A half free), but the role that does NOT
// issue in a phase gets 32 of its MFMAs there and the issuing role only 16:   waves 0 - 3:  T | 16 | b2 | burst, 16 | b3 | 32      waves 4 - 7:  T | 16 | b2 | 32 | b3 | burst, 16
// (ASYM = 0: the loop's present form — both roles 32 | b2 | burst / -, 16 | b3 | - / burst, 16)
template <int ASYM, int INFLIGHT>
__global__ __launch_bounds__(512, 2) void role_probe(const char* __restrict__ src, size_t region, float* __restrict__ sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    half8_t a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    for (int i = 0; i < 8; ++i) { a[i] += (_Float16)(lane & 3); b[i] -= (_Float16)(lane & 1); }
    float4_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = float4_t{0.f, 0.f, 0.f, 0.f};
    const unsigned mask = (unsigned)(region - 1), cu_off = (unsigned)(((size_t)blockIdx.x * 8 + wave) * (region / 2048)) & mask;
    auto burst = [&](int it) {
#pragma unroll
        for (int g = 0; g < 8; ++g) {
            const unsigned off = (cu_off + ((unsigned)(it * 8 + g) * 64 + lane) * 16) & mask;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + off), (lds_ptr_t)(smem + (wave * 8 + g) * 1024), 16, 0, 0);
        }
    };
    auto mf = [&](int n16) {                                   // n16 x 16 independent MFMAs
        for (int k = 0; k < n16; ++k)
#pragma unroll
            for (int r = 0; r < 2; ++r)
#pragma unroll
                for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    };
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_waitcnt((INFLIGHT & 15) | (7 << 4) | (15 << 8) | ((INFLIGHT >> 4) << 14));
        __builtin_amdgcn_s_barrier();
        if (ASYM) {
            mf(1);
            __builtin_amdgcn_s_barrier();
            if (wave < 4) { burst(it); mf(1); } else mf(2);
            __builtin_amdgcn_s_barrier();
            if (wave < 4) mf(2); else { burst(it); mf(1); }
        } else {
            mf(2);
            __builtin_amdgcn_s_barrier();
            if (wave < 4) burst(it);
            mf(1);
            __builtin_amdgcn_s_barrier();
            if (wave >= 4) burst(it);
            mf(1);
        }
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    sink[blockIdx.x * 512 + tid] = s;
}
template <int ASYM, int INFLIGHT>
void run_role(const char* name, const char* src, size_t region, float* sink) {
    const int iters = 2000, grid = 256;
    hipFuncSetAttribute((const void*)role_probe<ASYM, INFLIGHT>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    role_probe<ASYM, INFLIGHT><<<grid, 512, 128 * 1024>>>(src, region, sink, 50);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    role_probe<ASYM, INFLIGHT><<<grid, 512, 128 * 1024>>>(src, region, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-110s %8.1f ns per iteration\n", name, ms * 1e6 / iters);
}

// ---- what a THREE-stage 256 x 128 tile would run per K-tile: 32 MFMAs + 6 pieces per wave (48 KB per CU), ONE barrier, K-tile t + 2 requested at the top of t
// (waves 0 - 3) / after 16 MFMAs (waves 4 - 7); two of these = the FLOPs of one 256 x 256 K-tile (R0 above)
template <int INFLIGHT>
__global__ __launch_bounds__(512, 2) void w3_probe(const char* __restrict__ src, size_t region, float* __restrict__ sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    half8_t a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    for (int i = 0; i < 8; ++i) { a[i] += (_Float16)(lane & 3); b[i] -= (_Float16)(lane & 1); }
    float4_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = float4_t{0.f, 0.f, 0.f, 0.f};
    const unsigned mask = (unsigned)(region - 1), cu_off = (unsigned)(((size_t)blockIdx.x * 8 + wave) * (region / 2048)) & mask;
    auto burst = [&](int it) {
#pragma unroll
        for (int g = 0; g < 6; ++g) {
            const unsigned off = (cu_off + ((unsigned)(it * 6 + g) * 64 + lane) * 16) & mask;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + off), (lds_ptr_t)(smem + (wave * 6 + g) * 1024), 16, 0, 0);
        }
    };
    auto mf16 = [&]() {
#pragma unroll
        for (int r = 0; r < 2; ++r)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    };
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_waitcnt((INFLIGHT & 15) | (7 << 4) | (15 << 8) | ((INFLIGHT >> 4) << 14));
        __builtin_amdgcn_s_barrier();
        if (wave < 4) burst(it);
        mf16();
        if (wave >= 4) burst(it);
        mf16();
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    sink[blockIdx.x * 512 + tid] = s;
}
template <int INFLIGHT>
void run_w3(const char* name, const char* src, size_t region, float* sink) {
    const int iters = 4000, grid = 256;
    hipFuncSetAttribute((const void*)w3_probe<INFLIGHT>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    w3_probe<INFLIGHT><<<grid, 512, 128 * 1024>>>(src, region, sink, 50);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    w3_probe<INFLIGHT><<<grid, 512, 128 * 1024>>>(src, region, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-110s %8.1f ns per iteration, %8.1f ns per 256 x 256 K-tile equivalent\n", name, ms * 1e6 / iters, 2 * ms * 1e6 / iters);
}

// ---- "B direct": what a K-loop would issue whose B fragments go from L2 straight to registers (per wave: its own 32 columns, 4 global_load_dwordx4 per K-tile, no
// duplicates in a 1 x 8 wave arrangement) and whose A tile alone goes through LDS (4 pieces per wave, 32 KB per K-tile: FOUR stages fit the 128 KB the two 64 KB stages
// take now, i.e. three K-tiles of lead and ONE barrier per K-tile) — at the price of every wave reading the whole A tile: 32 ds_read_b128 per wave and K-tile instead of 24.
// FORM 0: the present loop's pattern WITH its 24 fragment reads (3 barriers, roles, 8 pieces per wave); FORM 1: B direct (1 barrier, 4 pieces + 4 register loads + 32 reads).
template <int FORM, int INFLIGHT, int BLOAD = 1>
__global__ __launch_bounds__(512, 2) void bd_probe(const char* __restrict__ src, size_t region, const char* __restrict__ wsrc, float* __restrict__ sink, int iters) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const int tid = threadIdx.x, lane = tid & 63, wave = __builtin_amdgcn_readfirstlane(tid >> 6);
    half8_t a = {1, 2, 3, 4, 5, 6, 7, 8}, b = {8, 7, 6, 5, 4, 3, 2, 1};
    for (int i = 0; i < 8; ++i) { a[i] += (_Float16)(lane & 3); b[i] -= (_Float16)(lane & 1); }
    float4_t acc[8];
    for (int i = 0; i < 8; ++i) acc[i] = float4_t{0.f, 0.f, 0.f, 0.f};
    const unsigned mask = (unsigned)(region - 1), cu_off = (unsigned)(((size_t)blockIdx.x * 8 + wave) * (region / 2048)) & mask;
    constexpr int NP = FORM ? 4 : 8;
    auto burst = [&](int it) {
#pragma unroll
        for (int g = 0; g < NP; ++g) {
            const unsigned off = (cu_off + ((unsigned)(it * NP + g) * 64 + lane) * 16) & mask;
            __builtin_amdgcn_global_load_lds((gbl_ptr_t)(src + off), (lds_ptr_t)(smem + (wave * 8 + g) * 1024), 16, 0, 0);
        }
    };
    // B fragments of a wave: 16 rows of W (row stride 1536 B) x 64 B per instruction, two column blocks x two k-steps; the weights are a 4 MB L2-resident panel set
    half8_t bf[4];
    auto bload = [&](int it) {
#pragma unroll
        for (int g = 0; g < 4; ++g) {
            // (the workgroup's weight panel: one of 12 column tiles of a 3072 x 768 weight, as neighbouring workgroups of a c_fc launch read them)
            const char* ptr = wsrc + ((size_t)((blockIdx.x % 12) * 256 + wave * 32 + (g & 1) * 16 + (lane & 15)) * 1536 + (lane >> 4) * 16 + (size_t)((it * 2 + (g >> 1)) % 24) * 64);
            if (BLOAD) asm volatile("global_load_dwordx4 %0, %1, off" : "=v"(bf[g]) : "v"(ptr));
        }
    };
    half8_t fr[4];
    const unsigned lbase = (unsigned)(size_t)smem;    // (generic -> the low 32 bits are the LDS offset on this target)
    auto reads = [&](int n, int salt) {
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (k < n) {
                const unsigned addr = lbase + (((wave * 7 + salt * 4 + k) * 1024 + lane * 16) & (128 * 1024 - 1));
                asm volatile("ds_read_b128 %0, %1" : "=v"(fr[k]) : "v"(addr));
            }
    };
    auto mf = [&](int n8) {
        for (int k = 0; k < n8; ++k)
#pragma unroll
            for (int i = 0; i < 8; ++i) acc[i] = __builtin_amdgcn_mfma_f32_16x16x32_f16(a, b, acc[i], 0, 0, 0);
    };
    for (int it = 0; it < iters; ++it) {
        __builtin_amdgcn_s_waitcnt((INFLIGHT & 15) | (7 << 4) | (15 << 8) | ((INFLIGHT >> 4) << 14));
        __builtin_amdgcn_s_barrier();
        if (FORM == 1) {
            bload(it);
            burst(it);
#pragma unroll
            for (int g = 0; g < 8; ++g) { reads(4, g); mf(1); }
        } else {
#pragma unroll
            for (int g = 0; g < 4; ++g) { reads(3, g); mf(1); }
            __builtin_amdgcn_s_barrier();
            if (wave < 4) burst(it);
            reads(3, 4); mf(1); reads(3, 5); mf(1);
            __builtin_amdgcn_s_barrier();
            if (wave >= 4) burst(it);
            reads(3, 6); mf(1); reads(3, 7); mf(1);
        }
        asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory");
    }
    float s = 0.f;
    for (int i = 0; i < 8; ++i) s += acc[i][0] + acc[i][3];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    for (int g = 0; g < 4; ++g) s += (float)bf[g][0] + (float)fr[g][0];
    sink[blockIdx.x * 512 + tid] = s;
}
template <int FORM, int INFLIGHT, int BLOAD = 1>
void run_bd(const char* name, const char* src, size_t region, float* sink) {
    const int iters = 2000, grid = 256;
    hipFuncSetAttribute((const void*)bd_probe<FORM, INFLIGHT, BLOAD>, hipFuncAttributeMaxDynamicSharedMemorySize, 128 * 1024);
    bd_probe<FORM, INFLIGHT, BLOAD><<<grid, 512, 128 * 1024>>>(src, region, src, sink, 50);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0);
    bd_probe<FORM, INFLIGHT, BLOAD><<<grid, 512, 128 * 1024>>>(src, region, src, sink, iters);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-118s %8.1f ns per iteration\n", name, ms * 1e6 / iters);
}

int main(int argc, char** argv) {
    if (argc > 1 && argv[1][0] == 'b') {
        const size_t big = (size_t)512 << 20, small = (size_t)4 << 20;
        char* src; float* sink;
        hipMalloc(&src, big); hipMalloc(&sink, 256 * 512 * 4);
        hipMemset(src, 1, big);
        setvbuf(stdout, NULL, _IONBF, 0);
        printf("---- the present loop's issue pattern with its fragment reads (F0) against a B-direct loop's (F1); 64 MFMAs per wave and iteration either way\n");
        run_bd<0, 8>("F0  present: 3 barriers, roles, 8 pieces + 24 reads per wave; 4 MB source, one K-tile in flight", src, small, sink);
        run_bd<0, 8>("F0  present, 64 MB source (Infinity Cache)", src, (size_t)64 << 20, sink);
        run_bd<1, 8, 0>("F1a B direct WITHOUT its register loads: 1 barrier, 4 pieces + 32 reads per wave; 4 MB source, one K-tile in flight", src, small, sink);
        run_bd<1, 24, 0>("F1a 4 MB, three in flight", src, small, sink);
        run_bd<1, 8, 0>("F1a 64 MB (Infinity Cache), one in flight", src, (size_t)64 << 20, sink);
        run_bd<1, 24, 0>("F1a 64 MB, three in flight", src, (size_t)64 << 20, sink);
        run_bd<1, 8>("F1  B direct: + 4 register loads per wave (16 rows x 64 B each, from a 4 MB weight set); 4 MB source, one K-tile in flight", src, small, sink);
        run_bd<1, 16>("F1  4 MB, two K-tiles in flight", src, small, sink);
        run_bd<1, 24>("F1  4 MB, three K-tiles in flight", src, small, sink);
        return 0;
    }

    const size_t big = (size_t)512 << 20, small = (size_t)4 << 20;
    char *src, *dst; unsigned long long* cyc; float* sink;
    hipMalloc(&src, big); hipMalloc(&dst, big); hipMalloc(&cyc, 1024 * 8); hipMalloc(&sink, 256 * 512 * 4);
    hipMemset(src, 1, big); hipMemset(dst, 0, big);
    run<0>("0 partner idle", src, small, dst, cyc, sink);
    run<5>("5 partner: MFMAs too (shared pipe)", src, small, dst, cyc, sink);
    run<1>("1 partner: 8 LDS-DMA pieces per 64 MFMAs, L2-resident source", src, small, dst, cyc, sink);
    run<2>("2 partner: 8 LDS-DMA pieces per 64 MFMAs, 512 MB source", src, big, dst, cyc, sink);
    run<3>("3 partner: 16 nt stores (1 KB) per 64 MFMAs, 512 MB target", src, big, dst, cyc, sink);
    run<4>("4 partner: 24 ds_read_b128 per 64 MFMAs", src, small, dst, cyc, sink);
    run<6>("6 no partner work; the MFMA waves issue 1 piece per 8 MFMAs (512 MB source, one iteration in flight: latency-bound)", src, big, dst, cyc, sink);
    printf("---- symmetric: every wave 64 MFMAs + 8 pieces per iteration\n");
    run_sym<0, 0, 0>("S0  MFMAs only", src, small, sink);
    run_sym<0, 3, 0>("S0b MFMAs only, 3 barriers per iteration", src, small, sink);
    run_sym<1, 0, 8>("S1  burst of 8 at the top, L2-resident source, one iteration in flight, no barrier", src, small, sink);
    run_sym<2, 0, 8>("S2  one piece per 8 MFMAs, L2-resident, no barrier", src, small, sink);
    run_sym<3, 0, 8>("S3  roles (waves 0-3 at the top, 4-7 after 32 MFMAs), L2-resident, no barrier", src, small, sink);
    run_sym<3, 1, 8>("S4  roles, 1 barrier per iteration", src, small, sink);
    run_sym<3, 3, 8>("S5  roles, 3 barriers per iteration (staggered-refill places)", src, small, sink);
    run_sym<1, 0, 16>("S6  burst at the top, 512 MB source, two iterations in flight, no barrier", src, big, sink);
    run_sym<3, 3, 16>("S7  roles, 3 barriers, 512 MB source, two iterations in flight", src, big, sink);
    run_sym<2, 0, 16>("S8  one piece per 8 MFMAs, 512 MB source, two iterations in flight, no barrier", src, big, sink);
    printf("---- three barriers per iteration, role-symmetric (present loop) vs role-asymmetric MFMA placement\n");
    run_role<0, 8>("R0  present form: 32 | b2 | X burst, 16 | b3 | Y burst, 16            (L2-resident 4 MB)", src, small, sink);
    run_role<1, 8>("R1  asymmetric:   16 | b2 | X burst+16 / Y 32 | b3 | X 32 / Y burst+16  (L2-resident 4 MB)", src, small, sink);
    run_role<0, 8>("R2  present form, 64 MB source (Infinity Cache)", src, (size_t)64 << 20, sink);
    run_role<1, 8>("R3  asymmetric,   64 MB source (Infinity Cache)", src, (size_t)64 << 20, sink);
    printf("---- a three-stage 256 x 128 tile (one barrier, two K-tiles of lead): 32 MFMAs + 6 pieces per wave and iteration\n");
    run_w3<6>("W1  L2-resident 4 MB, one K-tile in flight beyond the awaited one", src, small, sink);
    run_w3<12>("W2  L2-resident 4 MB, two in flight", src, small, sink);
    run_w3<6>("W3  64 MB source (Infinity Cache), one in flight", src, (size_t)64 << 20, sink);
    run_w3<12>("W4  64 MB source (Infinity Cache), two in flight", src, (size_t)64 << 20, sink);
    printf("---- delivery rate of the LDS-DMA path by bytes in flight / source size / active CUs (burst at the top, no barrier; 64 KB per CU and iteration)\n");
    run_sym<1, 0, 16>("D1  L2-resident 4 MB, TWO iterations in flight (128 KB per CU)", src, small, sink);
    run_sym<1, 0, 24>("D2  L2-resident 4 MB, THREE iterations in flight", src, small, sink);
    run_sym<1, 0, 8>("D3  1 MB source, one iteration in flight", src, (size_t)1 << 20, sink);
    run_sym<1, 0, 16>("D4  1 MB source, two iterations in flight", src, (size_t)1 << 20, sink);
    run_sym<1, 0, 8>("D5  4 MB source, one iteration in flight, 64 workgroups (8 per XCD)", src, small, sink, 64);
    run_sym<1, 0, 8>("D6  4 MB source, one iteration in flight, 8 workgroups (1 per XCD)", src, small, sink, 8);
    run_sym<1, 0, 8>("D7  64 MB source (Infinity Cache), one iteration in flight", src, (size_t)64 << 20, sink);
    run_sym<1, 0, 16>("D8  64 MB source (Infinity Cache), two iterations in flight", src, (size_t)64 << 20, sink);
    return 0;
}
