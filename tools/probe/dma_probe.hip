// Probe: how fast can every CU of an MI355X pull the operand tiles of a persistent 256x256 fp16 GEMM into LDS by LDS-DMA, as a
// function of the ring depth / stage size?  Same addressing as pclip's linear_fast_kernel (tile (tm, tn): 256 rows of A[M,K] and
// 256 rows of B[N,K], K-tile by K-tile), no MFMAs, no fragment reads.   Build: hipcc -O3 --offload-arch=gfx950 dma_probe.hip -o dma_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
typedef _Float16 half_t;
typedef __attribute__((address_space(3))) void* lds_ptr_t;

template <int N>
__device__ __forceinline__ void wait_vm() { __builtin_amdgcn_s_waitcnt((N & 15) | (7 << 4) | (15 << 8) | ((N >> 4) << 14)); }
__device__ __forceinline__ void lds_barrier() { asm volatile("s_waitcnt lgkmcnt(0)" ::: "memory"); __builtin_amdgcn_s_barrier(); asm volatile("" ::: "memory"); }

// BKS = halves of K per stage (64 or 32), NS = ring slots, stage = 512 rows x BKS halves.  Pieces: a wave instruction moves 1 KB =
// (1024 / (BKS*2)) rows x BKS*2 bytes.
template <int BKS, int NS>
__global__ __launch_bounds__(512, 2) void probe(const half_t* __restrict__ A, const half_t* __restrict__ B, int M, int N, int K, int tiles_n,
                                                int ntiles, int* sink, int mode) {
#if defined(__HIP_DEVICE_COMPILE__)      // device-only builtins: the host pass must see an empty body or it drops the stub
    extern __shared__ __attribute__((aligned(16))) char smem[];
    const bool pf = mode & 8;
    mode &= 7;
    constexpr int ROWB = BKS * 2, RPP = 1024 / ROWB, STAGE = 512 * ROWB, PER = 512 / RPP / 8;   // pieces per wave per stage
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = gridDim.x, nt = K / BKS;
    int cnt = 0;
    // mode 0: tile = block id (consecutive tiles on different XCDs); 1: pclip's xcd_remap (32 consecutive tiles of a round on one
    // XCD); 2: every workgroup reads tile 0 (pure L2 hits); 3: xcd_remap + each XCD's 32 tiles form a 4 (rows) x 8 (columns)
    // super-tile when tiles_n % 8 == 0, i.e. an XCD keeps 4 A panels and 8 B panels hot; 4: like 1, but a round's XCD chunk is
    // column-major (same tn, consecutive tm)
    const int q = G >> 3, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    for (int base = 0; base < ntiles; base += G) {
        int tile = base + blockIdx.x;
        if (mode == 1 || mode == 3 || mode == 4) tile = base + xcd * q + idx;
        if (mode == 2) tile = 0;
        if (tile >= ntiles) break;
        int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        if (mode == 3 && tiles_n % 8 == 0) {
            const int per_row = tiles_n / 8;                   // super-tiles per band of 4 tile rows
            const int st = tile / 32, in = tile % 32;          // super-tile index, position inside
            tm = (st / per_row) * 4 + in / 8;
            tn = (st % per_row) * 8 + in % 8;
            if (tm * 256 >= M) { tm = tile / tiles_n; tn = tile - tm * tiles_n; }
        }
        if (mode == 4) { const int tiles_m = (M + 255) / 256; tn = tile / tiles_m; tm = tile - tn * tiles_m; }
        // this wave's pieces: PER pieces, piece j covers rows (wave*PER + j)*RPP .. of the 512-row stage (rows < 256: A, else B)
        __amdgpu_buffer_rsrc_t rs[2];
        rs[0] = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)tm * 256 * K), 0, 0x7fffffff, 0x00020000);
        rs[1] = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)tn * 256 * K), 0, 0x7fffffff, 0x00020000);
        int voff[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) {
            const int r = ((wave * PER + j) * RPP + lane / (ROWB / 16)) & 255;
            int rr = r;
            if (tm * 256 + rr >= M && (wave * PER + j) * RPP < 256) rr = M - 1 - tm * 256;
            voff[j] = (rr * K + (lane % (ROWB / 16)) * 8) * 2;
        }
        auto stage = [&](int t, int slot) {
#pragma unroll
            for (int j = 0; j < PER; ++j) {
                const bool isb = (wave * PER + j) * RPP >= 256;
                __builtin_amdgcn_raw_ptr_buffer_load_lds(isb ? rs[1] : rs[0], (lds_ptr_t)(smem + slot * STAGE + (wave * PER + j) * 1024), 16, voff[j], t * ROWB, 0, 0);
            }
        };
        // pf: each lane touches one 128-byte line (row = wave*64 + lane of the 512-row K-tile) of K-tile t + NS with a 4-byte LDS-DMA
        const int prow = wave * 64 + lane;
        const int pvoff = ((prow & 255) * K) * 2;
        const bool pb = prow >= 256;
        auto touch = [&](int t) {
            __builtin_amdgcn_raw_ptr_buffer_load_lds(pb ? rs[1] : rs[0], (lds_ptr_t)(smem + NS * STAGE), 4, pvoff, (t < nt ? t : nt - 1) * ROWB, 0, 0);
        };
        for (int t = 0; t < NS - 1 && t < nt; ++t) stage(t, t);
        int slot = 0, fill = NS - 1;
        for (int t = 0; t < nt; ++t) {
            const int ahead = nt - 1 - t;
            if (pf && NS == 2 && t > 0) wait_vm<1>();
            else if (NS >= 5 && ahead >= 3) wait_vm<3 * PER>();
            else if (NS >= 4 && ahead >= 2) wait_vm<2 * PER>();
            else if (NS >= 3 && ahead >= 1) wait_vm<PER>();
            else wait_vm<0>();
            lds_barrier();
            if (t + NS - 1 < nt) { stage(t + NS - 1, fill); if (pf && NS == 2) touch(t + 2); }
            fill = slot;
            cnt += smem[slot * STAGE + threadIdx.x * 16];
            slot = slot + 1 == NS ? 0 : slot + 1;
        }
        lds_barrier();
    }
    if (cnt == 12345678) sink[0] = cnt;
#endif
}

template <int BKS, int NS>
void run(const half_t* A, const half_t* B, int M, int N, int K, int* sink, const char* name, int mode) {
    const int tiles_m = (M + 255) / 256, tiles_n = N / 256, ntiles = tiles_m * tiles_n;
    const int lds = NS * 512 * BKS * 2 + 256;
    hipFuncSetAttribute((const void*)probe<BKS, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0);
        probe<BKS, NS><<<256, 512, lds>>>(A, B, M, N, K, tiles_n, ntiles, sink, mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r && ms < best) best = ms;
    }
    const double bytes = (double)ntiles * (K / 64) * 65536.0;
    printf("mode %d %-16s M=%d N=%d K=%d: %8.1f us  %6.2f TB/s aggregate  %5.1f GB/s per CU  (%.2f us per 64-wide K-tile)\n", mode, name, M, N, K, best * 1e3,
           bytes / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / 256 / 1e9, best * 1e3 / ((double)ntiles / 256 * (K / 64)));
}

// Variant: full 128-byte row segments (BK = 64) but only SR of the K-tile's 512 rows per stage (SR = 256: A part / B part alternate;
// SR = 128: the guide's half-tiles), NS ring slots, NS - 1 stages in flight.
template <int SR, int NS>
__global__ __launch_bounds__(512, 2) void probe_rows(const half_t* __restrict__ A, const half_t* __restrict__ B, int M, int N, int K, int tiles_n,
                                                     int ntiles, int* sink, int mode) {
#if defined(__HIP_DEVICE_COMPILE__)
    extern __shared__ __attribute__((aligned(16))) char smem[];
    constexpr int STAGE = SR * 128, PER = SR / 8 / 8, SPK = 512 / SR;      // pieces per wave per stage, stages per K-tile
    const int lane = threadIdx.x & 63, wave = __builtin_amdgcn_readfirstlane(threadIdx.x >> 6);
    const int G = gridDim.x, nst = (K / 64) * SPK;
    const int q = G >> 3, xcd = blockIdx.x & 7, idx = blockIdx.x >> 3;
    int cnt = 0;
    for (int base = 0; base < ntiles; base += G) {
        int tile = base + blockIdx.x;
        if (mode == 1) tile = base + xcd * q + idx;
        if (mode == 2) tile = 0;
        if (tile >= ntiles) break;
        int tm = tile / tiles_n, tn = tile - tm * tiles_n;
        if (mode >= 5) {
            // column groups: ng = mode - 3 (mode 5: 2 groups, 6: 3 - uneven XCD split not handled, 7: 4 groups); XCD x works on group x % ng,
            // the XCDs of a group walk its (tm, local tn) sequence 32 tiles at a time
            const int ng = mode == 5 ? 2 : 4, cg = tiles_n / ng, per = 8 / ng;
            const int grp = xcd % ng, xi = xcd / ng;
            const int s = ((base / G) * per + xi) * q + idx;
            tm = s / cg; tn = grp * cg + s % cg;
            if (tm * 256 >= M) continue;
        }
        __amdgpu_buffer_rsrc_t rs[2];
        rs[0] = __builtin_amdgcn_make_buffer_rsrc((void*)(A + (size_t)tm * 256 * K), 0, 0x7fffffff, 0x00020000);
        rs[1] = __builtin_amdgcn_make_buffer_rsrc((void*)(B + (size_t)tn * 256 * K), 0, 0x7fffffff, 0x00020000);
        int voff[PER];
#pragma unroll
        for (int j = 0; j < PER; ++j) voff[j] = (((wave * PER + j) * 8 + (lane >> 3)) * K + (lane & 7) * 8) * 2;   // row inside the stage's SR rows
        auto stage = [&](int s, int slot) {
            const int t = s / SPK, part = s - t * SPK;                       // rows part*SR .. of the 512-row K-tile
            const int row0 = (part * SR) & 255;
            const bool isb = part * SR >= 256;
#pragma unroll
            for (int j = 0; j < PER; ++j)
                __builtin_amdgcn_raw_ptr_buffer_load_lds(isb ? rs[1] : rs[0], (lds_ptr_t)(smem + slot * STAGE + (wave * PER + j) * 1024), 16, voff[j],
                                                         t * 128 + row0 * K * 2, 0, 0);
        };
        for (int s = 0; s < NS - 1 && s < nst; ++s) stage(s, s);
        int slot = 0, fill = NS - 1;
        for (int s = 0; s < nst; ++s) {
            const int ahead = nst - 1 - s;
            if (NS >= 9 && ahead >= 7) wait_vm<7 * PER>();
            else if (NS >= 8 && ahead >= 6) wait_vm<6 * PER>();
            else if (NS >= 7 && ahead >= 5) wait_vm<5 * PER>();
            else if (NS >= 6 && ahead >= 4) wait_vm<4 * PER>();
            else if (NS >= 5 && ahead >= 3) wait_vm<3 * PER>();
            else if (NS >= 4 && ahead >= 2) wait_vm<2 * PER>();
            else if (NS >= 3 && ahead >= 1) wait_vm<PER>();
            else wait_vm<0>();
            lds_barrier();
            if (s + NS - 1 < nst) stage(s + NS - 1, fill);
            fill = slot;
            cnt += smem[slot * STAGE + threadIdx.x * 16];
            slot = slot + 1 == NS ? 0 : slot + 1;
        }
        lds_barrier();
    }
    if (cnt == 12345678) sink[0] = cnt;
#endif
}

template <int SR, int NS>
void run_rows(const half_t* A, const half_t* B, int M, int N, int K, int* sink, int mode) {
    const int tiles_m = M / 256, tiles_n = N / 256, ntiles = tiles_m * tiles_n;   // full tiles only (no row clamp in this variant)
    const int lds = NS * SR * 128;
    hipFuncSetAttribute((const void*)probe_rows<SR, NS>, hipFuncAttributeMaxDynamicSharedMemorySize, lds);
    hipEvent_t e0, e1;
    hipEventCreate(&e0); hipEventCreate(&e1);
    float best = 1e30f;
    for (int r = 0; r < 4; ++r) {
        hipEventRecord(e0);
        probe_rows<SR, NS><<<256, 512, lds>>>(A, B, M, N, K, tiles_n, ntiles, sink, mode);
        hipEventRecord(e1);
        hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        if (r && ms < best) best = ms;
    }
    const double bytes = (double)ntiles * (K / 64) * 65536.0;
    printf("mode %d rows %3d x %d slots  M=%d N=%d K=%d: %8.1f us  %6.2f TB/s aggregate  %5.1f GB/s per CU  (%.2f us per 64-wide K-tile)\n", mode, SR, NS, M, N, K,
           best * 1e3, bytes / (best * 1e-3) / 1e12, bytes / (best * 1e-3) / 256 / 1e9, best * 1e3 / ((double)ntiles / 256 * (K / 64)));
}

int main() {
    const int M = 201728, NMAX = 3072, KMAX = 3072;
    half_t *A, *B; int* sink;
    hipMalloc(&A, (size_t)M * KMAX * 2); hipMalloc(&B, (size_t)NMAX * KMAX * 2); hipMalloc(&sink, 4);
    hipMemset(A, 0, (size_t)M * KMAX * 2); hipMemset(B, 0, (size_t)NMAX * KMAX * 2);
    for (auto [n, k] : std::vector<std::pair<int, int>>{{3072, 768}, {2048, 768}, {1024,768}, {1024, 3072}}) {
        run_rows<256, 3>(A, B, M, n, k, sink, 1);
        if (n % 512 == 0) run_rows<256, 3>(A, B, M, n, k, sink, 5);
        if (n % 1024 == 0) run_rows<256, 3>(A, B, M, n, k, sink, 7);
        for (int mode = 1; mode < 1; ++mode) {
            run<64, 2>(A, B, M, n, k, sink, "BK64 x 2 slots", mode);
            run_rows<256, 2>(A, B, M, n, k, sink, mode);
            run_rows<256, 3>(A, B, M, n, k, sink, mode);
            run_rows<256, 4>(A, B, M, n, k, sink, mode);
            run_rows<256, 5>(A, B, M, n, k, sink, mode);
            run_rows<128, 4>(A, B, M, n, k, sink, mode);
            run_rows<128, 6>(A, B, M, n, k, sink, mode);
            run_rows<128, 8>(A, B, M, n, k, sink, mode);
            run_rows<128, 9>(A, B, M, n, k, sink, mode);
        }
    }
    return 0;
}
