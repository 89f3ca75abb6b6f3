// Prototype build + classification in ONE launch (reference main.py:399-405 + utils.py:225-244 + main.py:190, the test-time path
// `z_img_proto = normalise(mean_k normalise(mem))` -> `P(zq, z_img_proto, z_text_proto, alpha, beta)` -> argmax).
// At EuroSAT's size (10 classes x 16 shots, 8100 queries, D = 512) both stages are latency: a trivial kernel replays at 1.9 us, proto_build at 4.0 us and classify at
// 6.4 us (tools/c2_probe.py).  Here the first N workgroups of the grid build one prototype each (class_sum + finish_prototype of pclip_proto_dev.h: the bits of
// pclip_proto_build_f16), publish the row (write-through stores at agent scope) and count themselves in sync[0]; the other workgroups run classify_small's body
// (pclip_classify_small.h: the bits of pclip_classify_f16) — the textual waves start at once, the visual waves request their queries FIRST, wait for sync[0] == N
// and read the prototype rows with agent-coherent loads.  Workgroups are dispatched in blockIdx order and the builders wait for nobody, so the wait cannot deadlock
// whatever the residency; sync[] is left zero by the last visual wave past the wait (one pair of words per stream).
// MEASURED SLOWER than the two launches (12.8 - 16.8 us against 10.3): the hand-over across XCDs costs more than the launch it removes (profiles/r05_c2_phases.txt) —
// the entry point stays for callers that want one graph node; the host wrapper takes the two launches unless asked.
#include "pclip_proto_dev.h"
#include "pclip_classify_small.h"
#include <stdlib.h>

namespace {

template <int NCH, int NT>
__global__ __launch_bounds__(256) void proto_classify_kernel(const half_t* __restrict__ mem, int K, int per_shot_norm, half_t* proto, float* proto_sq,
                                                             const half_t* __restrict__ q, const half_t* __restrict__ zt, int Q, int N, int D,
                                                             float alpha, float oma, float beta, float* __restrict__ p, int32_t* __restrict__ argmax,
                                                             float* __restrict__ topk_p, int32_t* __restrict__ topk_i, int k, int* sync, int wt) {
    extern __shared__ __attribute__((aligned(16))) char smem[];
    if ((int)blockIdx.x < N) {
        const int n = blockIdx.x, lane = threadIdx.x & 63;
        float acc[NCH][8];
        class_sum<NCH>(mem, n * K, n * K + K, D, per_shot_norm, acc, reinterpret_cast<float*>(smem));
        if ((threadIdx.x >> 6) == 0) {
            finish_prototype<NCH>(acc, (float)K, n, D, lane, proto, nullptr, proto_sq, wt != 0);
#if defined(PCLIP_RACE_STRESS)
            if ((__builtin_readcyclecounter() >> 4) & 1) for (int i = 0; i < 8; ++i) __builtin_amdgcn_s_sleep(127);   // a late builder: consumers must really wait
#endif
            // every lane's row stores are visible at agent scope before the count: an L2 write-back (agent release) — or, wt, the stores were write-through and
            // only have to be complete
            if (wt) __builtin_amdgcn_fence(__ATOMIC_RELEASE, "workgroup"); else __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            if (lane == 0) __hip_atomic_fetch_add(sync, 1, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        }
        return;
    }
    classify_small_body<NT, true, true>(smem, (int)blockIdx.x - N, (int)gridDim.x - N, q, proto, zt, Q, N, D, alpha, oma, beta, p, argmax, topk_p, topk_i, k, sync, N, wt);
}

template <int NCH, int NT>
int launch(const void* mem, int N, int K, int D, int per_shot_norm, void* proto, float* proto_sq, const void* q, const void* zt, int Q, float alpha, float oma,
           float beta, float* p, int32_t* argmax, float* topk_p, int32_t* topk_i, int k, int* sync, int cus, hipStream_t s) {
    const size_t lds_c = classify_small_lds(NT, 4), lds_b = (size_t)4 * NCH * 512 * 4;
    const size_t lds = lds_c > lds_b ? lds_c : lds_b;
    static const int wt = getenv("PCLIP_PROTO_CLASSIFY_WT") ? atoi(getenv("PCLIP_PROTO_CLASSIFY_WT")) : 1;
    const int ngroups = ceil_div(Q, 16);
    int consumers = ceil_div(ngroups, 2);                           // four waves: two (visual, textual) pairs per workgroup
    const int cap = 2 * cus - N;
    if (consumers > cap) consumers = cap > 1 ? cap : 1;
    proto_classify_kernel<NCH, NT><<<N + consumers, 256, lds, s>>>((const half_t*)mem, K, per_shot_norm, (half_t*)proto, proto_sq, (const half_t*)q,
                                                                  (const half_t*)zt, Q, N, D, alpha, oma, beta, p, argmax, topk_p, topk_i, k, sync, wt);
    return pclip_check_launch("proto_classify");
}

}  // namespace

extern "C" int pclip_proto_classify_applies(int N, int K, int D, int Q) {
    return N >= 1 && N <= 32 && K >= 1 && Q >= 1 && D >= 32 && D % 32 == 0 && D <= 1024;
}

extern "C" int pclip_proto_classify_f16(const void* mem, int N, int K, int D, int per_shot_norm, void* proto_f16, float* proto_sq, const void* q, const void* zt,
                                        int Q, float alpha, float one_minus_alpha, float beta, float* p, int32_t* argmax, float* topk_p, int32_t* topk_i,
                                        int topk, int32_t* sync, pclip_stream_t stream) {
    PCLIP_REQUIRE(mem && proto_f16 && q && zt && sync, "pclip_proto_classify_f16: null pointer");
    PCLIP_REQUIRE(pclip_proto_classify_applies(N, K, D, Q), "pclip_proto_classify_f16: shape N=%d K=%d D=%d Q=%d has no single-launch form (N <= 32, D %% 32 == 0, D <= 1024); "
                  "call pclip_proto_build_f16 + pclip_classify_f16", N, K, D, Q);
    PCLIP_REQUIRE(p || argmax || topk_p || topk_i, "pclip_proto_classify_f16: no output requested");
    PCLIP_REQUIRE(topk >= 0 && topk <= N && topk <= 16 && ((!topk_p && !topk_i) || topk > 0), "pclip_proto_classify_f16: topk=%d outside [0, min(N, 16)] (or top-k outputs without k)", topk);
    int cus = pclip_device_cus();
    if (cus <= 0) cus = 256;
    hipStream_t s = (hipStream_t)stream;
#define PCLIP_PC(NCH, NT) return launch<NCH, NT>(mem, N, K, D, per_shot_norm, proto_f16, proto_sq, q, zt, Q, alpha, one_minus_alpha, beta, p, argmax, topk_p, topk_i, topk, sync, cus, s)
    if (D <= 512) { if (N <= 16) PCLIP_PC(1, 1); else PCLIP_PC(1, 2); }
    if (N <= 16) PCLIP_PC(2, 1);
    PCLIP_PC(2, 2);
#undef PCLIP_PC
}
