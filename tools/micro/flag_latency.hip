// What does a tile pay to learn something from an earlier tile inside a streaming kernel?
// Every block copies one 32 KiB tile (like the downsweep) and in between publishes a word and polls the word of the
// block DIST tiles before it (relaxed agent-scope = sc1 accesses).  Reported: shader-clock cycles from "my loads have
// arrived and I published" until the predecessor's word was seen (0 polls = it was there already), per DIST.
//   chain = 0: the predecessor publishes as soon as ITS loads arrived (aggregate-style: no dependency chain)
//   chain = 1: the predecessor publishes only after it has seen ITS predecessor (inclusive-style serial chain)
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/flag_latency tools/micro/flag_latency.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d line %d\n", (int)e_, __LINE__); exit(2); } } while (0)

typedef uint32_t v4u __attribute__((ext_vector_type(4)));
constexpr int THREADS = 512, TILE16 = 2048;

template <bool CHAIN>
__global__ __launch_bounds__(THREADS) void k(const v4u *__restrict__ in, v4u *__restrict__ out, uint32_t *flags, uint32_t *wait_cycles,
                                             uint32_t *polls, uint32_t dist, uint32_t work)
{
    const uint32_t t = blockIdx.x;
    const v4u *src = in + (size_t)t * TILE16;
    v4u *dst = out + (size_t)t * TILE16;
    v4u v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = src[threadIdx.x + u * THREADS];
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    // some ALU work standing in for the ranking (keeps the block alive a few microseconds)
    uint32_t x = v[0].x;
    for (uint32_t i = 0; i < work; ++i) x = x * 1664525u + 1013904223u;
    v[0].x ^= (x == 0x9e3779b9u);
    if (threadIdx.x == 0) {
        const unsigned long long t0 = __builtin_amdgcn_s_memtime();
        if (!CHAIN) __hip_atomic_store(&flags[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        uint32_t n = 0;
        if (t >= dist) {
            while (__hip_atomic_load(&flags[t - dist], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) == 0u) {
                __builtin_amdgcn_s_sleep(1);
                if (++n > (1u << 20)) break;
            }
        }
        if (CHAIN) __hip_atomic_store(&flags[t], 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        wait_cycles[t] = (uint32_t)(__builtin_amdgcn_s_memtime() - t0);
        polls[t] = n;
    }
    __syncthreads();
#pragma unroll
    for (int u = 0; u < 4; ++u) dst[threadIdx.x + u * THREADS] = v[u];
}

int main()
{
    const size_t bytes = 1ull << 30;
    const uint32_t tiles = (uint32_t)(bytes / (TILE16 * 16));
    v4u *a, *b; uint32_t *flags, *wc, *pl;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&b, bytes));
    CK(hipMalloc(&flags, tiles * 4)); CK(hipMalloc(&wc, tiles * 4)); CK(hipMalloc(&pl, tiles * 4));
    CK(hipMemset(a, 1, bytes));
    std::vector<uint32_t> h(tiles), hp(tiles);
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    for (int chain = 0; chain < 2; ++chain)
        for (uint32_t work : {0u, 2000u})
            for (uint32_t dist : {1u, 8u, 64u}) {
                if (chain && dist != 1) continue;
                float ms = 0;
                for (int rep = 0; rep < 2; ++rep) {
                    CK(hipMemset(flags, 0, tiles * 4));
                    CK(hipEventRecord(e0));
                    if (chain) hipLaunchKernelGGL((k<true>), dim3(tiles), dim3(THREADS), 0, 0, a, b, flags, wc, pl, dist, work);
                    else hipLaunchKernelGGL((k<false>), dim3(tiles), dim3(THREADS), 0, 0, a, b, flags, wc, pl, dist, work);
                    CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
                    CK(hipEventElapsedTime(&ms, e0, e1));
                }
                CK(hipMemcpy(h.data(), wc, tiles * 4, hipMemcpyDeviceToHost));
                CK(hipMemcpy(hp.data(), pl, tiles * 4, hipMemcpyDeviceToHost));
                std::vector<uint32_t> s(h.begin() + 2048, h.end());
                std::sort(s.begin(), s.end());
                uint64_t zero_polls = 0, gaveup = 0;
                for (uint32_t i = 2048; i < tiles; ++i) { zero_polls += hp[i] == 0; gaveup += hp[i] > (1u << 20); }
                printf("chain %d work %4u dist %2u: kernel %.3f ms (%5.1f tiles/us)  wait cycles p10 %u p50 %u p90 %u p99 %u   first-poll hits %.1f%%  gave up %llu\n",
                       chain, work, dist, ms, tiles / ms / 1e3, s[s.size() / 10], s[s.size() / 2], s[s.size() * 9 / 10], s[s.size() * 99 / 100],
                       100.0 * zero_polls / (tiles - 2048), (unsigned long long)gaveup);
            }
    return 0;
}
