// Microbenchmark: cost of a 64-lane dword store that is split into k contiguous pieces
// landing in k far-apart regions (what a radix scatter does), vs one 256-B piece.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <vector>

template <int K, int ALIGNED>
__global__ __launch_bounds__(256) void scatter_store(uint32_t *out, size_t region_dwords, int iters)
{
    const int lane = threadIdx.x & 63;
    const size_t wave = (size_t)blockIdx.x * 4 + (threadIdx.x >> 6);
    const int piece = lane / (64 / K);          // which of the K regions this lane writes to
    const int within = lane % (64 / K);
    // every wave owns, in every region, a private contiguous stream
    size_t pos = (size_t)piece * region_dwords + wave * (size_t)iters * (64 / K) + within + (ALIGNED ? 0 : (piece * 5 + 3) % 13);
    for (int i = 0; i < iters; ++i) {
        out[pos] = (uint32_t)(pos + i);
        pos += 64 / K;
    }
}

template <int K, int ALIGNED>
float run(uint32_t *d, size_t total_dwords, int blocks, int iters)
{
    const size_t region = total_dwords / 64;   // 64 regions max
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    scatter_store<K, ALIGNED><<<blocks, 256>>>(d, region, iters);
    hipEventRecord(a);
    for (int r = 0; r < 5; ++r) scatter_store<K, ALIGNED><<<blocks, 256>>>(d, region, iters);
    hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b);
    return ms / 5;
}

int main()
{
    const size_t total = (size_t)1 << 30;   // 4 GiB
    uint32_t *d; hipMalloc(&d, total * 4 + 4096);
    const int blocks = 8192, iters = 512;   // 8192*4 waves * 512 iters * 256 B = 4 GiB
    const double gb = (double)blocks * 4 * iters * 256 / 1e9;
#define R(K, A) { float ms = run<K, A>(d, total, blocks, iters); printf("K=%2d aligned=%d  %.3f ms  %.0f GB/s\n", K, A, ms, gb / ms * 1e3); }
    R(1, 1) R(2, 1) R(4, 1) R(8, 1) R(16, 1) R(1, 0) R(2, 0) R(4, 0) R(8, 0) R(16, 0)
    return 0;
}
