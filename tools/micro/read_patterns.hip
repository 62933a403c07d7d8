// Which read pattern reaches the box's read ceiling?  4 GiB read-only kernels in the shapes an upsweep can take.
// build: hipcc --offload-arch=gfx950 -O3 -o tools/micro/read_patterns tools/micro/read_patterns.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdint>
#include <cstdlib>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d line %d\n", (int)e_, __LINE__); exit(2); } } while (0)
typedef uint32_t v4u __attribute__((ext_vector_type(4)));

// A: the stream micro: 256 threads, each block 16 KiB, 4 x 16-byte loads per thread, one batch
template <bool NT>
__global__ __launch_bounds__(256) void pat_a(const uint32_t *__restrict__ in, uint32_t *__restrict__ sink)
{
    const v4u *p = reinterpret_cast<const v4u *>(in) + (size_t)blockIdx.x * 1024 + threadIdx.x;
    uint32_t acc = 0;
    v4u v[4];
#pragma unroll
    for (int u = 0; u < 4; ++u) v[u] = NT ? __builtin_nontemporal_load(p + u * 256) : p[u * 256];
#pragma unroll
    for (int u = 0; u < 4; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    if (acc == 0x12345678u) sink[0] = acc;
}
// B: the upsweep as built: 512 threads per 8 tiles of 32 KiB, wave w walks tile w in BATCHES of GB dword loads
template <int GB, bool NT>
__global__ __launch_bounds__(512) void pat_b(const uint32_t *__restrict__ in, uint32_t *__restrict__ sink)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t *src = in + ((size_t)blockIdx.x * 8 + w) * 8192 + lane;
    uint32_t acc = 0;
#pragma unroll 1
    for (int j = 0; j < 8192; j += GB * 64) {
        uint32_t v[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) v[u] = NT ? __builtin_nontemporal_load(src + j + u * 64) : src[j + u * 64];
#pragma unroll
        for (int u = 0; u < GB; ++u) acc ^= v[u];
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// C: THREADS threads per TILES tiles of 32 KiB read cooperatively with 16-byte loads, BATCH loads in flight per thread
template <int THREADS, int TILES, int BATCH, bool NT>
__global__ __launch_bounds__(THREADS) void pat_c(const uint32_t *__restrict__ in, uint32_t *__restrict__ sink)
{
    constexpr int PER_THREAD = TILES * 2048 / THREADS;   // 16-byte loads per thread
    const v4u *p = reinterpret_cast<const v4u *>(in) + (size_t)blockIdx.x * (TILES * 2048) + threadIdx.x;
    uint32_t acc = 0;
#pragma unroll 1
    for (int j = 0; j < PER_THREAD; j += BATCH) {
        v4u v[BATCH];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) v[u] = NT ? __builtin_nontemporal_load(p + (size_t)(j + u) * THREADS) : p[(size_t)(j + u) * THREADS];
#pragma unroll
        for (int u = 0; u < BATCH; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}
// D: like B (wave = tile) but 16-byte loads: wave walks its tile in batches of GB x 1 KiB
template <int GB, bool NT>
__global__ __launch_bounds__(512) void pat_d(const uint32_t *__restrict__ in, uint32_t *__restrict__ sink)
{
    const int w = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const v4u *src = reinterpret_cast<const v4u *>(in + ((size_t)blockIdx.x * 8 + w) * 8192) + lane;
    uint32_t acc = 0;
#pragma unroll 1
    for (int j = 0; j < 32; j += GB) {
        v4u v[GB];
#pragma unroll
        for (int u = 0; u < GB; ++u) v[u] = NT ? __builtin_nontemporal_load(src + (j + u) * 64) : src[(j + u) * 64];
#pragma unroll
        for (int u = 0; u < GB; ++u) acc ^= v[u].x ^ v[u].y ^ v[u].z ^ v[u].w;
    }
    if (acc == 0x12345678u) sink[0] = acc;
}

template <typename F> static void run(const char *name, F launch)
{
    hipEvent_t e0, e1; CK(hipEventCreate(&e0)); CK(hipEventCreate(&e1));
    float best = 1e9f, sum = 0;
    for (int i = 0; i < 12; ++i) {
        CK(hipEventRecord(e0)); launch(); CK(hipEventRecord(e1)); CK(hipEventSynchronize(e1));
        float ms; CK(hipEventElapsedTime(&ms, e0, e1));
        if (i >= 2) { sum += ms; if (ms < best) best = ms; }
    }
    printf("%-64s best %.3f ms (%.2f TB/s)  mean %.3f ms\n", name, best, 4.294967296 / best, sum / 10);
}

int main()
{
    const size_t bytes = 4ull << 30;
    uint32_t *a, *sink;
    CK(hipMalloc(&a, bytes)); CK(hipMalloc(&sink, 256));
    CK(hipMemset(a, 1, bytes));
    const unsigned tiles = (unsigned)(bytes / 32768);
#define L(k, g, b) [&] { hipLaunchKernelGGL((k), dim3(g), dim3(b), 0, 0, a, sink); }
    run("A  stream micro (256 thr, 16 KiB/block, 4 x b128)", L(pat_a<false>, tiles * 2, 256));
    run("A  stream micro, nt", L(pat_a<true>, tiles * 2, 256));
    run("B  upsweep as built: wave = tile, 4 batches x 32 dword, nt", L((pat_b<32, true>), tiles / 8, 512));
    run("B  same, plain loads", L((pat_b<32, false>), tiles / 8, 512));
    run("B  wave = tile, 8 batches x 16 dword, nt", L((pat_b<16, true>), tiles / 8, 512));
    run("D  wave = tile, 4 batches x 8 b128, nt", L((pat_d<8, true>), tiles / 8, 512));
    run("D  wave = tile, 2 batches x 16 b128, nt", L((pat_d<16, true>), tiles / 8, 512));
    run("C  512 thr, 8 tiles cooperatively, 4 batches x 8 b128, nt", L((pat_c<512, 8, 8, true>), tiles / 8, 512));
    run("C  512 thr, 8 tiles cooperatively, 2 batches x 16 b128, nt", L((pat_c<512, 8, 16, true>), tiles / 8, 512));
    run("C  512 thr, 1 tile per block, 1 batch x 4 b128, nt", L((pat_c<512, 1, 4, true>), tiles, 512));
    run("C  256 thr, 1 tile per block, 1 batch x 8 b128, nt", L((pat_c<256, 1, 8, true>), tiles, 256));
    run("C  256 thr, 1 tile per block, 2 batches x 4 b128, nt", L((pat_c<256, 1, 4, true>), tiles, 256));
    run("C  512 thr, 2 tiles per block, 1 batch x 8 b128, nt", L((pat_c<512, 2, 8, true>), tiles / 2, 512));
    run("C  1024 thr, 8 tiles per block, 1 batch x 16 b128, nt", L((pat_c<1024, 8, 16, true>), tiles / 8, 1024));
    run("C  512 thr, 8 tiles, 1 batch x 32 b128, nt", L((pat_c<512, 8, 32, true>), tiles / 8, 512));
    return 0;
}
