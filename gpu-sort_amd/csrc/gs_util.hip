// gs_util.hip -- version/error strings, on-device input generators and
// size-independent result checks.  The generators restate oracle/oracle.c's
// counter-based streams so CPU and GPU inputs are identical;
// they stand in for the cuRAND fills of lsb/sort.cu:125-131,
// msb/src/test.cu:38-43 and msb/tests/data_gen.h:33-84.
#include "gs_device.hpp"
#include "gs_host.hpp"

namespace gs {

constexpr uint64_t GOLDEN64 = 0x9E3779B97F4A7C15ull;

__device__ __forceinline__ uint32_t uniform_at(uint64_t seed, uint64_t idx)
{
    return (uint32_t)(splitmix64(seed * GOLDEN64 + idx) >> 32);
}

// One instantiation per generator kind (wave-uniform `kind` branches inside a
// single kernel were mis-structurized by hipcc 7.2: the enumerated branch
// stored an undefined register).
template <int KIND>
__global__ __launch_bounds__(256) void generate_kernel(uint32_t *__restrict__ out, uint64_t n, uint64_t seed,
                                                       uint64_t start, int level)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint64_t idx = start + i;
        uint32_t k;
        if (KIND == GS_GEN_UNIFORM) {
            k = uniform_at(seed, idx);
        } else if (KIND == GS_GEN_ZIPF) {
            const uint64_t h = splitmix64((seed + 0x2545F491ull) * GOLDEN64 + idx);
            const uint32_t e = (uint32_t)(((h >> 32) * 24ull) >> 32);
            const uint32_t m = (uint32_t)h & ((1u << e) - 1u);
            k = ((1u << e) + m) * 0x9E3779B1u;
        } else if (KIND == GS_GEN_ENTROPY_AND) {
            k = (level < 1) ? 0u : uniform_at(seed, idx);
            for (int l = 1; l < level; ++l) k &= uniform_at(seed + 17ull * (uint64_t)l, idx);
        } else {
            k = (uint32_t)idx;
        }
        out[i] = k;
    }
}

// result[0] += adjacent inversions, [1] += sum splitmix64(key), [2] ^= same
__global__ __launch_bounds__(256) void check_sorted_kernel(const uint32_t *__restrict__ keys, uint64_t n, int descending,
                                                           unsigned long long *__restrict__ result)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long inv = 0, sum = 0, xr = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t a = keys[i];
        if (i + 1 < n) {
            const uint32_t b = keys[i + 1];
            inv += descending ? (a < b) : (a > b);
        }
        const uint64_t h = splitmix64(a);
        sum += h;
        xr ^= h;
    }
    // wave reduce, one atomic per wave
    for (int o = 32; o > 0; o >>= 1) {
        inv += __shfl_xor(inv, o, WAVE);
        sum += __shfl_xor(sum, o, WAVE);
        xr ^= __shfl_xor(xr, o, WAVE);
    }
    if (lane_id() == 0) {
        if (inv) atomicAdd(&result[0], inv);
        atomicAdd(&result[1], sum);
        atomicXor(&result[2], xr);
    }
}

__global__ __launch_bounds__(256) void check_pairs_enum_kernel(const uint32_t *__restrict__ keys_in,
                                                               const uint32_t *__restrict__ keys_sorted,
                                                               const uint32_t *__restrict__ vals, uint64_t n,
                                                               unsigned long long *__restrict__ result)
{
    const uint64_t stride = (uint64_t)gridDim.x * blockDim.x;
    unsigned long long bad = 0, sum = 0;
    for (uint64_t i = (uint64_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += stride) {
        const uint32_t v = vals[i];
        if (v >= n || keys_in[v] != keys_sorted[i]) ++bad;
        sum += v;
    }
    for (int o = 32; o > 0; o >>= 1) {
        bad += __shfl_xor(bad, o, WAVE);
        sum += __shfl_xor(sum, o, WAVE);
    }
    if (lane_id() == 0) {
        if (bad) atomicAdd(&result[0], bad);
        atomicAdd(&result[1], sum);
    }
}

static inline uint32_t stream_grid(uint64_t n)
{
    const uint64_t b = (n + 255) / 256;
    return (uint32_t)(b < 1 ? 1 : (b > 8192 ? 8192 : b));
}

}  // namespace gs

namespace gs {
thread_local gs_profile *tl_profile = nullptr;

static hipEvent_t take_event(gs_profile *p)
{
    hipEvent_t e;
    if (!p->pool.empty()) { e = p->pool.back(); p->pool.pop_back(); return e; }
    hipEventCreate(&e);
    return e;
}

KernelTimer::KernelTimer(int id, hipStream_t stream) : p(tl_profile), s(stream)
{
    if (!p) return;
    span.id = id;
    span.a = take_event(p);
    span.b = take_event(p);
    hipEventRecord(span.a, s);
}
KernelTimer::~KernelTimer()
{
    if (!p) return;
    hipEventRecord(span.b, s);
    p->spans.push_back(span);
}
}  // namespace gs

namespace gs {
__global__ void zero_kernel(unsigned long long *p, uint32_t n64)
{
    for (uint32_t i = blockIdx.x * blockDim.x + threadIdx.x; i < n64; i += gridDim.x * blockDim.x) p[i] = 0ull;
}
hipError_t zero_async(void *p, size_t bytes, hipStream_t s)
{
    const uint32_t n64 = (uint32_t)(bytes / 8);
    if (n64 == 0) return hipSuccess;
    const uint32_t blocks = (n64 + 255u) / 256u;
    hipLaunchKernelGGL(zero_kernel, dim3(blocks < 1024u ? blocks : 1024u), dim3(256), 0, s, (unsigned long long *)p, n64);
    return hipGetLastError();
}
}  // namespace gs

using namespace gs;

extern "C" {

int gs_version(void) { return GS_VERSION; }

gs_profile *gs_profile_create(void) { return new gs_profile(); }

void gs_profile_destroy(gs_profile *p)
{
    if (!p) return;
    if (tl_profile == p) tl_profile = nullptr;
    for (auto &sp : p->spans) { hipEventDestroy(sp.a); hipEventDestroy(sp.b); }
    for (auto e : p->pool) hipEventDestroy(e);
    delete p;
}

void gs_profile_begin(gs_profile *p) { tl_profile = p; }
void gs_profile_end(void) { tl_profile = nullptr; }

int gs_profile_read(gs_profile *p, double total_ms[GS_K_COUNT], uint64_t launches[GS_K_COUNT])
{
    GS_CLEAR_STALE_ERROR();
    if (!p) return hipErrorInvalidValue;
    for (auto &sp : p->spans) {
        hipError_t e = hipEventSynchronize(sp.b);
        if (e != hipSuccess) return (int)e;
        float ms = 0.f;
        e = hipEventElapsedTime(&ms, sp.a, sp.b);
        if (e != hipSuccess) return (int)e;
        p->ms[sp.id] += ms;
        p->launches[sp.id] += 1;
        p->pool.push_back(sp.a);
        p->pool.push_back(sp.b);
    }
    p->spans.clear();
    for (int i = 0; i < GS_K_COUNT; ++i) {
        if (total_ms) total_ms[i] = p->ms[i];
        if (launches) launches[i] = p->launches[i];
    }
    return hipSuccess;
}

const char *gs_kernel_name(int id)
{
    static const char *names[GS_K_COUNT] = {"lsb_upsweep", "lsb_scan", "lsb_downsweep", "msb_histogram", "msb_classify",
                                            "msb_partition", "msb_local_sort", "shard", "other", "lsb_pass"};
    return (id >= 0 && id < GS_K_COUNT) ? names[id] : "?";
}

const char *gs_error_string(int err) { return hipGetErrorString((hipError_t)err); }

int gs_generate_u32(uint32_t *d_out, uint64_t num_items, int kind, uint64_t seed, uint64_t start_index, int level,
                    void *stream)
{
    GS_CLEAR_STALE_ERROR();
    if (kind < GS_GEN_UNIFORM || kind > GS_GEN_ENUMERATED) return hipErrorInvalidValue;
    if (num_items == 0) return hipSuccess;
    const dim3 g(stream_grid(num_items)), b(256);
    hipStream_t s = (hipStream_t)stream;
    switch (kind) {
    case GS_GEN_UNIFORM: hipLaunchKernelGGL(generate_kernel<GS_GEN_UNIFORM>, g, b, 0, s, d_out, num_items, seed, start_index, level); break;
    case GS_GEN_ZIPF: hipLaunchKernelGGL(generate_kernel<GS_GEN_ZIPF>, g, b, 0, s, d_out, num_items, seed, start_index, level); break;
    case GS_GEN_ENTROPY_AND: hipLaunchKernelGGL(generate_kernel<GS_GEN_ENTROPY_AND>, g, b, 0, s, d_out, num_items, seed, start_index, level); break;
    default: hipLaunchKernelGGL(generate_kernel<GS_GEN_ENUMERATED>, g, b, 0, s, d_out, num_items, seed, start_index, level); break;
    }
    return (int)hipGetLastError();
}

int gs_check_sorted_u32(const uint32_t *d_keys, uint64_t num_items, int descending, uint64_t *d_result, void *stream)
{
    GS_CLEAR_STALE_ERROR();
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = zero_async(d_result, 3 * sizeof(uint64_t), s);
    if (e != hipSuccess) return (int)e;
    if (num_items == 0) return hipSuccess;
    hipLaunchKernelGGL(check_sorted_kernel, dim3(stream_grid(num_items)), dim3(256), 0, s, d_keys, num_items,
                       descending, (unsigned long long *)d_result);
    return (int)hipGetLastError();
}

int gs_check_pairs_enumerated_u32(const uint32_t *d_keys_in, const uint32_t *d_keys_sorted, const uint32_t *d_vals,
                                  uint64_t num_items, uint64_t *d_result, void *stream)
{
    GS_CLEAR_STALE_ERROR();
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = zero_async(d_result, 2 * sizeof(uint64_t), s);
    if (e != hipSuccess) return (int)e;
    if (num_items == 0) return hipSuccess;
    hipLaunchKernelGGL(check_pairs_enum_kernel, dim3(stream_grid(num_items)), dim3(256), 0, s, d_keys_in,
                       d_keys_sorted, d_vals, num_items, (unsigned long long *)d_result);
    return (int)hipGetLastError();
}

}  // extern "C"
