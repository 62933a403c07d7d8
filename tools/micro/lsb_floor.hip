// lsb_floor.hip -- measured floor of the LSB downsweep in its own launch shape.
//
// What bounds `lsb_downsweep` from below if its RANKING were free?  A synthetic kernel in the same launch shape
// (one workgroup per tile of 8192 keys, 512 threads x 16 keys, 41 KiB of LDS -> three workgroups per CU, tiles
// handed out in XCD-contiguous slices) does everything the real kernel does to memory --
//   the tile's 16 wave-striped key loads per lane, the spine / prefix16 / totals reads of wave 0, one random-slot LDS
//   write + one in-order LDS read per key (the exchange), the gbase lookup, and the REAL scatter: every key goes to
//   the address the stable partition gives it (the output is compared bit for bit with the real kernel's) --
// while the ranks are PRECOMPUTED: the input of tile t holds the tile's keys so that the key loaded into slot s belongs
// at tile rank pi(s) = (s * M) mod TILE (a fixed bijection, one multiply).  The variants add the real kernel's vector
// work back, piece by piece:
//   F0  copy through LDS, linear stores                 (no scatter, no ranking)
//   F1  F0 with the real scatter
//   F2  F1 + the 32-instruction ballot match per key    (results folded into a value nothing depends on)
//   F3  F2 + the wave-private counter read / add per key (the LDS traffic of the ranking)
// plus the real kernel on the same box.  Also for tiles of 16384 keys (1024 threads, two workgroups per CU -- the
// synthetic kernel has no pos[] registers) and 32768 keys.
//
// build (from the repo root; tools/micro/Makefile-free on purpose):
//   hipcc --offload-arch=gfx950 -O3 -std=c++17 -Iinclude -Igpu-sort_amd/csrc tools/micro/lsb_floor.hip \
//         -o tools/micro/lsb_floor -Lgpu-sort_amd/lib -lgpusort -Wl,-rpath,'$ORIGIN/../../gpu-sort_amd/lib'
// run:   tools/micro/lsb_floor [log2n=30] [shift=8]
#include "gs_device.hpp"
#include "gs_lsb.hpp"
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>

using namespace gs;

#define CK(x) do { hipError_t e_ = (hipError_t)(x); if (e_ != hipSuccess) { fprintf(stderr, "%s:%d: %s -> %s\n", __FILE__, __LINE__, #x, hipGetErrorString(e_)); exit(1); } } while (0)

// ---------------------------------------------------------------- set-up: per-tile stable partition by digit ----
// one workgroup per tile; thread d collects the keys of digit d in order (broadcast LDS reads).  Slow and simple.
template <int TILE>
__global__ __launch_bounds__(256) void make_inputs(const uint32_t *__restrict__ raw, uint32_t *__restrict__ sorted_tiles,
                                                   uint32_t *__restrict__ permuted_tiles, uint32_t shift, uint32_t minv)
{
    __shared__ uint32_t keys[TILE];
    __shared__ uint32_t base[256];
    const size_t lo = (size_t)blockIdx.x * TILE;
    for (int i = threadIdx.x; i < TILE; i += 256) keys[i] = raw[lo + i];
    __syncthreads();
    const uint32_t d = threadIdx.x;
    uint32_t c = 0;
    for (int i = 0; i < TILE; ++i) c += ((keys[i] >> shift) & 255u) == d;
    base[d] = c;
    __syncthreads();
    uint32_t at = 0;
    for (uint32_t j = 0; j < d; ++j) at += base[j];
    __syncthreads();
    for (int i = 0; i < TILE; ++i) {
        const uint32_t k = keys[i];
        if (((k >> shift) & 255u) == d) {
            sorted_tiles[lo + at] = k;                                     // rank r = at
            permuted_tiles[lo + ((at * minv) & (uint32_t)(TILE - 1))] = k;   // loaded into slot s with pi(s) = r
            ++at;
        }
    }
}

// ---------------------------------------------------------------- the synthetic downsweep ----
// Shape: WAVES x 64 threads, KPT keys per thread, OCC waves per SIMD asked of the compiler.  The wave-private counters
// of the F3 variants live in their own 1 KiB rows where three (8192-key tile) or two (8 waves x 32 keys) workgroups
// still fit a CU with them; the 16-wave shape overlays them on the staging buffer (one more barrier), as a real
// kernel of that shape would have to (64 KiB of staging + 16 KiB of counters would leave ONE workgroup per CU).
template <int WAVES, int KPT, bool COUNTERS>
struct FloorSmem {
    static constexpr bool ALIAS = WAVES > 8;
    uint32_t whist[ALIAS ? 1 : WAVES][RADIX];
    uint32_t gbase[RADIX];
    uint32_t ex[RADIX];
    uint32_t stage[WAVES * WAVE * KPT];
};

template <int WAVES, int KPT, int OCC, bool SCATTER, bool MATCH, bool COUNTERS, int MODE = 0>
__global__ __launch_bounds__(WAVES * WAVE, OCC) void floor_kernel(
    const uint32_t *__restrict__ keys_in, uint32_t *__restrict__ keys_out, const uint32_t *__restrict__ spine,
    const uint16_t *__restrict__ prefix16, const uint32_t *__restrict__ totals, uint32_t *__restrict__ sink, PassParams p,
    uint32_t mul)
{
    constexpr int TILE = WAVES * WAVE * KPT;
    constexpr uint32_t PER8K = TILE / LSB_TILE;        // 8192-key tiles of the library's upsweep per tile of this kernel
    constexpr bool ALIAS = FloorSmem<WAVES, KPT, COUNTERS>::ALIAS;
    __shared__ __attribute__((aligned(16))) FloorSmem<WAVES, KPT, COUNTERS> sm;
    const uint32_t full_tiles = p.n / (uint32_t)TILE;
    if (blockIdx.x >= full_tiles) return;
    const uint32_t t = tile_of_item(blockIdx.x, full_tiles);
    const int lane = lane_id(), w = wave_id();
    uint32_t wbits = p.bits;
    asm volatile("" : "+v"(wbits));
    auto digit = [&](uint32_t k) { return __builtin_amdgcn_ubfe(k, p.shift, wbits); };

    __builtin_amdgcn_s_setprio(3);
    const uint32_t wbase = (uint32_t)w * (WAVE * KPT) + lane;
    const uint32_t *kin = keys_in + (size_t)t * TILE;
    uint32_t key[KPT];
#pragma unroll
    for (int i = 0; i < KPT; ++i) key[i] = kin[wbase + i * WAVE];
    __builtin_amdgcn_s_setprio(0);

    // wave 0: digit starts + this tile's offsets, exactly as the real kernel reads them (parked in LDS until the
    // tile-local run starts are known)
    if (SCATTER && w == 0) {
        const uint4 tot = reinterpret_cast<const uint4 *>(totals)[lane];
        const uint32_t lane_sum = tot.x + tot.y + tot.z + tot.w;
        const uint32_t exs = wave_inclusive_scan(lane_sum) - lane_sum;
        uint32_t dstart[4];
        dstart[0] = exs; dstart[1] = dstart[0] + tot.x; dstart[2] = dstart[1] + tot.y; dstart[3] = dstart[2] + tot.z;
        const uint32_t t8 = t * PER8K;
        const uint32_t *sp = spine + (uint32_t)(4 * lane) * p.grid + t8 / LSB_CHUNK;
        const uint2 pf = reinterpret_cast<const uint2 *>(prefix16 + (size_t)t8 * RADIX)[lane];
        reinterpret_cast<uint4 *>(sm.gbase)[lane] = make_uint4(dstart[0] + sp[0] + (pf.x & 0xffffu), dstart[1] + sp[p.grid] + (pf.x >> 16),
                                                              dstart[2] + sp[2 * p.grid] + (pf.y & 0xffffu), dstart[3] + sp[3 * p.grid] + (pf.y >> 16));
    }
    uint32_t *my = ALIAS ? &sm.stage[w * RADIX] : sm.whist[COUNTERS ? w : 0];
    if (COUNTERS) {
#pragma unroll
        for (int i = lane; i < RADIX; i += WAVE) my[i] = 0;
    }
    if (threadIdx.x < RADIX) sm.ex[threadIdx.x] = 0;
    if (!COUNTERS && !ALIAS && p.n == 1u) sm.whist[w][lane] = 1;   // never true: keeps the counters' LDS allocated

    uint32_t acc = 0;
    if (MATCH) {
        uint32_t d_prev = 0, plo = 0, phi = 0;
#pragma unroll
        for (int i = 0; i <= KPT; ++i) {
            uint32_t d_cur = 0, clo = 0, chi = 0;
            if (i < KPT) {
                d_cur = digit(key[i]);
                match_digit(d_cur, clo, chi);
            }
            if (i > 0) {
                const uint32_t lower = count_lower(plo, phi);
                if (COUNTERS) {
                    acc += my[d_prev] + lower;
                    if (lower == 0)
                        __hip_atomic_fetch_add(&my[d_prev], (uint32_t)(__popc(plo) + __popc(phi)), __ATOMIC_RELAXED,
                                               __HIP_MEMORY_SCOPE_WAVEFRONT);
                } else {
                    acc += lower;
                }
            }
            d_prev = d_cur; plo = clo; phi = chi;
            __builtin_amdgcn_sched_barrier(0);     // one round at a time: nothing of later rounds is hoisted into registers
        }
    }
    if (COUNTERS && ALIAS) __syncthreads();            // the counters' rows become staging space
    // the exchange: the key loaded into slot s belongs at tile rank pi(s) (precomputed by the input's layout)
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t s = wbase + i * WAVE;
        sm.stage[(s * mul) & (uint32_t)(TILE - 1)] = key[i];
    }
    __syncthreads();
    // where every digit run starts in the staged tile (the real kernel knows it from its counters and pays one LDS read
    // per key for its wave bases instead)
    if (SCATTER) {
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t slot = (uint32_t)w * (WAVE * KPT) + i * WAVE + lane;
            const uint32_t kk = sm.stage[slot], prev = sm.stage[slot ? slot - 1 : 0];
            if (slot == 0 || digit(prev) != digit(kk)) sm.ex[digit(kk)] = slot;
        }
    }
    if (SCATTER) {
        __syncthreads();
        if (w == 0) {
            const uint4 e = reinterpret_cast<const uint4 *>(sm.ex)[lane], g = reinterpret_cast<const uint4 *>(sm.gbase)[lane];
            reinterpret_cast<uint4 *>(sm.gbase)[lane] = make_uint4((g.x - e.x) << 2, (g.y - e.y) << 2, (g.z - e.z) << 2, (g.w - e.w) << 2);
        }
    }
    __syncthreads();
    // experiment modes (timing only, output not comparable): 1 = every digit's stream shifted by d x 4352 bytes (breaks the
    // 16 MiB lockstep of the 256 write frontiers of uniform keys), 2 = every stream folded into an 8 KiB window (2 MiB in
    // all: the scatter never leaves the L2s -- what the request path costs without HBM)
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t slot = (uint32_t)w * (WAVE * KPT) + i * WAVE + lane;   // wave-contiguous slots, as the real kernel stores
        const uint32_t kk = sm.stage[slot];
        uint32_t off;
        if (SCATTER) off = sm.gbase[digit(kk)] + slot * 4u;       // n <= 2^30: 32-bit byte offsets, like the real !BIG kernel
        else off = (t * (uint32_t)TILE + slot) * 4u;
        if (MODE == 1) {
            const uint64_t o64 = (uint64_t)off + (uint64_t)digit(kk) * 4352u;
            *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(keys_out) + o64) = kk;
            continue;
        }
        if (MODE == 2) off = (off & 0x1fffu) + digit(kk) * 0x2000u;
        *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(keys_out) + off) = kk;
    }
    if (MATCH && acc == 0x9e3779b9u) sink[0] = acc;   // keeps the ranking work alive; practically never taken
}

// the library's pass parameters, rebuilt from its C ABI (the tool links against libgpusort.so's exported symbols only)
static PassParams floor_params(uint64_t n, int shift, int bits)
{
    PassParams p{};
    uint32_t grid, tile, tpc;
    gs_lsb_geometry(n, 0, &grid, &tile, &tpc);
    p.n = (uint32_t)n;
    p.num_tiles = (uint32_t)((n + tile - 1) / tile);
    p.grid = grid;
    p.ds_grid = (uint32_t)(n / tile);
    p.shift = (uint32_t)shift;
    p.bits = (uint32_t)bits;
    p.mask = (1u << bits) - 1u;
    return p;
}

static float time_ms(hipEvent_t a, hipEvent_t b) { float ms = 0; CK(hipEventSynchronize(b)); CK(hipEventElapsedTime(&ms, a, b)); return ms; }

__global__ void diff_kernel(const uint32_t *a, const uint32_t *b, size_t n, unsigned long long *out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    unsigned long long c = 0;
    for (; i < n; i += (size_t)gridDim.x * blockDim.x) c += a[i] != b[i];
    if (c) atomicAdd(out, c);
}
static unsigned long long differences(const uint32_t *a, const uint32_t *b, size_t n)
{
    unsigned long long *d, h = 0;
    CK(hipMalloc(&d, 8)); CK(hipMemset(d, 0, 8));
    diff_kernel<<<4096, 256>>>(a, b, n, d);
    CK(hipMemcpy(&h, d, 8, hipMemcpyDeviceToHost)); CK(hipFree(d));
    return h;
}

template <int WAVES, int KPT, int OCC, bool SCATTER, bool MATCH, bool COUNTERS, int MODE = 0>
static void run_variant(const char *name, const uint32_t *in, uint32_t *out, const uint32_t *expect, uint32_t *spine, uint16_t *prefix16,
                        uint32_t *totals, uint32_t *sink, const PassParams &p, uint32_t mul, size_t n, int reps)
{
    constexpr int TILE = WAVES * WAVE * KPT;
    const dim3 grid((unsigned)(n / TILE)), block(WAVES * WAVE);
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    CK(hipMemset(out, 0xff, n * 4));
    float best = 1e9f, sum = 0;
    for (int r = 0; r < reps + 1; ++r) {
        CK(hipEventRecord(a));
        floor_kernel<WAVES, KPT, OCC, SCATTER, MATCH, COUNTERS, MODE><<<grid, block>>>(in, out, spine, prefix16, totals, sink, p, mul);
        CK(hipEventRecord(b));
        CK(hipGetLastError());
        const float ms = time_ms(a, b);
        if (r) { sum += ms; best = ms < best ? ms : best; }
    }
    const unsigned long long bad = MODE == 0 ? differences(out, expect, n) : 0;
    const double gb = 8.0 * n / 1e9;
    printf("%-34s %2d waves x %2d keys, tile %5d  avg %.3f ms  min %.3f ms  %.0f GB/s (%.3f of 8 TB/s)  %s\n", name, WAVES, KPT, TILE, sum / reps, best,
           gb / (sum / reps) * 1e3, gb / (sum / reps) * 1e3 / 8000.0,
           bad ? "OUTPUT DIFFERS" : (MODE == 0 ? "output bit-exact" : "(addresses altered: timing only)"));
    if (bad) printf("   !! %llu differing keys\n", bad);
    fflush(stdout);
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
}

static uint32_t inverse_mod_pow2(uint32_t m, uint32_t mod)
{
    for (uint32_t x = 1; x < mod; x += 2)
        if (((x * m) & (mod - 1)) == 1u) return x;
    return 0;
}

template <int WAVES, int KPT, int OCC>
static void run_geometry(const uint32_t *raw, uint32_t *sorted_tiles, uint32_t *permuted, uint32_t *out, uint32_t *expect, void *temp,
                         size_t temp_bytes, size_t n, int shift, int reps)
{
    constexpr int TILE = WAVES * WAVE * KPT;
    const uint32_t mul = 2731u, minv = inverse_mod_pow2(mul, TILE);   // pi(s) = s * 2731 mod TILE
    make_inputs<TILE><<<dim3((unsigned)(n / TILE)), dim3(256)>>>(raw, sorted_tiles, permuted, (uint32_t)shift, minv);
    CK(hipGetLastError()); CK(hipDeviceSynchronize());
    uint32_t *spine, *totals; uint16_t *prefix16;
    CK(gs_lsb_workspace_layout(temp, n, &spine, &totals, &prefix16));
    // the pass's counts through the library's upsweep + scan (of the input the downsweep that follows reads: the
    // 8192-key tiles of the tile-sorted and of the permuted input hold different keys when TILE is 16384)
    CK(gs_lsb_upsweep_u32(temp, temp_bytes, sorted_tiles, n, shift, 8, 0, GS_KEY_U32, nullptr));
    CK(gs_lsb_scan_spine(temp, temp_bytes, n, nullptr));
    // expectation: the real downsweep on the tile-sorted input (a stable partition does not care how tiles are cut)
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float sum = 0, best = 1e9f;
    for (int r = 0; r < reps + 1; ++r) {
        CK(hipEventRecord(a));
        CK(gs_lsb_downsweep_u32(temp, temp_bytes, sorted_tiles, expect, nullptr, nullptr, n, shift, 8, 0, GS_KEY_U32, GS_KEY_U32, nullptr));
        CK(hipEventRecord(b));
        const float ms = time_ms(a, b);
        if (r) { sum += ms; best = ms < best ? ms : best; }
    }
    printf("%-34s tile %5d  avg %.3f ms  min %.3f ms  (tile-sorted input: its ranks are trivial, its scatter is the real one)\n",
           "real lsb_downsweep, sorted tiles", LSB_TILE, sum / reps, best);
    CK(gs_lsb_upsweep_u32(temp, temp_bytes, permuted, n, shift, 8, 0, GS_KEY_U32, nullptr));
    CK(gs_lsb_scan_spine(temp, temp_bytes, n, nullptr));
    sum = 0; best = 1e9f;
    for (int r = 0; r < reps + 1; ++r) {
        CK(hipEventRecord(a));
        CK(gs_lsb_downsweep_u32(temp, temp_bytes, permuted, out, nullptr, nullptr, n, shift, 8, 0, GS_KEY_U32, GS_KEY_U32, nullptr));
        CK(hipEventRecord(b));
        const float ms = time_ms(a, b);
        if (r) { sum += ms; best = ms < best ? ms : best; }
    }
    printf("%-34s tile %5d  avg %.3f ms  min %.3f ms  %.0f GB/s (%.3f of 8 TB/s)\n", "real lsb_downsweep, permuted tiles", LSB_TILE, sum / reps, best,
           8.0 * n / 1e9 / (sum / reps) * 1e3, 8.0 * n / 1e9 / (sum / reps) * 1e3 / 8000.0);
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));

    PassParams p = floor_params(n, shift, 8);
    uint32_t *sink;
    CK(hipMalloc(&sink, 256));
    run_variant<WAVES, KPT, OCC, false, false, false>("F0 copy through LDS, linear", permuted, out, sorted_tiles, spine, prefix16, totals, sink, p, mul, n, reps);
    run_variant<WAVES, KPT, OCC, true, false, false>("F1 + real scatter", permuted, out, expect, spine, prefix16, totals, sink, p, mul, n, reps);
    run_variant<WAVES, KPT, OCC, true, true, false>("F2 + ballot match (32 VALU/key)", permuted, out, expect, spine, prefix16, totals, sink, p, mul, n, reps);
    run_variant<WAVES, KPT, OCC, true, true, true>("F3 + counter read/add per key", permuted, out, expect, spine, prefix16, totals, sink, p, mul, n, reps);
    run_variant<WAVES, KPT, OCC, false, true, true>("F0 + match + counters, linear", permuted, out, sorted_tiles, spine, prefix16, totals, sink, p, mul, n, reps);
    if (WAVES == 8 && KPT == 16) {
        run_variant<WAVES, KPT, OCC, true, false, false, 1>("F1, streams shifted d x 4352 B", permuted, out, expect, spine, prefix16, totals, sink, p, mul, n, reps);
        run_variant<WAVES, KPT, OCC, true, false, false, 2>("F1, streams folded into 2 MiB", permuted, out, expect, spine, prefix16, totals, sink, p, mul, n, reps);
    }
    CK(hipFree(sink));
}


// ---------------------------------------------------------------- scatter patterns without keys ----
// The same launch shape and LDS exchange, but the destination is a pure function of (tile, slot): every tile writes
// R = TILE / L runs of L keys, run r of tile t at  r * stream_stride + t * L * 4 (+ a per-stream misalignment of
// 4 * (r % 32) bytes when MISALIGN) -- what a pass over PERFECTLY uniform keys would write.  L = 32 is the real pass
// (256 streams, 128-byte runs); the sweep over L shows what run length the memory system wants, MISALIGN what the
// 128-byte-line straddling of real runs costs, FOLD = 1 keeps every stream inside an 8 KiB window at its own stride
// (same pages, nothing reaches HBM), FOLD = 2 packs those windows densely (one page).
template <int LOG2L, bool MISALIGN, int FOLD, bool WORK = false>
__global__ __launch_bounds__(LSB_THREADS, 6) void pattern_kernel(const uint32_t *__restrict__ keys_in, uint32_t *__restrict__ keys_out,
                                                                uint32_t n, uint32_t mul)
{
    constexpr int TILE = LSB_TILE, L = 1 << LOG2L, R = TILE / L;
    __shared__ __attribute__((aligned(16))) uint32_t stage[TILE];
    __shared__ uint32_t pad_[2560];       // 42 KiB in all: three workgroups per CU, like the real kernel
    const uint32_t full_tiles = n / (uint32_t)TILE;
    if (blockIdx.x >= full_tiles) return;
    const uint32_t t = tile_of_item(blockIdx.x, full_tiles);
    const int lane = lane_id(), w = wave_id();
    __builtin_amdgcn_s_setprio(3);
    const uint32_t wbase = (uint32_t)w * (WAVE * LSB_KPT) + lane;
    const uint32_t *kin = keys_in + (size_t)t * TILE;
    uint32_t key[LSB_KPT];
#pragma unroll
    for (int i = 0; i < LSB_KPT; ++i) key[i] = kin[wbase + i * WAVE];
    __builtin_amdgcn_s_setprio(0);
    if (n == 1u) pad_[threadIdx.x] = 1;
    uint32_t acc = 0;
    if (WORK) {   // the real kernel's ranking work on the loaded keys (ballot match + wave-private counter read / add), results unused
        uint32_t *my = pad_ + w * RADIX;
#pragma unroll
        for (int i = lane; i < RADIX; i += WAVE) my[i] = 0;
        uint32_t d_prev = 0, plo = 0, phi = 0;
#pragma unroll
        for (int i = 0; i <= LSB_KPT; ++i) {
            uint32_t d_cur = 0, clo = 0, chi = 0;
            if (i < LSB_KPT) {
                d_cur = (key[i] >> 8) & 255u;
                match_digit(d_cur, clo, chi);
            }
            if (i > 0) {
                const uint32_t lower = count_lower(plo, phi);
                acc += my[d_prev] + lower;
                if (lower == 0)
                    __hip_atomic_fetch_add(&my[d_prev], (uint32_t)(__popc(plo) + __popc(phi)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            d_prev = d_cur; plo = clo; phi = chi;
        }
    }
#pragma unroll
    for (int i = 0; i < LSB_KPT; ++i) stage[((wbase + i * WAVE) * mul) & (uint32_t)(TILE - 1)] = key[i];
    __syncthreads();
    const uint64_t stride = (uint64_t)n * 4u / R;                  // bytes between streams
    __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int i = 0; i < LSB_KPT; ++i) {
        const uint32_t slot = (uint32_t)w * (WAVE * LSB_KPT) + i * WAVE + lane;
        const uint32_t r = slot >> LOG2L, j = slot & (uint32_t)(L - 1);
        uint64_t within = ((uint64_t)t * L + j) * 4u + (MISALIGN ? 4u * (r & 31u) : 0u);
        uint64_t off;
        if (FOLD == 0) off = (uint64_t)r * stride + within;
        else if (FOLD == 1) off = (uint64_t)r * stride + (within & 0x1fffu);
        else off = (uint64_t)r * 0x2000u + (within & 0x1fffu);
        *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(keys_out) + off) = stage[slot];
    }
    if (WORK && acc == 0x9e3779b9u) keys_out[n] = acc;     // practically never
}

template <int LOG2L, bool MISALIGN, int FOLD, bool WORK = false>
static void run_pattern(const uint32_t *in, uint32_t *out, size_t n, int reps)
{
    hipEvent_t a, b;
    CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
    float best = 1e9f, sum = 0;
    for (int r = 0; r < reps + 1; ++r) {
        CK(hipEventRecord(a));
        pattern_kernel<LOG2L, MISALIGN, FOLD, WORK><<<dim3((unsigned)(n / LSB_TILE)), dim3(LSB_THREADS)>>>(in, out, (uint32_t)n, 2731u);
        CK(hipEventRecord(b));
        CK(hipGetLastError());
        const float ms = time_ms(a, b);
        if (r) { sum += ms; best = ms < best ? ms : best; }
    }
    printf("pattern%s: %5d streams x runs of %4d keys (%5d B)%s%s  avg %.3f ms  min %.3f ms  %.0f GB/s\n", WORK ? " + ranking work" : "", LSB_TILE >> LOG2L, 1 << LOG2L, 4 << LOG2L,
           MISALIGN ? ", misaligned" : ", aligned   ", FOLD == 0 ? "                    " : (FOLD == 1 ? ", 8 KiB windows@stride" : ", 8 KiB windows packed"),
           sum / reps, best, 8.0 * n / 1e9 / (sum / reps) * 1e3);
    fflush(stdout);
    CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
}

static void run_patterns(const uint32_t *in, uint32_t *out, size_t n, int reps)
{
    run_pattern<13, false, 0>(in, out, n, reps);   // one stream: linear
    run_pattern<9, false, 0>(in, out, n, reps);
    run_pattern<8, false, 0>(in, out, n, reps);
    run_pattern<7, false, 0>(in, out, n, reps);
    run_pattern<6, false, 0>(in, out, n, reps);
    run_pattern<5, false, 0>(in, out, n, reps);    // the real pass's shape, runs aligned to their 128-byte lines
    run_pattern<4, false, 0>(in, out, n, reps);
    run_pattern<3, false, 0>(in, out, n, reps);
    run_pattern<7, true, 0>(in, out, n, reps);
    run_pattern<6, true, 0>(in, out, n, reps);
    run_pattern<5, true, 0>(in, out, n, reps);     // ... straddling lines like real runs
    run_pattern<4, true, 0>(in, out, n, reps);
    run_pattern<5, false, 1>(in, out, n, reps);
    run_pattern<5, false, 2>(in, out, n, reps);
    run_pattern<5, true, 1>(in, out, n, reps);
    run_pattern<5, true, 2>(in, out, n, reps);
    run_pattern<13, false, 0, true>(in, out, n, reps);
    run_pattern<7, false, 0, true>(in, out, n, reps);
    run_pattern<6, false, 0, true>(in, out, n, reps);
    run_pattern<5, false, 0, true>(in, out, n, reps);
    run_pattern<4, false, 0, true>(in, out, n, reps);
    run_pattern<7, true, 0, true>(in, out, n, reps);
    run_pattern<6, true, 0, true>(in, out, n, reps);
    run_pattern<5, true, 0, true>(in, out, n, reps);
    run_pattern<4, true, 0, true>(in, out, n, reps);
}

int main(int argc, char **argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 30;
    const int shift = argc > 2 ? atoi(argv[2]) : 8;
    const int reps = 5;
    if (log2n < 20 || log2n > 30) { fprintf(stderr, "log2n in [20, 30]\n"); return 1; }
    const size_t n = (size_t)1 << log2n;
    uint32_t *raw, *sorted_tiles, *permuted, *out, *expect;
    CK(hipMalloc(&raw, n * 4)); CK(hipMalloc(&sorted_tiles, n * 4)); CK(hipMalloc(&permuted, n * 4));
    CK(hipMalloc(&out, n * 4 + 256 * 4352 + 4096)); CK(hipMalloc(&expect, n * 4));
    const size_t temp_bytes = gs_lsb_temp_bytes(n, 0);
    void *temp;
    CK(hipMalloc(&temp, temp_bytes));
    CK(gs_generate_u32(raw, n, GS_GEN_UNIFORM, 0, 0, 0, nullptr));
    CK(hipDeviceSynchronize());
    hipDeviceProp_t prop;
    CK(hipGetDeviceProperties(&prop, 0));
    printf("# lsb_floor: n = 2^%d uniform u32 keys, digit at shift %d, %s, %d CUs, %d timed launches per line; 8 B/key algorithmic\n", log2n, shift,
           prop.gcnArchName, prop.multiProcessorCount, reps);
    // the real kernel on the raw keys, for the box's reference point
    {
        CK(gs_lsb_upsweep_u32(temp, temp_bytes, raw, n, shift, 8, 0, GS_KEY_U32, nullptr));
        CK(gs_lsb_scan_spine(temp, temp_bytes, n, nullptr));
        hipEvent_t a, b;
        CK(hipEventCreate(&a)); CK(hipEventCreate(&b));
        float sum = 0, best = 1e9f, usum = 0;
        for (int r = 0; r < reps + 1; ++r) {
            CK(hipEventRecord(a));
            CK(gs_lsb_upsweep_u32(temp, temp_bytes, raw, n, shift, 8, 0, GS_KEY_U32, nullptr));
            CK(hipEventRecord(b));
            const float ums = time_ms(a, b);
            CK(gs_lsb_scan_spine(temp, temp_bytes, n, nullptr));
            CK(hipEventRecord(a));
            CK(gs_lsb_downsweep_u32(temp, temp_bytes, raw, out, nullptr, nullptr, n, shift, 8, 0, GS_KEY_U32, GS_KEY_U32, nullptr));
            CK(hipEventRecord(b));
            const float ms = time_ms(a, b);
            if (r) { sum += ms; usum += ums; best = ms < best ? ms : best; }
        }
        printf("%-34s tile %5d  avg %.3f ms  min %.3f ms  %.0f GB/s (%.3f of 8 TB/s)   [upsweep avg %.3f ms]\n", "real lsb_downsweep, raw keys", LSB_TILE,
               sum / reps, best, 8.0 * n / 1e9 / (sum / reps) * 1e3, 8.0 * n / 1e9 / (sum / reps) * 1e3 / 8000.0, usum / reps);
        CK(hipEventDestroy(a)); CK(hipEventDestroy(b));
    }
    if (argc > 3 && !strcmp(argv[3], "pmc")) {     // two kernels for a rocprofv3 --pmc pass: the real pass's shape, aligned / straddling
        run_pattern<5, false, 0>(raw, out, n, 2);
        run_pattern<5, true, 0>(raw, out, n, 2);
        run_pattern<13, false, 0>(raw, out, n, 2);
        return 0;
    }
    if (argc > 3 && !strcmp(argv[3], "probe")) {   // a few patterns for tools/box_probe.sh: what differs between boxes
        run_pattern<13, false, 0>(raw, out, n, 3);
        run_pattern<5, false, 0>(raw, out, n, 3);
        run_pattern<5, true, 0>(raw, out, n, 3);
        run_pattern<4, false, 0>(raw, out, n, 3);
        run_pattern<4, true, 0>(raw, out, n, 3);
        run_pattern<5, true, 0, true>(raw, out, n, 3);
        return 0;
    }
    run_patterns(raw, out, n, reps);
    run_geometry<8, 16, 6>(raw, sorted_tiles, permuted, out, expect, temp, temp_bytes, n, shift, reps);     // the real kernel's shape
    run_geometry<16, 16, 8>(raw, sorted_tiles, permuted, out, expect, temp, temp_bytes, n, shift, reps);    // 16384 keys, 64 VGPRs, 2 per CU
    run_geometry<8, 32, 4>(raw, sorted_tiles, permuted, out, expect, temp, temp_bytes, n, shift, reps);     // 16384 keys, 128 VGPRs, 2 per CU
    run_geometry<16, 32, 4>(raw, sorted_tiles, permuted, out, expect, temp, temp_bytes, n, shift, reps);    // 32768 keys, 1 per CU
    return 0;
}
