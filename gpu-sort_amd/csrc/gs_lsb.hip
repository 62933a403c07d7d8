// gs_lsb.hip -- stable LSD radix sort for gfx950 (MI355X), 8-bit digits, three
// kernels per pass: upsweep histogram -> spine scan -> downsweep scatter.
//
// Replaces (behaviour, not code) the CUB path the LSB driver calls:
//   cub::DeviceRadixSort::Sort{Keys,Pairs}[Descending]  lsb/cub/cub/device/device_radix_sort.cuh:248,595,754
//   DispatchRadixSort::InvokePasses / InvokePass        lsb/cub/cub/device/dispatch/dispatch_radix_sort.cuh:899-1166
//   AgentRadixSortUpsweep / RadixSortScanBins / AgentRadixSortDownsweep
//
// Design: every block owns one contiguous run of 8192-key tiles
// (even share), so the spine is only 256 x grid counters.  Ranking inside a
// tile is per-wavefront ballot/popcount matching on the 8-bit digit with a
// wave-private 256-bin LDS histogram; the tile is then staged through LDS in
// digit order so each wave writes contiguous runs to HBM.
#include "gs_device.hpp"
#include "gs_host.hpp"

namespace gs {

constexpr int LSB_THREADS = 512;                     // 8 waves
constexpr int LSB_WAVES = LSB_THREADS / WAVE;
constexpr int LSB_KPT = 16;                          // keys per thread per tile
constexpr int LSB_TILE = LSB_THREADS * LSB_KPT;      // 8192 keys = 32 KiB
constexpr int LSB_BLOCKS_PER_CU = 2;                 // 128 VGPRs -> 4 waves/SIMD -> 2 blocks of 8 waves
constexpr int MI355X_CUS = 256;
constexpr uint32_t LSB_MAX_GRID = MI355X_CUS * LSB_BLOCKS_PER_CU;

struct PassParams {
    uint32_t n;          // number of keys
    uint32_t num_tiles;  // ceil(n / LSB_TILE)
    uint32_t grid;       // blocks; block b owns tiles [tile_begin(b), tile_end(b))
    int shift;           // digit = (key >> shift) & mask
    uint32_t mask;
    int f32_in, f32_out;          // float twiddle on read / undo on write
    uint32_t xor_in, xor_out;     // uniform xor on read / write (sign flip, descending)
};

__device__ __forceinline__ void even_share(const PassParams &p, uint32_t b, uint32_t &t0, uint32_t &t1)
{
    const uint32_t q = p.num_tiles / p.grid, r = p.num_tiles % p.grid;
    t0 = b * q + (b < r ? b : r);
    t1 = t0 + q + (b < r ? 1u : 0u);
}

// ---------------------------------------------------------------- upsweep --
// Per-block digit histogram of the block's tile range -> spine[d * grid + b].
// Wave-private 256-bin LDS histograms (ds_add_u32, no return value needed).
template <bool VEC>
__global__ __launch_bounds__(LSB_THREADS) void lsb_upsweep_kernel(const uint32_t *__restrict__ keys,
                                                                  uint32_t *__restrict__ spine, PassParams p)
{
    __shared__ uint32_t hist[LSB_WAVES][RADIX];
    const int tid = threadIdx.x, w = wave_id();
    for (int i = tid; i < LSB_WAVES * RADIX; i += LSB_THREADS) (&hist[0][0])[i] = 0;
    uint32_t t0, t1;
    even_share(p, blockIdx.x, t0, t1);
    const uint64_t lo = (uint64_t)t0 * LSB_TILE;
    uint64_t hi = (uint64_t)t1 * LSB_TILE;
    if (hi > p.n) hi = p.n;
    const uint32_t len = (uint32_t)(hi - lo);
    const uint32_t *src = keys + lo;
    __syncthreads();

    uint32_t *my = hist[w];
    auto count = [&](uint32_t raw) {
        const uint32_t k = twiddle_in(raw, p.f32_in, p.xor_in);
        atomicAdd(&my[(k >> p.shift) & p.mask], 1u);
    };
    uint32_t done = 0;
    if (VEC) {
        const uint4 *src4 = reinterpret_cast<const uint4 *>(src);
        const uint32_t nvec = len >> 2;
        uint32_t v = tid;
        for (; v + 3u * LSB_THREADS < nvec; v += 4u * LSB_THREADS) {
            const uint4 a = src4[v], b = src4[v + LSB_THREADS], c = src4[v + 2 * LSB_THREADS],
                        d = src4[v + 3 * LSB_THREADS];
            count(a.x); count(a.y); count(a.z); count(a.w);
            count(b.x); count(b.y); count(b.z); count(b.w);
            count(c.x); count(c.y); count(c.z); count(c.w);
            count(d.x); count(d.y); count(d.z); count(d.w);
        }
        for (; v < nvec; v += LSB_THREADS) {
            const uint4 a = src4[v];
            count(a.x); count(a.y); count(a.z); count(a.w);
        }
        done = nvec << 2;
    }
    for (uint32_t i = done + tid; i < len; i += LSB_THREADS) count(src[i]);
    __syncthreads();
    if (tid < RADIX) {
        uint32_t s = 0;
#pragma unroll
        for (int j = 0; j < LSB_WAVES; ++j) s += hist[j][tid];
        spine[(uint32_t)tid * p.grid + blockIdx.x] = s;
    }
}

// ------------------------------------------------------------------- scan --
// One block per digit row: exclusive prefix over the row's `grid` block counts
// (in place) and the row total.  The 256-entry scan over the totals is done
// in the downsweep prologue, so the spine scan is fully parallel.
__global__ __launch_bounds__(256) void lsb_scan_kernel(uint32_t *__restrict__ spine, uint32_t *__restrict__ totals,
                                                       uint32_t grid)
{
    __shared__ uint32_t scratch[8];
    uint32_t *row = spine + (size_t)blockIdx.x * grid;
    const uint32_t ipt = (grid + 255u) / 256u;
    const uint32_t lo = threadIdx.x * ipt;
    const uint32_t hi = (lo + ipt < grid) ? lo + ipt : grid;
    uint32_t s = 0;
    for (uint32_t i = lo; i < hi; ++i) s += row[i];
    uint32_t total;
    uint32_t run = block_exclusive_scan_256(s, scratch, &total);
    for (uint32_t i = lo; i < hi; ++i) {
        const uint32_t c = row[i];
        row[i] = run;
        run += c;
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = total;
}

// -------------------------------------------------------------- downsweep --
// Stable scatter of the block's tiles.  Per tile:
//   1. wave-striped coalesced load (key i of lane l of wave w sits at
//      tile + w*1024 + i*64 + l, so position order = (w, i, l));
//   2. rank: for each i, ballot-match the digit across the wave, rank within
//      the wave = popcount of lower matching lanes + the wave's running count
//      of that digit (wave-private LDS histogram);
//   3. digit threads turn the 8 wave histograms into tile-absolute bases and
//      advance the block's running global offset per digit;
//   4. keys go to LDS at their tile rank, are read back in rank order and
//      stored to base[digit] + slot: consecutive lanes hit consecutive
//      addresses inside each digit run.  Values follow the same slots.
template <bool HAS_VALUES>
__global__ __launch_bounds__(LSB_THREADS, 4) void lsb_downsweep_kernel(
    const uint32_t *__restrict__ keys_in, uint32_t *__restrict__ keys_out, const uint32_t *__restrict__ vals_in,
    uint32_t *__restrict__ vals_out, const uint32_t *__restrict__ spine, const uint32_t *__restrict__ totals,
    PassParams p)
{
    __shared__ uint32_t whist[LSB_WAVES][RADIX];  // wave-private digit counters / bases
    __shared__ uint32_t stage[LSB_TILE];          // tile in rank order (keys, then values)
    __shared__ uint32_t gbase[RADIX];             // global offset of digit run minus tile-local start
    __shared__ uint32_t scratch[8];

    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    uint32_t t0, t1;
    even_share(p, blockIdx.x, t0, t1);

    // global offset of this block's first key of digit `tid`
    uint32_t bin_offset = 0;
    {
        const uint32_t tot = (tid < RADIX) ? totals[tid] : 0u;
        const uint32_t ex = block_exclusive_scan_256(tot, scratch, nullptr);
        if (tid < RADIX) bin_offset = ex + spine[(uint32_t)tid * p.grid + blockIdx.x];
    }

    uint32_t *my = whist[w];
    for (uint32_t t = t0; t < t1; ++t) {
        const uint64_t tile_base = (uint64_t)t * LSB_TILE;
        const uint32_t valid = (p.n - tile_base < (uint64_t)LSB_TILE) ? (uint32_t)(p.n - tile_base) : (uint32_t)LSB_TILE;
        const uint32_t wbase = (uint32_t)w * (WAVE * LSB_KPT) + lane;
        const uint32_t *kin = keys_in + tile_base;

        uint32_t key[LSB_KPT], val[LSB_KPT], pos[LSB_KPT];
        if (valid == LSB_TILE) {
#pragma unroll
            for (int i = 0; i < LSB_KPT; ++i) key[i] = kin[wbase + i * WAVE];
            if (HAS_VALUES) {
                const uint32_t *vin = vals_in + tile_base;
#pragma unroll
                for (int i = 0; i < LSB_KPT; ++i) val[i] = vin[wbase + i * WAVE];
            }
#pragma unroll
            for (int i = 0; i < LSB_KPT; ++i) key[i] = twiddle_in(key[i], p.f32_in, p.xor_in);
        } else {
            // last, partial tile: pad with all-ones keys (largest digit, ranked
            // after every real key of that digit because they sit at the tail)
#pragma unroll
            for (int i = 0; i < LSB_KPT; ++i) {
                const uint32_t idx = wbase + i * WAVE;
                key[i] = (idx < valid) ? twiddle_in(kin[idx], p.f32_in, p.xor_in) : 0xffffffffu;
            }
            if (HAS_VALUES) {
                const uint32_t *vin = vals_in + tile_base;
#pragma unroll
                for (int i = 0; i < LSB_KPT; ++i) {
                    const uint32_t idx = wbase + i * WAVE;
                    val[i] = (idx < valid) ? vin[idx] : 0u;
                }
            }
        }

        // 2. rank inside the wave
#pragma unroll
        for (int i = lane; i < RADIX; i += WAVE) my[i] = 0;
        __builtin_amdgcn_wave_barrier();
#pragma unroll
        for (int i = 0; i < LSB_KPT; ++i) {
            const uint32_t d = (key[i] >> p.shift) & p.mask;
            const uint64_t peers = match_digit(d);
            const uint32_t lower = count_lower(peers);
            const uint32_t cnt = (uint32_t)__popcll(peers);
            const uint32_t prev = my[d];
            pos[i] = prev + lower;
            __builtin_amdgcn_wave_barrier();
            if (lower == cnt - 1u) my[d] = prev + cnt;
            __builtin_amdgcn_wave_barrier();
        }
        __syncthreads();

        // 3. wave histograms -> tile-absolute bases; advance global offsets
        {
            uint32_t c[LSB_WAVES], tot = 0;
            if (tid < RADIX) {
#pragma unroll
                for (int j = 0; j < LSB_WAVES; ++j) { c[j] = whist[j][tid]; }
#pragma unroll
                for (int j = 0; j < LSB_WAVES; ++j) { const uint32_t x = c[j]; c[j] = tot; tot += x; }
            }
            const uint32_t ex = block_exclusive_scan_256(tot, scratch, nullptr);
            if (tid < RADIX) {
#pragma unroll
                for (int j = 0; j < LSB_WAVES; ++j) whist[j][tid] = c[j] + ex;
                gbase[tid] = bin_offset - ex;
                bin_offset += tot;
            }
        }
        __syncthreads();

        // 4. keys -> LDS in rank order -> global
#pragma unroll
        for (int i = 0; i < LSB_KPT; ++i) {
            const uint32_t d = (key[i] >> p.shift) & p.mask;
            pos[i] += my[d];
            stage[pos[i]] = key[i];
        }
        __syncthreads();
        uint32_t dst[LSB_KPT];
#pragma unroll
        for (int i = 0; i < LSB_KPT; ++i) {
            const uint32_t slot = (uint32_t)tid + i * LSB_THREADS;
            const uint32_t k = stage[slot];
            dst[i] = gbase[(k >> p.shift) & p.mask] + slot;
            if (slot < valid) keys_out[dst[i]] = twiddle_out(k, p.f32_out, p.xor_out);
        }
        if (HAS_VALUES) {
            __syncthreads();
#pragma unroll
            for (int i = 0; i < LSB_KPT; ++i) stage[pos[i]] = val[i];
            __syncthreads();
#pragma unroll
            for (int i = 0; i < LSB_KPT; ++i) {
                const uint32_t slot = (uint32_t)tid + i * LSB_THREADS;
                if (slot < valid) vals_out[dst[i]] = stage[slot];
            }
        }
        __syncthreads();
    }
}

// ------------------------------------------------------------------- host --

static inline uint32_t lsb_num_tiles(uint64_t n) { return (uint32_t)((n + LSB_TILE - 1) / LSB_TILE); }
static inline uint32_t lsb_grid(uint64_t n)
{
    const uint32_t t = lsb_num_tiles(n);
    return t < LSB_MAX_GRID ? (t ? t : 1u) : LSB_MAX_GRID;
}

static void twiddle_masks(int key_type, int descending, bool first, bool last, PassParams &p)
{
    // keys are stored twiddled between passes; the first pass maps in, the last maps out
    const uint32_t sign = (key_type == GS_KEY_I32) ? 0x80000000u : 0u;
    const uint32_t flip = descending ? 0xffffffffu : 0u;
    p.f32_in = (first && key_type == GS_KEY_F32) ? 1 : 0;
    p.f32_out = (last && key_type == GS_KEY_F32) ? 1 : 0;
    p.xor_in = first ? (sign ^ flip) : 0u;
    p.xor_out = last ? (sign ^ flip) : 0u;
}

static PassParams make_params(uint64_t n, int shift, int bits)
{
    PassParams p{};
    p.n = (uint32_t)n;
    p.num_tiles = lsb_num_tiles(n);
    p.grid = lsb_grid(n);
    p.shift = shift;
    p.mask = (1u << bits) - 1u;
    return p;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static inline size_t spine_bytes(uint64_t n) { return align256((size_t)RADIX * lsb_grid(n) * sizeof(uint32_t)); }

int lsb_upsweep(const uint32_t *keys, uint32_t *spine, const PassParams &p, hipStream_t s)
{
    const bool vec = ((uintptr_t)keys & 15u) == 0;
    KernelTimer kt(GS_K_LSB_UPSWEEP, s);
    if (vec) hipLaunchKernelGGL(lsb_upsweep_kernel<true>, dim3(p.grid), dim3(LSB_THREADS), 0, s, keys, spine, p);
    else hipLaunchKernelGGL(lsb_upsweep_kernel<false>, dim3(p.grid), dim3(LSB_THREADS), 0, s, keys, spine, p);
    return (int)hipGetLastError();
}

int lsb_scan(uint32_t *spine, uint32_t *totals, uint32_t grid, hipStream_t s)
{
    KernelTimer kt(GS_K_LSB_SCAN, s);
    hipLaunchKernelGGL(lsb_scan_kernel, dim3(RADIX), dim3(256), 0, s, spine, totals, grid);
    return (int)hipGetLastError();
}

int lsb_downsweep(const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, const uint32_t *spine,
                  const uint32_t *totals, const PassParams &p, hipStream_t s)
{
    KernelTimer kt(GS_K_LSB_DOWNSWEEP, s);
    if (vin)
        hipLaunchKernelGGL(lsb_downsweep_kernel<true>, dim3(p.grid), dim3(LSB_THREADS), 0, s, kin, kout, vin, vout,
                           spine, totals, p);
    else
        hipLaunchKernelGGL(lsb_downsweep_kernel<false>, dim3(p.grid), dim3(LSB_THREADS), 0, s, kin, kout, vin, vout,
                           spine, totals, p);
    return (int)hipGetLastError();
}

}  // namespace gs

using namespace gs;

extern "C" {

size_t gs_lsb_temp_bytes(uint64_t num_items, int /*has_values*/)
{
    return spine_bytes(num_items) + align256(RADIX * sizeof(uint32_t));
}

void gs_lsb_geometry(uint64_t num_items, int /*has_values*/, uint32_t *grid, uint32_t *tile)
{
    if (grid) *grid = lsb_grid(num_items);
    if (tile) *tile = LSB_TILE;
}

int gs_lsb_upsweep_u32(const uint32_t *d_keys_in, uint32_t *d_spine, uint64_t num_items, int shift, int bits,
                       int descending, int key_type_in, void *stream)
{
    if (num_items >= (1ull << 32) || bits < 1 || bits > 8 || shift < 0 || shift + bits > 32) return hipErrorInvalidValue;
    if (num_items == 0) return hipSuccess;
    PassParams p = make_params(num_items, shift, bits);
    twiddle_masks(key_type_in, descending, true, true, p);
    return lsb_upsweep(d_keys_in, d_spine, p, (hipStream_t)stream);
}

int gs_lsb_scan_spine(uint32_t *d_spine, uint32_t *d_totals, uint64_t num_items, int /*has_values*/, void *stream)
{
    if (num_items >= (1ull << 32)) return hipErrorInvalidValue;
    if (num_items == 0) return hipSuccess;
    return lsb_scan(d_spine, d_totals, lsb_grid(num_items), (hipStream_t)stream);
}

int gs_lsb_downsweep_u32(const uint32_t *d_keys_in, uint32_t *d_keys_out, const uint32_t *d_vals_in,
                         uint32_t *d_vals_out, const uint32_t *d_spine, const uint32_t *d_totals, uint64_t num_items,
                         int shift, int bits, int descending, int key_type_in, int key_type_out, void *stream)
{
    if (num_items >= (1ull << 32) || bits < 1 || bits > 8 || shift < 0 || shift + bits > 32) return hipErrorInvalidValue;
    if ((d_vals_in == nullptr) != (d_vals_out == nullptr)) return hipErrorInvalidValue;
    if (num_items == 0) return hipSuccess;
    PassParams p = make_params(num_items, shift, bits);
    PassParams in{}, out{};
    twiddle_masks(key_type_in, descending, true, false, in);
    twiddle_masks(key_type_out, descending, false, true, out);
    p.f32_in = in.f32_in; p.xor_in = in.xor_in;
    p.f32_out = out.f32_out; p.xor_out = out.xor_out;
    return lsb_downsweep(d_keys_in, d_keys_out, d_vals_in, d_vals_out, d_spine, d_totals, p, (hipStream_t)stream);
}

int gs_lsb_sort_u32(void *d_temp, size_t temp_bytes, uint32_t *d_keys[2], uint32_t *d_vals[2], int *selector,
                    uint64_t num_items, int begin_bit, int end_bit, int descending, int key_type, void *stream)
{
    if (!selector || (*selector != 0 && *selector != 1) || !d_keys) return hipErrorInvalidValue;
    if (begin_bit < 0 || end_bit > 32 || begin_bit > end_bit) return hipErrorInvalidValue;
    if (num_items >= (1ull << 32)) return hipErrorInvalidValue;
    if (key_type < GS_KEY_U32 || key_type > GS_KEY_F32) return hipErrorInvalidValue;
    if (num_items == 0 || begin_bit == end_bit) return hipSuccess;
    if (!d_temp || temp_bytes < gs_lsb_temp_bytes(num_items, d_vals != nullptr)) return hipErrorInvalidValue;
    if (!d_keys[0] || !d_keys[1] || (d_vals && (!d_vals[0] || !d_vals[1]))) return hipErrorInvalidValue;

    hipStream_t s = (hipStream_t)stream;
    uint32_t *spine = (uint32_t *)d_temp;
    uint32_t *totals = (uint32_t *)((char *)d_temp + spine_bytes(num_items));
    const int num_bits = end_bit - begin_bit;
    const int num_passes = (num_bits + RADIX_BITS - 1) / RADIX_BITS;
    int sel = *selector;
    for (int pass = 0; pass < num_passes; ++pass) {
        const int shift = begin_bit + pass * RADIX_BITS;
        const int bits = (end_bit - shift < RADIX_BITS) ? end_bit - shift : RADIX_BITS;
        PassParams p = make_params(num_items, shift, bits);
        twiddle_masks(key_type, descending, pass == 0, pass == num_passes - 1, p);
        const uint32_t *kin = d_keys[sel];
        uint32_t *kout = d_keys[sel ^ 1];
        const uint32_t *vin = d_vals ? d_vals[sel] : nullptr;
        uint32_t *vout = d_vals ? d_vals[sel ^ 1] : nullptr;
        int e;
        if ((e = lsb_upsweep(kin, spine, p, s))) return e;
        if ((e = lsb_scan(spine, totals, p.grid, s))) return e;
        if ((e = lsb_downsweep(kin, kout, vin, vout, spine, totals, p, s))) return e;
        sel ^= 1;
    }
    *selector = sel;
    return hipSuccess;
}

}  // extern "C"
