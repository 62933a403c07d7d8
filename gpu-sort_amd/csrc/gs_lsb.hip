// gs_lsb.hip -- stable LSD radix sort for gfx950 (MI355X), 8-bit digits, three
// kernels per pass: upsweep histogram -> spine scan -> downsweep scatter.
//
// Replaces (behaviour, not code) the CUB path the LSB driver calls:
//   cub::DeviceRadixSort::Sort{Keys,Pairs}[Descending]  lsb/cub/cub/device/device_radix_sort.cuh:248,595,754
//   DispatchRadixSort::InvokePasses / InvokePass        lsb/cub/cub/device/dispatch/dispatch_radix_sort.cuh:899-1166
//   AgentRadixSortUpsweep / RadixSortScanBins / AgentRadixSortDownsweep
//
// Decomposition.  The keys are cut into tiles of 8192; 8
// consecutive tiles form a chunk.
//   upsweep   one block per chunk, one WAVE per tile: per-tile digit counts in
//             a wave-private LDS histogram.  Output: spine[d][chunk] = keys of
//             digit d in the chunk (u32) and prefix16[tile][d] = keys of digit d
//             in the chunk's earlier tiles (u16), so that every tile knows its
//             own global offsets while the scanned structure stays chunk-sized.
//   scan      one block per digit row of the spine + digit totals.
//   downsweep one block per TILE.  Blocks are dispatched in order, so the
//             resident blocks work on consecutive tiles and the blocks of one
//             XCD on a contiguous slice of them: for each digit the chip writes
//             one compact window at a time and neighbouring runs meet in the
//             same L2, while the blocks drift out of phase so loads, ranking
//             and stores of different blocks overlap.  Measured on MI355X:
//             2.05 ms per pass at 2^30 keys, against 3.2 ms for persistent
//             blocks that each own a long run of tiles and march in lockstep.
#include "gs_device.hpp"
#include "gs_lsb.hpp"
#include <cstdlib>
#include <cstring>

namespace gs {

// ---------------------------------------------------------------- upsweep --
#ifndef UPSWEEP_BATCH
#define UPSWEEP_BATCH 32   // dword loads in flight per lane (a tile is 128 per lane)
#endif
#ifndef UPSWEEP_SUB
#define UPSWEEP_SUB 4      // histogram copies per wave (power of two)
#endif
// Plain dword loads in batches beat 16-byte loads here (0.81 vs 0.84 ms at 2^30) and need no alignment.
__global__ __launch_bounds__(LSB_THREADS) void lsb_upsweep_kernel(const uint32_t *__restrict__ keys,
                                                                  uint32_t *__restrict__ spine,
                                                                  uint16_t *__restrict__ prefix16, PassParams p)
{
    // every wave counts into UPSWEEP_SUB copies of its histogram (lane & 3 picks one; rows padded by one word so
    // equal digits of different copies sit in different banks): under skew the lanes that share a hot digit
    // spread over four banks instead of queueing on one (Zipf keys: 1.40 -> ~1.0 ms at level 1 of the MSB sort)
    __shared__ uint32_t hist[LSB_WAVES][UPSWEEP_SUB][RADIX + 1];
    const int tid = threadIdx.x, w = wave_id(), lane = lane_id();
    uint32_t *my = hist[w][lane & (UPSWEEP_SUB - 1)];
    for (int i = lane; i < UPSWEEP_SUB * (RADIX + 1); i += WAVE) (&hist[w][0][0])[i] = 0;

    const uint32_t chunk = blockIdx.x;
    const uint32_t tile = chunk * LSB_CHUNK + (uint32_t)w;
    if (tile < p.num_tiles) {
        const uint64_t lo = (uint64_t)tile * LSB_TILE;
        const uint32_t len = (p.n - lo < (uint64_t)LSB_TILE) ? (uint32_t)(p.n - lo) : (uint32_t)LSB_TILE;
        const uint32_t *src = keys + lo;
        auto count = [&](uint32_t raw) {
            const uint32_t k = twiddle_in(raw, p.f32_in, p.xor_in);
            hist_add(my, __builtin_amdgcn_ubfe(k, p.shift, p.bits));        // wave-private ds_add_u32
        };
        // batches of dword loads from clamped indices: one code path for full, partial and misaligned tiles
        // (measured as fast as an unclamped unrolled variant; a loop of one guarded load per trip would pay
        // one HBM round trip per 64 keys)
        constexpr int GB = UPSWEEP_BATCH;
        const uint32_t last = len - 1u;
#pragma unroll 1
        for (uint32_t j = 0; j < len; j += GB * WAVE) {
            uint32_t v[GB];
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                const uint32_t idx = j + u * WAVE + lane;
                v[u] = __builtin_nontemporal_load(&src[idx < last ? idx : last]);   // streaming hint: 0.80 -> 0.76 ms
            }
#pragma unroll
            for (int u = 0; u < GB; ++u)
                if (j + u * WAVE + lane < len) count(v[u]);
        }
    }
    __syncthreads();
    if (tid < RADIX) {
        uint32_t run = 0;
#pragma unroll
        for (int j = 0; j < LSB_WAVES; ++j) {
            const uint32_t t = chunk * LSB_CHUNK + (uint32_t)j;
            if (t < p.num_tiles) prefix16[(size_t)t * RADIX + tid] = (uint16_t)run;
            uint32_t c = 0;
#pragma unroll
            for (int q = 0; q < UPSWEEP_SUB; ++q) c += hist[j][q][tid];
            run += c;
        }
        spine[(uint32_t)tid * p.grid + chunk] = run;
    }
}

// ---- small arrays (up to LSB_SMALL_TILES tiles): with one wave per tile a handful of waves would each
// walk 8192 keys in four dependent batches (15 us however small the array).  Here a whole workgroup counts
// one tile (16 keys per thread, one batch) and leaves the tile's RAW counts in prefix16; the scan kernel
// below turns them into the chunk-relative prefixes and the spine the downsweep expects.
constexpr uint32_t LSB_SMALL_TILES = 2048;    // 16 Mi keys
__global__ __launch_bounds__(LSB_THREADS) void lsb_upsweep_small_kernel(const uint32_t *__restrict__ keys,
                                                                        uint16_t *__restrict__ prefix16, PassParams p)
{
    __shared__ uint32_t hist[LSB_WAVES][RADIX];
    const int tid = threadIdx.x, w = wave_id(), lane = lane_id();
    uint32_t *my = hist[w];
#pragma unroll
    for (int i = lane; i < RADIX; i += WAVE) my[i] = 0;
    const uint32_t tile = blockIdx.x;
    const uint64_t lo = (uint64_t)tile * LSB_TILE;
    const uint32_t len = (p.n - lo < (uint64_t)LSB_TILE) ? (uint32_t)(p.n - lo) : (uint32_t)LSB_TILE;
    const uint32_t *src = keys + lo;
    const uint32_t last = len - 1u;
    uint32_t v[LSB_KPT];
#pragma unroll
    for (int u = 0; u < LSB_KPT; ++u) {
        const uint32_t idx = (uint32_t)w * (WAVE * LSB_KPT) + u * WAVE + lane;
        v[u] = src[idx < last ? idx : last];
    }
#pragma unroll
    for (int u = 0; u < LSB_KPT; ++u) {
        const uint32_t idx = (uint32_t)w * (WAVE * LSB_KPT) + u * WAVE + lane;
        if (idx < len) hist_add(my, __builtin_amdgcn_ubfe(twiddle_in(v[u], p.f32_in, p.xor_in), p.shift, p.bits));
    }
    __syncthreads();
    if (tid < RADIX) {
        uint32_t c = 0;
#pragma unroll
        for (int j = 0; j < LSB_WAVES; ++j) c += hist[j][tid];
        prefix16[(size_t)tile * RADIX + tid] = (uint16_t)c;        // <= 8192: a full tile of one digit is 0x2000
    }
}

// one block per digit, thread c = chunk c (grid <= 256): raw tile counts -> in-chunk exclusive prefixes
// (in place), chunk totals -> exclusive scan over the chunks (spine row) and the digit total
__global__ __launch_bounds__(RADIX) void lsb_scan_small_kernel(uint32_t *__restrict__ spine, uint32_t *__restrict__ totals,
                                                               uint16_t *__restrict__ prefix16, uint32_t grid, uint32_t num_tiles)
{
    __shared__ uint32_t scratch[8];
    const uint32_t d = blockIdx.x, c = threadIdx.x;
    uint32_t run = 0;
    if (c < grid) {
#pragma unroll
        for (int j = 0; j < LSB_CHUNK; ++j) {
            const uint32_t t = c * LSB_CHUNK + j;
            if (t < num_tiles) {
                uint16_t *q = prefix16 + (size_t)t * RADIX + d;
                const uint32_t cnt = *q;
                *q = (uint16_t)run;
                run += cnt;
            }
        }
    }
    uint32_t total = 0;
    const uint32_t ex = block_exclusive_scan_256(run, scratch, &total);
    if (c < grid) spine[(size_t)d * grid + c] = ex;
    if (c == 0) totals[d] = total;
}

// ------------------------------------------------- single-sweep histogram --
// Digit totals of ALL passes in one read of the keys (single-sweep mode): the digit of
// pass q is taken from the twiddled key, which is what the later passes see.
struct Hist4Params {
    uint32_t n;
    int num_passes;
    uint32_t shift[4], bits[4];
    int f32_in;
    uint32_t xor_in;
};
template <bool VEC>
__global__ __launch_bounds__(LSB_THREADS) void lsb_hist4_kernel(const uint32_t *__restrict__ keys,
                                                                uint32_t *__restrict__ totals4, Hist4Params hp)
{
    __shared__ uint32_t h[LSB_WAVES][4][RADIX];
    const int tid = threadIdx.x, w = wave_id();
    for (int i = tid; i < LSB_WAVES * 4 * RADIX; i += LSB_THREADS) (&h[0][0][0])[i] = 0;
    __syncthreads();
    auto count = [&](uint32_t raw) {
        const uint32_t k = twiddle_in(raw, hp.f32_in, hp.xor_in);
#pragma unroll
        for (int q = 0; q < 4; ++q)
            if (q < hp.num_passes) hist_add(h[w][q], __builtin_amdgcn_ubfe(k, hp.shift[q], hp.bits[q]));
    };
    const uint32_t stride = gridDim.x * LSB_THREADS;
    uint32_t done = 0;
    if (VEC) {
        const uint4 *k4 = reinterpret_cast<const uint4 *>(keys);
        const uint32_t nvec = hp.n >> 2;
        uint32_t v = blockIdx.x * LSB_THREADS + tid;
        for (; v + 3u * stride < nvec && v + 3u * stride >= v; v += 4u * stride) {
            const uint4 a = k4[v], b = k4[v + stride], c = k4[v + 2 * stride], d = k4[v + 3 * stride];
            count(a.x); count(a.y); count(a.z); count(a.w);
            count(b.x); count(b.y); count(b.z); count(b.w);
            count(c.x); count(c.y); count(c.z); count(c.w);
            count(d.x); count(d.y); count(d.z); count(d.w);
        }
        for (; v < nvec; v += stride) { const uint4 a = k4[v]; count(a.x); count(a.y); count(a.z); count(a.w); }
        done = nvec << 2;
    }
    for (uint64_t i = (uint64_t)done + blockIdx.x * LSB_THREADS + tid; i < hp.n; i += stride) count(keys[i]);
    __syncthreads();
    for (int i = tid; i < 4 * RADIX; i += LSB_THREADS) {
        uint32_t s = 0;
#pragma unroll
        for (int j = 0; j < LSB_WAVES; ++j) s += (&h[j][0][0])[i];
        if (s) atomicAdd(&totals4[i], s);
    }
}

// ------------------------------------------------------------------- scan --
// One block per digit row: exclusive prefix over the row's `grid` chunk counts
// (in place) and the row total.  The 256-entry scan over the totals is done
// in the downsweep prologue, so the spine scan is fully parallel.  The row is
// swept in coalesced segments of 1024 entries with a running carry.
constexpr int SCAN_THREADS = 1024;
__global__ __launch_bounds__(SCAN_THREADS) void lsb_scan_kernel(uint32_t *__restrict__ spine,
                                                                uint32_t *__restrict__ totals, uint32_t grid)
{
    __shared__ uint32_t wsum[SCAN_THREADS / WAVE];
    uint32_t *row = spine + (size_t)blockIdx.x * grid;
    const int w = wave_id(), lane = lane_id();
    uint32_t carry = 0;
    for (uint32_t base = 0; base < grid; base += SCAN_THREADS) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t c = (i < grid) ? row[i] : 0u;
        const uint32_t inc = wave_inclusive_scan(c);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        // 16 wave sums: every wave scans them redundantly in its first 16 lanes
        const uint32_t ws = (lane < SCAN_THREADS / WAVE) ? wsum[lane] : 0u;
        const uint32_t wincl = wave_inclusive_scan(ws);
        const uint32_t wbase_ = __shfl(wincl - ws, w, WAVE);
        const uint32_t seg_total = __shfl(wincl, SCAN_THREADS / WAVE - 1, WAVE);
        if (i < grid) row[i] = carry + wbase_ + inc - c;
        carry += seg_total;
        __syncthreads();
    }
    if (threadIdx.x == 0) totals[blockIdx.x] = carry;
}

// -------------------------------------------------------------- downsweep --
// Stable scatter, one tile at a time:
//   1. wave-striped coalesced load (key i of lane l of wave w sits at
//      tile + w*1024 + i*64 + l, so position order = (w, i, l)); HBM latency
//      is covered by the other blocks resident on the CU;
//   2. rank inside the wave: the set of lanes holding the same digit (ballot
//      match) gives the rank inside the group by popcount of the lower lanes;
//      the wave's running count of the digit (wave-private LDS histogram) gives
//      the rank of the group.  Every lane reads the count, the first lane of the
//      group then adds the group size with a no-return LDS atomic; LDS executes
//      one wave's operations in order, so round i+1 sees round i's add without a
//      wait.  The match set comes either from 8 VALU ballots (match_digit) or
//      from LDS: each lane ORs its lane bit into the wave's mask entry of its
//      digit, reads the entry back and clears its bit again.  Both are exact;
//      `valu_rounds` splits the 16 rounds between the two pipes;
//   3. the 8 wave histograms become tile-absolute bases per (wave, digit) (4 digits
//      per lane, b128 LDS accesses, DPP scan), by wave 0 alone (keys only, 3
//      blocks/CU) or redundantly by every wave for its own row, which removes a
//      barrier and the serial section (pairs, 2 blocks/CU); wave 0 publishes, per
//      digit, the tile's global base = digit start + scanned chunk count + prefix16;
//   4. keys (and values) go to LDS at their tile rank and are read back in rank
//      order: consecutive lanes hit consecutive addresses inside a digit run.
// Two (pairs) or three (keys only) block barriers per tile.
template <bool HAS_VALUES>
struct DownsweepSmem {
    uint32_t whist[LSB_WAVES][RADIX];                     // wave-private digit counters, then bases (byte offsets)
    uint16_t wbase[HAS_VALUES ? LSB_WAVES : 1][RADIX];    // pairs: tile-absolute base of (wave, digit), < 8192
    uint32_t gbase[RADIX];                                // global offset of digit run - tile-local start
    uint32_t stage[LSB_TILE * (HAS_VALUES ? 2 : 1)];      // tile in rank order; pairs interleaved {key,val}
};

#ifdef GS_EXP_PHASES
// experiment builds only (tools/phase_exp.py): shader-clock length of every phase of wave 0
__device__ uint32_t gs_phase_buf[131072 * 16];   // [block][phase], n <= 2^30
#define GS_PHASE(k)                                                                          \
    do {                                                                                     \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                        \
        if (tid == 0 && blockIdx.x < 131072u) gs_phase_buf[blockIdx.x * 16 + (k)] = (uint32_t)(now_ - tprev_); \
        tprev_ = now_;                                                                       \
    } while (0)
#define GS_PHASE_WAIT(what) asm volatile("s_waitcnt " what ::: "memory")
#else
#define GS_PHASE(k) do { } while (0)
#define GS_PHASE_WAIT(what) do { } while (0)
#endif

// The kernel is VALU-bound on MI355X (about 900 vector instructions per wave and tile, 57 % of
// them the ballot match; measured with tools/phase_exp.py and an ISA count), so the template
// parameters exist to keep instructions out of the hot variants:
// TAIL = false: one of the array's FULL tiles.
// TAIL = true: one block handles the last, partial tile (guarded loads); being
// last in key order, its keys of digit d sit at the very end of digit d's global
// range, so it needs only the digit totals.  Splitting it off keeps the guarded
// path's registers out of the hot kernel.
// TW: key transform on read / write.  0 = none (u32 ascending, and every middle pass: keys
// travel twiddled between passes), 1 = xor mask (signed keys, descending), 2 = float + xor.
// BIG = false: n <= 2^30, so byte offsets into the output fit 32 bits and a store needs no
// 64-bit address arithmetic.
// FUSED = true (single-sweep mode, see the host section): no upsweep/scan ran for this pass;
// the tile learns its global offsets by decoupled look-back over `status`, one 32-bit word
// per (tile, digit): bits 31:30 = 0 empty / 1 tile count / 2 inclusive prefix, bits 29:0 the
// value.  Each word is one self-validating granule written by one relaxed agent-scope store
// and read by relaxed agent-scope loads (cdna_hip_programming.md Guideline 16, form R2).
constexpr uint32_t ST_AGG = 1u << 30, ST_INC = 2u << 30, ST_VAL = (1u << 30) - 1u;

template <bool HAS_VALUES, bool TAIL, int TW, bool BIG, bool FUSED = false>
__global__ __launch_bounds__(LSB_THREADS, HAS_VALUES ? 4 : 6) void lsb_downsweep_kernel(
    const uint32_t *__restrict__ keys_in, uint32_t *__restrict__ keys_out, const uint32_t *__restrict__ vals_in,
    uint32_t *__restrict__ vals_out, const uint32_t *__restrict__ spine, const uint16_t *__restrict__ prefix16,
    const uint32_t *__restrict__ totals, PassParams p, uint32_t *__restrict__ status = nullptr,
    uint32_t *__restrict__ error_word = nullptr)
{
    __shared__ __attribute__((aligned(16))) DownsweepSmem<HAS_VALUES> sm;
    constexpr bool ALLWAVE = HAS_VALUES;   // see step 3

    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const uint32_t full_tiles = p.n / (uint32_t)LSB_TILE;
    auto tw_in = [&](uint32_t k) { return TW == 0 ? k : twiddle_in(k, TW == 2 ? p.f32_in : 0, p.xor_in); };
    auto tw_out = [&](uint32_t k) { return TW == 0 ? k : twiddle_out(k, TW == 2 ? p.f32_out : 0, p.xor_out); };
    // the digit width lives in a vector register: v_bfe_u32 takes one scalar operand (the shift)
    uint32_t wbits = p.bits;
    asm volatile("" : "+v"(wbits));
    auto digit = [&](uint32_t k) { return __builtin_amdgcn_ubfe(k, p.shift, wbits); };

    uint32_t *my = sm.whist[w];
    const uint16_t *mybase = sm.wbase[w];
    const uint32_t wbase = (uint32_t)w * (WAVE * LSB_KPT) + lane;
    const uint32_t tail_valid = p.n - full_tiles * (uint32_t)LSB_TILE;   // used when TAIL

    if (!TAIL && blockIdx.x >= full_tiles) return;
    const uint32_t t = TAIL ? full_tiles : tile_of_item(blockIdx.x, full_tiles);
#ifdef GS_EXP_PHASES
    unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
    const unsigned long long t0_ = tprev_, r0_ = __builtin_amdgcn_s_memrealtime();
#endif
    // keys only: the waves that issue loads and stores get priority over the ones that rank, so the memory
    // pipes are fed as early as possible (1.87 -> 1.82 ms; with values it costs 7 %, so pairs keep the default)
    if (!HAS_VALUES) __builtin_amdgcn_s_setprio(3);
    const uint64_t tile_base = (uint64_t)t * LSB_TILE;
    const uint32_t valid = TAIL ? tail_valid : (uint32_t)LSB_TILE;

    // 1. wave-striped coalesced load
    uint32_t key[LSB_KPT], val[HAS_VALUES ? LSB_KPT : 1], pos[LSB_KPT];
    {
        const uint32_t *kin = keys_in + tile_base;
        if (!TAIL) {
#pragma unroll
            for (int i = 0; i < LSB_KPT; ++i) key[i] = kin[wbase + i * WAVE];
        } else {
            // pad with keys whose twiddled form is all ones (largest digit; ranked after
            // every real key of that digit because they sit at the tail)
            const uint32_t pad = twiddle_out(0xffffffffu, TW == 2 ? p.f32_in : 0, TW ? p.xor_in : 0u);
#pragma unroll
            for (int i = 0; i < LSB_KPT; ++i) {
                const uint32_t idx = wbase + i * WAVE;
                key[i] = pad;
                if (idx < tail_valid) key[i] = kin[idx];
            }
        }
    }
    GS_PHASE(0);                                   // load issue
    if (!HAS_VALUES) __builtin_amdgcn_s_setprio(0);
    if (HAS_VALUES) {
        const uint32_t *vin = vals_in + tile_base;
#pragma unroll
        for (int i = 0; i < LSB_KPT; ++i) {
            const uint32_t idx = wbase + i * WAVE;
            val[i] = 0;
            if (!TAIL || idx < valid) val[i] = vin[idx];
        }
    }
    // (after the key loads are in flight) wave 0, lane l: global start of digits 4l..4l+3 (exclusive scan of the totals)
    // (TAIL: inclusive scan; the tile's own counts are subtracted later)
    uint32_t dstart[4] = {0, 0, 0, 0};
    if (w == 0) {
        const uint4 tot = reinterpret_cast<const uint4 *>(totals)[lane];
        const uint32_t lane_sum = tot.x + tot.y + tot.z + tot.w;
        const uint32_t ex = wave_inclusive_scan(lane_sum) - lane_sum;
        dstart[0] = ex + (TAIL ? tot.x : 0u);
        dstart[1] = dstart[0] + (TAIL ? tot.y : tot.x);
        dstart[2] = dstart[1] + (TAIL ? tot.z : tot.y);
        dstart[3] = dstart[2] + (TAIL ? tot.w : tot.z);
    }

    // this tile's global offsets (wave 0): scanned chunk count + count of the chunk's earlier tiles
    uint32_t tbase[4] = {0, 0, 0, 0};
    if (!TAIL && !FUSED && w == 0) {
        const uint32_t *sp = spine + (uint32_t)(4 * lane) * p.grid + t / LSB_CHUNK;
        const uint2 pf = reinterpret_cast<const uint2 *>(prefix16 + (size_t)t * RADIX)[lane];
        tbase[0] = sp[0] + (pf.x & 0xffffu);
        tbase[1] = sp[p.grid] + (pf.x >> 16);
        tbase[2] = sp[2 * p.grid] + (pf.y & 0xffffu);
        tbase[3] = sp[3 * p.grid] + (pf.y >> 16);
    }

    // global base of digit run = digit start + tile offset - tile-local start (wave 0, lane l: digits 4l..4l+3)
    auto publish_gbase = [&](const uint32_t (&ex)[4], const uint32_t (&run)[4]) {
        if (FUSED && !TAIL) {
            // decoupled look-back (wave 0, lane l owns digits 4l..4l+3): publish this tile's
            // counts, add up the predecessors' words walking backwards until an inclusive
            // prefix is met, publish the own inclusive prefix.  Tiles are dispatched in order,
            // so predecessors are resident or finished; every spin is bounded all the same.
            uint32_t *mine = status + (size_t)t * RADIX + 4 * lane;
            if (t == 0) {
#pragma unroll
                for (int q = 0; q < 4; ++q) __hip_atomic_store(mine + q, ST_INC | run[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            } else {
#pragma unroll
                for (int q = 0; q < 4; ++q) __hip_atomic_store(mine + q, ST_AGG | run[q], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                uint32_t jq[4] = {t - 1, t - 1, t - 1, t - 1};
                bool done[4] = {false, false, false, false};
                uint32_t spins = 0;
                for (;;) {
                    uint32_t e[4];
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        e[q] = done[q] ? 0u
                                       : __hip_atomic_load(status + (size_t)jq[q] * RADIX + 4 * lane + q, __ATOMIC_RELAXED,
                                                           __HIP_MEMORY_SCOPE_AGENT);
                    bool waiting = false;
#pragma unroll
                    for (int q = 0; q < 4; ++q) {
                        if (done[q]) continue;
                        const uint32_t f = e[q] >> 30;
                        if (f == 0) { waiting = true; continue; }           // not published yet: retry
                        tbase[q] += e[q] & ST_VAL;
                        if (f == 2) done[q] = true; else --jq[q];             // tile 0 always publishes inclusive
                    }
                    const bool all_done = done[0] && done[1] && done[2] && done[3];
                    if (__builtin_amdgcn_ballot_w64(!all_done) == 0) break;
                    if (__builtin_amdgcn_ballot_w64(waiting) != 0) {
                        __builtin_amdgcn_s_sleep(2);
                        if (++spins > (1u << 22)) {                          // never hang the GPU
                            if (lane == 0 && error_word) atomicOr(error_word, 1u);
                            break;
                        }
                    }
                }
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    __hip_atomic_store(mine + q, ST_INC | ((tbase[q] + run[q]) & ST_VAL), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            }
        }
        uint32_t g[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = dstart[q] + tbase[q] - ex[q];
        if (TAIL) {   // keys of digit d end exactly at the inclusive total of d
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] -= run[q];
            // padded keys inflate the count of the largest digit only, and they are never stored
            const uint32_t pads = (uint32_t)LSB_TILE - valid, dmax = p.mask;
            if (lane == (int)(dmax >> 2)) {
#pragma unroll
                for (int q = 0; q < 4; ++q)
                    if ((dmax & 3u) == (uint32_t)q) g[q] += pads;
            }
        }
        if (!BIG) {   // byte offsets (mod 2^32; exact once the slot is added)
#pragma unroll
            for (int q = 0; q < 4; ++q) g[q] <<= 2;
        }
        reinterpret_cast<uint4 *>(sm.gbase)[lane] = make_uint4(g[0], g[1], g[2], g[3]);
    };

    // 2. rank inside the wave (the LDS count of round i is consumed one round later, so
    //    its latency hides behind the match of round i+1)
#pragma unroll
    for (int i = lane; i < RADIX; i += WAVE) my[i] = 0;
    GS_PHASE_WAIT("vmcnt(0)");
    GS_PHASE(1);                                   // load wait
#pragma unroll
    for (int i = 0; i < LSB_KPT; ++i) key[i] = tw_in(key[i]);
    {
        uint32_t d_prev = 0, plo = 0, phi = 0;
#pragma unroll
        for (int i = 0; i <= LSB_KPT; ++i) {
            uint32_t d_cur = 0, clo = 0, chi = 0;
            if (i < LSB_KPT) {
                d_cur = digit(key[i]);
                match_digit(d_cur, clo, chi);
            }
            if (i > 0) {
                const uint32_t lower = count_lower(plo, phi);
                pos[i - 1] = my[d_prev] + lower;            // LDS read, all lanes
                if (lower == 0)                             // first lane of the group adds the group size
                    __hip_atomic_fetch_add(&my[d_prev], (uint32_t)(__popc(plo) + __popc(phi)), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            d_prev = d_cur; plo = clo; phi = chi;
        }
    }
#pragma unroll
    for (int i = 0; i < LSB_KPT; ++i) {
        // finish the adds before the barrier, and make the keys opaque so their LDS
        // histogram addresses are recomputed after the barrier instead of kept live
        asm volatile("" : "+v"(pos[i]), "+v"(key[i]));
    }
    GS_PHASE_WAIT("lgkmcnt(0)");
    GS_PHASE(2);                                   // rank
    __syncthreads();
    GS_PHASE(3);                                   // barrier 1

    // 3. wave histograms -> tile-absolute base of every (wave, digit) + global base per digit.
    //    4 digits per lane, b128 LDS accesses, DPP scan of the 256 digit totals.
    if constexpr (ALLWAVE) {
        // every wave sums the 8 rows and keeps only its own row's bases (own row of `wbase`), so
        // there is no serial section and no second barrier: best at 2 blocks/CU (pairs)
        uint32_t run[4] = {0, 0, 0, 0}, below[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < LSB_WAVES; ++j) {
            const uint4 x = reinterpret_cast<const uint4 *>(sm.whist[j])[lane];
            run[0] += x.x; run[1] += x.y; run[2] += x.z; run[3] += x.w;
            if (j < w) { below[0] += x.x; below[1] += x.y; below[2] += x.z; below[3] += x.w; }
        }
        const uint32_t lane_sum = run[0] + run[1] + run[2] + run[3];
        uint32_t ex[4];
        ex[0] = wave_inclusive_scan(lane_sum) - lane_sum;
        ex[1] = ex[0] + run[0];
        ex[2] = ex[1] + run[1];
        ex[3] = ex[2] + run[2];
        reinterpret_cast<uint2 *>(sm.wbase[w])[lane] =
            make_uint2((ex[0] + below[0]) | ((ex[1] + below[1]) << 16), (ex[2] + below[2]) | ((ex[3] + below[3]) << 16));
        if (w == 0) publish_gbase(ex, run);
    } else {
        // wave 0 alone, two sweeps over the 8 rows (only one row in registers at a time), bases
        // written back in place as BYTE offsets into `stage`; the other waves wait at the barrier
        // while the CU's other two blocks run: best at 3 blocks/CU (keys only)
        if (w == 0) {
            uint32_t run[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < LSB_WAVES; ++j) {
                const uint4 x = reinterpret_cast<const uint4 *>(sm.whist[j])[lane];
                run[0] += x.x; run[1] += x.y; run[2] += x.z; run[3] += x.w;
            }
            const uint32_t lane_sum = run[0] + run[1] + run[2] + run[3];
            uint32_t ex[4];
            ex[0] = wave_inclusive_scan(lane_sum) - lane_sum;
            ex[1] = ex[0] + run[0];
            ex[2] = ex[1] + run[1];
            ex[3] = ex[2] + run[2];
            publish_gbase(ex, run);
            asm volatile("" ::: "memory");   // re-read the rows instead of keeping 32 registers live
            uint4 e4 = make_uint4(ex[0] << 2, ex[1] << 2, ex[2] << 2, ex[3] << 2);
#pragma unroll
            for (int j = 0; j < LSB_WAVES; ++j) {
                const uint4 x = reinterpret_cast<const uint4 *>(sm.whist[j])[lane];
                reinterpret_cast<uint4 *>(sm.whist[j])[lane] = e4;
                e4.x += x.x << 2; e4.y += x.y << 2; e4.z += x.z << 2; e4.w += x.w << 2;
            }
        }
        __syncthreads();
    }
    GS_PHASE(4);                                   // scan + barrier 2

    // 4. tile -> LDS in rank order -> global.  All 16 base reads are issued before the first
    //    write so the LDS round trip is paid once, not per key.
    {
        uint32_t wb[LSB_KPT];
#pragma unroll
        for (int i = 0; i < LSB_KPT; ++i) {
            const uint32_t d = digit(key[i]);
            wb[i] = ALLWAVE ? (uint32_t)mybase[d] : my[d];
        }
#pragma unroll
        for (int i = 0; i < LSB_KPT; ++i) {
            if (HAS_VALUES) {
                reinterpret_cast<uint2 *>(sm.stage)[pos[i] + wb[i]] = make_uint2(key[i], val[i]);
            } else {
                const uint32_t at = (pos[i] << 2) + wb[i];       // bytes
                *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(sm.stage) + at) = key[i];
            }
        }
    }
    GS_PHASE_WAIT("lgkmcnt(0)");
    GS_PHASE(5);                                   // LDS scatter
    __syncthreads();
    GS_PHASE(6);                                   // barrier 3
    if (!HAS_VALUES) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int i = 0; i < LSB_KPT; ++i) {
        const uint32_t slot = (uint32_t)tid + i * LSB_THREADS;
        uint32_t k, v = 0;
        if (HAS_VALUES) {
            const uint2 kv = reinterpret_cast<const uint2 *>(sm.stage)[slot];
            k = kv.x; v = kv.y;
        } else {
            k = sm.stage[slot];
        }
        const uint32_t g = sm.gbase[digit(k)];
        if (!TAIL || slot < valid) {
            if (BIG) {
                const uint32_t dst = g + slot;
                keys_out[dst] = tw_out(k);
                if (HAS_VALUES) vals_out[dst] = v;
            } else {
                const uint32_t off = g + slot * 4u;             // 32-bit byte offset: scalar base + vector offset
                *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(keys_out) + off) = tw_out(k);
                if (HAS_VALUES) *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(vals_out) + off) = v;
            }
        }
    }
    GS_PHASE(7);                                   // store issue
    GS_PHASE_WAIT("vmcnt(0)");
    GS_PHASE(8);                                   // store drain
#ifdef GS_EXP_PHASES
    if (tid == 0 && blockIdx.x < 131072u) {        // clock calibration: shader clocks vs 100 MHz real time
        gs_phase_buf[blockIdx.x * 16 + 9] = (uint32_t)(__builtin_amdgcn_s_memtime() - t0_);
        gs_phase_buf[blockIdx.x * 16 + 10] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - r0_);
    }
#endif
}

// ------------------------------------------------------------------- host --

static inline uint32_t lsb_num_tiles(uint64_t n) { return (uint32_t)((n + LSB_TILE - 1) / LSB_TILE); }
static inline uint32_t lsb_grid(uint64_t n)
{
    const uint32_t t = lsb_num_tiles(n);
    const uint32_t g = (t + LSB_CHUNK - 1u) / LSB_CHUNK;
    return g ? g : 1u;
}
// downsweep blocks: one per full tile
static inline uint32_t lsb_ds_grid(uint64_t n)
{
    const uint32_t full = (uint32_t)(n / LSB_TILE);
    return full ? full : 1u;
}

void lsb_twiddle_masks(int key_type, int descending, bool first, bool last, PassParams &p)
{
    // keys are stored twiddled between passes; the first pass maps in, the last maps out
    const uint32_t sign = (key_type == GS_KEY_I32) ? 0x80000000u : 0u;
    const uint32_t flip = descending ? 0xffffffffu : 0u;
    p.f32_in = (first && key_type == GS_KEY_F32) ? 1 : 0;
    p.f32_out = (last && key_type == GS_KEY_F32) ? 1 : 0;
    p.xor_in = first ? (sign ^ flip) : 0u;
    p.xor_out = last ? (sign ^ flip) : 0u;
}

PassParams lsb_make_params(uint64_t n, int shift, int bits)
{
    PassParams p{};
    p.n = (uint32_t)n;
    p.num_tiles = lsb_num_tiles(n);
    p.grid = lsb_grid(n);
    p.ds_grid = lsb_ds_grid(n);
    p.shift = (uint32_t)shift;
    p.bits = (uint32_t)bits;
    p.mask = (1u << bits) - 1u;
    return p;
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
static inline size_t spine_bytes(uint64_t n) { return align256((size_t)RADIX * lsb_grid(n) * sizeof(uint32_t)); }
static inline size_t totals_bytes() { return align256(RADIX * sizeof(uint32_t)); }
static inline size_t prefix16_bytes(uint64_t n) { return align256((size_t)lsb_num_tiles(n) * RADIX * sizeof(uint16_t)); }
// single-sweep mode (n <= 2^30: 30-bit prefixes): digit totals of 4 passes + error word, and the look-back words
static inline bool fused_enabled()
{
    static const char *e = getenv("GS_LSB_MODE");   // read once per process: sizes and strategy stay consistent
    return e && strcmp(e, "fused") == 0;
}
static inline bool fused_possible(uint64_t n) { return fused_enabled() && n <= (1ull << 30) && n >= (uint64_t)LSB_TILE; }
static inline size_t totals4_bytes() { return align256(4 * RADIX * sizeof(uint32_t)) + 256; }
static inline size_t status_bytes(uint64_t n) { return fused_possible(n) ? align256((n / LSB_TILE) * RADIX * sizeof(uint32_t)) : 0; }

size_t lsb_temp_bytes(uint64_t n) { return spine_bytes(n) + totals_bytes() + prefix16_bytes(n) + totals4_bytes() + status_bytes(n); }
LsbWorkspace lsb_carve(void *temp, uint64_t n)
{
    char *c = (char *)temp;
    LsbWorkspace ws;
    ws.spine = (uint32_t *)c;
    ws.totals = (uint32_t *)(c + spine_bytes(n));
    ws.prefix16 = (uint16_t *)(c + spine_bytes(n) + totals_bytes());
    ws.totals4 = (uint32_t *)(c + spine_bytes(n) + totals_bytes() + prefix16_bytes(n));
    ws.error_word = ws.totals4 + 4 * RADIX;
    ws.status = fused_possible(n) ? (uint32_t *)((char *)ws.totals4 + totals4_bytes()) : nullptr;
    return ws;
}

int lsb_upsweep(const uint32_t *keys, uint32_t *spine, uint16_t *prefix16, const PassParams &p, hipStream_t s)
{
    KernelTimer kt(GS_K_LSB_UPSWEEP, s);
    hipLaunchKernelGGL(lsb_upsweep_kernel, dim3(p.grid), dim3(LSB_THREADS), 0, s, keys, spine, prefix16, p);
    return (int)hipGetLastError();
}

int lsb_scan(uint32_t *spine, uint32_t *totals, uint32_t grid, hipStream_t s)
{
    KernelTimer kt(GS_K_LSB_SCAN, s);
    hipLaunchKernelGGL(lsb_scan_kernel, dim3(RADIX), dim3(SCAN_THREADS), 0, s, spine, totals, grid);
    return (int)hipGetLastError();
}

template <int TW, bool BIG>
static void launch_downsweep(const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, const uint32_t *spine,
                             const uint16_t *prefix16, const uint32_t *totals, const PassParams &p, hipStream_t s)
{
    const dim3 block(LSB_THREADS);
    if (vin)
        hipLaunchKernelGGL((lsb_downsweep_kernel<true, false, TW, BIG>), dim3(p.ds_grid), block, 0, s, kin, kout, vin, vout,
                           spine, prefix16, totals, p, (uint32_t *)nullptr, (uint32_t *)nullptr);
    else
        hipLaunchKernelGGL((lsb_downsweep_kernel<false, false, TW, BIG>), dim3(p.ds_grid), block, 0, s, kin, kout, vin, vout,
                           spine, prefix16, totals, p, (uint32_t *)nullptr, (uint32_t *)nullptr);
}

static void launch_downsweep_tail(const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
                                  const uint32_t *totals, const PassParams &p, hipStream_t s)
{
    const dim3 block(LSB_THREADS);   // one block, the general variant
    if (vin)
        hipLaunchKernelGGL((lsb_downsweep_kernel<true, true, 2, true>), dim3(1), block, 0, s, kin, kout, vin, vout,
                           (const uint32_t *)nullptr, (const uint16_t *)nullptr, totals, p, (uint32_t *)nullptr,
                           (uint32_t *)nullptr);
    else
        hipLaunchKernelGGL((lsb_downsweep_kernel<false, true, 2, true>), dim3(1), block, 0, s, kin, kout, vin, vout,
                           (const uint32_t *)nullptr, (const uint16_t *)nullptr, totals, p, (uint32_t *)nullptr,
                           (uint32_t *)nullptr);
}

int lsb_downsweep(const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, const uint32_t *spine,
                  const uint16_t *prefix16, const uint32_t *totals, const PassParams &p, hipStream_t s)
{
    KernelTimer kt(GS_K_LSB_DOWNSWEEP, s);
    if (p.n >= (uint32_t)LSB_TILE) {   // full tiles
        const int tw = (p.f32_in || p.f32_out) ? 2 : ((p.xor_in | p.xor_out) ? 1 : 0);
        const bool big = p.n > (1u << 30);
#define GS_DS(TW_, BIG_) launch_downsweep<TW_, BIG_>(kin, kout, vin, vout, spine, prefix16, totals, p, s)
        if (big) { if (tw == 2) GS_DS(2, true); else if (tw == 1) GS_DS(1, true); else GS_DS(0, true); }
        else { if (tw == 2) GS_DS(2, false); else if (tw == 1) GS_DS(1, false); else GS_DS(0, false); }
#undef GS_DS
    }
    if (p.n % (uint32_t)LSB_TILE) launch_downsweep_tail(kin, kout, vin, vout, totals, p, s);   // the partial last tile
    return (int)hipGetLastError();
}

// LSB pass strategy.  "three" = upsweep -> scan -> downsweep per pass (the north_star's
// formulation; 48 B/key for 4 passes).  "fused" = single-sweep: ONE histogram kernel gives the
// digit totals of every pass, and each pass is a single scatter whose tiles get their offsets
// by decoupled look-back (36 B/key); available for n <= 2^30.  Selected by GS_LSB_MODE
// ("three" | "fused"); results are identical.
static bool lsb_use_fused(uint64_t n) { return fused_possible(n); }

template <typename Route>
static int lsb_run_passes(const LsbWorkspace &ws, uint64_t num_items, int begin_bit, int end_bit, int descending,
                          int key_type, bool pairs, hipStream_t s, Route route)
{
    const int num_bits = end_bit - begin_bit;
    const int num_passes = (num_bits + RADIX_BITS - 1) / RADIX_BITS;
    const bool fused = lsb_use_fused(num_items) && ws.status;
    for (int pass = 0; pass < num_passes; ++pass) {
        const int shift = begin_bit + pass * RADIX_BITS;
        const int bits = (end_bit - shift < RADIX_BITS) ? end_bit - shift : RADIX_BITS;
        PassParams p = lsb_make_params(num_items, shift, bits);
        lsb_twiddle_masks(key_type, descending, pass == 0, pass == num_passes - 1, p);
        const uint32_t *kin, *vin;
        uint32_t *kout, *vout;
        route(pass, num_passes, kin, kout, vin, vout);
        if (!pairs) { vin = nullptr; vout = nullptr; }
        int e;
        if (!fused) {
            if (p.num_tiles <= LSB_SMALL_TILES) {   // latency-bound sizes: workgroup per tile, chunk prefixes in the scan
                { KernelTimer kt(GS_K_LSB_UPSWEEP, s);
                  hipLaunchKernelGGL(lsb_upsweep_small_kernel, dim3(p.num_tiles), dim3(LSB_THREADS), 0, s, kin, ws.prefix16, p); }
                { KernelTimer kt(GS_K_LSB_SCAN, s);
                  hipLaunchKernelGGL(lsb_scan_small_kernel, dim3(RADIX), dim3(RADIX), 0, s, ws.spine, ws.totals, ws.prefix16, p.grid,
                                     p.num_tiles); }
                if ((e = (int)hipGetLastError())) return e;
            } else {
            if ((e = lsb_upsweep(kin, ws.spine, ws.prefix16, p, s))) return e;
            if ((e = lsb_scan(ws.spine, ws.totals, p.grid, s))) return e;
            }
            if ((e = lsb_downsweep(kin, kout, vin, vout, ws.spine, ws.prefix16, ws.totals, p, s))) return e;
            continue;
        }
        if (pass == 0) {   // digit totals of all passes from one read of the input
            Hist4Params hp{};
            hp.n = p.n; hp.num_passes = num_passes; hp.f32_in = p.f32_in; hp.xor_in = p.xor_in;
            for (int q = 0; q < num_passes; ++q) {
                hp.shift[q] = (uint32_t)(begin_bit + q * RADIX_BITS);
                hp.bits[q] = (uint32_t)((end_bit - (int)hp.shift[q] < RADIX_BITS) ? end_bit - (int)hp.shift[q] : RADIX_BITS);
            }
            hipError_t me = zero_async(ws.totals4, totals4_bytes(), s);
            if (me != hipSuccess) return (int)me;
            KernelTimer kt(GS_K_LSB_UPSWEEP, s);
            const uint32_t g = p.num_tiles < 2048u ? p.num_tiles : 2048u;
            if (((uintptr_t)kin & 15u) == 0)
                hipLaunchKernelGGL(lsb_hist4_kernel<true>, dim3(g), dim3(LSB_THREADS), 0, s, kin, ws.totals4, hp);
            else
                hipLaunchKernelGGL(lsb_hist4_kernel<false>, dim3(g), dim3(LSB_THREADS), 0, s, kin, ws.totals4, hp);
        }
        hipError_t me = zero_async(ws.status, status_bytes(num_items), s);
        if (me != hipSuccess) return (int)me;
        const uint32_t *tot = ws.totals4 + pass * RADIX;
        {
            KernelTimer kt(GS_K_LSB_DOWNSWEEP, s);
            const dim3 block(LSB_THREADS);
            if (vin)
                hipLaunchKernelGGL((lsb_downsweep_kernel<true, false, 2, false, true>), dim3(p.ds_grid), block, 0, s, kin, kout,
                                   vin, vout, (const uint32_t *)nullptr, (const uint16_t *)nullptr, tot, p, ws.status,
                                   ws.error_word);
            else
                hipLaunchKernelGGL((lsb_downsweep_kernel<false, false, 2, false, true>), dim3(p.ds_grid), block, 0, s, kin, kout,
                                   vin, vout, (const uint32_t *)nullptr, (const uint16_t *)nullptr, tot, p, ws.status,
                                   ws.error_word);
            if (p.n % (uint32_t)LSB_TILE) launch_downsweep_tail(kin, kout, vin, vout, tot, p, s);
        }
        if ((e = (int)hipGetLastError())) return e;
    }
    return hipSuccess;
}

}  // namespace gs

using namespace gs;

extern "C" {

#ifdef GS_EXP_PHASES
int gs_exp_phases(uint32_t *host_out, uint32_t blocks)
{
    GS_CLEAR_STALE_ERROR();
    return (int)hipMemcpyFromSymbol(host_out, HIP_SYMBOL(gs_phase_buf), (size_t)blocks * 16 * sizeof(uint32_t));
}
#endif

size_t gs_lsb_temp_bytes(uint64_t num_items, int /*has_values*/)
{
    return lsb_temp_bytes(num_items);
}

void gs_lsb_geometry(uint64_t num_items, int /*has_values*/, uint32_t *grid, uint32_t *tile, uint32_t *tiles_per_chunk)
{
    if (grid) *grid = lsb_grid(num_items);
    if (tile) *tile = LSB_TILE;
    if (tiles_per_chunk) *tiles_per_chunk = LSB_CHUNK;
}

int gs_lsb_upsweep_u32(void *d_temp, size_t temp_bytes, const uint32_t *d_keys_in, uint64_t num_items, int shift,
                       int bits, int descending, int key_type_in, void *stream)
{
    GS_CLEAR_STALE_ERROR();
    if (num_items >= (1ull << 32) || bits < 1 || bits > 8 || shift < 0 || shift + bits > 32) return hipErrorInvalidValue;
    if (!d_temp || temp_bytes < gs_lsb_temp_bytes(num_items, 0)) return hipErrorInvalidValue;
    if (num_items == 0) return hipSuccess;
    PassParams p = lsb_make_params(num_items, shift, bits);
    lsb_twiddle_masks(key_type_in, descending, true, true, p);
    const LsbWorkspace ws = lsb_carve(d_temp, num_items);
    return lsb_upsweep(d_keys_in, ws.spine, ws.prefix16, p, (hipStream_t)stream);
}

int gs_lsb_scan_spine(void *d_temp, size_t temp_bytes, uint64_t num_items, void *stream)
{
    GS_CLEAR_STALE_ERROR();
    if (num_items >= (1ull << 32)) return hipErrorInvalidValue;
    if (!d_temp || temp_bytes < gs_lsb_temp_bytes(num_items, 0)) return hipErrorInvalidValue;
    if (num_items == 0) return hipSuccess;
    const LsbWorkspace ws = lsb_carve(d_temp, num_items);
    return lsb_scan(ws.spine, ws.totals, lsb_grid(num_items), (hipStream_t)stream);
}

int gs_lsb_downsweep_u32(void *d_temp, size_t temp_bytes, const uint32_t *d_keys_in, uint32_t *d_keys_out,
                         const uint32_t *d_vals_in, uint32_t *d_vals_out, uint64_t num_items, int shift, int bits,
                         int descending, int key_type_in, int key_type_out, void *stream)
{
    GS_CLEAR_STALE_ERROR();
    if (num_items >= (1ull << 32) || bits < 1 || bits > 8 || shift < 0 || shift + bits > 32) return hipErrorInvalidValue;
    if ((d_vals_in == nullptr) != (d_vals_out == nullptr)) return hipErrorInvalidValue;
    if (!d_temp || temp_bytes < gs_lsb_temp_bytes(num_items, 0)) return hipErrorInvalidValue;
    if (num_items == 0) return hipSuccess;
    PassParams p = lsb_make_params(num_items, shift, bits);
    PassParams in{}, out{};
    lsb_twiddle_masks(key_type_in, descending, true, false, in);
    lsb_twiddle_masks(key_type_out, descending, false, true, out);
    p.f32_in = in.f32_in; p.xor_in = in.xor_in;
    p.f32_out = out.f32_out; p.xor_out = out.xor_out;
    const LsbWorkspace ws = lsb_carve(d_temp, num_items);
    return lsb_downsweep(d_keys_in, d_keys_out, d_vals_in, d_vals_out, ws.spine, ws.prefix16, ws.totals, p,
                         (hipStream_t)stream);
}

int gs_lsb_workspace_layout(void *d_temp, uint64_t num_items, uint32_t **d_spine, uint32_t **d_totals,
                            uint16_t **d_prefix16)
{
    GS_CLEAR_STALE_ERROR();
    const LsbWorkspace ws = lsb_carve(d_temp, num_items);
    if (d_spine) *d_spine = ws.spine;
    if (d_totals) *d_totals = ws.totals;
    if (d_prefix16) *d_prefix16 = ws.prefix16;
    return hipSuccess;
}

int gs_lsb_sort_u32(void *d_temp, size_t temp_bytes, uint32_t *d_keys[2], uint32_t *d_vals[2], int *selector,
                    uint64_t num_items, int begin_bit, int end_bit, int descending, int key_type, void *stream)
{
    GS_CLEAR_STALE_ERROR();
    if (!selector || (*selector != 0 && *selector != 1) || !d_keys) return hipErrorInvalidValue;
    if (begin_bit < 0 || end_bit > 32 || begin_bit > end_bit) return hipErrorInvalidValue;
    if (num_items >= (1ull << 32)) return hipErrorInvalidValue;
    if (key_type < GS_KEY_U32 || key_type > GS_KEY_F32) return hipErrorInvalidValue;
    if (num_items == 0 || begin_bit == end_bit) return hipSuccess;
    if (!d_temp || temp_bytes < gs_lsb_temp_bytes(num_items, d_vals != nullptr)) return hipErrorInvalidValue;
    if (!d_keys[0] || !d_keys[1] || (d_vals && (!d_vals[0] || !d_vals[1]))) return hipErrorInvalidValue;

    int sel = *selector;
    if (num_items <= small_sort_capacity(d_vals != nullptr) && !fused_enabled()) {
        // fits one workgroup: one launch (CUB's single-tile path); the result lands where the passes would leave it
        const int passes = (end_bit - begin_bit + RADIX_BITS - 1) / RADIX_BITS, fin = sel ^ (passes & 1);
        PassParams tw{};
        lsb_twiddle_masks(key_type, descending, true, true, tw);
        const int e = small_stable_sort(d_temp, temp_bytes, d_keys[sel], d_keys[fin], d_vals ? d_vals[sel] : nullptr,
                                        d_vals ? d_vals[fin] : nullptr, (uint32_t)num_items, begin_bit, end_bit, tw.f32_in,
                                        tw.xor_in, tw.f32_out, tw.xor_out, (hipStream_t)stream);
        if (e) return e;
        *selector = fin;
        return hipSuccess;
    }
    const LsbWorkspace ws = lsb_carve(d_temp, num_items);
    const int e = lsb_run_passes(ws, num_items, begin_bit, end_bit, descending, key_type, d_vals != nullptr,
                                 (hipStream_t)stream,
                                 [&](int, int, const uint32_t *&kin, uint32_t *&kout, const uint32_t *&vin, uint32_t *&vout) {
                                     kin = d_keys[sel]; kout = d_keys[sel ^ 1];
                                     vin = d_vals ? d_vals[sel] : nullptr; vout = d_vals ? d_vals[sel ^ 1] : nullptr;
                                     sel ^= 1;
                                 });
    if (e) return e;
    *selector = sel;
    return hipSuccess;
}

// Non-overwriting form: the input arrays stay untouched and the result lands in the
// output arrays; the extra ping-pong buffers live in the workspace (CUB's
// is_overwrite_okay == false, dispatch_radix_sort.cuh:1099-1129).
size_t gs_lsb_copy_temp_bytes(uint64_t num_items, int has_values)
{
    const size_t buf = align256((size_t)num_items * sizeof(uint32_t));
    return align256(lsb_temp_bytes(num_items)) + buf * (has_values ? 2 : 1);
}

int gs_lsb_sort_copy_u32(void *d_temp, size_t temp_bytes, const uint32_t *d_keys_in, uint32_t *d_keys_out,
                         const uint32_t *d_vals_in, uint32_t *d_vals_out, uint64_t num_items, int begin_bit, int end_bit,
                         int descending, int key_type, void *stream)
{
    GS_CLEAR_STALE_ERROR();
    if (begin_bit < 0 || end_bit > 32 || begin_bit > end_bit) return hipErrorInvalidValue;
    if (num_items >= (1ull << 32)) return hipErrorInvalidValue;
    if (key_type < GS_KEY_U32 || key_type > GS_KEY_F32) return hipErrorInvalidValue;
    if ((d_vals_in == nullptr) != (d_vals_out == nullptr)) return hipErrorInvalidValue;
    if (num_items == 0) return hipSuccess;
    const bool pairs = d_vals_in != nullptr;
    if (!d_keys_in || !d_keys_out) return hipErrorInvalidValue;
    if (!d_temp || temp_bytes < gs_lsb_copy_temp_bytes(num_items, pairs)) return hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    if (begin_bit == end_bit) {   // zero passes: the output is a copy of the input
        hipError_t e = hipMemcpyAsync(d_keys_out, d_keys_in, num_items * sizeof(uint32_t), hipMemcpyDeviceToDevice, s);
        if (e == hipSuccess && pairs)
            e = hipMemcpyAsync(d_vals_out, d_vals_in, num_items * sizeof(uint32_t), hipMemcpyDeviceToDevice, s);
        return (int)e;
    }
    if (num_items <= small_sort_capacity(pairs) && !fused_enabled()) {   // one workgroup, straight from IN to OUT
        PassParams tw{};
        lsb_twiddle_masks(key_type, descending, true, true, tw);
        return small_stable_sort(d_temp, temp_bytes, d_keys_in, d_keys_out, d_vals_in, d_vals_out, (uint32_t)num_items, begin_bit,
                                 end_bit, tw.f32_in, tw.xor_in, tw.f32_out, tw.xor_out, s);
    }
    const LsbWorkspace ws = lsb_carve(d_temp, num_items);
    const size_t buf = align256((size_t)num_items * sizeof(uint32_t));
    uint32_t *tk = (uint32_t *)((char *)d_temp + align256(lsb_temp_bytes(num_items)));
    uint32_t *tv = (uint32_t *)((char *)tk + buf);
    return lsb_run_passes(ws, num_items, begin_bit, end_bit, descending, key_type, pairs, s,
                          [&](int pass, int num_passes, const uint32_t *&kin, uint32_t *&kout, const uint32_t *&vin,
                              uint32_t *&vout) {
                              // pass k writes OUT when an even number of passes follows it, else the scratch
                              const bool to_out = ((num_passes - 1 - pass) & 1) == 0;
                              kin = (pass == 0) ? d_keys_in : (to_out ? tk : d_keys_out);
                              vin = (pass == 0) ? d_vals_in : (to_out ? tv : d_vals_out);
                              kout = to_out ? d_keys_out : tk;
                              vout = to_out ? d_vals_out : tv;
                          });
}

}  // extern "C"
