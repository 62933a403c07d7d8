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
#include <cstdlib>

namespace gs {

constexpr int LSB_THREADS = 512;                     // 8 waves
constexpr int LSB_WAVES = LSB_THREADS / WAVE;
constexpr int LSB_KPT = 16;                          // keys per thread per tile
constexpr int LSB_TILE = LSB_THREADS * LSB_KPT;      // 8192 keys = 32 KiB
constexpr int LSB_BLOCKS_PER_CU = 2;                 // 128 VGPRs -> 4 waves/SIMD -> 2 blocks of 8 waves
constexpr int MI355X_CUS = 256;
constexpr int MI355X_XCDS = 8;
constexpr uint32_t LSB_RESIDENT = MI355X_CUS * LSB_BLOCKS_PER_CU;   // blocks in flight on the chip
constexpr uint32_t LSB_MAX_CHUNK = 8;                               // tiles per block at large n

struct PassParams {
    uint32_t n;          // number of keys
    uint32_t num_tiles;  // ceil(n / LSB_TILE)
    uint32_t grid;       // blocks = chunks; chunk c owns tiles [c*chunk, min((c+1)*chunk, num_tiles))
    uint32_t chunk;      // tiles per chunk
    int shift;           // digit = (key >> shift) & mask
    uint32_t mask;
    uint32_t bits;       // digit width (<= 8)
    int f32_in, f32_out;          // float twiddle on read / undo on write
    uint32_t xor_in, xor_out;     // uniform xor on read / write (sign flip, descending)
    uint32_t valu_rounds;         // bit i set: round i matches with VALU ballots, else through LDS
};

// Block -> chunk of consecutive tiles.  Blocks are dispatched in blockIdx order
// and dealt round-robin over the 8 XCDs, so at any moment the resident blocks work
// on one window of consecutive chunks: for every digit their output runs are
// neighbours in memory (one contiguous window per digit instead of one stream per
// block spread over the whole array).  Within a round of LSB_RESIDENT blocks the
// XCD with label (b % 8) takes a contiguous slice of the window, so the partial
// 128-B lines at the ends of neighbouring runs meet in the same L2.  Speed only:
// any bijection block <-> chunk gives the same result.
__device__ __forceinline__ uint32_t chunk_of_block(const PassParams &p, uint32_t b)
{
    const uint32_t round = b / LSB_RESIDENT;
    if ((round + 1u) * LSB_RESIDENT > p.grid) return b;   // ragged last round: identity
    const uint32_t r = b % LSB_RESIDENT;
    return round * LSB_RESIDENT + (r % MI355X_XCDS) * (LSB_RESIDENT / MI355X_XCDS) + r / MI355X_XCDS;
}

__device__ __forceinline__ void chunk_tiles(const PassParams &p, uint32_t c, uint32_t &t0, uint32_t &t1)
{
    t0 = c * p.chunk;
    t1 = t0 + p.chunk;
    if (t1 > p.num_tiles) t1 = p.num_tiles;
}

// ---------------------------------------------------------------- upsweep --
// Per-chunk digit histogram of the chunk's tile range -> spine[d * grid + chunk].
// Wave-private 256-bin LDS histograms (ds_add_u32, no return value needed).
template <bool VEC>
__global__ __launch_bounds__(LSB_THREADS) void lsb_upsweep_kernel(const uint32_t *__restrict__ keys,
                                                                  uint32_t *__restrict__ spine, PassParams p)
{
    __shared__ uint32_t hist[LSB_WAVES][RADIX];
    const int tid = threadIdx.x, w = wave_id();
    for (int i = tid; i < LSB_WAVES * RADIX; i += LSB_THREADS) (&hist[0][0])[i] = 0;
    const uint32_t chunk = chunk_of_block(p, blockIdx.x);
    uint32_t t0, t1;
    chunk_tiles(p, chunk, t0, t1);
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
        spine[(uint32_t)tid * p.grid + chunk] = s;
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
// Stable scatter of the block's tiles.  Per tile:
//   1. wave-striped coalesced load (key i of lane l of wave w sits at
//      tile + w*1024 + i*64 + l, so position order = (w, i, l)); the NEXT
//      tile's loads are issued before this tile is ranked, so HBM latency is
//      covered by the ranking of the current tile;
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
//   3. wave 0 turns the 8 wave histograms into tile-absolute bases (4 digits
//      per lane, b128 LDS accesses, DPP scan) and advances the per-digit output
//      cursors;
//   4. keys (and values) go to LDS at their tile rank and are read back in rank
//      order: consecutive lanes hit consecutive addresses inside a digit run.
//
// Write combining (CARRY).  A tile holds ~32 keys per digit, so a plain
// scatter writes 128-byte runs at arbitrary 4-byte offsets: almost every
// 64-byte memory segment is written in two pieces, a tile apart in time, and
// the partial pieces cost HBM efficiency (measured: 2.9 ms per pass against
// 1.8 ms for the same kernel with 1-KiB runs).  So each block keeps, per
// digit, the keys of the last incomplete 64-byte segment in LDS (`carry`, at
// most 15 keys) and stores only up to the last 64-byte boundary; the carried
// keys go out in front of the next tile's run of that digit.  Output order
// inside a digit is unchanged, so the sort stays stable.  Only the first and
// the last segment of a digit per CHUNK are still partial.
//
// Three block barriers per tile.
constexpr uint32_t SEG = 16;   // keys per 64-byte segment

template <bool HAS_VALUES, bool CARRY>
struct DownsweepSmem {
    unsigned long long wmask[LSB_WAVES][RADIX];           // wave-private lane masks per digit (zero between rounds)
    uint32_t whist[LSB_WAVES][RADIX];                     // wave-private digit counters -> bases
    uint2 tab_run[RADIX];                                 // {global base - slot bias, first slot NOT flushed}
    uint2 tab_carry[CARRY ? RADIX : 1];                   // {global cursor, carried keys to flush now}
    uint32_t stage[LSB_TILE * (HAS_VALUES ? 2 : 1)];      // tile in rank order; pairs interleaved {key,val}
    uint32_t carry[CARRY ? RADIX * SEG * (HAS_VALUES ? 2 : 1) : 1];   // [digit][SEG] keys (pairs: {key,val})
};

// TAIL = false: the block's FULL tiles, software-pipelined, write-combined.
// TAIL = true: one block handles the array's last, partial tile (guarded loads,
// direct scatter); being last in key order, its keys of digit d sit at the very
// end of digit d's global range, so it needs only the digit totals.  Splitting
// it off keeps the guarded path's registers out of the hot kernel.
template <bool HAS_VALUES, bool TAIL>
__global__ __launch_bounds__(LSB_THREADS, 4) void lsb_downsweep_kernel(
    const uint32_t *__restrict__ keys_in, uint32_t *__restrict__ keys_out, const uint32_t *__restrict__ vals_in,
    uint32_t *__restrict__ vals_out, const uint32_t *__restrict__ spine, const uint32_t *__restrict__ totals,
    PassParams p)
{
    constexpr bool CARRY = !TAIL;
    __shared__ __attribute__((aligned(16))) DownsweepSmem<HAS_VALUES, CARRY> sm;

    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    const uint32_t full_tiles = p.n / (uint32_t)LSB_TILE;
    const uint32_t chunk = TAIL ? 0u : chunk_of_block(p, blockIdx.x);
    uint32_t t0, t1;
    if (TAIL) {
        t0 = full_tiles;
        t1 = full_tiles + 1;
    } else {
        chunk_tiles(p, chunk, t0, t1);
        if (t1 > full_tiles) t1 = full_tiles;
        if (t0 >= t1) return;
    }

    // wave 0, lane l: output cursor (global index of the next key to store) of digits
    // 4l..4l+3 for this block, and the number of keys currently carried per digit
    // (TAIL: inclusive scan of the totals; the tile's own counts are subtracted later)
    uint32_t cursor[4] = {0, 0, 0, 0}, carried[4] = {0, 0, 0, 0};
    if (w == 0) {
        const uint4 tot = reinterpret_cast<const uint4 *>(totals)[lane];
        const uint32_t lane_sum = tot.x + tot.y + tot.z + tot.w;
        const uint32_t ex = wave_inclusive_scan(lane_sum) - lane_sum;
        if (TAIL) {
            cursor[0] = ex + tot.x;
            cursor[1] = cursor[0] + tot.y;
            cursor[2] = cursor[1] + tot.z;
            cursor[3] = cursor[2] + tot.w;
        } else {
            const uint32_t *sp = spine + (uint32_t)(4 * lane) * p.grid + chunk;
            cursor[0] = ex + sp[0];
            cursor[1] = ex + tot.x + sp[p.grid];
            cursor[2] = ex + tot.x + tot.y + sp[2 * p.grid];
            cursor[3] = ex + tot.x + tot.y + tot.z + sp[3 * p.grid];
        }
    }

    uint32_t *my = sm.whist[w];
    unsigned long long *mm = sm.wmask[w];
    const uint32_t half_bit = 1u << (lane & 31);
    // which of the 16 rounds match on the VALU instead of through LDS (balances the two pipes)
    const uint32_t valu_rounds = p.valu_rounds;
#pragma unroll
    for (int i = lane; i < RADIX; i += WAVE) mm[i] = 0ull;   // invariant: all zero between rounds
    const uint32_t wbase = (uint32_t)w * (WAVE * LSB_KPT) + lane;
    const uint32_t tail_valid = p.n - full_tiles * (uint32_t)LSB_TILE;   // used when TAIL
    uint32_t knext[LSB_KPT];

    // keys are fetched one tile ahead; values are fetched at the top of their own tile
    // and are not needed before the staging step, a full ranking later
    auto load_keys = [&](uint32_t t) {
        const uint32_t *kin = keys_in + (uint64_t)t * LSB_TILE;
        if (!TAIL) {
#pragma unroll
            for (int i = 0; i < LSB_KPT; ++i) knext[i] = kin[wbase + i * WAVE];
        } else {
            // pad with keys whose twiddled form is all ones (largest digit; ranked after
            // every real key of that digit because they sit at the tail)
            const uint32_t pad = twiddle_out(0xffffffffu, p.f32_in, p.xor_in);
#pragma unroll
            for (int i = 0; i < LSB_KPT; ++i) {
                const uint32_t idx = wbase + i * WAVE;
                knext[i] = pad;
                if (idx < tail_valid) knext[i] = kin[idx];
            }
        }
    };

    // store carried keys of every digit: entry (d, m) -> global[cursor_d + m] for m < count_d
    auto flush_carry = [&]() {
#pragma unroll
        for (int k = 0; k < (int)(RADIX * SEG) / LSB_THREADS; ++k) {
            const uint32_t e = (uint32_t)tid + k * LSB_THREADS;
            const uint32_t d = e / SEG, m = e % SEG;
            const uint2 t = sm.tab_carry[d];
            if (m < t.y) {
                if (HAS_VALUES) {
                    const uint2 kv = reinterpret_cast<const uint2 *>(sm.carry)[e];
                    keys_out[t.x + m] = twiddle_out(kv.x, p.f32_out, p.xor_out);
                    vals_out[t.x + m] = kv.y;
                } else {
                    keys_out[t.x + m] = twiddle_out(sm.carry[e], p.f32_out, p.xor_out);
                }
            }
        }
    };

    load_keys(t0);
    for (uint32_t t = t0; t < t1; ++t) {
        const uint64_t tile_base = (uint64_t)t * LSB_TILE;
        const uint32_t valid = TAIL ? tail_valid : (uint32_t)LSB_TILE;

        uint32_t key[LSB_KPT], val[LSB_KPT], pos[LSB_KPT];
#pragma unroll
        for (int i = 0; i < LSB_KPT; ++i) key[i] = twiddle_in(knext[i], p.f32_in, p.xor_in);
        if (HAS_VALUES) {
            const uint32_t *vin = vals_in + tile_base;
#pragma unroll
            for (int i = 0; i < LSB_KPT; ++i) {
                const uint32_t idx = wbase + i * WAVE;
                val[i] = 0;
                if (!TAIL || idx < valid) val[i] = vin[idx];
            }
        }
        if (!TAIL && t + 1 < t1) load_keys(t + 1);   // in flight while this tile is ranked

        // 2. rank inside the wave (the LDS mask of round i is consumed one round later, so
        //    its latency hides behind the issue of round i+1)
#pragma unroll
        for (int i = lane; i < RADIX; i += WAVE) my[i] = 0;
        {
            uint32_t d_prev = 0, plo = 0, phi = 0;
#pragma unroll
            for (int i = 0; i <= LSB_KPT; ++i) {
                uint32_t d_cur = 0, clo = 0, chi = 0;
                if (i < LSB_KPT) {
                    d_cur = __builtin_amdgcn_ubfe(key[i], p.shift, p.bits);
                    if ((valu_rounds >> i) & 1u) {
                        match_digit(d_cur, clo, chi);
                    } else {
                        // each lane sets / reads / clears its own 32-bit half of the 64-bit entry
                        uint32_t *half = reinterpret_cast<uint32_t *>(&mm[d_cur]) + (lane >> 5);
                        __hip_atomic_fetch_or(half, half_bit, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        const unsigned long long m =
                            __hip_atomic_load(&mm[d_cur], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        __hip_atomic_store(half, 0u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
                        clo = (uint32_t)m;
                        chi = (uint32_t)(m >> 32);
                    }
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
        __syncthreads();

        // 3. wave histograms -> tile-absolute bases; advance the output cursors.
        //    Two sweeps over the 8 rows (b128 LDS reads are cheap) keep only one
        //    row in registers at a time.
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
            if (TAIL) {   // keys of digit d end exactly at the inclusive total of d
#pragma unroll
                for (int q = 0; q < 4; ++q) cursor[q] -= run[q];
                // padded keys inflate the count of the largest digit only, and they are never stored
                const uint32_t pads = (uint32_t)LSB_TILE - valid, dmax = p.mask;
                if (lane == (int)(dmax >> 2)) {
#pragma unroll
                    for (int q = 0; q < 4; ++q)
                        if ((dmax & 3u) == (uint32_t)q) cursor[q] += pads;
                }
            }
            uint2 tr[4], tc[4];
#pragma unroll
            for (int q = 0; q < 4; ++q) {
                if (CARRY) {
                    // stream of digit = carried keys then this tile's run; store up to the last
                    // 64-byte boundary it reaches, carry the rest
                    const uint32_t c = carried[q], g = cursor[q];
                    const uint32_t total = c + run[q];
                    const uint32_t end_al = (g + total) & ~(SEG - 1u);
                    const uint32_t flush = (end_al > g) ? end_al - g : 0u;
                    tr[q] = make_uint2(g + c - ex[q], ex[q] + flush - c);   // run key at slot j: j < .y -> global[.x + j]
                    tc[q] = make_uint2(g, flush ? c : 0u);                  //              else -> carry[j - .y]
                    cursor[q] = g + flush;
                    carried[q] = total - flush;
                } else {
                    tr[q] = make_uint2(cursor[q] - ex[q], 0xffffffffu);
                    cursor[q] += run[q];
                }
            }
            reinterpret_cast<uint4 *>(sm.tab_run)[2 * lane] = make_uint4(tr[0].x, tr[0].y, tr[1].x, tr[1].y);
            reinterpret_cast<uint4 *>(sm.tab_run)[2 * lane + 1] = make_uint4(tr[2].x, tr[2].y, tr[3].x, tr[3].y);
            if (CARRY) {
                reinterpret_cast<uint4 *>(sm.tab_carry)[2 * lane] = make_uint4(tc[0].x, tc[0].y, tc[1].x, tc[1].y);
                reinterpret_cast<uint4 *>(sm.tab_carry)[2 * lane + 1] = make_uint4(tc[2].x, tc[2].y, tc[3].x, tc[3].y);
            }
            asm volatile("" ::: "memory");   // re-read the rows instead of keeping 32 registers live
            uint4 e4 = make_uint4(ex[0], ex[1], ex[2], ex[3]);
#pragma unroll
            for (int j = 0; j < LSB_WAVES; ++j) {
                const uint4 x = reinterpret_cast<const uint4 *>(sm.whist[j])[lane];
                reinterpret_cast<uint4 *>(sm.whist[j])[lane] = e4;
                e4.x += x.x; e4.y += x.y; e4.z += x.z; e4.w += x.w;
            }
        }
        __syncthreads();

        // 4. tile -> LDS in rank order; meanwhile the carried keys that complete a segment go out
#pragma unroll
        for (int i = 0; i < LSB_KPT; ++i) {
            const uint32_t d = __builtin_amdgcn_ubfe(key[i], p.shift, p.bits);
            const uint32_t at = pos[i] + my[d];
            if (HAS_VALUES) reinterpret_cast<uint2 *>(sm.stage)[at] = make_uint2(key[i], val[i]);
            else sm.stage[at] = key[i];
        }
        if (CARRY) flush_carry();
        __syncthreads();
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
            const uint32_t d = __builtin_amdgcn_ubfe(k, p.shift, p.bits);
            const uint2 tr = sm.tab_run[d];
            if (!CARRY) {
                if (slot < valid) {
                    keys_out[tr.x + slot] = twiddle_out(k, p.f32_out, p.xor_out);
                    if (HAS_VALUES) vals_out[tr.x + slot] = v;
                }
            } else if ((int32_t)(slot - tr.y) < 0) {
                keys_out[tr.x + slot] = twiddle_out(k, p.f32_out, p.xor_out);
                if (HAS_VALUES) vals_out[tr.x + slot] = v;
            } else {
                const uint32_t e = d * SEG + (slot - tr.y);
                if (HAS_VALUES) reinterpret_cast<uint2 *>(sm.carry)[e] = make_uint2(k, v);
                else sm.carry[e] = k;
            }
        }
        // no barrier here: the next tile writes `stage`, the tables and reads `carry` only
        // after its two barriers, which every thread reaches after finishing this read-out
    }

    if (CARRY) {   // end of the chunk: store what is still carried (one partial segment per digit)
        __syncthreads();
        if (w == 0) {
            reinterpret_cast<uint4 *>(sm.tab_carry)[2 * lane] = make_uint4(cursor[0], carried[0], cursor[1], carried[1]);
            reinterpret_cast<uint4 *>(sm.tab_carry)[2 * lane + 1] = make_uint4(cursor[2], carried[2], cursor[3], carried[3]);
        }
        __syncthreads();
        flush_carry();
    }
}

// ------------------------------------------------------------------- host --

static inline uint32_t lsb_num_tiles(uint64_t n) { return (uint32_t)((n + LSB_TILE - 1) / LSB_TILE); }
// tiles per chunk: 8 at large n (spine = 256 x n/65536 counters, 0.4 % of the keys'
// bytes), fewer when that would leave the chip without enough blocks
static inline uint32_t lsb_chunk(uint64_t n)
{
    static const char *e = getenv("GS_LSB_CHUNK");   // experiments only
    const uint32_t t = lsb_num_tiles(n);
    if (e && atoi(e) > 0) return (uint32_t)atoi(e);
    uint32_t c = t / (2u * LSB_RESIDENT);
    if (c < 1u) c = 1u;
    if (c > LSB_MAX_CHUNK) c = LSB_MAX_CHUNK;
    return c;
}
static inline uint32_t lsb_grid(uint64_t n)
{
    const uint32_t t = lsb_num_tiles(n), c = lsb_chunk(n);
    const uint32_t g = (t + c - 1u) / c;
    return g ? g : 1u;
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
    p.chunk = lsb_chunk(n);
    p.shift = shift;
    p.mask = (1u << bits) - 1u;
    p.bits = (uint32_t)bits;
    { static const char *e = getenv("GS_VALU_ROUNDS"); p.valu_rounds = e ? (uint32_t)strtoul(e, nullptr, 0) : 0x0000u; }
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
    hipLaunchKernelGGL(lsb_scan_kernel, dim3(RADIX), dim3(SCAN_THREADS), 0, s, spine, totals, grid);
    return (int)hipGetLastError();
}

int lsb_downsweep(const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, const uint32_t *spine,
                  const uint32_t *totals, const PassParams &p, hipStream_t s)
{
    KernelTimer kt(GS_K_LSB_DOWNSWEEP, s);
    const dim3 block(LSB_THREADS);
    if (p.n >= (uint32_t)LSB_TILE) {   // full tiles
        if (vin)
            hipLaunchKernelGGL((lsb_downsweep_kernel<true, false>), dim3(p.grid), block, 0, s, kin, kout, vin, vout,
                               spine, totals, p);
        else
            hipLaunchKernelGGL((lsb_downsweep_kernel<false, false>), dim3(p.grid), block, 0, s, kin, kout, vin, vout,
                               spine, totals, p);
    }
    if (p.n % (uint32_t)LSB_TILE) {    // the partial last tile, if any
        if (vin)
            hipLaunchKernelGGL((lsb_downsweep_kernel<true, true>), dim3(1), block, 0, s, kin, kout, vin, vout, spine,
                               totals, p);
        else
            hipLaunchKernelGGL((lsb_downsweep_kernel<false, true>), dim3(1), block, 0, s, kin, kout, vin, vout, spine,
                               totals, p);
    }
    return (int)hipGetLastError();
}

}  // namespace gs

using namespace gs;

extern "C" {

size_t gs_lsb_temp_bytes(uint64_t num_items, int /*has_values*/)
{
    return spine_bytes(num_items) + align256(RADIX * sizeof(uint32_t));
}

void gs_lsb_geometry(uint64_t num_items, int /*has_values*/, uint32_t *grid, uint32_t *tile, uint32_t *tiles_per_chunk)
{
    if (grid) *grid = lsb_grid(num_items);
    if (tile) *tile = LSB_TILE;
    if (tiles_per_chunk) *tiles_per_chunk = lsb_chunk(num_items);
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
