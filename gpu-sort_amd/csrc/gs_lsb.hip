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
#ifndef GS_EXP_SLEEP_MODE
#define GS_EXP_SLEEP_MODE 0
#endif
#ifndef GS_EXP_SLEEP_MIN_TILES
#define GS_EXP_SLEEP_MIN_TILES 49152u
#endif
#ifndef GS_EXP_SLEEP_PAIRS
#define GS_EXP_SLEEP_PAIRS 0
#endif
#include <cstdlib>
#include <cstring>

namespace gs {

// ---------------------------------------------------------------- upsweep --
#ifndef UPSWEEP_BATCH
#define UPSWEEP_BATCH 64   // dword loads in flight per lane (a tile is 128 per lane).  In-process A/B at 2^30 keys (round 3, tools/ab_inproc.py):
                           // 16 -> 0.751 ms, 32 -> 0.709-0.721, 64 -> 0.685-0.694, 128 -> 0.707 ms per launch (64: 150 VGPRs, one workgroup per CU)
#endif
#ifndef UPSWEEP_SUB
#define UPSWEEP_SUB 4      // histogram copies per wave (power of two)
#endif
// relaxed agent-scope accesses = `sc1` loads / write-through stores: what workgroups of one launch may exchange
// without fences (cdna_hip_programming.md Guideline 16, forms R1 / R2; compiler-visible, so hipcc counts their waits)
__device__ __forceinline__ uint32_t ld_agent(const uint32_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ uint64_t ld_agent(const uint64_t *p) { return __hip_atomic_load(p, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }
__device__ __forceinline__ void st_agent(uint64_t *p, uint64_t v) { __hip_atomic_store(p, v, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT); }

// NEXT: the wave also counts the digit of the FOLLOWING pass (one plain histogram copy per wave) and the block adds
// its sums to `next_totals` -- the pipelined pass needs the digit totals before its first tile is scattered.
// PIPE: the block is one role of lsb_pipe_pass_kernel: key loads with the default cache policy (they must stay in the
// Infinity Cache for the downsweep role; the streaming hint would keep them out), results published write-through.
template <bool NEXT, bool PIPE>
struct UpsweepSmem {
    // every wave counts into UPSWEEP_SUB copies of its histogram (lane & 3 picks one; rows padded by one word so
    // equal digits of different copies sit in different banks): under skew the lanes that share a hot digit
    // spread over four banks instead of queueing on one (Zipf keys: 1.40 -> ~1.0 ms at level 1 of the MSB sort)
    uint32_t hist[LSB_WAVES][UPSWEEP_SUB][RADIX + 1];
    uint32_t hist2[NEXT ? LSB_WAVES : 1][NEXT ? RADIX : 1];
    alignas(8) uint16_t pre[PIPE ? LSB_WAVES : 1][PIPE ? RADIX : 4];   // prefix16 rows on their way to 8-byte stores
    alignas(8) uint32_t tot[PIPE ? RADIX : 2];
};

// Plain dword loads in batches beat 16-byte loads here (0.81 vs 0.84 ms at 2^30) and need no alignment.
// PLAIN: the keys need no transform on the way in (u32 ascending, and every pass after the first: keys travel
// twiddled between passes), so the full-tile path below is load, v_bfe, address, ds_add and nothing else.
template <bool NEXT, bool PIPE, bool PLAIN = false>
__device__ __forceinline__ void upsweep_chunk(UpsweepSmem<NEXT, PIPE> &sm, const uint32_t *__restrict__ keys, uint32_t chunk,
                                              uint32_t *__restrict__ spine, uint16_t *__restrict__ prefix16,
                                              uint32_t *__restrict__ cc, uint32_t *__restrict__ next_totals, const PassParams &p,
                                              const PipeParams &q)
{
    const int tid = threadIdx.x, w = wave_id(), lane = lane_id();
    uint32_t *my = sm.hist[w][lane & (UPSWEEP_SUB - 1)];
    for (int i = lane; i < UPSWEEP_SUB * (RADIX + 1); i += WAVE) (&sm.hist[w][0][0])[i] = 0;
    if (NEXT)
        for (int i = lane; i < RADIX; i += WAVE) sm.hist2[w][i] = 0;

    const uint32_t tile = chunk * LSB_CHUNK + (uint32_t)w;
    if (tile < p.num_tiles) {
        const uint64_t lo = (uint64_t)tile * LSB_TILE;
        const uint32_t len = (p.n - lo < (uint64_t)LSB_TILE) ? (uint32_t)(p.n - lo) : (uint32_t)LSB_TILE;
        const uint32_t *src = keys + lo;
        auto count = [&](uint32_t raw) {
            const uint32_t k = twiddle_in(raw, p.f32_in, p.xor_in);
            hist_add(my, __builtin_amdgcn_ubfe(k, p.shift, p.bits));        // wave-private ds_add_u32
            if (NEXT) hist_add(sm.hist2[w], __builtin_amdgcn_ubfe(k, q.next_shift, q.next_bits));
        };
        constexpr int GB = UPSWEEP_BATCH;
        if (!NEXT && !PIPE && len == (uint32_t)LSB_TILE) {
            // full tile (all but the array's last one): no clamps, no guards, and the test for a digit shared by the
            // whole wave (a hot bucket, constant high bytes: 64 lanes would queue on 4 counters) is made on two keys
            // of the batch instead of on each -- 15 -> 4 vector instructions per key
            uint32_t wbits = p.bits;
            asm volatile("" : "+v"(wbits));   // v_bfe_u32 takes one scalar operand (the shift)
            auto digit_of = [&](uint32_t raw) {
                return __builtin_amdgcn_ubfe(PLAIN ? raw : twiddle_in(raw, p.f32_in, p.xor_in), p.shift, wbits);
            };
#pragma unroll 1
            for (uint32_t j = 0; j < (uint32_t)LSB_TILE; j += GB * WAVE) {
                const uint32_t *at = src + j + lane;
                uint32_t v[GB];
#pragma unroll
                for (int u = 0; u < GB; ++u) v[u] = __builtin_nontemporal_load(at + u * WAVE);
                const uint32_t da = digit_of(v[0]), db = digit_of(v[GB / 2]);
                const bool hot = __builtin_amdgcn_ballot_w64(da == __builtin_amdgcn_readfirstlane(da)) == ~0ull ||
                                 __builtin_amdgcn_ballot_w64(db == __builtin_amdgcn_readfirstlane(db)) == ~0ull;
                if (hot) {
#pragma unroll
                    for (int u = 0; u < GB; ++u) hist_add(my, digit_of(v[u]));
                } else {
#pragma unroll
                    for (int u = 0; u < GB; ++u) atomicAdd(&my[digit_of(v[u])], 1u);
                }
            }
        } else {
        // batches of dword loads from clamped indices: one code path for partial and misaligned tiles
        // (a loop of one guarded load per trip would pay one HBM round trip per 64 keys)
        const uint32_t last = len - 1u;
#pragma unroll 1
        for (uint32_t j = 0; j < len; j += GB * WAVE) {
            uint32_t v[GB];
#pragma unroll
            for (int u = 0; u < GB; ++u) {
                const uint32_t idx = j + u * WAVE + lane;
                const uint32_t *at = &src[idx < last ? idx : last];
                v[u] = PIPE ? *at : __builtin_nontemporal_load(at);   // streaming hint: 0.80 -> 0.76 ms
            }
#pragma unroll
            for (int u = 0; u < GB; ++u)
                if (j + u * WAVE + lane < len) count(v[u]);
        }
        }
    }
    __syncthreads();
#if defined(GS_EXP_UPS) && GS_EXP_UPS == 4
    if (!PIPE && !NEXT && (chunk & 7u) != 0u) return;     // timing experiment: only one workgroup in eight writes its results
#endif
#if defined(GS_EXP_UPS) && GS_EXP_UPS >= 1 && GS_EXP_UPS <= 3
    // timing experiments only (results land in the wrong layout): 1 = the chunk's prefix16 rows as ONE 16-byte store per
    // digit thread (4 KiB per workgroup in four wave instructions instead of 32), 2 = also the spine as one 1 KiB row per
    // chunk, 3 = no result stores at all
    if (!PIPE && !NEXT && tid < RADIX) {
        uint32_t run = 0, pk[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < LSB_WAVES; ++j) {
            pk[j >> 1] |= (run & 0xffffu) << (16 * (j & 1));
            uint32_t c = 0;
#pragma unroll
            for (int u = 0; u < UPSWEEP_SUB; ++u) c += sm.hist[j][u][tid];
            run += c;
        }
#if GS_EXP_UPS < 3
        reinterpret_cast<uint4 *>(prefix16 + (size_t)chunk * LSB_CHUNK * RADIX)[tid] = make_uint4(pk[0], pk[1], pk[2], pk[3]);
#if GS_EXP_UPS == 2
        spine[(size_t)chunk * RADIX + tid] = run;
#else
        spine[(uint32_t)tid * p.grid + chunk] = run;
#endif
#else
        if (run == 0xffffffffu) spine[0] = pk[0] + pk[1] + pk[2] + pk[3];
#endif
        return;
    }
#endif
    if (tid < RADIX) {
        uint32_t run = 0;
#pragma unroll
        for (int j = 0; j < LSB_WAVES; ++j) {
            const uint32_t t = chunk * LSB_CHUNK + (uint32_t)j;
            if (PIPE) sm.pre[j][tid] = (uint16_t)run;
            else if (t < p.num_tiles) prefix16[(size_t)t * RADIX + tid] = (uint16_t)run;
            uint32_t c = 0;
#pragma unroll
            for (int u = 0; u < UPSWEEP_SUB; ++u) c += sm.hist[j][u][tid];
            run += c;
        }
        if (PIPE) sm.tot[tid] = run | (q.tag << 28);       // run <= 65536
        else spine[(uint32_t)tid * p.grid + chunk] = run;
        if (NEXT) {
            uint32_t s2 = 0;
#pragma unroll
            for (int j = 0; j < LSB_WAVES; ++j) s2 += sm.hist2[j][tid];
            if (s2) atomicAdd(&next_totals[tid], s2);
        }
    }
    if (PIPE) {
        // publish: the chunk's prefix16 rows (wave w = tile w, 8 bytes per lane), drained by every storing wave,
        // then -- behind the workgroup's barrier -- the tagged count words the scanner role polls
        __syncthreads();
        uint32_t tid2 = threadIdx.x;
        asm volatile("" : "+v"(tid2));   // recomputed from scratch: otherwise `tile` lives (and spills) across the counting loop
        const uint32_t tile2 = chunk * LSB_CHUNK + (tid2 >> 6);
        if (tile2 < p.num_tiles)
            st_agent(reinterpret_cast<uint64_t *>(prefix16 + (size_t)tile2 * RADIX) + lane,
                     reinterpret_cast<const uint64_t *>(sm.pre[tile2 - chunk * LSB_CHUNK])[lane]);
        asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
        __syncthreads();
        if (tid < RADIX / 2)
            st_agent(reinterpret_cast<uint64_t *>(cc + (size_t)chunk * RADIX) + tid, reinterpret_cast<const uint64_t *>(sm.tot)[tid]);
    }
}

#ifndef GS_EXP_UPS_WPE
#define GS_EXP_UPS_WPE 1
#endif
template <bool NEXT, bool PLAIN = false>
__global__ __launch_bounds__(LSB_THREADS, GS_EXP_UPS_WPE) void lsb_upsweep_kernel(const uint32_t *__restrict__ keys,
                                                                  uint32_t *__restrict__ spine,
                                                                  uint16_t *__restrict__ prefix16,
                                                                  uint32_t *__restrict__ next_totals, PassParams p, PipeParams q)
{
    // blocks are dispatched round-robin over the 8 XCDs; the blocks of one XCD take CONSECUTIVE chunks, so the 16 chunk
    // totals that share a 64-byte line of a spine row are merged in one L2 instead of leaving eight L2s as partial
    // lines (0.736 -> 0.708 ms per launch at 2^30 keys)
    __shared__ UpsweepSmem<NEXT, false> sm;
    upsweep_chunk<NEXT, false, PLAIN>(sm, keys, chunk_of_block(blockIdx.x, p.grid), spine, prefix16, nullptr, next_totals, p, q);
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
#ifdef GS_EXP_ALLWAVE_KEYS
    uint16_t wbase[LSB_WAVES][RADIX];
#else
    uint16_t wbase[HAS_VALUES ? LSB_WAVES : 1][RADIX];    // pairs: tile-absolute base of (wave, digit), < 8192
#endif
    uint32_t gbase[RADIX];                                // global offset of digit run - tile-local start
    uint32_t stage[LSB_TILE * (HAS_VALUES ? 2 : 1)];      // tile in rank order; pairs interleaved {key,val}
    uint32_t dead;                                        // pipelined pass only: wave 0's wait gave up -> the tile stores nothing
};

#ifdef GS_EXP_PHASES
// experiment builds only (tools/phase_exp.py): shader-clock length of every phase of wave 0
__device__ uint32_t gs_phase_buf[131072 * 16];   // [block][phase], n <= 2^30
#define GS_PHASE(k)                                                                          \
    do {                                                                                     \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                        \
        if (tid == 0 && t < 131072u) gs_phase_buf[t * 16 + (k)] = (uint32_t)(now_ - tprev_); \
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
// PIPE = true: the tile is one block of lsb_pipe_pass_kernel; its chunk's scanned counts come from the scanner
// role as {tag, value} granules (`sc`), its in-chunk prefixes from the upsweep role (`prefix16`), both published
// write-through inside the same launch and read here with agent-scope loads.
constexpr uint32_t PIPE_SPIN_LIMIT = 1u << 18;   // polls (each >= one memory round trip) before a wait gives up

template <bool HAS_VALUES, bool TAIL, int TW, bool BIG, bool PIPE>
__device__ __forceinline__ void downsweep_tile(DownsweepSmem<HAS_VALUES> &sm, const uint32_t t,
    const uint32_t *__restrict__ keys_in, uint32_t *__restrict__ keys_out, const uint32_t *__restrict__ vals_in,
    uint32_t *__restrict__ vals_out, const uint32_t *__restrict__ spine, const uint16_t *__restrict__ prefix16,
    const uint32_t *__restrict__ totals, const PassParams &p, const uint64_t *__restrict__ sc, uint32_t tag,
    uint32_t *__restrict__ error_word, const uint32_t tid_ = threadIdx.x)
{
#ifdef GS_EXP_ALLWAVE_KEYS
    constexpr bool ALLWAVE = true;          // experiment: every wave computes its own bases for keys too (no second barrier)
#else
    constexpr bool ALLWAVE = HAS_VALUES;   // see step 3
#endif

    [[maybe_unused]] const int tid = (int)tid_;
    const int lane = (int)(tid_ & 63u), w = (int)(tid_ >> 6);
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
    (void)full_tiles;

#ifdef GS_EXP_PHASES
    unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
    const unsigned long long t0_ = tprev_, r0_ = __builtin_amdgcn_s_memrealtime();
#endif
    // keys only: the waves that issue loads and stores get priority over the ones that rank, so the memory
    // pipes are fed as early as possible (1.87 -> 1.82 ms; with values it costs 7 %, so pairs keep the default)
#ifdef GS_EXP_SLEEP_START
    __builtin_amdgcn_s_sleep(GS_EXP_SLEEP_START);
#endif
    if (!HAS_VALUES) __builtin_amdgcn_s_setprio(3);
    const uint64_t tile_base = (uint64_t)t * LSB_TILE;
    const uint32_t valid = TAIL ? tail_valid : (uint32_t)LSB_TILE;

    // pipelined pass: the chunk's scanned counts are requested first, so they are back before the keys are
    uint64_t scg[4] = {0, 0, 0, 0};
    const uint64_t *scrow = nullptr;
    if (PIPE && w == 0) {
        scrow = sc + (size_t)(t / LSB_CHUNK) * RADIX + 4 * lane;
#pragma unroll
        for (int q = 0; q < 4; ++q) scg[q] = ld_agent(scrow + q);
    }

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
#ifndef GS_EXP_RANK_PRIO
#define GS_EXP_RANK_PRIO 0
#endif
    if (!HAS_VALUES) __builtin_amdgcn_s_setprio(GS_EXP_RANK_PRIO);
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
    if (!TAIL && !PIPE && w == 0) {
        const uint32_t *sp = spine + (uint32_t)(4 * lane) * p.grid + t / LSB_CHUNK;
        const uint2 pf = reinterpret_cast<const uint2 *>(prefix16 + (size_t)t * RADIX)[lane];
        tbase[0] = sp[0] + (pf.x & 0xffffu);
        tbase[1] = sp[p.grid] + (pf.x >> 16);
        tbase[2] = sp[2 * p.grid] + (pf.y & 0xffffu);
        tbase[3] = sp[3 * p.grid] + (pf.y >> 16);
    }

    if (PIPE && !TAIL && w == 0) {
        // every granule carries this pass's tag once the scanner has written it (normally long ago: the upsweep
        // role runs PIPE_LEAD_CHUNKS ahead); only then may the prefix16 row be read (it was published before the
        // counts the scanner waited for).  Bounded: a wait that gives up flags the sort instead of hanging the GPU --
        // and the tile then stores NOTHING: its offsets would come from untagged words (stale values of another pass
        // can exceed n: an out-of-bounds scatter), so the whole workgroup leaves behind the first barrier.
        uint32_t spins = 0;
        bool gave_up = false;
        for (;;) {
            const bool ok = (uint32_t)(scg[0] >> 32) == tag && (uint32_t)(scg[1] >> 32) == tag &&
                            (uint32_t)(scg[2] >> 32) == tag && (uint32_t)(scg[3] >> 32) == tag;
            if (__builtin_amdgcn_ballot_w64(!ok) == 0) break;
            if (++spins > PIPE_SPIN_LIMIT) {
                if (lane == 0) atomicOr(error_word, 1u);
                gave_up = true;
                break;
            }
            __builtin_amdgcn_s_sleep(8);
#pragma unroll
            for (int q = 0; q < 4; ++q) scg[q] = ld_agent(scrow + q);
        }
        if (lane == 0) sm.dead = gave_up ? 1u : 0u;
        const uint64_t pf = ld_agent(reinterpret_cast<const uint64_t *>(prefix16 + (size_t)t * RADIX) + lane);
        tbase[0] = (uint32_t)scg[0] + (uint32_t)(pf & 0xffffu);
        tbase[1] = (uint32_t)scg[1] + (uint32_t)((pf >> 16) & 0xffffu);
        tbase[2] = (uint32_t)scg[2] + (uint32_t)((pf >> 32) & 0xffffu);
        tbase[3] = (uint32_t)scg[3] + (uint32_t)(pf >> 48);
    }

    // global base of digit run = digit start + tile offset - tile-local start (wave 0, lane l: digits 4l..4l+3)
    auto publish_gbase = [&](const uint32_t (&ex)[4], const uint32_t (&run)[4]) {
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
    if (PIPE && !TAIL && sm.dead) return;          // the wait for this tile's offsets gave up: no global store (all waves alike)

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
            } else if (ALLWAVE) {
                sm.stage[pos[i] + wb[i]] = key[i];
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
    // Pacing (round 3, in-process A/B on the same buffers, tools/ab_inproc.py): every wave pauses 32 x 64 cycles (~0.9 us) between
    // barrier 3 and its 16 stores.  What it buys depends on where the driver placed the arrays: on placements where the kernel
    // runs 1.89-1.94 ms per launch without the pause it runs 1.78-1.79 ms with it; on placements where it runs 1.74 ms without,
    // the pause costs 0.7-1.5 % (1.755-1.77 ms) -- profiles/r03_ab_pacing.txt has both kinds, from the same box and process
    // sequence.  So the pause takes 8 % off the slow placements and the spread between placements shrinks from 11 % to 2 %.
    // Pauses of 8/16/24 recover less of the slow case (1.91/1.88/1.81 ms), 40-64 cost more of the fast one; pausing only the odd
    // waves (behind a scalar branch) 1.85 ms.  Only from 2^29 keys up: below, the launch is not bound by the memory system
    // and the pause is latency (2^28 keys: +1 %; 2^22-2^24: +5 %).  Pairs gain nothing from it (3.57-3.62 ms with 16/32, 3.71 with
    // 64, 3.58-3.60 without).  GS_EXP_SLEEP* override all of it for experiments.
#ifndef GS_EXP_SLEEP
#define GS_EXP_SLEEP 32
#define GS_EXP_SLEEP_MODE_DEFAULT 0
#else
#define GS_EXP_SLEEP_MODE_DEFAULT GS_EXP_SLEEP_MODE
#endif
    if ((!HAS_VALUES || GS_EXP_SLEEP_PAIRS) && full_tiles >= GS_EXP_SLEEP_MIN_TILES) {
        // s_sleep is a scalar instruction: it must sit behind a SCALAR branch (a branch on a vector condition only masks lanes
        // and the wave sleeps all the same), hence the readfirstlane
        [[maybe_unused]] const int ws = __builtin_amdgcn_readfirstlane(w);
#if GS_EXP_SLEEP_MODE_DEFAULT == 3
        if (ws & 1) __builtin_amdgcn_s_sleep(GS_EXP_SLEEP); else __builtin_amdgcn_s_sleep(GS_EXP_SLEEP_B);
#elif GS_EXP_SLEEP_MODE_DEFAULT == 4
        if (ws >= 4) __builtin_amdgcn_s_sleep(GS_EXP_SLEEP);
#elif GS_EXP_SLEEP_MODE_DEFAULT == 1
        if (ws & 1) __builtin_amdgcn_s_sleep(GS_EXP_SLEEP);
#elif GS_EXP_SLEEP_MODE_DEFAULT == 0
        __builtin_amdgcn_s_sleep(GS_EXP_SLEEP);
#endif
    }
#ifndef GS_EXP_STORE_PRIO
#define GS_EXP_STORE_PRIO 3
#endif
    if (!HAS_VALUES) __builtin_amdgcn_s_setprio(GS_EXP_STORE_PRIO);
#pragma unroll
    for (int i = 0; i < LSB_KPT; ++i) {
        // a wave stores 1024 CONSECUTIVE slots (not every 512th 64-slot group): its 16 store instructions walk ~32
        // neighbouring digit runs in order instead of touching ~48 runs all over the output -- keys 1.92 -> 1.81 ms,
        // pairs 4.37 -> 4.15 ms per pass on the same box (the output pages are reused by consecutive instructions)
        const uint32_t slot = (uint32_t)w * (WAVE * LSB_KPT) + i * WAVE + lane;
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
#ifdef GS_EXP_DRAIN
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");   // experiment: the wave stays until its stores are acknowledged
#endif
    GS_PHASE_WAIT("vmcnt(0)");
    GS_PHASE(8);                                   // store drain
#ifdef GS_EXP_PHASES
    if (tid == 0 && t < 131072u) {                 // clock calibration: shader clocks vs 100 MHz real time; who and where
        gs_phase_buf[t * 16 + 9] = (uint32_t)(__builtin_amdgcn_s_memtime() - t0_);
        gs_phase_buf[t * 16 + 10] = (uint32_t)(__builtin_amdgcn_s_memrealtime() - r0_);
        gs_phase_buf[t * 16 + 11] = (uint32_t)r0_;
        gs_phase_buf[t * 16 + 12] = blockIdx.x;
        gs_phase_buf[t * 16 + 13] = __builtin_amdgcn_s_getreg((31 << 11) | 4);    // HW_ID
        gs_phase_buf[t * 16 + 14] = __builtin_amdgcn_s_getreg((31 << 11) | 20);   // XCC_ID
    }
#endif
}

// The arguments the first instructions need (key count, digit position, spine row length, the key pointers) come first
// and as plain scalars: with -mllvm -amdgpu-kernarg-preload-count they arrive in SGPRs with the wave, so the key loads
// are issued without a scalar-load round trip (two dependent ones before).  `p` still carries everything else.
template <bool HAS_VALUES, bool TAIL, int TW, bool BIG>
__global__ __launch_bounds__(LSB_THREADS, HAS_VALUES ? 4 : 6) void lsb_downsweep_kernel(
    const uint32_t n_, const uint32_t shift_, const uint32_t bits_, const uint32_t grid_,
    const uint32_t *__restrict__ keys_in, uint32_t *__restrict__ keys_out, const uint32_t *__restrict__ totals,
    const uint32_t *__restrict__ spine, const uint16_t *__restrict__ prefix16, const uint32_t *__restrict__ vals_in,
    uint32_t *__restrict__ vals_out, PassParams p)
{
    __shared__ __attribute__((aligned(16))) DownsweepSmem<HAS_VALUES> sm;
    p.n = n_; p.shift = shift_; p.bits = bits_; p.grid = grid_;
    const uint32_t full_tiles = p.n / (uint32_t)LSB_TILE;
#if defined(GS_EXP_TPB) || defined(GS_EXP_PERSIST)
    // experiment builds only (profiles/r02_lsb_structure_experiments.txt): a block walks several tiles, plainly one after the other
    if (!TAIL) {
#ifdef GS_EXP_STAGGER
        if (((blockIdx.x / 256u) % 3u) >= 1u) __builtin_amdgcn_s_sleep(GS_EXP_STAGGER);
        if (((blockIdx.x / 256u) % 3u) >= 2u) __builtin_amdgcn_s_sleep(GS_EXP_STAGGER);
#endif
#ifdef GS_EXP_TPB
        for (uint32_t j = 0; j < (uint32_t)GS_EXP_TPB; ++j) {
            const uint32_t item = blockIdx.x * (uint32_t)GS_EXP_TPB + j;
#else
        for (uint32_t j = 0;; ++j) {
            const uint32_t item = blockIdx.x + j * gridDim.x;
#endif
            if (item >= full_tiles) return;
            uint32_t tz = threadIdx.x;            // opaque: nothing thread-derived stays live across tiles
            asm volatile("" : "+v"(tz));
            downsweep_tile<HAS_VALUES, TAIL, TW, BIG, false>(sm, tile_of_item(item, full_tiles), keys_in, keys_out, vals_in, vals_out,
                                                             spine, prefix16, totals, p, nullptr, 0u, nullptr, tz);
            __syncthreads();
        }
        return;
    }
#endif
    if (!TAIL && blockIdx.x >= full_tiles) return;   // (n_ is preloaded: two scalar instructions, no memory wait)
#ifdef GS_EXP_WIDE_MIN_TILES
    const uint32_t t = TAIL ? full_tiles : full_tiles < GS_EXP_WIDE_MIN_TILES ? tile_of_item(blockIdx.x, full_tiles) : tile_of_item_wide(blockIdx.x, full_tiles);
#else
    const uint32_t t = TAIL ? full_tiles : tile_of_item_wide(blockIdx.x, full_tiles);
#endif
    downsweep_tile<HAS_VALUES, TAIL, TW, BIG, false>(sm, t, keys_in, keys_out, vals_in, vals_out, spine, prefix16, totals, p,
                                                     nullptr, 0u, nullptr);
}


// ---------------------------------------------------------- pipelined pass --
// One launch = one pass, the north_star's three steps as ROLES of its workgroups:
//   blocks 0..7        scanner: walk the chunk rows in order and turn the tagged counts the upsweep role publishes
//                      (`cc`) into exclusive prefixes over the earlier chunks (`sc`), 32 digit columns per block;
//   upsweep blocks     one chunk each (upsweep_chunk): counts, in-chunk prefixes, and the NEXT pass's digit totals;
//   downsweep blocks   one tile each (downsweep_tile).
// Blocks are dispatched in index order, and the index order puts every upsweep block PIPE_LEAD_CHUNKS chunks (1024
// tiles) ahead of the downsweep blocks of the same keys: 8 upsweep blocks, then the 64 downsweep blocks of 64 tiles
// whose counts were taken two groups earlier.  So (a) nothing a block waits for depends on a block dispatched after
// it (the scanner and the upsweep blocks it needs have lower indices), (b) the downsweep reads keys the upsweep
// brought into the Infinity Cache ~15 us earlier: 4 of the pass's 12 B/key do not reach HBM (tools/micro/mall_reuse.hip:
// copy + look-ahead read 1.73 ms, copy + unrelated read 2.42 ms, copy alone 1.60 ms), (c) the upsweep's memory waits
// overlap the downsweep's ranking on the same CUs.  Needs the digit totals BEFORE the launch: the previous pass's
// upsweep gathers them (NEXT).  Every wait is bounded and reports through `error_word`.
// Scanner role.  The upsweep role publishes 8.75 chunk rows per microsecond at 2^30 keys and a dependent read takes
// 4-8 us under the pass's own load, so far more than 100 rows must be in flight: the 8 scanner blocks (one per XCD
// round-robin slot) are 64 INDEPENDENT waves, each owning 4 digit columns; inside a wave the 16 lanes of a DPP row
// share one digit and take 4 consecutive rows each, so a batch is 64 rows, its prefix is a DPP row scan (no LDS, no
// barrier), and PIPE_SCAN_AHEAD batches are requested before the oldest is consumed (256 rows in flight per wave).
// A batch is written once all its rows carry the tag: it must stay shorter than the upsweep's lead (deadlock
// otherwise: downsweep blocks waiting for rows whose upsweep blocks are dispatched behind them).
constexpr int PIPE_SCAN_K = 4;                           // rows per lane and batch
constexpr int PIPE_SCAN_AHEAD = 3;                       // batches requested ahead of the one being consumed
constexpr uint32_t PIPE_ROW_PAD = 16 * PIPE_SCAN_K;      // batch length; cc / sc have this many rows beyond the last chunk
static_assert(PIPE_ROW_PAD < PIPE_LEAD_CHUNKS, "a scanner batch must be shorter than the upsweep's lead");

// inclusive scan inside each DPP row of 16 lanes
__device__ __forceinline__ uint32_t row16_inclusive_scan(uint32_t x)
{
    uint32_t v = x;
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x111, 0xf, 0xf, false);  // row_shr:1
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x112, 0xf, 0xf, false);  // row_shr:2
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)x, 0x113, 0xf, 0xf, false);  // row_shr:3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x114, 0xf, 0xe, false);  // row_shr:4, banks 1-3
    v += (uint32_t)__builtin_amdgcn_update_dpp(0, (int)v, 0x118, 0xf, 0xc, false);  // row_shr:8, banks 2-3
    return v;
}

// `rows` is a multiple of the batch: the upsweep role publishes zero counts for the rows past the last chunk.
__device__ __forceinline__ void pipe_scan_rows(uint32_t slice, const uint32_t *__restrict__ cc, uint64_t *__restrict__ sc,
                                               uint32_t rows, uint32_t tag, uint32_t *__restrict__ error_word)
{
    constexpr int K = PIPE_SCAN_K, A = PIPE_SCAN_AHEAD;
    const int lane = lane_id();
    const uint32_t d = slice * 32u + (uint32_t)wave_id() * 4u + ((uint32_t)lane >> 4);   // my digit column
    const uint32_t g = (uint32_t)lane & 15u;                                              // my row lane
    const uint32_t nb = rows / PIPE_ROW_PAD;
    // 32-bit byte offsets from the scalar bases (<= 64 MiB and 128 MiB at 2^32 keys), rows at constant distances
    auto off_of = [&](uint32_t b) { return ((b * PIPE_ROW_PAD + g * K) * (uint32_t)RADIX + d) * 4u; };
    auto request = [&](uint32_t (&v)[K], uint32_t b) {
        uint32_t off = off_of(b);
        asm volatile("" : "+v"(off));
#pragma unroll
        for (int k = 0; k < K; ++k)
            v[k] = ld_agent(reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(cc) + (off + (uint32_t)k * RADIX * 4u)));
    };
    uint32_t run = 0;
    // returns false when the wait for a batch gave up: the scanner then publishes NOTHING more (rows tagged as valid but
    // summed from untagged counts would be trusted by every tile behind them); the tiles that depend on the missing
    // rows time out in turn and store nothing
    auto consume = [&](uint32_t (&v)[K], uint32_t b) -> bool {
        uint32_t spins = 0;
        for (;;) {
            bool bad = false;
#pragma unroll
            for (int k = 0; k < K; ++k) bad |= (v[k] >> 28) != tag;
            if (__builtin_amdgcn_ballot_w64(bad) == 0) break;
            if (++spins > PIPE_SPIN_LIMIT) {
                if (lane == 0) atomicOr(error_word, 2u);
                return false;
            }
            __builtin_amdgcn_s_sleep(8);
            request(v, b);
        }
        uint32_t tot = 0;
#pragma unroll
        for (int k = 0; k < K; ++k) { v[k] &= 0x0fffffffu; tot += v[k]; }
        const uint32_t inc = row16_inclusive_scan(tot);
        uint32_t acc = run + inc - tot;
        run += (uint32_t)__shfl((int)inc, lane | 15, WAVE);   // the row's total
        const uint32_t off8 = off_of(b) * 2u;
#pragma unroll
        for (int k = 0; k < K; ++k) {
            st_agent(reinterpret_cast<uint64_t *>(reinterpret_cast<char *>(sc) + (off8 + (uint32_t)k * RADIX * 8u)),
                     ((uint64_t)tag << 32) | (uint64_t)acc);
            acc += v[k];
        }
        return true;
    };
    static_assert(A == 3, "the ring below is written out for three batches ahead");
    uint32_t v0[K], v1[K], v2[K], v3[K];
    if (nb > 0) request(v0, 0);
    if (nb > 1) request(v1, 1);
    if (nb > 2) request(v2, 2);
#pragma unroll 1
    for (uint32_t b = 0; b < nb; b += 4) {
        if (b + 3 < nb) request(v3, b + 3);
        if (!consume(v0, b)) return;
        if (b + 1 >= nb) break;
        if (b + 4 < nb) request(v0, b + 4);
        if (!consume(v1, b + 1)) return;
        if (b + 2 >= nb) break;
        if (b + 5 < nb) request(v1, b + 5);
        if (!consume(v2, b + 2)) return;
        if (b + 3 >= nb) break;
        if (b + 6 < nb) request(v2, b + 6);
        if (!consume(v3, b + 3)) return;
    }
}

template <bool HAS_VALUES>
union PipeSmem {
    DownsweepSmem<HAS_VALUES> ds;
    UpsweepSmem<true, true> us;
    UpsweepSmem<false, true> us1;
};

template <bool HAS_VALUES, int TW, bool BIG>
__global__ __launch_bounds__(LSB_THREADS, HAS_VALUES ? 4 : 6) void lsb_pipe_pass_kernel(
    const uint32_t *__restrict__ keys_in, uint32_t *__restrict__ keys_out, const uint32_t *__restrict__ vals_in,
    uint32_t *__restrict__ vals_out, uint32_t *__restrict__ cc, uint64_t *__restrict__ sc, uint16_t *__restrict__ prefix16,
    const uint32_t *__restrict__ totals, uint32_t *__restrict__ next_totals, uint32_t *__restrict__ error_word, PassParams p,
    PipeParams q)
{
    __shared__ __attribute__((aligned(16))) PipeSmem<HAS_VALUES> sm;
    uint32_t x = blockIdx.x;
    if (x < MI355X_XCDS) {       // the scanner: a whole round of the XCD round-robin, so that downsweep item i sits on XCD i % 8
        pipe_scan_rows(x, cc, sc, q.scan_rows, q.tag, error_word);
        return;
    }
    x -= MI355X_XCDS;
    uint32_t chunk;
    if (x < q.lead_chunks) {
        chunk = x;
    } else {
        x -= q.lead_chunks;
        const uint32_t g = x / PIPE_GROUP_BLOCKS, r = x % PIPE_GROUP_BLOCKS, sub = r / PIPE_SUB_BLOCKS, s = r % PIPE_SUB_BLOCKS;
        if (s >= 8u) {
            const uint32_t full_tiles = p.n / (uint32_t)LSB_TILE;
            const uint32_t item = g * (uint32_t)LSB_GROUP + sub * 64u + (s - 8u);
            if (item >= full_tiles) return;
            downsweep_tile<HAS_VALUES, false, TW, BIG, true>(sm.ds, tile_of_item(item, full_tiles), keys_in, keys_out, vals_in,
                                                             vals_out, nullptr, prefix16, totals, p, sc, q.tag, error_word);
            return;
        }
        chunk = q.lead_chunks + g * (uint32_t)(LSB_GROUP / LSB_CHUNK) + sub * 8u + s;
    }
    if (chunk >= p.grid) {   // rows past the last chunk: zero counts, so the scanner's batches are whole
        if (chunk < q.scan_rows && threadIdx.x < RADIX / 2)
            st_agent(reinterpret_cast<uint64_t *>(cc + (size_t)chunk * RADIX) + threadIdx.x, ((uint64_t)(q.tag << 28) << 32) | (q.tag << 28));
        return;
    }
    if (chunk + 1u == q.test_drop_chunk_plus1) return;       // test hook: this chunk's counts are never published
    if (q.next_bits) upsweep_chunk<true, true>(sm.us, keys_in, chunk, nullptr, prefix16, cc, next_totals, p, q);
    else upsweep_chunk<false, true>(sm.us1, keys_in, chunk, nullptr, prefix16, cc, next_totals, p, q);
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
// pipelined passes: tagged chunk counts, scanned granules, digit totals per pass, error word -- one block, zeroed per sort
static inline size_t cc_bytes(uint64_t n) { return align256((size_t)RADIX * (lsb_grid(n) + PIPE_ROW_PAD) * sizeof(uint32_t)); }
static inline size_t sc_bytes(uint64_t n) { return align256((size_t)RADIX * (lsb_grid(n) + PIPE_ROW_PAD) * sizeof(uint64_t)); }
static inline size_t ptotals_bytes() { return align256(5 * RADIX * sizeof(uint32_t)) + 256; }
static inline size_t pipe_bytes_full(uint64_t n) { return cc_bytes(n) + sc_bytes(n) + ptotals_bytes(); }

// LSB pass strategy.  "three" (default) = upsweep -> scan -> downsweep as three launches per pass.  "pipe" = the
// first pass as three launches, every later pass as ONE launch whose workgroups take the three roles
// (lsb_pipe_pass_kernel), for arrays of more than LSB_SMALL_TILES tiles.  Same steps, same results;
// GS_LSB_MODE=three|pipe selects.  Measured on MI355X at 2^30 keys: a pipelined pass takes
// 4.1-5.4 ms against 2.7 ms for the three launches -- an upsweep workgroup occupies a downsweep slot for 15-30 us --
// so it is an opt-in experiment, kept because it is bit-exact, tested, and the basis of that measurement.
static inline bool pipe_enabled()
{
    static const char *e = getenv("GS_LSB_MODE");   // read once per process
    return e && strcmp(e, "pipe") == 0;
}

// smallest array (in tiles) that takes pipelined passes; GS_LSB_PIPE_MIN_TILES lowers it so tests reach the kernel with small inputs
static inline uint32_t pipe_min_tiles()
{
    static const uint32_t v = [] { const char *e = getenv("GS_LSB_PIPE_MIN_TILES"); return e ? (uint32_t)strtoul(e, nullptr, 10) : LSB_SMALL_TILES + 1u; }();
    return v;
}

static inline bool pipe_size_ok(uint64_t n) { return pipe_enabled() && lsb_num_tiles(n) >= pipe_min_tiles() && n > small_sort_capacity(false); }

// the pipelined passes' block exists only in processes that asked for them (the error word always does)
static inline size_t pipe_bytes(uint64_t n) { return pipe_enabled() ? pipe_bytes_full(n) : ptotals_bytes(); }
size_t lsb_temp_bytes(uint64_t n) { return spine_bytes(n) + totals_bytes() + prefix16_bytes(n) + pipe_bytes(n); }
LsbWorkspace lsb_carve(void *temp, uint64_t n)
{
    char *c = (char *)temp;
    LsbWorkspace ws;
    ws.spine = (uint32_t *)c;
    ws.totals = (uint32_t *)(c + spine_bytes(n));
    ws.prefix16 = (uint16_t *)(c + spine_bytes(n) + totals_bytes());
    char *pb = c + spine_bytes(n) + totals_bytes() + prefix16_bytes(n);
    const bool pe = pipe_enabled();
    ws.cc = pe ? (uint32_t *)pb : nullptr;
    ws.sc = pe ? (uint64_t *)(pb + cc_bytes(n)) : nullptr;
    ws.ptotals = (uint32_t *)(pb + (pe ? cc_bytes(n) + sc_bytes(n) : 0));
    ws.error_word = ws.ptotals + 5 * RADIX;
    return ws;
}

int lsb_upsweep(const uint32_t *keys, uint32_t *spine, uint16_t *prefix16, const PassParams &p, hipStream_t s)
{
    KernelTimer kt(GS_K_LSB_UPSWEEP, s);
    const dim3 grid(p.grid);
    if (!p.f32_in && !p.xor_in)
        hipLaunchKernelGGL((lsb_upsweep_kernel<false, true>), grid, dim3(LSB_THREADS), 0, s, keys, spine, prefix16,
                           (uint32_t *)nullptr, p, PipeParams{});
    else
        hipLaunchKernelGGL((lsb_upsweep_kernel<false, false>), grid, dim3(LSB_THREADS), 0, s, keys, spine, prefix16,
                           (uint32_t *)nullptr, p, PipeParams{});
    return (int)hipGetLastError();
}

// the same, also gathering the digit totals of the pass that follows (into next_totals, zeroed by the caller)
static int lsb_upsweep_next(const uint32_t *keys, uint32_t *spine, uint16_t *prefix16, uint32_t *next_totals, const PassParams &p,
                            const PipeParams &q, hipStream_t s)
{
    KernelTimer kt(GS_K_LSB_UPSWEEP, s);
    hipLaunchKernelGGL(lsb_upsweep_kernel<true>, dim3(p.grid), dim3(LSB_THREADS), 0, s, keys, spine, prefix16, next_totals, p, q);
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
#if defined(GS_EXP_TPB)
    const dim3 grid((p.ds_grid + GS_EXP_TPB - 1) / GS_EXP_TPB);
#elif defined(GS_EXP_PERSIST)
    const dim3 grid(p.ds_grid < (uint32_t)GS_EXP_PERSIST ? p.ds_grid : (uint32_t)GS_EXP_PERSIST);
#else
    const dim3 grid(p.ds_grid);
#endif
    if (vin)
        hipLaunchKernelGGL((lsb_downsweep_kernel<true, false, TW, BIG>), grid, block, 0, s, p.n, p.shift, p.bits, p.grid, kin, kout,
                           totals, spine, prefix16, vin, vout, p);
    else
        hipLaunchKernelGGL((lsb_downsweep_kernel<false, false, TW, BIG>), grid, block, 0, s, p.n, p.shift, p.bits, p.grid, kin, kout,
                           totals, spine, prefix16, vin, vout, p);
}

static void launch_downsweep_tail(const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
                                  const uint32_t *totals, const PassParams &p, hipStream_t s)
{
    const dim3 block(LSB_THREADS);   // one block, the general variant
    if (vin)
        hipLaunchKernelGGL((lsb_downsweep_kernel<true, true, 2, true>), dim3(1), block, 0, s, p.n, p.shift, p.bits, p.grid, kin, kout,
                           totals, (const uint32_t *)nullptr, (const uint16_t *)nullptr, vin, vout, p);
    else
        hipLaunchKernelGGL((lsb_downsweep_kernel<false, true, 2, true>), dim3(1), block, 0, s, p.n, p.shift, p.bits, p.grid, kin, kout,
                           totals, (const uint32_t *)nullptr, (const uint16_t *)nullptr, vin, vout, p);
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

template <int TW, bool BIG>
static void launch_pipe_pass(const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, const LsbWorkspace &ws,
                             const uint32_t *totals, uint32_t *next_totals, const PassParams &p, const PipeParams &q, uint32_t blocks,
                             hipStream_t s)
{
    const dim3 block(LSB_THREADS);
    if (vin)
        hipLaunchKernelGGL((lsb_pipe_pass_kernel<true, TW, BIG>), dim3(blocks), block, 0, s, kin, kout, vin, vout, ws.cc, ws.sc,
                           ws.prefix16, totals, next_totals, ws.error_word, p, q);
    else
        hipLaunchKernelGGL((lsb_pipe_pass_kernel<false, TW, BIG>), dim3(blocks), block, 0, s, kin, kout, vin, vout, ws.cc, ws.sc,
                           ws.prefix16, totals, next_totals, ws.error_word, p, q);
}

// one pipelined pass: all full tiles in one launch (+ the partial last tile, which needs the digit totals only)
static int lsb_pipe_pass(const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, const LsbWorkspace &ws,
                         const uint32_t *totals, uint32_t *next_totals, const PassParams &p, const PipeParams &q, hipStream_t s)
{
    KernelTimer kt(GS_K_LSB_PASS, s);
    const uint32_t full = p.n / (uint32_t)LSB_TILE, groups = (full + LSB_GROUP - 1u) / LSB_GROUP;
    const uint32_t blocks = MI355X_XCDS + q.lead_chunks + groups * PIPE_GROUP_BLOCKS;
    const int tw = (p.f32_in || p.f32_out) ? 2 : ((p.xor_in | p.xor_out) ? 1 : 0);
    const bool big = p.n > (1u << 30);
#define GS_PP(TW_, BIG_) launch_pipe_pass<TW_, BIG_>(kin, kout, vin, vout, ws, totals, next_totals, p, q, blocks, s)
    if (big) { if (tw == 2) GS_PP(2, true); else if (tw == 1) GS_PP(1, true); else GS_PP(0, true); }
    else { if (tw == 2) GS_PP(2, false); else if (tw == 1) GS_PP(1, false); else GS_PP(0, false); }
#undef GS_PP
    if (p.n % (uint32_t)LSB_TILE) launch_downsweep_tail(kin, kout, vin, vout, totals, p, s);
    return (int)hipGetLastError();
}

template <typename Route>
static int lsb_run_passes(const LsbWorkspace &ws, uint64_t num_items, int begin_bit, int end_bit, int descending,
                          int key_type, bool pairs, hipStream_t s, Route route)
{
    const int num_bits = end_bit - begin_bit;
    const int num_passes = (num_bits + RADIX_BITS - 1) / RADIX_BITS;
    const bool pipe = num_passes > 1 && pipe_size_ok(num_items);
    if (pipe) {   // tags, totals and the error word start from zero
        const hipError_t me = zero_async(ws.cc, pipe_bytes(num_items), s);
        if (me != hipSuccess) return (int)me;
    } else if (pipe_size_ok(num_items)) {   // a size gs_lsb_pipe_status reads the word for: keep it meaningful
        const hipError_t me = zero_async(ws.error_word, 8, s);
        if (me != hipSuccess) return (int)me;
    }
    for (int pass = 0; pass < num_passes; ++pass) {
        const int shift = begin_bit + pass * RADIX_BITS;
        const int bits = (end_bit - shift < RADIX_BITS) ? end_bit - shift : RADIX_BITS;
        PassParams p = lsb_make_params(num_items, shift, bits);
        lsb_twiddle_masks(key_type, descending, pass == 0, pass == num_passes - 1, p);
        const uint32_t *kin, *vin;
        uint32_t *kout, *vout;
        route(pass, num_passes, kin, kout, vin, vout);
        if (!pairs) { vin = nullptr; vout = nullptr; }
        PipeParams q{};
        if (pipe) {
            q.tag = (uint32_t)pass + 1u;
            q.lead_chunks = PIPE_LEAD_CHUNKS;
            q.scan_rows = (p.grid + PIPE_ROW_PAD - 1u) / PIPE_ROW_PAD * PIPE_ROW_PAD;   // whole scanner batches (<= grid + pad - 1)
            if (const char *e = getenv("GS_LSB_PIPE_TEST_DROP")) q.test_drop_chunk_plus1 = (uint32_t)strtoul(e, nullptr, 10) + 1u;   // test hook
            if (pass + 1 < num_passes) {
                q.next_shift = (uint32_t)(shift + RADIX_BITS);
                q.next_bits = (uint32_t)((end_bit - (int)q.next_shift < RADIX_BITS) ? end_bit - (int)q.next_shift : RADIX_BITS);
            }
        }
        int e;
        if (pipe && pass > 0) {
            uint32_t *next_totals = ws.ptotals + (size_t)(pass + 1) * RADIX;
            if ((e = lsb_pipe_pass(kin, kout, vin, vout, ws, ws.ptotals + (size_t)pass * RADIX, next_totals, p, q, s))) return e;
            continue;
        }
        if (p.num_tiles <= LSB_SMALL_TILES && !pipe) {   // latency-bound sizes: workgroup per tile, chunk prefixes in the scan
            { KernelTimer kt(GS_K_LSB_UPSWEEP, s);
              hipLaunchKernelGGL(lsb_upsweep_small_kernel, dim3(p.num_tiles), dim3(LSB_THREADS), 0, s, kin, ws.prefix16, p); }
            { KernelTimer kt(GS_K_LSB_SCAN, s);
              hipLaunchKernelGGL(lsb_scan_small_kernel, dim3(RADIX), dim3(RADIX), 0, s, ws.spine, ws.totals, ws.prefix16, p.grid,
                                 p.num_tiles); }
            if ((e = (int)hipGetLastError())) return e;
        } else {
            if (pipe) { if ((e = lsb_upsweep_next(kin, ws.spine, ws.prefix16, ws.ptotals + RADIX, p, q, s))) return e; }
            else if ((e = lsb_upsweep(kin, ws.spine, ws.prefix16, p, s))) return e;
            if ((e = lsb_scan(ws.spine, ws.totals, p.grid, s))) return e;
        }
        if ((e = lsb_downsweep(kin, kout, vin, vout, ws.spine, ws.prefix16, ws.totals, p, s))) return e;
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

int gs_lsb_pipe_status(void *d_temp, uint64_t num_items, uint32_t *h_status, void *stream)
{
    GS_CLEAR_STALE_ERROR();
    if (!d_temp || !h_status || num_items >= (1ull << 32)) return hipErrorInvalidValue;
    *h_status = 0;
    if (!pipe_size_ok(num_items)) return hipSuccess;   // such arrays never take pipelined passes
    const LsbWorkspace ws = lsb_carve(d_temp, num_items);
    hipError_t e = hipMemcpyAsync(h_status, ws.error_word, sizeof(uint32_t), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    return (int)e;
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
    if (num_items <= small_sort_capacity(d_vals != nullptr)) {
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
    if (num_items <= small_sort_capacity(pairs)) {   // one workgroup, straight from IN to OUT
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
