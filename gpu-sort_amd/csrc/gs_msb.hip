// gs_msb.hip -- unstable MSD hybrid radix sort for gfx950 (MI355X), ascending.
//
// Replaces (behaviour, not code) rdxsrt_unstable_sort and its kernels
// (msb/src/sort/gpu_radix_sort.h:197-507, msb/src/sort/cuda_radix_sort.h,
// msb/src/sort/gpu_radix_sort.cu):
//   - most-significant byte first; a bucket that fits one workgroup's LDS is
//     finished there by a local LSD sort on its remaining bits and written
//     straight to the result buffer (M7), bigger buckets are partitioned on
//     their next byte (M3 histogram, M4 offsets + classification, M6 scatter);
//   - adjacent tiny sub-buckets are merged into one local-sort task while their
//     sum stays below 3000 keys and re-sorted on one more byte (M4 rules,
//     cuda_radix_sort.h:1084-1087, cuda_radix_sort_config.h:9);
//   - values are only defined up to permutation inside equal keys (unstable);
//   - for 32-bit keys the result ends in the caller's INPUT arrays
//     (gpu_radix_sort.h:359-360).
//
// What is different by design (MI355X-first):
//   - the whole schedule lives on the device: bucket lists, tile prefixes and
//     task lists are built by the classify kernel with atomics, and every level
//     launches fixed upper-bound grids whose surplus blocks exit at once; the
//     reference's three blocking device->host round trips per pass
//     (gpu_radix_sort.cu:31-49) are gone, and so are its 8 cudaMallocs per call
//     (one caller-provided workspace);
//   - the top-byte partition is one stable LSB pass (gs_lsb.hip) at shift 24;
//   - 160 KiB of LDS per CU lets a workgroup finish buckets of up to 17408 keys
//     (reference: 9216), so uniform 2^30 keys need two partitions + one local
//     sort = 32 B/key of HBM traffic (reference thresholds: 44 B/key);
//   - ranking inside the local sort is the wave64 ballot/popcount match of the
//     LSB path; the scatter of an unstable partition ranks with LDS atomics.
#include "gs_device.hpp"
#include <new>
#include "gs_lsb.hpp"
#include <type_traits>
#include <vector>
#include <cstdlib>

namespace gs {

constexpr int MSB_THREADS = 512;
constexpr int MSB_WAVES = MSB_THREADS / WAVE;
constexpr int MSB_KPT = 16;
constexpr int MSB_TILE = MSB_THREADS * MSB_KPT;    // 8192 keys per partition tile
constexpr int MSB_NCLASS = 4;                      // local-sort size classes (reference: 7-9 configs)
constexpr uint32_t MSB_MERGE = 3000;               // merge adjacent sub-buckets while the sum is below this
constexpr uint32_t MSB_MAX_GRID = 16384;           // blocks per launch; kernels stride over longer lists
// local-sort classes: threads x keys per thread = capacity 2048, 4608, 9216, 17408.  The two big
// classes run 1024 threads so that two workgroups per CU give 32 waves.
#ifndef GS_LS3_THREADS
#define GS_LS3_THREADS 1024
#define GS_LS3_KPT 17
#endif
#ifndef GS_LS2_THREADS
#define GS_LS2_THREADS 512
#define GS_LS2_KPT 18
#endif
#ifndef GS_LS1_THREADS
#define GS_LS1_THREADS 512
#define GS_LS1_KPT 9
#endif
#ifndef GS_LS0_THREADS
#define GS_LS0_THREADS 512
#define GS_LS0_KPT 4
#endif
__host__ __device__ constexpr int msb_class_threads(int c) { return c == 0 ? GS_LS0_THREADS : c == 1 ? GS_LS1_THREADS : c == 2 ? GS_LS2_THREADS : GS_LS3_THREADS; }
__host__ __device__ constexpr int msb_class_kpt(int c) { return c == 0 ? GS_LS0_KPT : c == 1 ? GS_LS1_KPT : c == 2 ? GS_LS2_KPT : GS_LS3_KPT; }
__host__ __device__ constexpr uint32_t msb_class_cap(int c) { return (uint32_t)(msb_class_kpt(c) * msb_class_threads(c)); }
// tiles of a range of x keys; x + MSB_TILE - 1 would wrap for ranges within one tile of 2^32
__host__ __device__ constexpr uint32_t msb_tiles_of(uint32_t x) { return x / (uint32_t)MSB_TILE + (x % (uint32_t)MSB_TILE ? 1u : 0u); }
// pairs keep {key,value} in LDS: their 17408 class takes 139 KiB, i.e. one 1024-thread workgroup per CU (still 7 ms
// cheaper at 2^30 uniform pairs than the third partition level that a largest class of 9216 made necessary)
#ifndef GS_PAIR_CLASSES
#define GS_PAIR_CLASSES 4      // 4: pairs also get the 17408 class (139 KiB of LDS, one 1024-thread workgroup per CU)
#endif
__host__ __device__ constexpr int msb_num_classes(bool has_values) { return has_values ? GS_PAIR_CLASSES : 4; }

struct MsbBucket { uint32_t offset, size, tile_start, tiles; }; // a bucket still to be partitioned (output offset, keys, its tiles)
struct MsbPiece { uint32_t lo, size, tile_start, bucket; };     // multi-GPU: a bucket arrives in one piece per source rank
struct MsbTile { uint32_t lo, valid, bucket, pad; };            // one tile of a level: keys [lo, lo + valid)
struct MsbTask { uint32_t offset, size, sort_bits, pad; };      // a range to finish with a local sort
struct MsbLevel {
    unsigned long long packed;           // hi32: buckets to partition at this level, lo32: their tiles
    uint32_t task_count[MSB_NCLASS];     // local-sort tasks emitted by this level's classification
    uint32_t flagged;                    // != 0: the one-pass local sort left tasks to the general kernel (a plain store:
                                         // thousands of atomics on one word would cost a millisecond)
    uint32_t overflow;                   // level 0's record only: != 0 once ANY device-side append of the sort was clamped by a list
                                         // capacity (a bucket, tile or task record dropped: the result is then wrong).  "Never by
                                         // sizing" (msb_max_*) is an argument; this word is the check.
    unsigned long long unused1;
    unsigned long long keys;             // level 0: the array's size (census)
    unsigned long long unused2;
    uint32_t census_blocks, pad;         // slots of MsbWs::census this level's classification wrote
};
// Census (gs_msb_census: what bench.py prices the MSB sort's algorithmic bytes with).  Every classification block sums what
// its buckets pass on and leaves ONE record; the reader adds them up.  (Three or four global atomics per bucket on the
// level's counters were half of the classification's time once a level had thousands of buckets: Zipf 2^30, 0.46 -> 0.25 ms.)
#define MSB_OVERFLOW(ws_) ((ws_).level[0].overflow = 1u)      /* plain store, idempotent; see MsbLevel::overflow */
struct MsbCensusSlot { unsigned long long next_keys, task_keys, pivot_keys, pivot_buckets; };
constexpr uint32_t MSB_CLASSIFY_GRID = 4096;   // classification blocks per launch at most = census slots per level
// Heavy hitters (skewed inputs: BASELINE configs[3], Zipf).  A bucket in which ONE key value holds at least half of the
// keys -- and the keys below and above it each fit a local sort -- is finished where it stands instead of being
// partitioned byte by byte down to the last level (the reference moves such keys at every remaining pass, only its
// shared-memory atomics are spared: cuda_radix_sort.h:438-441; CUB skips a pass only when ONE digit holds everything,
// agent_radix_sort_downsweep.cuh:694-707):
//   expand    picks the candidate value (majority of three samples of the bucket),
//   upsweep   counts the keys equal to / below it next to the digit histogram,
//   classify  decides (eq >= size / 2, less and greater <= the largest local sort) and emits the two stranger ranges as
//             local-sort tasks,
//   scatter   such a bucket's tiles move ONLY the strangers (to their final ranges in the level's destination buffer)
//             and write the value over the middle range of the result buffer (in place where the bucket already
//             lies in the result buffer: only the slots a stranger held are touched).
// 8 B/key (upsweep read + tile read; + 4 B/key of writes where the bucket lies in the other buffer) once, instead of
// 12 B/key per remaining level.
struct MsbPivot { uint32_t cand, eq, less, flag, cur_less, cur_greater, examine, pad1; };   // examine: the samples made it a candidate

// A level is partitioned like an LSB pass over its tiles (no atomics, deterministic): the tiles of
// all its buckets are numbered consecutively, 8 consecutive tiles form a chunk,
//   spine[d][c]       keys of digit d in chunk c, scanned in place over the whole level,
//   prefix16[g][d]    keys of digit d in the tiles of g's chunk before g,
// so E(g, d) = spine[d][g / 8] + prefix16[g][d] counts digit d in ALL level tiles before g, and a
// bucket's tile g finds its offset inside sub-bucket d as E(g, d) - E(first tile of the bucket, d).
struct MsbWs {
    MsbLevel *level;                     // [5]
    MsbBucket *buckets[2];               // level L uses buckets[L & 1]
    MsbTile *tiles;                      // tile records of the current level
    MsbCensusSlot *census;               // [MSB_LEVELS][MSB_CLASSIFY_GRID]
    uint32_t *cursors;                   // [max_buckets][256]: sub-bucket start - E(first tile, d)
    uint32_t *spine;                     // [256][stride]
    uint16_t *prefix16;                  // [max_tiles][256]
    MsbTask *tasks[MSB_NCLASS];
    MsbPiece *pieces;                    // [extra_pieces] (gs_msb_finish_u32 only)
    MsbPivot *pivots;                    // [max_buckets] heavy-hitter state of the current level's buckets
    // capacities of the lists above.  The sizing makes them sufficient; all the same every device-side append and
    // every reader of a device-side count is bounded by them, so that a wrong count (a bug) gives a wrong result a
    // test can catch instead of an out-of-bounds access (a GPU memory fault can take the whole node down)
    uint32_t max_buckets, max_tasks, max_tiles, stride;
    // geometry of the kernels that work on this workspace: the u32 kernels of this file (tiles of 8192 keys, 32-bit keys,
    // local-sort classes of 2048 / 4608 / 9216 / 17408) or the wide ones further down (64-bit keys and / or values: tiles
    // of 4096 elements, classes of 2048 / 8192); expand, scan and classify serve both through these fields
    uint32_t tile_shift, key_bits, caps[MSB_NCLASS];
};
constexpr int MSB_LEVELS = 10;           // level records: a 64-bit key has 8 byte levels (+ one the classification may write past the last)
__host__ __device__ inline uint32_t ws_tiles_of(const MsbWs &ws, uint32_t x)
{
    return (x >> ws.tile_shift) + ((x & ((1u << ws.tile_shift) - 1u)) ? 1u : 0u);
}

// Tiles of a bucket (round 3): a bucket of MSB_ALIGN_MIN_TILES tiles or more (2 M keys: the level-1 buckets of a large sort) that does not start on a MSB_TILE_ALIGN-element boundary
// gets a SHORTER first tile, so that all its other tiles start on one: a wave's 256-byte loads then cover two cache lines, not
// three.  With every bucket's top byte holding exactly n/256 keys (all tiles aligned like the LSB sort's) the level-1 histogram of
// 2^30 keys took 0.75 ms instead of 0.84 and the scatter 1.71 instead of 1.78 (tools/align_exp.py).  Small buckets keep one ragged
// tile (their last): a second one would cost them more than the alignment gives.
#ifndef GS_MSB_ALIGN_MIN_TILES
#define GS_MSB_ALIGN_MIN_TILES 256
#endif
constexpr uint32_t MSB_TILE_ALIGN = 64, MSB_ALIGN_MIN_TILES = GS_MSB_ALIGN_MIN_TILES;
__host__ __device__ inline uint32_t ws_first_tile(const MsbWs &ws, uint32_t off, uint32_t size, uint32_t min_tiles = MSB_ALIGN_MIN_TILES)
{
    const uint32_t tl = 1u << ws.tile_shift, r = off & (MSB_TILE_ALIGN - 1u);
    if (r == 0u || size < min_tiles * tl) return size < tl ? size : tl;
    return tl - r;
}
// the pieces a bucket arrives in (multi-GPU finish): every tile of that level may be ragged anyway (one workgroup per tile in the
// ragged launch), so pieces of 16 tiles or more are aligned
constexpr uint32_t MSB_PIECE_ALIGN_MIN_TILES = 16;
__host__ __device__ inline uint32_t msb_piece_first_tile(uint32_t off, uint32_t size)
{
    const uint32_t tl = (uint32_t)MSB_TILE, r = off & (MSB_TILE_ALIGN - 1u);
    if (r == 0u || size < MSB_PIECE_ALIGN_MIN_TILES * tl) return size < tl ? size : tl;
    return tl - r;
}
__host__ __device__ inline uint32_t msb_piece_tiles(uint32_t off, uint32_t size) { return 1u + msb_tiles_of(size - msb_piece_first_tile(off, size)); }
__host__ __device__ inline uint32_t ws_tiles_of_at(const MsbWs &ws, uint32_t off, uint32_t size)   // size >= 1
{
    return 1u + ws_tiles_of(ws, size - ws_first_tile(ws, off, size));
}

static inline size_t align256(size_t x) { return (x + 255) & ~(size_t)255; }
// `extra`: pieces of the multi-GPU finish (each may add a ragged tile), 0 otherwise
static inline uint32_t msb_max_buckets(uint64_t n, bool has_values, uint32_t extra = 0, uint32_t cap_max = 0)
{
    return (uint32_t)(n / (cap_max ? cap_max : msb_class_cap(msb_num_classes(has_values) - 1))) + RADIX + 1 + extra;
}
// `extra_tasks`: segments of a segmented sort (each may be one task)
static inline uint32_t msb_max_tasks(uint64_t n, bool has_values, uint32_t extra = 0, uint32_t extra_tasks = 0, uint32_t cap_max = 0)
{
    // a task is either >= MSB_MERGE keys or is followed by something that did not fit: <= 2n/MERGE,
    // plus up to 256 per partitioned bucket
    // (+ two stranger ranges per heavy-hitter bucket)
    return (uint32_t)(2 * n / MSB_MERGE) + 3 * msb_max_buckets(n, has_values, extra, cap_max) + 2 * RADIX + extra_tasks;
}
// tiles of a level: n / TILE full ones + one ragged tile per bucket (+ a second one for the buckets of MSB_ALIGN_MIN_TILES tiles or
// more, see ws_first_tile), padded to whole chunks + one spare chunk
static inline uint32_t msb_max_tiles(uint64_t n, bool has_values, uint32_t extra = 0, uint32_t cap_max = 0, uint32_t tile = MSB_TILE)
{
    const uint64_t t = n / tile + msb_max_buckets(n, has_values, extra, cap_max) + n / ((uint64_t)MSB_ALIGN_MIN_TILES * tile) + 2;
    return (uint32_t)((t / MSB_WAVES + 2) * MSB_WAVES);
}
// `wide_cap` != 0: the geometry of the wide kernels (tiles of 4096 elements, largest local sort of `wide_cap` elements)
static size_t msb_ws_bytes(uint64_t n, bool has_values, uint32_t extra = 0, uint32_t extra_tasks = 0, uint32_t wide_cap = 0)
{
    const uint32_t tile = wide_cap ? 4096u : (uint32_t)MSB_TILE;
    const size_t mb = msb_max_buckets(n, has_values, extra, wide_cap), mt = msb_max_tasks(n, has_values, extra, extra_tasks, wide_cap),
                 ml = msb_max_tiles(n, has_values, extra, wide_cap, tile);
    return align256(MSB_LEVELS * sizeof(MsbLevel)) + 2 * align256(mb * sizeof(MsbBucket)) + align256(ml * sizeof(MsbTile)) +
           align256(mb * RADIX * sizeof(uint32_t)) + align256((size_t)RADIX * (ml / MSB_WAVES) * sizeof(uint32_t)) +
           align256(ml * RADIX * sizeof(uint16_t)) + MSB_NCLASS * align256(mt * sizeof(MsbTask)) +
           align256((size_t)extra * sizeof(MsbPiece)) + align256(mb * sizeof(MsbPivot)) +
           align256((size_t)MSB_LEVELS * MSB_CLASSIFY_GRID * sizeof(MsbCensusSlot));
}
static MsbWs msb_carve(void *temp, uint64_t n, bool has_values, uint32_t extra = 0, uint32_t extra_tasks = 0, uint32_t wide_cap = 0,
                       uint32_t key_bits = 32)
{
    MsbWs ws;
    const uint32_t tile = wide_cap ? 4096u : (uint32_t)MSB_TILE;
    ws.max_buckets = msb_max_buckets(n, has_values, extra, wide_cap);
    ws.max_tasks = msb_max_tasks(n, has_values, extra, extra_tasks, wide_cap);
    // test hook (tests/test_msb_gpu.py::test_list_overflow_is_reported): a smaller task-list bound than the sizing gives,
    // so that the overflow path can be exercised at all.  Read on every call; never set in production.
    if (const char *e = getenv("GS_MSB_TEST_MAX_TASKS")) {
        const uint32_t lim = (uint32_t)strtoul(e, nullptr, 10);
        if (lim >= 1u && lim < ws.max_tasks) ws.max_tasks = lim;
    }
    ws.max_tiles = msb_max_tiles(n, has_values, extra, wide_cap, tile);
    ws.stride = ws.max_tiles / MSB_WAVES;
    ws.tile_shift = wide_cap ? 12u : 13u;
    ws.key_bits = key_bits;
    for (int q = 0; q < MSB_NCLASS; ++q) ws.caps[q] = wide_cap ? (q == 0 ? 2048u : wide_cap) : msb_class_cap(q);
    char *c = (char *)temp;
    ws.level = (MsbLevel *)c; c += align256(MSB_LEVELS * sizeof(MsbLevel));
    for (int i = 0; i < 2; ++i) { ws.buckets[i] = (MsbBucket *)c; c += align256((size_t)ws.max_buckets * sizeof(MsbBucket)); }
    ws.tiles = (MsbTile *)c; c += align256((size_t)ws.max_tiles * sizeof(MsbTile));
    ws.cursors = (uint32_t *)c; c += align256((size_t)ws.max_buckets * RADIX * sizeof(uint32_t));
    ws.spine = (uint32_t *)c; c += align256((size_t)RADIX * ws.stride * sizeof(uint32_t));
    ws.prefix16 = (uint16_t *)c; c += align256((size_t)ws.max_tiles * RADIX * sizeof(uint16_t));
    for (int i = 0; i < MSB_NCLASS; ++i) { ws.tasks[i] = (MsbTask *)c; c += align256((size_t)ws.max_tasks * sizeof(MsbTask)); }
    ws.pieces = (MsbPiece *)c; c += align256((size_t)extra * sizeof(MsbPiece));
    ws.pivots = (MsbPivot *)c; c += align256((size_t)ws.max_buckets * sizeof(MsbPivot));
    ws.census = (MsbCensusSlot *)c;
    return ws;
}

// ------------------------------------------------------------------- init --
__global__ void msb_init_kernel(MsbWs ws, uint32_t n)
{
    const int t = threadIdx.x;
    if (t < MSB_LEVELS) {
        MsbLevel z{};
        if (t == 0) { z.packed = (1ull << 32) | ws_tiles_of(ws, n); z.keys = n; }
        ws.level[t] = z;
    }
    if (t == 0) ws.buckets[0][0] = MsbBucket{0u, n, 0u, ws_tiles_of(ws, n)};
}

// direct path for arrays that fit one workgroup: a single task on all 32 bits
__global__ void msb_single_task_kernel(MsbWs ws, uint32_t n, int cls)
{
    if (threadIdx.x == 0) {
        ws.tasks[cls][0] = MsbTask{0u, n, 32u, 0u};
        ws.level[0].task_count[cls] = 1;
    }
}

// tile records of level L from its bucket list (one block per bucket and step)
// `pivot_src` != nullptr: also pick the bucket's heavy-hitter candidate from its keys (see MsbPivot)
__global__ __launch_bounds__(256) void msb_expand_kernel(MsbWs ws, int L, const uint32_t *__restrict__ pivot_src = nullptr)
{
    uint32_t nb = (uint32_t)(ws.level[L].packed >> 32);
    if (nb > ws.max_buckets) { nb = ws.max_buckets; MSB_OVERFLOW(ws); }
    for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const MsbBucket B = ws.buckets[L & 1][b];
        if (pivot_src && threadIdx.x < WAVE) {
            // candidate = majority of three samples; the bucket is only examined further (the upsweep's per-key
            // compares cost it a third of its time) if at least 3 of 16 evenly spaced samples agree with it -- a value
            // holding half of the bucket fails that with probability 0.002, one holding 5 % passes with 0.04
            const uint32_t q = B.size / 4u;
            const uint32_t a0 = pivot_src[B.offset + q], a1 = pivot_src[B.offset + 2u * q], a2 = pivot_src[B.offset + 3u * q];
            const uint32_t cand = (a1 == a2) ? a1 : a0;
            const uint32_t lane = threadIdx.x;
            const uint32_t smp = pivot_src[B.offset + (uint32_t)(((unsigned long long)B.size * (2u * (lane & 15u) + 1u)) >> 5)];
            const uint32_t hits = (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(lane < 16u && smp == cand));
            if (lane == 0) ws.pivots[b] = MsbPivot{cand, 0u, 0u, 0u, 0u, 0u, hits >= 3u ? 1u : 0u, 0u};
        }
        const uint32_t tl = 1u << ws.tile_shift, v0 = ws_first_tile(ws, B.offset, B.size);
        for (uint32_t t = threadIdx.x; t < B.tiles; t += blockDim.x) {
            const uint32_t before = t ? v0 + (t - 1u) * tl : 0u, left = B.size - before;       // (tile 0 may be the short one)
            const uint32_t len = t ? (left < tl ? left : tl) : v0;
            if (B.tile_start + t < ws.max_tiles)
                ws.tiles[B.tile_start + t] = MsbTile{B.offset + before, len, b, 0u};
            else MSB_OVERFLOW(ws);
        }
    }
}

// multi-GPU finish: the level's buckets arrive in pieces (one per source rank) that lie anywhere in
// the source buffer; a bucket's pieces own consecutive tile ranges
__global__ __launch_bounds__(256) void msb_expand_pieces_kernel(MsbWs ws, uint32_t npieces)
{
    for (uint32_t q = blockIdx.x; q < npieces; q += gridDim.x) {
        const MsbPiece P = ws.pieces[q];
        const uint32_t tiles = msb_piece_tiles(P.lo, P.size), v0 = msb_piece_first_tile(P.lo, P.size);
        for (uint32_t t = threadIdx.x; t < tiles; t += blockDim.x) {
            const uint32_t before = t ? v0 + (t - 1u) * (uint32_t)MSB_TILE : 0u, left = P.size - before;
            const uint32_t len = t ? (left < (uint32_t)MSB_TILE ? left : (uint32_t)MSB_TILE) : v0;
            if (P.tile_start + t < ws.max_tiles)
                ws.tiles[P.tile_start + t] = MsbTile{P.lo + before, len, P.bucket, 0u};
            else MSB_OVERFLOW(ws);
        }
    }
}

// ---------------------------------------------------------------- upsweep --
// digit of a key: byte `shift/8`, or -- multi-GPU sharding -- the destination rank looked up
// from the key's top bits: remap[twiddle_in(key) >> rshift]
struct DigitSel {
    int shift;                  // used when remap == nullptr
    const uint8_t *remap;       // destination of each top-bits bin, or nullptr
    int rshift;
    int f32_in;                 // key transform of the remap lookup, and of the keys themselves when tw_in is set
    uint32_t xor_in;
    int bits;                   // digit width (8; fewer in the last pass of a segmented sort on a bit sub-range)
    int tw_in;                  // the keys are still raw: transform them on load (first pass of a segmented sort)
};
constexpr int SHARD_MAX_BITS = 12;   // 4096 bins of the key space
// `tab`: the remap table staged in LDS by load_remap (global gathers of a 4 KiB table cost more
// than the rest of the kernel)
__device__ __forceinline__ void load_remap(const DigitSel &ds, uint8_t *tab)
{
    if (ds.remap) {
        const uint32_t nb = 1u << (32 - ds.rshift);
        for (uint32_t i = threadIdx.x; i < nb; i += blockDim.x) tab[i] = ds.remap[i];
        __syncthreads();
    }
}

template <bool REMAP>
__device__ __forceinline__ uint32_t msb_digit(const DigitSel &ds, const uint8_t *tab, uint32_t k)
{
    if (REMAP) return tab[twiddle_in(k, ds.f32_in, ds.xor_in) >> ds.rshift];
    return __builtin_amdgcn_ubfe(k, (uint32_t)ds.shift, (uint32_t)ds.bits);
}

// M3 as an upsweep: one block per chunk of 8 level tiles, one WAVE per tile (wave-private LDS
// counters, 32 dword loads in flight per lane: bucket offsets are not 16-byte aligned).
// PIVOT: also count the keys equal to / below the bucket's heavy-hitter candidate (see MsbPivot).
template <bool REMAP, bool PIVOT = false>
__global__ __launch_bounds__(MSB_THREADS) void msb_upsweep_kernel(MsbWs ws, int L, const uint32_t *__restrict__ src, DigitSel ds)
{
    constexpr int SUB = 4;      // histogram copies per wave, padded rows (see lsb_upsweep_kernel)
    __shared__ uint32_t lh[MSB_WAVES][SUB][RADIX + 1];
    __shared__ uint8_t tab[REMAP ? (1 << SHARD_MAX_BITS) : 4];
    if (REMAP) load_remap(ds, tab);
    uint32_t ntiles = (uint32_t)ws.level[L].packed;
    if (ntiles > ws.max_tiles - MSB_WAVES) ntiles = ws.max_tiles - MSB_WAVES;      // never (see MsbWs)
    const uint32_t nchunks = ntiles / MSB_WAVES + 1;          // covers tile index `ntiles` too (see classify)
    const int tid = threadIdx.x, w = wave_id(), lane = lane_id();
    uint32_t *my = lh[w][lane & (SUB - 1)];
#ifndef GS_MSB_UPS_BATCH
#define GS_MSB_UPS_BATCH 32
#endif
    constexpr int BATCH = GS_MSB_UPS_BATCH;
    for (uint32_t c0 = blockIdx.x; c0 < nchunks; c0 += gridDim.x) {
        // the blocks of one XCD take consecutive chunks: neighbours in the spine rows meet in one L2 (see lsb_upsweep_kernel;
        // a block that loops strides by MSB_MAX_GRID, a multiple of 8, so it stays on its residue class)
        const uint32_t c = chunk_of_block(c0, nchunks);
        for (int i = lane; i < SUB * (RADIX + 1); i += WAVE) (&lh[w][0][0])[i] = 0;
        const uint32_t g = c * MSB_WAVES + (uint32_t)w;
        if (g < ntiles) {
            const MsbTile T = ws.tiles[g];
            const uint32_t *p = src + T.lo;
            uint32_t cand = 0, n_eq = 0, n_less = 0;
            bool examine = false;
            if (PIVOT) { cand = ws.pivots[T.bucket].cand; examine = ws.pivots[T.bucket].examine != 0u; }
            auto count = [&](uint32_t k) {
                if (!REMAP && ds.tw_in) k = twiddle_in(k, ds.f32_in, ds.xor_in);
                hist_add(my, msb_digit<REMAP>(ds, tab, k));
                if (PIVOT && examine) {   // wave-uniform tallies: one compare per key, the rest is scalar
                    n_eq += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(k == cand));
                    n_less += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(k < cand));
                }
            };
            if (!REMAP && T.valid == (uint32_t)MSB_TILE && !(PIVOT && examine)) {
                // full tile, plain digit, no heavy-hitter tallies (nearly every tile of a level): the lean path of the LSB upsweep --
                // no per-key test for a digit shared by the whole wave (made on two keys of the batch instead), 4 vector
                // instructions per key instead of 15
                uint32_t wbits = (uint32_t)ds.bits;
                asm volatile("" : "+v"(wbits));
                auto digit_of = [&](uint32_t raw) {
                    return __builtin_amdgcn_ubfe(ds.tw_in ? twiddle_in(raw, ds.f32_in, ds.xor_in) : raw, (uint32_t)ds.shift, wbits);
                };
#ifndef GS_MSB_UPS_LEAN_BATCH
#define GS_MSB_UPS_LEAN_BATCH 32
#endif
                constexpr int LB = GS_MSB_UPS_LEAN_BATCH;      // the lean path's own batch (the other paths keep BATCH)
#pragma unroll 1
                for (int j = 0; j < MSB_TILE / WAVE; j += LB) {
                    uint32_t v[LB];
#pragma unroll
                    for (int u = 0; u < LB; ++u) v[u] = __builtin_nontemporal_load(&p[(j + u) * WAVE + lane]);
                    const uint32_t da = digit_of(v[0]), db = digit_of(v[LB / 2]);
                    const bool hot = __builtin_amdgcn_ballot_w64(da == __builtin_amdgcn_readfirstlane(da)) == ~0ull ||
                                     __builtin_amdgcn_ballot_w64(db == __builtin_amdgcn_readfirstlane(db)) == ~0ull;
                    if (hot) {
#pragma unroll
                        for (int u = 0; u < LB; ++u) hist_add(my, digit_of(v[u]));
                    } else {
#pragma unroll
                        for (int u = 0; u < LB; ++u) atomicAdd(&my[digit_of(v[u])], 1u);
                    }
                }
            } else if (T.valid == (uint32_t)MSB_TILE) {
#pragma unroll
                for (int j = 0; j < MSB_TILE / WAVE; j += BATCH) {
                    uint32_t v[BATCH];
#pragma unroll
                    for (int u = 0; u < BATCH; ++u) v[u] = __builtin_nontemporal_load(&p[(j + u) * WAVE + lane]);
#pragma unroll
                    for (int u = 0; u < BATCH; ++u) count(v[u]);
                }
            } else {
                // ragged tile: the same batches from clamped indices (one guarded load per trip would pay one
                // HBM round trip per 64 keys -- a level can hold tens of thousands of ragged tiles)
                const uint32_t last = T.valid - 1u;
#pragma unroll 1
                for (uint32_t j = 0; j < T.valid; j += BATCH * WAVE) {
                    uint32_t v[BATCH];
#pragma unroll
                    for (int u = 0; u < BATCH; ++u) {
                        const uint32_t idx = j + u * WAVE + lane;
                        v[u] = __builtin_nontemporal_load(&p[idx < last ? idx : last]);
                    }
#pragma unroll
                    for (int u = 0; u < BATCH; ++u)
                        if (j + u * WAVE + lane < T.valid) count(v[u]);
                }
            }
            // the tile's tallies go into its own record (both <= 8192); the classification sums them per bucket.  (Two global
            // atomics per tile on the bucket's counters made the tiles of one hot bucket queue on two addresses: 2^28 keys, half
            // of them one value: 1.14 ms for the level's histograms instead of 0.46.)
            if (PIVOT && examine && lane == 0) ws.tiles[g].pad = n_eq | (n_less << 16);
        }
        __syncthreads();
        if (tid < RADIX) {
            uint32_t run = 0;
#pragma unroll
            for (int j = 0; j < MSB_WAVES; ++j) {
                ws.prefix16[(size_t)(c * MSB_WAVES + j) * RADIX + tid] = (uint16_t)run;
                uint32_t cq = 0;
#pragma unroll
                for (int q = 0; q < SUB; ++q) cq += lh[j][q][tid];
                run += cq;
            }
            ws.spine[(size_t)tid * ws.stride + c] = run;
        }
        __syncthreads();
    }
}

// exclusive scan of every spine row over the level's chunks (one block per digit)
__global__ __launch_bounds__(1024) void msb_scan_kernel(MsbWs ws, int L)
{
    __shared__ uint32_t wsum[16];
    uint32_t nchunks = (uint32_t)ws.level[L].packed / MSB_WAVES + 1;
    if (nchunks > ws.stride) nchunks = ws.stride;              // never (see MsbWs)
    uint32_t *row = ws.spine + (size_t)blockIdx.x * ws.stride;
    const int w = wave_id(), lane = lane_id();
    uint32_t carry = 0;
    for (uint32_t base = 0; base < nchunks; base += 1024) {
        const uint32_t i = base + threadIdx.x;
        const uint32_t c = (i < nchunks) ? row[i] : 0u;
        const uint32_t inc = wave_inclusive_scan(c);
        if (lane == 63) wsum[w] = inc;
        __syncthreads();
        const uint32_t s = (lane < 16) ? wsum[lane] : 0u;
        const uint32_t wincl = wave_inclusive_scan(s);
        const uint32_t wbase_ = __shfl(wincl - s, w, WAVE);
        const uint32_t seg_total = __shfl(wincl, 15, WAVE);
        if (i < nchunks) row[i] = carry + wbase_ + inc - c;
        carry += seg_total;
        __syncthreads();
    }
}

// --------------------------------------------------------------- classify --
// M4: per bucket of level L turn the 256 sub-bucket counts into absolute offsets (the downsweep's
// cursors) and decide what happens to every sub-bucket next:
//   empty                      -> nothing
//   > largest local capacity   -> bucket of level L+1 (partitioned on the next byte)
//   otherwise                  -> local-sort task; adjacent sub-buckets are merged
//                                 greedily while the sum stays < MSB_MERGE
//                                 (a merged task also re-sorts this byte).
// The counts are differences of the scanned upsweep: E(g, d) over the bucket's tile range.
// LAST (byte 0): only the cursors are needed, the scatter finishes everything.
// `counts0`: level 0 of the sort reads the LSB pass's digit totals instead
// (and needs no cursors: the LSB downsweep does that scatter).
// PIVOT: a bucket dominated by one key value is finished by the heavy-hitter path instead (see MsbPivot).
#ifdef GS_EXP_CLS
__device__ unsigned long long gs_cls_stamp[16 * 8];   // experiment: s_memrealtime stamps of block 0, per level
#define CLS_STAMP(k) do { if (blockIdx.x == 0 && threadIdx.x == 0) gs_cls_stamp[L * 16 + (k)] = __builtin_amdgcn_s_memrealtime(); } while (0)
#else
#define CLS_STAMP(k) do { } while (0)
#endif
template <bool LAST, bool PIVOT = false>
__global__ __launch_bounds__(256) void msb_classify_kernel(MsbWs ws, int L, const uint32_t *__restrict__ counts0, int nclass)
{
    __shared__ uint32_t scratch[8];
    __shared__ uint32_t s_cnt[RADIX], s_abs[RADIX], s_task[RADIX], s_nsub[RADIX];
    __shared__ uint8_t s_large[RADIX];
    __shared__ uint32_t s_tot[2], s_ccnt[MSB_NCLASS], s_cbase[MSB_NCLASS], s_ksum[2], s_piv[2];
    __shared__ uint32_t s_p[RADIX], s_ln[RADIX], s_rank2idx[RADIX], s_mark[RADIX], s_jump[2][RADIX];   // the merge (see below)
    __shared__ unsigned long long s_base64;
    uint32_t nb = (uint32_t)(ws.level[L].packed >> 32);
    if (nb > ws.max_buckets) { nb = ws.max_buckets; MSB_OVERFLOW(ws); }   // never (see MsbWs)
    const int d = threadIdx.x;
    MsbCensusSlot acc{0ull, 0ull, 0ull, 0ull};                   // thread 0's running census of this block's buckets
    CLS_STAMP(0);
    const uint32_t cap_max = nclass > 0 ? ws.caps[nclass - 1] : 0xffffffffu;   // nclass 0: cursors only (LAST)
    const uint32_t rb = ws.key_bits - 8u - 8u * (uint32_t)L;     // bits below this level's byte
    for (uint32_t b = blockIdx.x; b < nb; b += gridDim.x) {
        const MsbBucket B = ws.buckets[L & 1][b];
        if (PIVOT) {
            MsbPivot P = ws.pivots[b];
            if (P.examine) {   // uniform for the block: sum the tallies the upsweep left in the bucket's tile records
                if (d < 2) s_piv[d] = 0;
                __syncthreads();
                uint32_t eq = 0, less = 0;
                for (uint32_t t = (uint32_t)d; t < B.tiles; t += RADIX) {
                    const uint32_t w = ws.tiles[B.tile_start + t].pad;
                    eq += w & 0xffffu; less += w >> 16;
                }
                eq = wave_inclusive_scan(eq); less = wave_inclusive_scan(less);
                if (lane_id() == 63) { atomicAdd(&s_piv[0], eq); atomicAdd(&s_piv[1], less); }
                __syncthreads();
                P.eq = s_piv[0]; P.less = s_piv[1];
                __syncthreads();
            }
            const uint32_t greater = B.size - P.eq - P.less;
            if (P.examine && P.eq >= B.size - P.eq && P.less <= cap_max && greater <= cap_max) {   // uniform for the block
                if (d == 0) {
                    ws.pivots[b].eq = P.eq; ws.pivots[b].less = P.less;
                    ws.pivots[b].flag = 1u;
                    acc.pivot_buckets += 1ull; acc.pivot_keys += B.size; acc.task_keys += P.less + greater;
                    // the strangers share the bucket's upper bytes only: their tasks sort this level's byte too
                    const uint32_t offs[2] = {B.offset, B.offset + P.less + P.eq}, sizes[2] = {P.less, greater};
                    for (int q = 0; q < 2; ++q) {
                        if (sizes[q] == 0) continue;
                        int cls = 0;
                        while (ws.caps[cls] < sizes[q]) ++cls;
                        const uint32_t at = atomicAdd(&ws.level[L].task_count[cls], 1u);
                        if (at < ws.max_tasks) ws.tasks[cls][at] = MsbTask{offs[q], sizes[q], rb + 8u, 0u};
                        else MSB_OVERFLOW(ws);
                    }
                }
                continue;
            }
        }
        uint32_t c, e0 = 0;
        if (counts0) {
            c = counts0[d];
        } else {
            const uint32_t g0 = B.tile_start, g1 = g0 + B.tiles;
            const uint32_t *srow = ws.spine + (size_t)d * ws.stride;
            e0 = srow[g0 / MSB_WAVES] + ws.prefix16[(size_t)g0 * RADIX + d];
            c = srow[g1 / MSB_WAVES] + ws.prefix16[(size_t)g1 * RADIX + d] - e0;
        }
        CLS_STAMP(1);
        const uint32_t ex = block_exclusive_scan_256(c, scratch, nullptr);
        CLS_STAMP(2);
        const uint32_t abs = B.offset + ex;
        if (!counts0) ws.cursors[(size_t)b * RADIX + d] = abs - e0;
        if (LAST) continue;
        s_cnt[d] = c; s_abs[d] = abs; s_task[d] = 0; s_nsub[d] = 0; s_large[d] = 0;
        if (d < MSB_NCLASS) s_ccnt[d] = 0;
        if (d < 2) s_ksum[d] = 0;
        // no sub-bucket below the merge threshold: nothing can merge, every thread settles its own digit (the walk below costs
        // 25-40 us per block, scalar code or not: a single wave doing serial work)
        const int any_small = __syncthreads_or(c != 0u && c < (uint32_t)MSB_MERGE);
        CLS_STAMP(3);
        if (!any_small) {
            const bool lg = c > cap_max;
            s_large[d] = lg ? 1 : 0;
            s_task[d] = (c != 0u && !lg) ? c : 0u;
            s_nsub[d] = (c != 0u && !lg) ? 1u : 0u;
        } else {
            // The reference's greedy merge (cuda_radix_sort.h:1084: a run of adjacent sub-buckets grows while its sum stays below
            // MSB_MERGE; a sub-bucket too large for a local sort ends it) without walking the 256 counts one by one (25-40 us for
            // a single wave): with P = prefix sums of the counts that may merge, a run that starts at i ends at the last j with
            // P[j] - (P[i] - c_i) < MSB_MERGE and no large sub-bucket in (i, j] -- a binary search per digit; the run starts are
            // the digits reachable from the first one through "next start after my run", marked by pointer doubling.
            const bool lg = c > cap_max, normal = c != 0u && !lg;
            const uint32_t p_ex = block_exclusive_scan_256(normal ? c : 0u, scratch, nullptr);
            const uint32_t ln_ex = block_exclusive_scan_256((lg ? 0x10000u : 0u) | (normal ? 1u : 0u), scratch, &s_tot[0]);
            const uint32_t p_in = p_ex + (normal ? c : 0u), ln_in = ln_ex + ((lg ? 0x10000u : 0u) | (normal ? 1u : 0u));
            s_p[d] = p_in; s_ln[d] = ln_in; s_mark[d] = 0u;
            if (normal) s_rank2idx[ln_ex & 0xffffu] = (uint32_t)d;      // the r-th digit that may merge
            __syncthreads();
            const uint32_t n_normal = s_tot[0] & 0xffffu;
            uint32_t end = (uint32_t)d, jump = RADIX;
            if (normal) {
                // last j >= d with P[j] - p_ex < MSB_MERGE and as many large digits up to j as up to d (both hold, then fail)
                uint32_t lo = (uint32_t)d, hi = RADIX - 1u;
                const uint32_t larges = ln_in >> 16;
                while (lo < hi) {
                    const uint32_t mid = (lo + hi + 1u) >> 1;
                    const bool ok = s_p[mid] - p_ex < (uint32_t)MSB_MERGE && (s_ln[mid] >> 16) == larges;
                    if (ok) lo = mid; else hi = mid - 1u;
                }
                end = lo;                                               // == d when the digit alone reaches the threshold
                const uint32_t r = s_ln[end] & 0xffffu;                 // digits that may merge up to `end` = rank of the next one
                jump = r < n_normal ? s_rank2idx[r] : (uint32_t)RADIX;
            }
            s_jump[0][d] = jump;
            if (d == 0 && n_normal) s_mark[s_rank2idx[0]] = 1u;
            __syncthreads();
#pragma unroll 1
            for (int k = 0; k < 8; ++k) {                               // reach 2^k starts further per round
                const uint32_t j = s_jump[k & 1][d];
                if (j < (uint32_t)RADIX) {
                    if (s_mark[d]) s_mark[j] = 1u;                      // marks only ever go from 0 to 1: no ordering needed
                    s_jump[(k + 1) & 1][d] = s_jump[k & 1][j];
                } else {
                    s_jump[(k + 1) & 1][d] = (uint32_t)RADIX;
                }
                __syncthreads();
            }
            const bool start = normal && s_mark[d] != 0u;
            s_large[d] = lg ? 1 : 0;
            s_task[d] = start ? s_p[end] - p_ex : 0u;
            s_nsub[d] = start ? (s_ln[end] & 0xffffu) - (ln_ex & 0xffffu) : 0u;
        }
        __syncthreads();
        CLS_STAMP(4);
        // one global atomic per block and list (not per entry): count the block's new
        // buckets / tiles and its tasks per class in LDS, reserve the ranges, then fill
        uint32_t new_bucket = 0xffffffffu;
        const bool is_large = s_large[d] != 0;
        const uint32_t tsize = s_task[d];
        int cls = 0;
        if (tsize) while (ws.caps[cls] < tsize) ++cls;
        const uint32_t tiles = is_large ? ws_tiles_of_at(ws, abs, c) : 0u;
        // exclusive prefixes inside the block (bucket index, tile index) keep tile_start sorted
        const uint32_t bidx = block_exclusive_scan_256(is_large ? 1u : 0u, scratch, &s_tot[0]);
        const uint32_t tidx = block_exclusive_scan_256(tiles, scratch, &s_tot[1]);
        uint32_t task_local = 0;
        if (tsize) task_local = atomicAdd(&s_ccnt[cls], 1u);          // LDS
        {   // census sums (a bucket holds < 2^32 keys)
            const uint32_t kl = wave_reduce_sum(is_large ? c : 0u), kt = wave_reduce_sum(tsize);
            if (lane_id() == 0) { if (kl) atomicAdd(&s_ksum[0], kl); if (kt) atomicAdd(&s_ksum[1], kt); }
        }
        __syncthreads();
        CLS_STAMP(5);
        if (d == 0) {
            unsigned long long old = 0;
            if (s_tot[0]) old = atomicAdd(&ws.level[L + 1].packed, ((unsigned long long)s_tot[0] << 32) | s_tot[1]);
            s_base64 = old;
            // census: keys passed on to the next level / handed to local sorts
            acc.next_keys += s_ksum[0]; acc.task_keys += s_ksum[1];
        }
        if (d < MSB_NCLASS) {
            const uint32_t k = s_ccnt[d];
            s_cbase[d] = k ? atomicAdd(&ws.level[L].task_count[d], k) : 0u;
        }
        __syncthreads();
        CLS_STAMP(6);
        if (is_large) {
            new_bucket = (uint32_t)(s_base64 >> 32) + bidx;
            if (new_bucket < ws.max_buckets)
                ws.buckets[(L + 1) & 1][new_bucket] = MsbBucket{abs, c, (uint32_t)s_base64 + tidx, tiles};
            else MSB_OVERFLOW(ws);
        } else if (tsize) {
            if (s_cbase[cls] + task_local < ws.max_tasks)
                ws.tasks[cls][s_cbase[cls] + task_local] = MsbTask{abs, tsize, rb + (s_nsub[d] > 1 ? 8u : 0u), 0u};
            else MSB_OVERFLOW(ws);
        }
    }
    CLS_STAMP(7);
    if (!LAST && d == 0 && blockIdx.x < MSB_CLASSIFY_GRID) {   // (the last level passes nothing on: its records would be zeros)
        ws.census[(size_t)L * MSB_CLASSIFY_GRID + blockIdx.x] = acc;
        if (blockIdx.x == 0) ws.level[L].census_blocks = gridDim.x < MSB_CLASSIFY_GRID ? gridDim.x : MSB_CLASSIFY_GRID;
    }
}

// ---------------------------------------------------------------- scatter --
// M6: counting-sort scatter of one tile on byte `shift/8` (or on the destination rank).  The tile
// is ranked exactly like an LSB downsweep tile (wave-striped load, wave64 ballot match +
// wave-private LDS counters, wave 0 turns the 8 rows into bases) and finds its slice of every
// sub-bucket from the scanned upsweep (cursor + spine + prefix16), so a level is a segmented
// LSB pass: no global atomics (measured: the reference's scheme of one fetch-add per tile and
// digit on the bucket's cursor row serialises on 8 cache lines and costs 2.9 ms per level at
// 2^30 keys against 2.0 ms), and the partition is deterministic and stable.  A bucket's ragged last
// tile pads with digit 255: the pads come last in tile order, so they land behind every real
// key and are never stored; the upsweep never counted them.
template <bool HAS_VALUES, bool REMAP>
struct ScatterSmem {
    uint32_t whist[MSB_WAVES][RADIX];                     // wave-private counters, then bases
    uint16_t wbase[(HAS_VALUES && !REMAP) ? MSB_WAVES : 1][RADIX];   // all-wave scan only (pairs, 2 blocks/CU)
    uint32_t gbase[RADIX];
    uint32_t stage[MSB_TILE * (HAS_VALUES ? 2 : 1)];
    uint8_t tab[REMAP ? (1 << SHARD_MAX_BITS) : 4];
};

template <bool HAS_VALUES, bool REMAP, bool TWOUT, bool FULL, bool BIG>
__device__ __forceinline__ void msb_scatter_tile(ScatterSmem<HAS_VALUES, REMAP> &sm, const DigitSel &ds,
                                                 const uint32_t *__restrict__ cursor, const uint32_t *__restrict__ spine_c,
                                                 uint32_t stride, const uint16_t *__restrict__ pfx,
                                                 const uint32_t *__restrict__ src_k, uint32_t *__restrict__ dst_k,
                                                 const uint32_t *__restrict__ src_v, uint32_t *__restrict__ dst_v, uint32_t lo,
                                                 uint32_t valid, int f32_out, uint32_t xor_out)
{
    constexpr bool small = !BIG;                     // n <= 2^30: 32-bit byte offsets into the output
    constexpr bool ALLWAVE = HAS_VALUES && !REMAP;   // with the 4 KiB remap table the extra rows would cost a block per CU
    constexpr int ESH = HAS_VALUES ? 3 : 2;           // log2 of a staged element's size
    const int lane = lane_id(), w = wave_id();
    uint32_t *my = sm.whist[w];
    const uint16_t *mybase = sm.wbase[ALLWAVE ? w : 0];
    const uint32_t wbase = (uint32_t)w * (WAVE * MSB_KPT) + lane;
    uint32_t key[MSB_KPT], val[HAS_VALUES ? MSB_KPT : 1], pos[MSB_KPT];
    if (!HAS_VALUES) __builtin_amdgcn_s_setprio(3);      // loads and stores before ranking (see lsb_downsweep_kernel)
    const uint32_t *pk = src_k + lo, *pv = HAS_VALUES ? src_v + lo : nullptr;   // scalar bases: loads take lane offset + immediate
#pragma unroll
    for (int i = 0; i < MSB_KPT; ++i) {
        const uint32_t idx = wbase + i * WAVE;
        const uint32_t off = (FULL ? idx : (idx < valid ? idx : valid - 1u)) * 4u;   // clamped, never predicated (pads are masked by `digit`)
        key[i] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(pk) + off);
    }
    if (HAS_VALUES) {
#pragma unroll
        for (int i = 0; i < MSB_KPT; ++i) {
            const uint32_t idx = wbase + i * WAVE;
            const uint32_t off = (FULL ? idx : (idx < valid ? idx : valid - 1u)) * 4u;
            val[i] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(pv) + off);
        }
    }
    if (!REMAP && ds.tw_in) {   // raw keys (segmented sort, first pass): they travel transformed from here on
#pragma unroll
        for (int i = 0; i < MSB_KPT; ++i) key[i] = twiddle_in(key[i], ds.f32_in, ds.xor_in);
    }
    if (!HAS_VALUES) __builtin_amdgcn_s_setprio(0);
    // global start of this tile's slice of every sub-bucket (wave 0): cursor + E(tile, d)
    uint32_t tbase[4] = {0, 0, 0, 0};
    if (w == 0) {
        const uint4 cur = reinterpret_cast<const uint4 *>(cursor)[lane];
        const uint32_t *sp = spine_c + (size_t)(4 * lane) * stride;
        const uint2 pf = reinterpret_cast<const uint2 *>(pfx)[lane];
        tbase[0] = cur.x + sp[0] + (pf.x & 0xffffu);
        tbase[1] = cur.y + sp[stride] + (pf.x >> 16);
        tbase[2] = cur.z + sp[2 * (size_t)stride] + (pf.y & 0xffffu);
        tbase[3] = cur.w + sp[3 * (size_t)stride] + (pf.y >> 16);
    }
    auto digit = [&](uint32_t k, uint32_t idx) {
        const uint32_t d = msb_digit<REMAP>(ds, sm.tab, k);
        return (FULL || idx < valid) ? d : 255u;
    };
#pragma unroll
    for (int i = lane; i < RADIX; i += WAVE) my[i] = 0;
    {
        uint32_t d_prev = 0, plo = 0, phi = 0;
#pragma unroll
        for (int i = 0; i <= MSB_KPT; ++i) {
            uint32_t d_cur = 0, clo = 0, chi = 0;
            if (i < MSB_KPT) {
                d_cur = digit(key[i], wbase + i * WAVE);
                match_digit(d_cur, clo, chi);
            }
            if (i > 0) {
                const uint32_t lower = count_lower(plo, phi);
                pos[i - 1] = my[d_prev] + lower;
                if (lower == 0)
                    __hip_atomic_fetch_add(&my[d_prev], (uint32_t)(__popc(plo) + __popc(phi)), __ATOMIC_RELAXED,
                                           __HIP_MEMORY_SCOPE_WAVEFRONT);
            }
            d_prev = d_cur; plo = clo; phi = chi;
        }
    }
#pragma unroll
    for (int i = 0; i < MSB_KPT; ++i) asm volatile("" : "+v"(pos[i]), "+v"(key[i]));
    __syncthreads();

    // wave 0, lane l: digits 4l..4l+3.  The tile's offsets were loaded long ago (tbase, below).
    auto reserve = [&](const uint32_t (&ex)[4], const uint32_t (&run)[4]) {
        (void)run;
        uint32_t g[4];
#pragma unroll
        for (int q = 0; q < 4; ++q) g[q] = small ? ((tbase[q] - ex[q]) << 2) : (tbase[q] - ex[q]);
        reinterpret_cast<uint4 *>(sm.gbase)[lane] = make_uint4(g[0], g[1], g[2], g[3]);
    };
    if constexpr (ALLWAVE) {
        uint32_t run[4] = {0, 0, 0, 0}, below[4] = {0, 0, 0, 0};
#pragma unroll
        for (int j = 0; j < MSB_WAVES; ++j) {
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
        if (w == 0) reserve(ex, run);
        // no barrier: a wave only reads its own `wbase` row, and `whist` is not written again
    } else {
        if (w == 0) {
            uint32_t run[4] = {0, 0, 0, 0};
#pragma unroll
            for (int j = 0; j < MSB_WAVES; ++j) {
                const uint4 x = reinterpret_cast<const uint4 *>(sm.whist[j])[lane];
                run[0] += x.x; run[1] += x.y; run[2] += x.z; run[3] += x.w;
            }
            const uint32_t lane_sum = run[0] + run[1] + run[2] + run[3];
            uint32_t ex[4];
            ex[0] = wave_inclusive_scan(lane_sum) - lane_sum;
            ex[1] = ex[0] + run[0];
            ex[2] = ex[1] + run[1];
            ex[3] = ex[2] + run[2];
            reserve(ex, run);
            asm volatile("" ::: "memory");
            uint4 e4 = make_uint4(ex[0] << ESH, ex[1] << ESH, ex[2] << ESH, ex[3] << ESH);   // byte offsets into `stage`
#pragma unroll
            for (int j = 0; j < MSB_WAVES; ++j) {
                const uint4 x = reinterpret_cast<const uint4 *>(sm.whist[j])[lane];
                reinterpret_cast<uint4 *>(sm.whist[j])[lane] = e4;
                e4.x += x.x << ESH; e4.y += x.y << ESH; e4.z += x.z << ESH; e4.w += x.w << ESH;
            }
        }
        __syncthreads();
    }
    {
        uint32_t wb[MSB_KPT];
#pragma unroll
        for (int i = 0; i < MSB_KPT; ++i) {
            const uint32_t d = digit(key[i], wbase + i * WAVE);
            wb[i] = ALLWAVE ? (uint32_t)mybase[d] : my[d];
        }
#pragma unroll
        for (int i = 0; i < MSB_KPT; ++i) {
            const uint32_t at = ALLWAVE ? ((pos[i] + wb[i]) << ESH) : ((pos[i] << ESH) + wb[i]);   // bytes
            if (HAS_VALUES) *reinterpret_cast<uint2 *>(reinterpret_cast<char *>(sm.stage) + at) = make_uint2(key[i], val[i]);
            else *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(sm.stage) + at) = key[i];
        }
    }
    __syncthreads();
#ifndef GS_MSB_SCATTER_SLEEP
#define GS_MSB_SCATTER_SLEEP 0
#endif
    if (!HAS_VALUES && GS_MSB_SCATTER_SLEEP && (w & 1)) __builtin_amdgcn_s_sleep(GS_MSB_SCATTER_SLEEP);   // see lsb_downsweep_kernel
    if (!HAS_VALUES) __builtin_amdgcn_s_setprio(3);
#pragma unroll
    for (int i = 0; i < MSB_KPT; ++i) {
        const uint32_t slot = (uint32_t)w * (WAVE * MSB_KPT) + i * WAVE + lane;   // wave-contiguous (see lsb_downsweep_kernel)
        uint32_t k, v = 0;
        if (HAS_VALUES) {
            const uint2 kv = reinterpret_cast<const uint2 *>(sm.stage)[slot];
            k = kv.x; v = kv.y;
        } else {
            k = sm.stage[slot];
        }
        if (FULL || slot < valid) {
            const uint32_t gb = sm.gbase[msb_digit<REMAP>(ds, sm.tab, k)];
            const uint32_t ko = TWOUT ? twiddle_out(k, f32_out, xor_out) : k;
            if (small) {
                const uint32_t off = gb + slot * 4u;
                *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(dst_k) + off) = ko;
                if (HAS_VALUES) *reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(dst_v) + off) = v;
            } else {
                const uint32_t dst = gb + slot;
                dst_k[dst] = ko;
                if (HAS_VALUES) dst_v[dst] = v;
            }
        }
    }
}

// One tile of a bucket the heavy-hitter path finishes (keys only; see MsbPivot): strangers go to their final ranges
// in `other` (the level's destination buffer, where this level's local sorts read), the candidate value goes to the
// tile's share of the middle range of `result`.  The tile writes `result` only inside its own key range, so the
// bucket may lie in `result` itself (`in_place`); then only slots that held a stranger need the value -- unless the
// keys still carry their order-preserving transform, which the stored value must not (`write_all`).
template <bool FULL>
__device__ __forceinline__ void msb_pivot_tile(uint32_t *__restrict__ scratch /* LDS, >= 2 * MSB_WAVES + 2 words */,
                                               MsbPivot *__restrict__ P, const MsbBucket &B, const uint32_t *__restrict__ src_k,
                                               uint32_t *__restrict__ other, uint32_t *__restrict__ result, uint32_t lo,
                                               uint32_t valid, int f32_out, uint32_t xor_out, bool write_all)
{
    const int lane = lane_id(), w = wave_id();
    const uint32_t wbase = (uint32_t)w * (WAVE * MSB_KPT) + lane;
    uint32_t key[MSB_KPT];
    const uint32_t *pk = src_k + lo;
#pragma unroll
    for (int i = 0; i < MSB_KPT; ++i) {
        const uint32_t idx = wbase + i * WAVE;
        key[i] = pk[FULL ? idx : (idx < valid ? idx : valid - 1u)];
    }
    const uint32_t cand = P->cand, mid_lo = B.offset + P->less, mid_hi = mid_lo + P->eq;
    const uint32_t cand_out = twiddle_out(cand, f32_out, xor_out);
    // strangers of the tile: one reservation per side and tile (a level-2 bucket of a Zipf input holds ~8000 of them: one
    // global atomic per key serialises on two words per bucket -- measured 4.0 ms for the level against 0.83 ms; one
    // returning atomic per wave instead of the two barriers: 1.3 ms), ranks by ballot
    uint32_t wl = 0, wg = 0;   // wave-uniform counts
#pragma unroll
    for (int i = 0; i < MSB_KPT; ++i) {
        const bool in = FULL || wbase + i * WAVE < valid;
        wl += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(in && key[i] < cand));
        wg += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(in && key[i] > cand));
    }
    if (lane == 0) { scratch[w] = wl; scratch[MSB_WAVES + w] = wg; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tl = 0, tg = 0;
#pragma unroll
        for (int j = 0; j < MSB_WAVES; ++j) { const uint32_t a = scratch[j], b2 = scratch[MSB_WAVES + j]; scratch[j] = tl; scratch[MSB_WAVES + j] = tg; tl += a; tg += b2; }
        scratch[2 * MSB_WAVES] = tl ? atomicAdd(&P->cur_less, tl) : 0u;
        scratch[2 * MSB_WAVES + 1] = tg ? atomicAdd(&P->cur_greater, tg) : 0u;
    }
    __syncthreads();
    uint32_t nl = B.offset + scratch[2 * MSB_WAVES] + scratch[w];                           // next free slot, from the front
    uint32_t ng = B.offset + B.size - 1u - scratch[2 * MSB_WAVES + 1] - scratch[MSB_WAVES + w];   // from the back
#pragma unroll
    for (int i = 0; i < MSB_KPT; ++i) {
        const uint32_t idx = wbase + i * WAVE;
        const bool in = FULL || idx < valid;
        const uint32_t k = key[i], at = lo + idx;
        const unsigned long long ml = __builtin_amdgcn_ballot_w64(in && k < cand), mg = __builtin_amdgcn_ballot_w64(in && k > cand);
        if (in) {
            if (k < cand) other[nl + count_lower_mask(ml)] = k;
            else if (k > cand) other[ng - count_lower_mask(mg)] = k;
            if (at >= mid_lo && at < mid_hi && (write_all || k != cand)) result[at] = cand_out;
        }
        nl += (uint32_t)__popcll(ml);
        ng -= (uint32_t)__popcll(mg);
    }
}

// The same for (key, value) pairs (round 3): the values of the candidate's keys have to move too, so nothing is written in place.
// Strangers go to their final ranges in `other` with their values; a pair of the candidate goes to a slot of the bucket's middle
// range that the tile reserves (a third reservation per tile, P->cur_eq; any order will do: the sort is unstable) -- in `result`
// when that is not the buffer being read (level 1: result == other), else in `other`, from where msb_pivot_copyback_kernel moves
// the middle range to `result` once the level's scatter has finished (level 2).
template <bool FULL>
__device__ __forceinline__ void msb_pivot_tile_pairs(uint32_t *__restrict__ scratch /* LDS, >= 3 * MSB_WAVES + 3 words */,
                                                     MsbPivot *__restrict__ P, const MsbBucket &B, const uint32_t *__restrict__ src_k,
                                                     const uint32_t *__restrict__ src_v, uint32_t *__restrict__ other_k,
                                                     uint32_t *__restrict__ other_v, uint32_t *__restrict__ mid_k,
                                                     uint32_t *__restrict__ mid_v, uint32_t lo, uint32_t valid, int f32_out, uint32_t xor_out)
{
    const int lane = lane_id(), w = wave_id();
    const uint32_t wbase = (uint32_t)w * (WAVE * MSB_KPT) + lane;
    uint32_t key[MSB_KPT], val[MSB_KPT];
    const uint32_t *pk = src_k + lo, *pv = src_v + lo;
#pragma unroll
    for (int i = 0; i < MSB_KPT; ++i) {
        const uint32_t idx = wbase + i * WAVE, at = FULL ? idx : (idx < valid ? idx : valid - 1u);
        key[i] = pk[at]; val[i] = pv[at];
    }
    const uint32_t cand = P->cand, mid_lo = B.offset + P->less;
    const uint32_t cand_out = twiddle_out(cand, f32_out, xor_out);
    uint32_t wl = 0, wg = 0, we = 0;   // wave-uniform counts
#pragma unroll
    for (int i = 0; i < MSB_KPT; ++i) {
        const bool in = FULL || wbase + i * WAVE < valid;
        wl += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(in && key[i] < cand));
        wg += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(in && key[i] > cand));
        we += (uint32_t)__popcll(__builtin_amdgcn_ballot_w64(in && key[i] == cand));
    }
    if (lane == 0) { scratch[w] = wl; scratch[MSB_WAVES + w] = wg; scratch[2 * MSB_WAVES + w] = we; }
    __syncthreads();
    if (threadIdx.x == 0) {
        uint32_t tl = 0, tg = 0, te = 0;
#pragma unroll
        for (int j = 0; j < MSB_WAVES; ++j) {
            const uint32_t a = scratch[j], b2 = scratch[MSB_WAVES + j], c = scratch[2 * MSB_WAVES + j];
            scratch[j] = tl; scratch[MSB_WAVES + j] = tg; scratch[2 * MSB_WAVES + j] = te;
            tl += a; tg += b2; te += c;
        }
        scratch[3 * MSB_WAVES] = tl ? atomicAdd(&P->cur_less, tl) : 0u;
        scratch[3 * MSB_WAVES + 1] = tg ? atomicAdd(&P->cur_greater, tg) : 0u;
        scratch[3 * MSB_WAVES + 2] = te ? atomicAdd(&P->pad1, te) : 0u;       // pad1: the cursor of the middle range
    }
    __syncthreads();
    uint32_t nl = B.offset + scratch[3 * MSB_WAVES] + scratch[w];                               // next free slot, from the front
    uint32_t ng = B.offset + B.size - 1u - scratch[3 * MSB_WAVES + 1] - scratch[MSB_WAVES + w]; // from the back
    uint32_t ne = mid_lo + scratch[3 * MSB_WAVES + 2] + scratch[2 * MSB_WAVES + w];             // in the middle
#pragma unroll
    for (int i = 0; i < MSB_KPT; ++i) {
        const bool in = FULL || wbase + i * WAVE < valid;
        const uint32_t k = key[i];
        const unsigned long long ml = __builtin_amdgcn_ballot_w64(in && k < cand), mg = __builtin_amdgcn_ballot_w64(in && k > cand),
                                 me = __builtin_amdgcn_ballot_w64(in && k == cand);
        if (in) {
            if (k < cand) { const uint32_t at = nl + count_lower_mask(ml); other_k[at] = k; other_v[at] = val[i]; }
            else if (k > cand) { const uint32_t at = ng - count_lower_mask(mg); other_k[at] = k; other_v[at] = val[i]; }
            else { const uint32_t at = ne + count_lower_mask(me); if (mid_k) mid_k[at] = cand_out; mid_v[at] = val[i]; }
        }
        nl += (uint32_t)__popcll(ml);
        ng -= (uint32_t)__popcll(mg);
        ne += (uint32_t)__popcll(me);
    }
}

// Level 2, pairs: the middle ranges of the heavy-hitter buckets lie in the level's destination buffer (values only); the result
// buffer gets the key value and the values, every tile of such a bucket moving its own slice of the middle range.
__global__ __launch_bounds__(MSB_THREADS) void msb_pivot_copyback_kernel(MsbWs ws, int L, const uint32_t *__restrict__ from_v,
                                                                         uint32_t *__restrict__ result_k, uint32_t *__restrict__ result_v,
                                                                         int f32_out, uint32_t xor_out)
{
    const unsigned long long packed = ws.level[L].packed;
    if (blockIdx.x >= (uint32_t)packed) return;
    const MsbTile T = ws.tiles[blockIdx.x];
    const MsbPivot P = ws.pivots[T.bucket];
    if (!P.flag) return;
    const MsbBucket B = ws.buckets[L & 1][T.bucket];
    const uint32_t mid_lo = B.offset + P.less, mid_hi = mid_lo + P.eq, cand_out = twiddle_out(P.cand, f32_out, xor_out);
    for (uint32_t i = threadIdx.x; i < T.valid; i += MSB_THREADS) {
        const uint32_t at = T.lo + i;
        if (at >= mid_lo && at < mid_hi) { result_k[at] = cand_out; result_v[at] = from_v[at]; }
    }
}

// FULL = true: one block per level tile, dispatched in order; ragged tiles are skipped.
// FULL = false: one block per bucket, for its ragged last tile (if any).  Two kernels keep the
// guarded path's registers out of the hot one (as in the LSB downsweep).
// PIVOT: tiles of buckets the classification flagged for the heavy-hitter path take msb_pivot_tile (`result_k` = the
// sort's result buffer, `pivot_f32_out` / `pivot_xor_out` = the key transform to undo in what it stores).
template <bool HAS_VALUES, bool REMAP, bool TWOUT, bool FULL, bool BIG, bool PIVOT = false>
__global__ __launch_bounds__(MSB_THREADS, HAS_VALUES ? 4 : 6) void msb_scatter_kernel(
    MsbWs ws, int L, const uint32_t *__restrict__ src_k, uint32_t *__restrict__ dst_k, const uint32_t *__restrict__ src_v,
    uint32_t *__restrict__ dst_v, DigitSel ds, int f32_out, uint32_t xor_out, int ragged_anywhere,
    uint32_t *__restrict__ result_k = nullptr, int pivot_f32_out = 0, uint32_t pivot_xor_out = 0u, uint32_t *__restrict__ result_v = nullptr)
{
    __shared__ __attribute__((aligned(16))) ScatterSmem<HAS_VALUES, REMAP> sm;
    const unsigned long long packed = ws.level[L].packed;
    // one tile: nothing to do if it is not this kernel's kind (FULL: full tiles, else: ragged ones) -- uniform for the workgroup
    auto one_tile = [&](uint32_t g) {
        const MsbTile T = ws.tiles[g];
        if (FULL ? T.valid != (uint32_t)MSB_TILE : T.valid == (uint32_t)MSB_TILE) return;
        if (PIVOT && ws.pivots[T.bucket].flag) {
            const MsbBucket B = ws.buckets[L & 1][T.bucket];
            const bool in_place = result_k == src_k;
            if constexpr (HAS_VALUES) {
                // (not in place: the level's destination IS the result buffer, level 1)
                msb_pivot_tile_pairs<FULL>(sm.gbase, ws.pivots + T.bucket, B, src_k, src_v, dst_k, dst_v, in_place ? (uint32_t *)nullptr : result_k,
                                           in_place ? dst_v : result_v, T.lo, T.valid, pivot_f32_out, pivot_xor_out);
            } else {
                msb_pivot_tile<FULL>(sm.gbase, ws.pivots + T.bucket, B, src_k, dst_k, result_k, T.lo, T.valid, pivot_f32_out, pivot_xor_out,
                                     !in_place || pivot_f32_out != 0 || pivot_xor_out != 0u);
            }
            return;
        }
        if (REMAP) load_remap(ds, sm.tab);
        msb_scatter_tile<HAS_VALUES, REMAP, TWOUT, FULL, BIG>(sm, ds, ws.cursors + (size_t)T.bucket * RADIX, ws.spine + g / MSB_WAVES,
                                                              ws.stride, ws.prefix16 + (size_t)g * RADIX, src_k, dst_k, src_v, dst_v,
                                                              T.lo, T.valid, f32_out, xor_out);
    };
    if (FULL) {
        if (blockIdx.x >= (uint32_t)packed) return;
#ifdef GS_MSB_WIDE_GROUP
        one_tile(tile_of_item_wide(blockIdx.x, (uint32_t)packed));
#else
        one_tile(tile_of_item(blockIdx.x, (uint32_t)packed));   // neighbouring tiles on one XCD: their runs meet in one L2
#endif
    } else if (ragged_anywhere == 1) {                    // buckets in pieces: one block per tile, full ones skipped
        if (blockIdx.x >= (uint32_t)packed) return;
        one_tile(blockIdx.x);
    } else {
        // one workgroup per bucket: its last tile and -- where the bucket's tiles were aligned (ws_first_tile) -- its first one may
        // be ragged.  (Two workgroups per bucket, one per candidate: the launch of 12.5 K instead of 6.2 K workgroups, most of
        // which leave at once, took 0.154 instead of 0.088 ms at level 2 of a 2^30-key Zipf sort.)
        // With few buckets (level 1: 256) two workgroups per bucket after all (`ragged_anywhere` == 2): the two tiles of a bucket one
        // after the other made the launch twice as long (17 -> 34 us).
        const bool two = ragged_anywhere == 2;
        const uint32_t b = two ? blockIdx.x >> 1 : blockIdx.x;
        if (b >= (uint32_t)(packed >> 32)) return;
        const MsbBucket B = ws.buckets[L & 1][b];
        const bool short_first = B.tiles >= 2u && ws_first_tile(ws, B.offset, B.size) != (1u << ws.tile_shift);
        if (two) {
            if (blockIdx.x & 1u) { if (short_first) one_tile(B.tile_start); }
            else one_tile(B.tile_start + B.tiles - 1u);
        } else {
            one_tile(B.tile_start + B.tiles - 1u);
            if (short_first) {
                __syncthreads();
                one_tile(B.tile_start);
            }
        }
    }
}

template <bool HAS_VALUES, bool REMAP, bool TWOUT>
static void launch_scatter(const MsbWs &ws, int L, uint32_t tiles_ub, uint32_t buckets_ub, bool big, const uint32_t *sk, uint32_t *dk,
                           const uint32_t *sv, uint32_t *dv, const DigitSel &ds, int f32_out, uint32_t xor_out, hipStream_t s,
                           bool ragged_anywhere = false, uint32_t *pivot_result = nullptr, int pivot_f32_out = 0,
                           uint32_t pivot_xor_out = 0u, uint32_t *pivot_result_v = nullptr)
{
    const dim3 blk(MSB_THREADS);
    const bool two_per_bucket = !ragged_anywhere && buckets_ub <= 1024u;      // see msb_scatter_kernel
    const dim3 rg(ragged_anywhere ? tiles_ub : two_per_bucket ? 2u * buckets_ub : buckets_ub);
    const int ra = ragged_anywhere ? 1 : two_per_bucket ? 2 : 0;
    if constexpr (!REMAP) {
        if (pivot_result) {   // heavy-hitter buckets may exist at this level
            if (big) {
                hipLaunchKernelGGL((msb_scatter_kernel<HAS_VALUES, false, TWOUT, true, true, true>), dim3(tiles_ub), blk, 0, s, ws, L, sk, dk, sv,
                                   dv, ds, f32_out, xor_out, 0, pivot_result, pivot_f32_out, pivot_xor_out, pivot_result_v);
                hipLaunchKernelGGL((msb_scatter_kernel<HAS_VALUES, false, TWOUT, false, true, true>), rg, blk, 0, s, ws, L, sk, dk, sv, dv, ds,
                                   f32_out, xor_out, ra, pivot_result, pivot_f32_out, pivot_xor_out, pivot_result_v);
            } else {
                hipLaunchKernelGGL((msb_scatter_kernel<HAS_VALUES, false, TWOUT, true, false, true>), dim3(tiles_ub), blk, 0, s, ws, L, sk, dk, sv,
                                   dv, ds, f32_out, xor_out, 0, pivot_result, pivot_f32_out, pivot_xor_out, pivot_result_v);
                hipLaunchKernelGGL((msb_scatter_kernel<HAS_VALUES, false, TWOUT, false, false, true>), rg, blk, 0, s, ws, L, sk, dk, sv, dv, ds,
                                   f32_out, xor_out, ra, pivot_result, pivot_f32_out, pivot_xor_out, pivot_result_v);
            }
            if constexpr (HAS_VALUES) {
                // pairs whose bucket lies in the result buffer itself: the middle ranges went to the destination buffer (values);
                // now that every tile has been read, they move to the result
                if (pivot_result == sk)
                    hipLaunchKernelGGL(msb_pivot_copyback_kernel, dim3(tiles_ub), blk, 0, s, ws, L, (const uint32_t *)dv, pivot_result,
                                       pivot_result_v, pivot_f32_out, pivot_xor_out);
            }
            return;
        }
    }
    if (big) {
        hipLaunchKernelGGL((msb_scatter_kernel<HAS_VALUES, REMAP, TWOUT, true, true>), dim3(tiles_ub), blk, 0, s, ws, L, sk, dk, sv, dv,
                           ds, f32_out, xor_out, 0);
        hipLaunchKernelGGL((msb_scatter_kernel<HAS_VALUES, REMAP, TWOUT, false, true>), rg, blk, 0, s, ws, L, sk, dk, sv, dv, ds,
                           f32_out, xor_out, ra);
    } else {
        hipLaunchKernelGGL((msb_scatter_kernel<HAS_VALUES, REMAP, TWOUT, true, false>), dim3(tiles_ub), blk, 0, s, ws, L, sk, dk, sv, dv,
                           ds, f32_out, xor_out, 0);
        hipLaunchKernelGGL((msb_scatter_kernel<HAS_VALUES, REMAP, TWOUT, false, false>), rg, blk, 0, s, ws, L, sk, dk, sv, dv, ds,
                           f32_out, xor_out, ra);
    }
}

// ------------------------------------------------------------- local sort --
// M7: finish one range of <= THREADS*KPT keys inside a workgroup: LSD passes of 8 bits over its
// low `sort_bits` bits, entirely in registers + LDS, then one coalesced store to the result
// buffer.  The first pass ranks with LDS fetch-adds (nothing is ordered yet, so it need not be
// stable -- the reference does the same, cuda_radix_sort.h:1419-1481), the later ones with the
// LSB downsweep's wave64 ballot/popcount match.  The wave-private counters live in the staging
// buffer itself (the keys are in registers while they are needed), so a 17408-key range costs
// 68 KiB of LDS and two 1024-thread workgroups fit a CU.
// digit width of the first (order-free) local pass: about one bin per key, and the histogram fits
// the staging buffer it is overlaid on (2048 -> 11, 4608 -> 12, 9216 -> 13, 17408 -> 14 bits)
__host__ __device__ constexpr int local_b1(int cap) { return cap >= 16384 ? 14 : cap >= 8192 ? 13 : cap >= 4096 ? 12 : 11; }
template <int THREADS, int KPT, bool HAS_VALUES>
struct LocalSmem {
    static constexpr int WAVES = THREADS / WAVE;
    union {
        uint32_t stage[KPT * THREADS * (HAS_VALUES ? 2 : 1)];
        // first pass: one shared histogram of 2^b1 u32 counters, or 2^(b1' + 2) one-byte counters (4 per word)
        uint32_t hist[1 << (local_b1(KPT * THREADS * (HAS_VALUES ? 2 : 1)) > local_b1(KPT * THREADS) ? local_b1(KPT * THREADS * (HAS_VALUES ? 2 : 1)) : local_b1(KPT * THREADS))];
        uint32_t whist[WAVES][RADIX];        // later passes: wave-private counters
    };
    uint32_t wtot[16];
};

// M7, one pass plan per task (B = sort_bits): the first pass takes min(B, 11..14) bits and ranks with LDS
// fetch-adds on ONE shared histogram (nothing is ordered yet, so it need not be stable -- the
// reference does the same, cuda_radix_sort.h:1419-1481); the remaining bits go in stable passes of
// at most 8 bits ranked by the wave64 ballot match with as many ballots as the digit has bits (a
// 16-bit task of the largest class is 14 + 2).  Counters and histogram live in the staging buffer itself (the
// keys are in registers while they are needed), so a 17408-key range costs 68 KiB of LDS and two
// 1024-thread workgroups fit a CU.  The next task's keys are requested before the current one is
// stored (their registers are free by then).
// STABLE (segmented sort): no order-free pass -- every pass is a stable ballot-match pass -- and the task's
// bits start at bit `pad` (= begin_bit) of the key.
// MODE: LS_ALL = the general pass plan for every task of the list; LS_ONEPASS = only the one-pass byte-counter
// sort (see below), tasks it cannot take (too many bits, or a bin that reaches 255 keys) are flagged in their
// `pad` word; LS_FLAGGED = the general plan for the flagged tasks.  The MSB sort launches LS_ONEPASS then
// LS_FLAGGED per class: two lean kernels instead of one that holds both plans (and spills at 64 VGPRs).
enum { LS_ALL = 0, LS_ONEPASS = 1, LS_FLAGGED = 2, LS_DEDUPE = 3, LS_DEDUPE_ALL = 4 };
constexpr uint32_t LS_FLAG = 0x80000000u;
constexpr uint32_t LS_DONE = 0x40000000u;         // finished by the few-distinct-values plan: the general plan skips the task
// LS_DEDUPE / LS_DEDUPE_ALL (round 3): the plan for tasks with FEW DISTINCT values (BASELINE configs[3]: after two partition levels a
// Zipf task holds ~8 K keys of ~256 values in a 16-bit space -- too many per bin for the one-pass byte counters, and two passes in
// the general plan).  A 2^B-bit map of the values present -> the rank of every present value among them (prefix popcount) -> one
// counter per DISTINCT value -> one order-free pass, whatever B <= 16 is.  LS_DEDUPE takes flagged tasks (after LS_ONEPASS),
// LS_DEDUPE_ALL any task (classes with no one-pass plan); both only look at tasks of more bits than the general plan's first pass
// takes, only when the level showed skew, and only when a 64-key sample shows repeats; a task with more than DD_MAX distinct values is
// left untouched for the general plan, a finished one gets LS_DONE.
constexpr uint32_t DD_BITW = 4096;                // map words: {16 present-bits, 16-bit prefix}; 2^16 values at most
constexpr uint32_t DD_MAX = 2048;                 // distinct values at most
// Counters: DD_CNT words shared out as R = 2 ... 16 REPLICAS per distinct value (R = the largest power of two with distinct * R <= DD_CNT),
// lane l adds to replica l % R.  Fetch-adds of one wave instruction on the SAME counter are served one after the other, and a Zipf task's
// heaviest value sits in 5 of 64 lanes on average: with one counter per value the count phase was a quarter of the task (0.36 ms of
// 1.72 ms at 2^30 Zipf keys went away in a timing experiment with artificially spread counters).  The replicas of a value are neighbours
// in the array, so the exclusive scan over the array still yields every (value, replica)'s first output slot.
constexpr uint32_t DD_CNT = 4096;
__host__ __device__ constexpr bool ls_is_dedupe(int mode) { return mode == LS_DEDUPE || mode == LS_DEDUPE_ALL; }
constexpr uint32_t LS_DD = 0x20000000u;           // the task's sample shows repeats: a candidate for the few-distinct-values plan

// A look at every task of a level that showed skew (some sub-bucket outgrew the local sorts and opened a next level), before
// its local sorts: one wave per task reads 64 evenly spaced keys.  A value that shows up three times among them will overflow a
// one-pass byte counter (it holds > 255 of <= 17408 keys): the task gets LS_FLAG, and the one-pass kernel leaves it alone without
// the wasted attempt (Zipf 2^30: 10 000 such tasks cost 0.29 ms of failed attempts).  Four samples with a twin: LS_DD (256 distinct
// values: ~14 of 64 samples have one; 2048: ~2).  Inside the sort kernels the same look was a dependent memory round trip per task
// on the critical path (4000 clocks of a 41 000-clock task, tools/ls_phases.py); here all tasks are looked at at once.
__global__ __launch_bounds__(256) void msb_task_sample_kernel(MsbWs ws, int L, const uint32_t *__restrict__ src_k, int has_values)
{
    if (!(L < 3 && (ws.level[L + 1].packed >> 32) != 0ull)) return;
    const uint32_t lane = lane_id(), wv = (blockIdx.x * blockDim.x + threadIdx.x) >> 6, nwv = (gridDim.x * blockDim.x) >> 6;
    bool flagged = false;
    for (int cls = 0; cls < MSB_NCLASS; ++cls) {
        uint32_t ntasks = ws.level[L].task_count[cls];
        if (ntasks > ws.max_tasks) ntasks = ws.max_tasks;
        const int b1 = local_b1((int)ws.caps[cls]), onepass_bits = local_b1((int)ws.caps[cls] * (has_values ? 2 : 1)) + 2;
        for (uint32_t t = wv; t < ntasks; t += nwv) {
            const MsbTask c = ws.tasks[cls][t];
            if (c.sort_bits <= (uint32_t)b1 || c.sort_bits > 16u || c.size < 64u) continue;
            // 8 clusters of 8 neighbouring keys (8 cache lines per task; 64 single keys were 64 lines -- 0.13 ms of traffic for the
            // 59 000 tasks of a 2^30-key Zipf sort).  Neighbours in a bucket arrived in input order, so for shuffled input they are
            // as good as any 64 keys; on input with runs they over-report repeats, which costs a failed attempt, never a wrong result
            const uint32_t smp = src_k[c.offset + (uint32_t)(((unsigned long long)(c.size - 8u) * (2u * (lane >> 3) + 1u)) >> 4) + (lane & 7u)];
            unsigned long long same = ~0ull;
            for (uint32_t b = 0; b < c.sort_bits; ++b) {
                const bool bit = (smp >> b) & 1u;
                const unsigned long long m = __builtin_amdgcn_ballot_w64(bit);
                same &= bit ? m : ~m;
            }
            const int mult = __popcll(same);
            uint32_t add = 0;
            if (c.sort_bits <= (uint32_t)onepass_bits && __builtin_amdgcn_ballot_w64(mult >= 3) != 0ull) { add |= LS_FLAG; flagged = true; }
            if (__popcll(__builtin_amdgcn_ballot_w64(mult >= 2)) >= 4) add |= LS_DD;
            if (add && lane == 0) ws.tasks[cls][t].pad = c.pad | add;
        }
    }
    if (flagged && lane == 0) ws.level[L].flagged = 1u;
}
#ifdef GS_EXP_LS_PHASES
// experiment builds only (tools/ls_phases.py): shader-clock length of every phase of wave 0, per task; [plan][class][task][16]
constexpr uint32_t LSP_TASKS = 65536;
__device__ uint32_t gs_ls_phase_buf[3 * MSB_NCLASS * LSP_TASKS * 16];
#define LSP_PLAN (MODE == LS_ONEPASS ? 0 : ls_is_dedupe(MODE) ? 2 : 1)
#define LSP(k)                                                                                                         \
    do {                                                                                                               \
        const unsigned long long now_ = __builtin_amdgcn_s_memtime();                                                  \
        if (tid == 0 && ti < LSP_TASKS) gs_ls_phase_buf[((LSP_PLAN * MSB_NCLASS + cls) * LSP_TASKS + ti) * 16 + (k)] += (uint32_t)(now_ - tprev_); \
        tprev_ = now_;                                                                                                 \
    } while (0)
#define LSP_WAIT(what) asm volatile("s_waitcnt " what ::: "memory")
#else
#define LSP(k) do { } while (0)
#define LSP_WAIT(what) do { } while (0)
#endif
// PLAIN: no key transform on the way in or out (u32 ascending): instantiated for the one-pass kernels, where it is 9 of ~30
// vector instructions per key.
template <int THREADS, int KPT, bool HAS_VALUES, bool STABLE = false, int MODE = LS_ALL, bool PLAIN = false>
#ifndef GS_LS_WPE_512x18
#define GS_LS_WPE_512x18 4
#endif
__global__ __launch_bounds__(THREADS, (THREADS == 1024 && !(HAS_VALUES && KPT > 9)) ? 8 : (THREADS == 512 && KPT == 18 && !HAS_VALUES) ? GS_LS_WPE_512x18 : 4) void msb_local_sort_kernel(
    MsbWs ws, int L, int cls, const uint32_t *__restrict__ src_k, uint32_t *__restrict__ dst_k, const uint32_t *__restrict__ src_v,
    uint32_t *__restrict__ dst_v, int f32_in, uint32_t xor_in, int f32_out, uint32_t xor_out)
{
    constexpr int WAVES = THREADS / WAVE;
    constexpr int LOCAL_B1 = local_b1(KPT * THREADS);
    // 2^ONEPASS_BITS byte counters fill (at most) the staging buffer: 13..16 bits
    constexpr int ONEPASS_BITS = local_b1(KPT * THREADS * (HAS_VALUES ? 2 : 1)) + 2;
    static_assert(KPT * THREADS >= (1 << LOCAL_B1) && KPT * THREADS >= WAVES * RADIX, "counters must fit the staging buffer");
    __shared__ __attribute__((aligned(16))) LocalSmem<THREADS, KPT, HAS_VALUES> sm;
    uint32_t ntasks = ws.level[L].task_count[cls];
    if (ntasks > ws.max_tasks) ntasks = ws.max_tasks;
    if ((MODE == LS_FLAGGED || MODE == LS_DEDUPE) && ws.level[L].flagged == 0u) return;     // nothing was left over (the usual case)
    const int tid = threadIdx.x, lane = lane_id(), w = wave_id();
    uint32_t *my = sm.whist[w];
    const uint32_t wbase0 = (uint32_t)w * (WAVE * KPT) + lane;
    // every use site takes a fresh, opaque copy of a cheap index base: otherwise the compiler keeps all
    // KPT derived indices and addresses live across the task loop and spills them (64-VGPR budget)
    auto fresh = [](uint32_t x) { asm volatile("" : "+v"(x)); return x; };
    static_assert(!(STABLE && MODE != LS_ALL), "the stable sort has one plan");
    // next task of this workgroup (stride gridDim.x) that this MODE processes; LS_ONEPASS flags the ones it leaves
    auto advance = [&](uint32_t from, MsbTask &out) {
        uint32_t t = from;
        for (; t < ntasks; t += gridDim.x) {
            const MsbTask c = ws.tasks[cls][t];
            // (the samples were looked at by msb_task_sample_kernel: LS_FLAG = a value heavy enough to overflow the one-pass
            // counters, LS_DD = repeats enough for the few-distinct-values plan)
            const bool mine = MODE == LS_ALL ? (STABLE || (c.pad & LS_DONE) == 0u)
                        : MODE == LS_ONEPASS ? (c.sort_bits > (uint32_t)LOCAL_B1 && c.sort_bits <= (uint32_t)ONEPASS_BITS && (c.pad & LS_FLAG) == 0u)
                        : MODE == LS_FLAGGED ? (c.pad & (LS_FLAG | LS_DONE)) == LS_FLAG
                        : ((c.pad & LS_DD) != 0u && (MODE == LS_DEDUPE_ALL || (c.pad & LS_FLAG) != 0u) && c.sort_bits > (uint32_t)LOCAL_B1 && c.sort_bits <= 16u);
            if (mine) {   // wave-uniform: keep the record in scalar registers (the 64-VGPR budget of the big classes is tight)
                out.offset = __builtin_amdgcn_readfirstlane(c.offset);
                out.size = __builtin_amdgcn_readfirstlane(c.size);
                out.sort_bits = __builtin_amdgcn_readfirstlane(c.sort_bits);
                out.pad = __builtin_amdgcn_readfirstlane(c.pad);
                break;
            }
            if (MODE == LS_ONEPASS && tid == 0 && (c.pad & LS_FLAG) == 0u) { ws.tasks[cls][t].pad = c.pad | LS_FLAG; ws.level[L].flagged = 1u; }
        }
        return t;
    };
    MsbTask T{};
    uint32_t ti = advance(blockIdx.x, T);
    if (ti >= ntasks) return;
    uint32_t key[KPT], val[HAS_VALUES ? KPT : 1], pos[KPT];
    // unconditional loads from clamped indices (a predicated load waits for its data before the
    // next one is issued: KPT round trips instead of one); padding is applied afterwards
    auto request = [&](const MsbTask &t) {
        const uint32_t *pk = src_k + t.offset, *pv = HAS_VALUES ? src_v + t.offset : nullptr;   // scalar bases
        const uint32_t last = t.size - 1u;
        const uint32_t wbase = fresh(wbase0);
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t idx = wbase + i * WAVE;
            const uint32_t off = (idx < last ? idx : last) * 4u;      // 32-bit byte offset: scalar base + vector offset
            key[i] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(pk) + off);
            if (HAS_VALUES) val[i] = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(pv) + off);
        }
    };
    request(T);
    [[maybe_unused]] bool zeroed = false;   // LS_ONEPASS: the staging buffer holds zeros (cleared by the store phase)
    for (;;) {
#ifdef GS_EXP_LS_PHASES
        unsigned long long tprev_ = __builtin_amdgcn_s_memtime();
        const unsigned long long t0_ = tprev_, r0_ = __builtin_amdgcn_s_memrealtime();
#endif
        MsbTask Tn = T;
        const uint32_t tn = advance(ti + gridDim.x, Tn);
        const bool has_next = tn < ntasks;
        LSP(0);                                           // next task record
        LSP_WAIT("vmcnt(0)");
        LSP(1);                                           // wait for this task's keys
        {
            const uint32_t wbase = fresh(wbase0);
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t idx = wbase + i * WAVE;
                const uint32_t k = PLAIN ? key[i] : twiddle_in(key[i], f32_in, xor_in);
                key[i] = (idx < T.size) ? k : 0xffffffffu;   // padding sorts last (keys are in twiddled form)
            }
        }
        const uint32_t B = T.sort_bits;
        // ---- (LS_ONEPASS; tasks with more bits than the general plan's first pass takes) one pass on all B bits
        // when 2^B one-BYTE counters fit the staging buffer (a 16-bit task of the
        // largest class: 64 KiB): fetch-add on the packed counter word returns the old count of the bin = the
        // key's rank inside it; then the word is re-read (final counts -> offset inside the word), the words
        // are summed and scanned, and the word base replaces the counts in place.  A bin that reaches 255 keys
        // (heavy duplicates) abandons the attempt: the keys are untouched and the general plan below runs.
        bool done = false;
        if constexpr (MODE == LS_ONEPASS) {
            const uint32_t nwords = (1u << B) >> 2, maskB = (1u << B) - 1u, wmask = maskB & ~3u;   // wmask: byte offset of a bin's word
            // the counters are zero here: the block's first task zeroes them below, every later one finds them zeroed by
            // the store phase of the task before (each thread clears the slots it has just read: no barrier, no extra loop)
            if (!zeroed) {
                for (uint32_t j = tid; j < nwords; j += THREADS) sm.hist[j] = 0;
                __syncthreads();
            }
            zeroed = false;
            LSP(2);                                       // (first task of the block) zero the counters
            uint32_t overflow = 0;
            {
                const uint32_t wbase = fresh(wbase0);
                // a bin shared by a whole wave (64 equal keys side by side) would queue 64 adds on one counter: looked for on
                // two of the wave's rounds instead of on each (Zipf tasks with such runs mostly fail the 64-key look above)
                const uint32_t ba = key[0] & maskB, bb = key[KPT / 2] & maskB;
                const bool hot = __builtin_amdgcn_ballot_w64(ba == __builtin_amdgcn_readfirstlane(ba)) == __builtin_amdgcn_ballot_w64(true) ||
                                 __builtin_amdgcn_ballot_w64(bb == __builtin_amdgcn_readfirstlane(bb)) == __builtin_amdgcn_ballot_w64(true);
                if (hot) {
#pragma unroll
                for (int i = 0; i < KPT; ++i) {
                    pos[i] = 0;
                    if (wbase + i * WAVE < T.size) {
                        const uint32_t bin = key[i] & maskB, sh = (bin & 3u) * 8u;
                        const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
                        const uint32_t b0 = __builtin_amdgcn_readfirstlane(bin);
                        uint32_t r;
                        if (__builtin_amdgcn_ballot_w64(bin == b0) == act) {      // one add for a wave-uniform bin
                            const uint32_t lower = count_lower_mask(act), cnt = (uint32_t)__popcll(act);
                            uint32_t old = 0;
                            if (lower == 0) old = atomicAdd(&sm.hist[b0 >> 2], cnt << sh);
                            old = (__builtin_amdgcn_readfirstlane(old) >> sh) & 255u;
                            r = old + lower;
                            overflow |= (old + cnt > 255u) ? 1u : 0u;
                        } else {
                            r = (atomicAdd(&sm.hist[bin >> 2], 1u << sh) >> sh) & 255u;
                            overflow |= (r >= 255u) ? 1u : 0u;
                        }
                        pos[i] = r;
                    }
                }
                } else {
                    // no guards: a pad (all ones) adds 0 to the last word; what it reads back is a real counter, so the
                    // overflow test below stays exact.  8 vector instructions per key.
                    uint32_t rmax = 0;
#pragma unroll
                    for (int i = 0; i < KPT; ++i) {
                        const uint32_t sh = (key[i] & 3u) << 3;
                        const uint32_t inc = (wbase + i * WAVE < T.size) ? (1u << sh) : 0u;
                        const uint32_t old = atomicAdd(reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(sm.hist) + (key[i] & wmask)), inc);
                        pos[i] = __builtin_amdgcn_ubfe(old, sh, 8u);
                        rmax = pos[i] > rmax ? pos[i] : rmax;
                    }
                    overflow = (rmax >= 255u) ? 1u : 0u;
                }
            }
            LSP_WAIT("lgkmcnt(0)");
            LSP(3);                                       // counting: one fetch-add per key
            const int ovf_ = __syncthreads_or((int)overflow);
            LSP(4);                                       // barrier
            if (!ovf_) {
                // exclusive scan of the word sums, written over the words (WPT consecutive words per thread).  A scanned word
                // keeps its four counts next to its base -- {base : 16, four 4-bit counts : 16} -- so that ONE more LDS read per
                // key yields both the word's base and the keys of the lower bins of the same word (a read of the raw counts
                // before the scan + a read of the bases after it cost the LDS-bound kernel a fourth random access per key and a
                // barrier).  A bin with more than 15 keys does not fit: the task is left to the general plan.
                constexpr uint32_t WPT = ((1u << ONEPASS_BITS) >> 2) / THREADS;
                static_assert(WPT >= 4 && WPT % 4 == 0, "words per thread");
                static_assert(KPT * THREADS <= 65536, "a word's base must fit 16 bits");
                // the ranks inside the bins are bytes: four per register while the scan needs the registers (64-VGPR budget)
                uint32_t rk[(KPT + 3) / 4];
#pragma unroll
                for (int j = 0; j < (KPT + 3) / 4; ++j) {
                    rk[j] = pos[4 * j];
                    if (4 * j + 1 < KPT) rk[j] |= pos[4 * j + 1] << 8;
                    if (4 * j + 2 < KPT) rk[j] |= pos[4 * j + 2] << 16;
                    if (4 * j + 3 < KPT) rk[j] |= pos[4 * j + 3] << 24;
                }
#pragma unroll
                for (int j = 0; j < (KPT + 3) / 4; ++j) asm volatile("" : "+v"(rk[j]));
#pragma unroll
                for (int i = 0; i < KPT; ++i) asm volatile("" : "+v"(key[i]));   // nothing derived from them stays live across the scan
                const bool act = (uint32_t)tid * WPT < nwords;
                uint32_t ssum = 0;
                if (act) {
#pragma unroll
                    for (uint32_t q = 0; q < WPT; q += 4) {
                        const uint4 c4 = reinterpret_cast<const uint4 *>(sm.hist)[((uint32_t)tid * WPT + q) >> 2];
                        ssum = __builtin_amdgcn_sad_u8(c4.x, 0u, ssum); ssum = __builtin_amdgcn_sad_u8(c4.y, 0u, ssum);
                        ssum = __builtin_amdgcn_sad_u8(c4.z, 0u, ssum); ssum = __builtin_amdgcn_sad_u8(c4.w, 0u, ssum);
                    }
                }
                const uint32_t inc = wave_inclusive_scan(ssum);
                if (lane == 63) sm.wtot[w] = inc;
                LSP(5);                                   // word sums + wave scan
                __syncthreads();
                LSP(6);                                   // barrier
                const uint32_t wsv = (lane < WAVES) ? sm.wtot[lane] : 0u;
                const uint32_t wincl = wave_inclusive_scan(wsv);
                uint32_t run = (uint32_t)__shfl((int)(wincl - wsv), w, WAVE) + inc - ssum;
                uint32_t wide = 0;                         // any count above 15
                auto pack = [&](uint32_t c) {              // bytes -> nibbles, base below them
                    wide |= c & 0xf0f0f0f0u;
                    const uint32_t nib = (c & 0xfu) | ((c >> 4) & 0xf0u) | ((c >> 8) & 0xf00u) | ((c >> 12) & 0xf000u);
                    const uint32_t e = run | (nib << 16);
                    run = __builtin_amdgcn_sad_u8(c, 0u, run);
                    return e;
                };
                if (act) {
#pragma unroll
                    for (uint32_t q = 0; q < WPT; q += 4) {
                        uint4 *p4 = reinterpret_cast<uint4 *>(sm.hist) + (((uint32_t)tid * WPT + q) >> 2);
                        const uint4 c4 = *p4;
                        uint4 e4;
                        e4.x = pack(c4.x); e4.y = pack(c4.y); e4.z = pack(c4.z); e4.w = pack(c4.w);
                        *p4 = e4;
                    }
                }
                LSP_WAIT("lgkmcnt(0)");
                LSP(7);                                   // bases written over the words
                const int wide_ = __syncthreads_or((int)(wide != 0u));
                LSP(8);                                   // barrier
                if (!wide_) {
#pragma unroll
                for (int i = 0; i < KPT; ++i) {
                    const uint32_t wd = *reinterpret_cast<const uint32_t *>(reinterpret_cast<const char *>(sm.hist) + (key[i] & wmask));
                    const uint32_t low = __builtin_amdgcn_ubfe(wd >> 16, 0u, (key[i] & 3u) << 2);          // the counts of the lower bins
                    pos[i] = ((rk[i >> 2] >> (8 * (i & 3))) & 255u) + (wd & 0xffffu) + (low & 0xfu) + ((low >> 4) & 0xfu) + (low >> 8);
                }
#pragma unroll
                for (int i = 0; i < KPT; ++i) asm volatile("" : "+v"(pos[i]));
                LSP(9);                                   // one lookup per key
                __syncthreads();                          // the counters are dead: the buffer takes the keys
                LSP(10);                                  // barrier
                {
                    // no guards: a pad goes to its own load slot, which lies behind the real keys
                    const uint32_t wbase = fresh(wbase0);
#pragma unroll
                    for (int i = 0; i < KPT; ++i) {
                        const uint32_t idx = wbase + i * WAVE;
                        const uint32_t at = idx < T.size ? pos[i] : idx;
                        if (HAS_VALUES) reinterpret_cast<uint2 *>(sm.stage)[at] = make_uint2(key[i], val[i]);
                        else sm.stage[at] = key[i];
                    }
                }
                LSP_WAIT("lgkmcnt(0)");
                LSP(11);                                  // keys into the staging buffer
                __syncthreads();
                LSP(12);                                  // barrier
                done = true;
                }
            }
        }
        if constexpr (ls_is_dedupe(MODE)) {
            static_assert((uint32_t)(KPT * THREADS * (HAS_VALUES ? 2 : 1)) >= DD_BITW + DD_CNT, "the map and the counters are overlaid on the staging buffer");
            static_assert(DD_BITW % THREADS == 0 && DD_CNT % (4 * THREADS) == 0 && KPT * THREADS <= 65536 && DD_MAX * 2 <= DD_CNT, "dedupe geometry");
            const uint32_t maskB = (1u << B) - 1u;
            uint32_t *const bitw = sm.stage, *const cnt = sm.stage + DD_BITW;
            // (keys only: every task after the block's first finds the buffer zeroed by the store phase of the task before)
            if (!zeroed) {
                for (uint32_t j = tid; j < DD_BITW + DD_CNT; j += THREADS) sm.stage[j] = 0;
                __syncthreads();
            }
            zeroed = false;
            LSP(2);                                       // zero the map and the counters + barrier
            {   // mark the values present (a plain read first: after the first rounds almost every bit is set already).  One key
                // after the other: with the reads of a thread batched (9 or 18 at a time) every read sees the early, empty map and
                // the ORs multiply -- 1.39 -> 1.46-1.49 ms for the Zipf local sorts
                const uint32_t wbase = fresh(wbase0);
#pragma unroll
                for (int i = 0; i < KPT; ++i) {
                    if (wbase + i * WAVE < T.size) {
                        const uint32_t v = key[i] & maskB, bit = 1u << (v & 15u);
                        if (!(bitw[v >> 4] & bit)) atomicOr(&bitw[v >> 4], bit);
                    }
                }
            }
            LSP_WAIT("lgkmcnt(0)");
            LSP(3);                                       // mark
            __syncthreads();
            LSP(4);                                       // barrier
            // ranks of the present values: exclusive prefix popcount over the map, kept in the upper half of every word
            constexpr uint32_t WPT = DD_BITW / THREADS;
            static_assert(WPT % 4 == 0, "map words per thread");
            uint32_t ssum = 0;
#pragma unroll
            for (uint32_t q = 0; q < WPT; q += 4) {
                const uint4 c4 = reinterpret_cast<const uint4 *>(bitw)[((uint32_t)tid * WPT + q) >> 2];
                ssum += (uint32_t)(__popc(c4.x) + __popc(c4.y) + __popc(c4.z) + __popc(c4.w));
            }
            uint32_t inc = wave_inclusive_scan(ssum);
            if (lane == 63) sm.wtot[w] = inc;
            __syncthreads();
            uint32_t wsv = (lane < WAVES) ? sm.wtot[lane] : 0u;
            uint32_t wincl = wave_inclusive_scan(wsv);
            uint32_t run = (uint32_t)__shfl((int)(wincl - wsv), w, WAVE) + inc - ssum;
            const uint32_t distinct = (uint32_t)__shfl((int)wincl, WAVES - 1, WAVE);
            LSP(5);                                       // map scan (one barrier inside)
            if (distinct <= DD_MAX) {                     // the same for every thread
#pragma unroll
                for (uint32_t q = 0; q < WPT; q += 4) {
                    uint4 *p4 = reinterpret_cast<uint4 *>(bitw) + (((uint32_t)tid * WPT + q) >> 2);
                    uint4 c4 = *p4;
                    const uint32_t px = run, py = px + (uint32_t)__popc(c4.x), pz = py + (uint32_t)__popc(c4.y), pw = pz + (uint32_t)__popc(c4.z);
                    run = pw + (uint32_t)__popc(c4.w);
                    c4.x |= px << 16; c4.y |= py << 16; c4.z |= pz << 16; c4.w |= pw << 16;
                    *p4 = c4;
                }
                __syncthreads();
                LSP(6);                                   // prefixes written + barrier
                // replicas per value (see DD_CNT): 16 up to 256 distinct values, 8 up to 512, 4 up to 1024, else 2
                const uint32_t rs = distinct <= 256u ? 4u : distinct <= 512u ? 3u : distinct <= 1024u ? 2u : 1u;
                const uint32_t rep = (uint32_t)lane & ((1u << rs) - 1u);
                {   // the old count of (value, replica) is the key's rank among the keys that share both (order-free)
                    const uint32_t wbase = fresh(wbase0);
#pragma unroll
                    for (int i = 0; i < KPT; ++i) {
                        pos[i] = 0;
                        if (wbase + i * WAVE < T.size) {
                            const uint32_t v = key[i] & maskB, wd = bitw[v >> 4];
                            const uint32_t id = (((wd >> 16) + (uint32_t)__popc(wd & ((1u << (v & 15u)) - 1u))) << rs) | rep;
#ifdef GS_EXP_DD_NOCONTEND      /* timing experiment only (wrong results): what do same-address fetch-adds cost? */
                            pos[i] = atomicAdd(&cnt[(id + (uint32_t)lane * 31u) & (DD_CNT - 1u)], 1u) | (id << 16);
#else
                            pos[i] = atomicAdd(&cnt[id], 1u) | (id << 16);
#endif
                        }
                    }
                }
#pragma unroll
                for (int i = 0; i < KPT; ++i) asm volatile("" : "+v"(pos[i]), "+v"(key[i]));
                LSP_WAIT("lgkmcnt(0)");
                LSP(7);                                   // count
                __syncthreads();
                LSP(8);                                   // barrier
                // exclusive scan of the counters in place
                constexpr uint32_t CPT = DD_CNT / THREADS;
                uint32_t csum = 0;
#pragma unroll
                for (uint32_t q = 0; q < CPT; q += 4) {
                    const uint4 c4 = reinterpret_cast<const uint4 *>(cnt)[((uint32_t)tid * CPT + q) >> 2];
                    csum += c4.x + c4.y + c4.z + c4.w;
                }
                inc = wave_inclusive_scan(csum);
                if (lane == 63) sm.wtot[w] = inc;
                __syncthreads();
                wsv = (lane < WAVES) ? sm.wtot[lane] : 0u;
                wincl = wave_inclusive_scan(wsv);
                run = (uint32_t)__shfl((int)(wincl - wsv), w, WAVE) + inc - csum;
#pragma unroll
                for (uint32_t q = 0; q < CPT; q += 4) {
                    uint4 *p4 = reinterpret_cast<uint4 *>(cnt) + (((uint32_t)tid * CPT + q) >> 2);
                    const uint4 c4 = *p4;
                    uint4 e4;
                    e4.x = run; e4.y = e4.x + c4.x; e4.z = e4.y + c4.y; e4.w = e4.z + c4.z;
                    run = e4.w + c4.w;
                    *p4 = e4;
                }
                __syncthreads();
                LSP(9);                                   // counter scan (two barriers)
#pragma unroll
                for (int i = 0; i < KPT; ++i) pos[i] = cnt[pos[i] >> 16] + (pos[i] & 0xffffu);
#pragma unroll
                for (int i = 0; i < KPT; ++i) asm volatile("" : "+v"(pos[i]));
                LSP(10);                                  // base lookup
                __syncthreads();                          // map and counters are dead: the buffer takes the keys
                {
                    const uint32_t wbase = fresh(wbase0);
#pragma unroll
                    for (int i = 0; i < KPT; ++i) {
                        const uint32_t idx = wbase + i * WAVE;
                        const uint32_t at = idx < T.size ? pos[i] : idx;      // a pad goes to its own load slot, behind the real keys
                        if (HAS_VALUES) reinterpret_cast<uint2 *>(sm.stage)[at] = make_uint2(key[i], val[i]);
                        else sm.stage[at] = key[i];
                    }
                }
                LSP_WAIT("lgkmcnt(0)");
                LSP(11);                                  // barrier + keys into the staging buffer
                __syncthreads();
                LSP(12);                                  // barrier
                done = true;
            } else {
                __syncthreads();                          // everyone has read wtot before the next task reuses it
            }
        }
        if constexpr (MODE == LS_ONEPASS) {
            if (!done && tid == 0) { ws.tasks[cls][ti].pad = T.pad | LS_FLAG; ws.level[L].flagged = 1u; }   // a bin overflowed
        } else if constexpr (ls_is_dedupe(MODE)) {
            if (done && tid == 0) ws.tasks[cls][ti].pad = T.pad | LS_DONE;
        } else {
        done = true;
        const uint32_t b1 = STABLE ? 0u : (B < (uint32_t)LOCAL_B1 ? B : (uint32_t)LOCAL_B1);
        bool in_regs = STABLE;                            // the keys of the first stable pass are still in registers
        if (!STABLE) {   // ---- first pass: shared histogram of 2^b1 bins, order-free ranks
            const uint32_t nbins = 1u << b1, mask1 = nbins - 1u;
            for (uint32_t j = tid; j < nbins; j += THREADS) sm.hist[j] = 0;
            __syncthreads();
            LSP(2);                                       // zero the histogram + barrier
            // the pads take no part in this pass (an order-free rank could put one in front of a real key
            // with all-ones digits): real keys fill [0, size), the pads stay implicit behind them
            {
                const uint32_t wbase = fresh(wbase0);
#pragma unroll
                for (int i = 0; i < KPT; ++i) {
                    pos[i] = 0;
                    if (wbase + i * WAVE < T.size) {
                        const uint32_t d = key[i] & mask1;
                        const unsigned long long act = __builtin_amdgcn_ballot_w64(true);
                        const uint32_t d0 = __builtin_amdgcn_readfirstlane(d);
                        if (__builtin_amdgcn_ballot_w64(d == d0) == act) {      // one add for a wave-uniform digit
                            const uint32_t lower = count_lower_mask(act);
                            uint32_t base = 0;
                            if (lower == 0) base = atomicAdd(&sm.hist[d0], (uint32_t)__popcll(act));
                            pos[i] = __builtin_amdgcn_readfirstlane(base) + lower;
                        } else {
                            pos[i] = atomicAdd(&sm.hist[d], 1u);
                        }
                    }
                }
            }
#pragma unroll
            for (int i = 0; i < KPT; ++i) asm volatile("" : "+v"(pos[i]), "+v"(key[i]));
            LSP_WAIT("lgkmcnt(0)");
            LSP(3);                                       // first pass: fetch-adds
            __syncthreads();
            LSP(4);                                       // barrier
            // exclusive scan of the bins in place: IPT consecutive bins per thread (fewer bins: fewer threads)
            constexpr uint32_t IPT = (1u << LOCAL_B1) / THREADS;
            const bool act = (uint32_t)tid * IPT < nbins;
            uint32_t ssum = 0;
            if (act) {
#pragma unroll
                for (uint32_t q = 0; q < IPT; q += 4) {
                    const uint4 c4 = reinterpret_cast<const uint4 *>(sm.hist)[((uint32_t)tid * IPT + q) >> 2];
                    ssum += c4.x + c4.y + c4.z + c4.w;
                }
            }
            const uint32_t inc = wave_inclusive_scan(ssum);
            if (lane == 63) sm.wtot[w] = inc;
            __syncthreads();
            const uint32_t wsv = (lane < WAVES) ? sm.wtot[lane] : 0u;
            const uint32_t wincl = wave_inclusive_scan(wsv);
            uint32_t run = (uint32_t)__shfl((int)(wincl - wsv), w, WAVE) + inc - ssum;
            if (act) {
#pragma unroll
                for (uint32_t q = 0; q < IPT; q += 4) {
                    uint4 *p4 = reinterpret_cast<uint4 *>(sm.hist) + (((uint32_t)tid * IPT + q) >> 2);
                    const uint4 c4 = *p4;
                    uint4 e4;
                    e4.x = run; e4.y = e4.x + c4.x; e4.z = e4.y + c4.y; e4.w = e4.z + c4.z;
                    run = e4.w + c4.w;
                    *p4 = e4;
                }
            }
            LSP_WAIT("lgkmcnt(0)");
            LSP(5);                                       // scan of the bins (one barrier inside)
            __syncthreads();
            LSP(6);                                       // barrier
#pragma unroll
            for (int i = 0; i < KPT; ++i) pos[i] += sm.hist[key[i] & mask1];
#pragma unroll
            for (int i = 0; i < KPT; ++i) asm volatile("" : "+v"(pos[i]));
            LSP(7);                                       // base lookup
            __syncthreads();                              // the histogram is dead: the buffer takes the keys
            LSP(8);                                       // barrier
            {
                const uint32_t wbase = fresh(wbase0);
#pragma unroll
                for (int i = 0; i < KPT; ++i) {
                    if (wbase + i * WAVE < T.size) {
                        if (HAS_VALUES) reinterpret_cast<uint2 *>(sm.stage)[pos[i]] = make_uint2(key[i], val[i]);
                        else sm.stage[pos[i]] = key[i];
                    }
                }
            }
            LSP_WAIT("lgkmcnt(0)");
            LSP(9);                                       // keys into the staging buffer
            __syncthreads();
            LSP(10);                                      // barrier
        }
        // ---- remaining bits: stable passes of <= 8 bits
        uint32_t rem = B - b1, np = (rem + 7u) / 8u, shift = b1 + (STABLE ? T.pad : 0u);
        while (rem) {
            const uint32_t b = (rem + np - 1u) / np, nd = 1u << b;
            const uint32_t wbase = fresh(wbase0);
            if (!in_regs) {
#pragma unroll
            for (int i = 0; i < KPT; ++i) {               // back into registers in position order
                const uint32_t idx = wbase + i * WAVE;
                if (HAS_VALUES) {
                    const uint2 kv = reinterpret_cast<const uint2 *>(sm.stage)[idx];
                    key[i] = kv.x; val[i] = kv.y;
                } else {
                    key[i] = sm.stage[idx];
                }
                if (idx >= T.size) key[i] = 0xffffffffu;  // the pads: last in position and largest in every digit
            }
            __syncthreads();                              // then the counters may overwrite the buffer
            }
            in_regs = false;
#pragma unroll
            for (int i = lane; i < RADIX; i += WAVE) my[i] = 0;
            auto rank_pass = [&](auto width) {
                uint32_t d_prev = 0, plo = 0, phi = 0;
#pragma unroll
                for (int i = 0; i <= KPT; ++i) {
                    uint32_t d_cur = 0, clo = 0, chi = 0;
                    if (i < KPT) {
                        d_cur = __builtin_amdgcn_ubfe(key[i], shift, b);
                        constexpr int MB = decltype(width)::value;
                        if (MB == 2) match_digit2(d_cur, clo, chi);
                        else if (MB == 4) match_digit4(d_cur, clo, chi);
                        else if (MB == 5) match_digit5(d_cur, clo, chi);
                        else match_digit(d_cur, clo, chi);
                    }
                    if (i > 0) {
                        const uint32_t lower = count_lower(plo, phi);
                        pos[i - 1] = my[d_prev] + lower;
                        if (lower == 0)
                            __hip_atomic_fetch_add(&my[d_prev], (uint32_t)(__popc(plo) + __popc(phi)), __ATOMIC_RELAXED,
                                                   __HIP_MEMORY_SCOPE_WAVEFRONT);
                    }
                    d_prev = d_cur; plo = clo; phi = chi;
                }
            };
            if (b <= 2u) rank_pass(std::integral_constant<int, 2>{});
            else if (b <= 4u) rank_pass(std::integral_constant<int, 4>{});
            else if (b <= 5u) rank_pass(std::integral_constant<int, 5>{});
            else rank_pass(std::integral_constant<int, 8>{});
#pragma unroll
            for (int i = 0; i < KPT; ++i) asm volatile("" : "+v"(pos[i]), "+v"(key[i]));
            __syncthreads();
            // digit-parallel scan by the first 256 threads: per digit, exclusive prefix over the waves'
            // counts (in place), then the exclusive scan of the digit totals folded into every row
            uint32_t tot = 0, inc = 0;
            if ((uint32_t)tid < nd) {
#pragma unroll
                for (int j = 0; j < WAVES; ++j) {
                    const uint32_t c = sm.whist[j][tid];
                    sm.whist[j][tid] = tot;
                    tot += c;
                }
            }
            if (tid < RADIX) {
                inc = wave_inclusive_scan(tot);
                if (lane == 63) sm.wtot[w] = inc;
            }
            __syncthreads();
            if ((uint32_t)tid < nd) {
                uint32_t ex = inc - tot;
                if (w > 0) ex += sm.wtot[0];
                if (w > 1) ex += sm.wtot[1];
                if (w > 2) ex += sm.wtot[2];
#pragma unroll
                for (int j = 0; j < WAVES; ++j) sm.whist[j][tid] += ex;
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < KPT; ++i) pos[i] += my[__builtin_amdgcn_ubfe(key[i], shift, b)];
#pragma unroll
            for (int i = 0; i < KPT; ++i) asm volatile("" : "+v"(pos[i]));
            __syncthreads();                              // the counters are dead: the buffer takes the keys
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                if (HAS_VALUES) reinterpret_cast<uint2 *>(sm.stage)[pos[i]] = make_uint2(key[i], val[i]);
                else sm.stage[pos[i]] = key[i];
            }
            __syncthreads();
            shift += b; rem -= b; --np;
        }
        LSP(11);                                          // the stable passes, all of them
        }   // general plan
        // ---- request the next task's keys, then store this one from the buffer
        if (has_next) request(Tn);
        LSP(13);                                          // request the next task's keys
        // Aligned stores (round 3): a task's output starts wherever the sub-bucket before it ends, so a wave's 256-byte streaming stores
        // straddled three cache lines, two of them partly.  Where the class has room (size + shift <= capacity) thread t stores the
        // elements t - shift, t - shift + THREADS, ... (shift = offset mod 64 elements): every wave's store starts on a 256-byte
        // boundary.  The buffer's slots are still read (and cleared) exactly once each: the slots in front of element 0 wrap to its end.
        constexpr uint32_t CAP = (uint32_t)(KPT * THREADS);
        const uint32_t ssh = (T.size + (T.offset & 63u) <= CAP) ? (T.offset & 63u) : 0u;
        if (done) {
        uint32_t *qk = dst_k + T.offset, *qv = HAS_VALUES ? dst_v + T.offset : nullptr;
        if (HAS_VALUES) {
            for (uint32_t jj = tid; jj < T.size + ssh; jj += THREADS) {
                const uint32_t j = jj - ssh;
                if (j >= T.size) continue;
                const uint2 kv = reinterpret_cast<const uint2 *>(sm.stage)[j];
                // whole lines of final output, written once: streaming stores (+0.3...0.7 % on the whole sort)
                __builtin_nontemporal_store(PLAIN ? kv.x : twiddle_out(kv.x, f32_out, xor_out), &qk[j]);
                __builtin_nontemporal_store(kv.y, &qv[j]);
            }
        } else {
            const uint32_t t0 = fresh((uint32_t)tid);
            // element t0 - shift + i * THREADS sits in the slot of the same number; only round 0 can lie in front of element 0
            // (THREADS > shift), and its slot is then the one CAP further on: one base register, the rest immediate offsets
            const uint32_t jb = t0 - ssh, s0 = jb < CAP ? jb : jb + CAP;
            pos[0] = sm.stage[s0];
#pragma unroll
            for (int i = 1; i < KPT; ++i) pos[i] = sm.stage[jb + i * THREADS];   // `pos` is free: batch the reads
            if constexpr ((MODE == LS_ONEPASS || ls_is_dedupe(MODE)) && !HAS_VALUES) {
                // the slots become the next task's byte counters (map and counters): each thread clears what it has just read
#pragma unroll
                for (int i = 1; i < KPT; ++i) sm.stage[jb + i * THREADS] = 0;
                sm.stage[s0] = 0;
                zeroed = true;
            }
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t j = jb + i * THREADS;
                if (j < T.size) __builtin_nontemporal_store(PLAIN ? pos[i] : twiddle_out(pos[i], f32_out, xor_out), reinterpret_cast<uint32_t *>(reinterpret_cast<char *>(qk) + j * 4u));
            }
        }
        }   // done
        LSP(14);                                          // read the buffer, issue the stores
        __syncthreads();
#ifdef GS_EXP_LS_PHASES
        if (tid == 0 && ti < LSP_TASKS) {                  // [15]: the task's whole time in shader clocks, barrier included
            gs_ls_phase_buf[((LSP_PLAN * MSB_NCLASS + cls) * LSP_TASKS + ti) * 16 + 15] += (uint32_t)(__builtin_amdgcn_s_memtime() - t0_);
            (void)r0_;
        }
#endif
        if (!has_next) break;
        T = Tn; ti = tn;
    }
}

// ------------------------------------------------------------------ shard --

// 2^bits-bin histogram of the keys' top bits (after the order-preserving twiddle), u64 counts
template <bool VEC>
__global__ __launch_bounds__(MSB_THREADS) void shard_hist_kernel(const uint32_t *__restrict__ keys, uint64_t n, int bits,
                                                                 unsigned long long *__restrict__ hist, int f32_in,
                                                                 uint32_t xor_in)
{
    __shared__ uint32_t lh[1 << SHARD_MAX_BITS];
    const uint32_t nb = 1u << bits;
    for (uint32_t i = threadIdx.x; i < nb; i += MSB_THREADS) lh[i] = 0;
    __syncthreads();
    const int rshift = 32 - bits;
    auto count = [&](uint32_t k) { atomicAdd(&lh[twiddle_in(k, f32_in, xor_in) >> rshift], 1u); };
    // a block never counts more than 2^32 keys (n < 2^32 per call site), so u32 LDS counters suffice
    const uint64_t stride = (uint64_t)gridDim.x * MSB_THREADS;
    uint64_t done = 0;
    if (VEC) {
        const uint4 *k4 = reinterpret_cast<const uint4 *>(keys);
        const uint64_t nvec = n >> 2;
        uint64_t v = (uint64_t)blockIdx.x * MSB_THREADS + threadIdx.x;
        for (; v + 3 * stride < nvec; v += 4 * stride) {
            const uint4 a = k4[v], b = k4[v + stride], c = k4[v + 2 * stride], d = k4[v + 3 * stride];
            count(a.x); count(a.y); count(a.z); count(a.w);
            count(b.x); count(b.y); count(b.z); count(b.w);
            count(c.x); count(c.y); count(c.z); count(c.w);
            count(d.x); count(d.y); count(d.z); count(d.w);
        }
        for (; v < nvec; v += stride) { const uint4 a = k4[v]; count(a.x); count(a.y); count(a.z); count(a.w); }
        done = nvec << 2;
    }
    for (uint64_t i = done + (uint64_t)blockIdx.x * MSB_THREADS + threadIdx.x; i < n; i += stride) count(keys[i]);
    __syncthreads();
    for (uint32_t i = threadIdx.x; i < nb; i += MSB_THREADS)
        if (lh[i]) atomicAdd(&hist[i], (unsigned long long)lh[i]);
}

// per-rank counts from the single bucket's cursor row (exclusive prefix of the counts)
__global__ void shard_counts_kernel(const uint32_t *__restrict__ row, uint32_t n, int num_ranks, unsigned long long *__restrict__ counts)
{
    const int r = threadIdx.x;
    if (r < num_ranks) counts[r] = (unsigned long long)((r + 1 < RADIX ? row[r + 1] : n) - row[r]);
}

// ------------------------------------------------------------------- host --

// `min_bits`: the bits a task of this launch has left (the level's remaining bits, or 8 more for a merged task).
// The one-pass sort only pays when it replaces TWO passes of the general plan (more bits than the class's first
// pass takes) and its byte counters fit; a class where neither task kind qualifies goes straight to LS_ALL.
static inline bool onepass_possible(int min_bits, int b1, int onepass_bits)
{
    return (min_bits > b1 && min_bits <= onepass_bits) || (min_bits + 8 > b1 && min_bits + 8 <= onepass_bits);
}
template <bool HAS_VALUES, bool STABLE = false>
static void launch_local_sorts(const MsbWs &ws, int L, uint32_t bound, const uint32_t *sk, uint32_t *dk, const uint32_t *sv,
                               uint32_t *dv, int f32_in, uint32_t xor_in, int f32_out, uint32_t xor_out, hipStream_t s,
                               int min_bits = 0, uint64_t num_items = 0, const uint32_t *known_tasks = nullptr, bool maybe_skew = true)
{
    KernelTimer kt(GS_K_MSB_LOCAL_SORT, s);
    // grid-stride over the task list; few blocks suffice for the big classes (empty launches are not free)
    // `known_tasks` (the host has looked, see MsbPeek): tasks per class -- empty classes are not launched at all
    auto grid_of = [&](int c) {
        uint64_t b = bound;
        if (known_tasks && known_tasks[c] < b) b = known_tasks[c];
        if (num_items && c > 0) {
            const uint64_t lim = num_items / msb_class_cap(c - 1) + RADIX;     // a class-c task holds > cap(c-1) keys
            if (lim < b) b = lim;
        }
#ifndef MSB_LS_MAX_GRID
#define MSB_LS_MAX_GRID MSB_MAX_GRID
#endif
        return (uint32_t)(b < MSB_LS_MAX_GRID ? (b ? b : 1) : MSB_LS_MAX_GRID);
    };
#define GS_LS1P(C, HV, M, P) hipLaunchKernelGGL((msb_local_sort_kernel<msb_class_threads(C), msb_class_kpt(C), HV, STABLE, M, P>), dim3(grid_of(C)), \
                                               dim3(msb_class_threads(C)), 0, s, ws, L, C, sk, dk, sv, dv, f32_in, xor_in, f32_out, xor_out)
#define GS_LS1(C, HV, M) GS_LS1P(C, HV, M, false)
    // (a resident grid of 1024 workgroups striding over the list instead of one workgroup per task, up to 16384: Zipf local sorts 1.69 -> 1.84 ms)
#define GS_LSDD(C, HV, M) hipLaunchKernelGGL((msb_local_sort_kernel<msb_class_threads(C), msb_class_kpt(C), HV, STABLE, M, false>),                \
                                             dim3(grid_of(C)), dim3(msb_class_threads(C)), 0, s, ws, L, C, sk, dk, sv, dv,                             \
                                             f32_in, xor_in, f32_out, xor_out)
    const bool plain = !f32_in && !xor_in && !f32_out && !xor_out;
    // GS_MSB_DEDUPE=0 switches the few-distinct-values plan off (A/B measurements; read once per process)
    static const bool dedupe_on = [] { const char *e = getenv("GS_MSB_DEDUPE"); return !(e && e[0] == '0'); }();
    // unstable sort: the one-pass kernel takes what it can and flags the rest for the general one
#define GS_LS(C, HV)                                                                                                  \
    do {                                                                                                              \
        constexpr bool DD = !STABLE && msb_class_cap(C) * (HV ? 2u : 1u) >= DD_BITW + DD_CNT;                           \
        const bool dd = DD && dedupe_on && maybe_skew && L < 3 && onepass_possible(min_bits, local_b1((int)msb_class_cap(C)), 16);          \
        if constexpr (STABLE) { GS_LS1(C, HV, LS_ALL); }                                                              \
        else if (!onepass_possible(min_bits, local_b1((int)msb_class_cap(C)), local_b1((int)msb_class_cap(C) * (HV ? 2 : 1)) + 2)) \
            { if constexpr (DD) { if (dd) GS_LSDD(C, HV, LS_DEDUPE_ALL); } GS_LS1(C, HV, LS_ALL); }                    \
        else { if (plain) GS_LS1P(C, HV, LS_ONEPASS, true); else GS_LS1(C, HV, LS_ONEPASS);                           \
               if constexpr (DD) { if (dd) GS_LSDD(C, HV, LS_DEDUPE); } GS_LS1(C, HV, LS_FLAGGED); }                   \
    } while (0)
    auto wanted = [&](int c) { return !known_tasks || known_tasks[c] != 0u; };
    if constexpr (!STABLE) {
        // the look at the tasks' samples (exits at once when the level showed no skew); one wave per task
        uint64_t waves = bound;
        if (known_tasks) waves = (uint64_t)known_tasks[0] + known_tasks[1] + known_tasks[2] + known_tasks[3];
        if (L < 3 && waves && maybe_skew) {
            const uint32_t blocks = (uint32_t)(waves / 4 + 1 < 2048 ? waves / 4 + 1 : 2048);
            hipLaunchKernelGGL(msb_task_sample_kernel, dim3(blocks), dim3(256), 0, s, ws, L, sk, HAS_VALUES ? 1 : 0);
        }
    }
    if (wanted(0)) GS_LS(0, HAS_VALUES);
    if (wanted(1)) GS_LS(1, HAS_VALUES);
    if (wanted(2)) GS_LS(2, HAS_VALUES);
    if ((!HAS_VALUES || GS_PAIR_CLASSES > 3) && wanted(3)) GS_LS(3, HAS_VALUES);
#undef GS_LS
#undef GS_LS1
#undef GS_LS1P
#undef GS_LSDD
}

// Levels 1..3 (partition on bytes 2, 1, 0 + the local sorts after each): keys travel between
// buf[1] (level-1 source) and buf[0] (level-1 destination and final result).  `npieces` != 0:
// the level-1 buckets and their pieces were written by the host (gs_msb_finish_u32) and the tile
// records come from the pieces.
// GS_MSB_PIVOT=0 switches the heavy-hitter path off (A/B measurements; read once per process)
static inline bool msb_pivot_enabled()
{
    static const bool on = [] { const char *e = getenv("GS_MSB_PIVOT"); return !(e && e[0] == '0'); }();
    return on;
}

// ---- a look at the next level's size (levels 2 and 3).  Every level is launched with grids sized for the worst case and
// its surplus blocks exit at once, but the launches of a level that turns out EMPTY still cost ~0.2 ms at 2^30 keys (131 K
// blocks to dispatch for the scatter alone) -- 5 % of a uniform sort, 30 % at 2^24 keys.  So, when the stream is not being
// captured into a graph: right after level L's classification a one-thread kernel copies level L+1's bucket / tile counts to
// a pinned host word and an event is recorded; the scatter and the local sorts of level L are enqueued (milliseconds of
// work), and only then the host waits for that event -- the device has long passed it -- and either skips the remaining
// levels or launches level L+1 with exact grids.  The device never idles; under graph capture (no host waits allowed)
// and with GS_MSB_PEEK=0 the worst-case grids are used as before.
__global__ void msb_peek_kernel(MsbWs ws, int L, unsigned long long *mailbox)   // L = the level about to be launched
{
    mailbox[0] = ws.level[L].packed;
    mailbox[1] = ((unsigned long long)ws.level[L - 1].task_count[1] << 32) | ws.level[L - 1].task_count[0];   // the tasks the
    mailbox[2] = ((unsigned long long)ws.level[L - 1].task_count[3] << 32) | ws.level[L - 1].task_count[2];   // level before it emitted
    __threadfence_system();
}
struct MsbPeek {
    unsigned long long *host = nullptr, *dev = nullptr;
    hipEvent_t ev = nullptr;
    int device = -1;
    bool armed = false;
};
static bool msb_peek_enabled()
{
    static const bool on = [] { const char *e = getenv("GS_MSB_PEEK"); return !(e && e[0] == '0'); }();
    return on;
}
// The calling thread's mailboxes, one per device it has sorted on (a host thread that drives several GPUs keeps them all:
// re-creating the pinned word on every switch would hipHostFree -- a device synchronisation -- inside every sort), released
// when the thread ends.
constexpr int MSB_PEEK_DEVICES = 16;
struct MsbPeekTable {
    MsbPeek slot[MSB_PEEK_DEVICES];
    ~MsbPeekTable()
    {
        for (MsbPeek &pk : slot) {
            if (pk.ev) (void)hipEventDestroy(pk.ev);
            if (pk.host) (void)hipHostFree(pk.host);
        }
    }
};
// the calling thread's mailbox for the current device, or nullptr (then the caller keeps the worst-case grids)
static MsbPeek *msb_peek_get(hipStream_t s)
{
    if (!msb_peek_enabled()) return nullptr;
    hipStreamCaptureStatus cs = hipStreamCaptureStatusNone;
    if (hipStreamIsCapturing(s, &cs) != hipSuccess || cs != hipStreamCaptureStatusNone) { (void)hipGetLastError(); return nullptr; }
    thread_local MsbPeekTable table;
    int dev = -1;
    if (hipGetDevice(&dev) != hipSuccess || dev < 0 || dev >= MSB_PEEK_DEVICES) { (void)hipGetLastError(); return nullptr; }
    MsbPeek &pk = table.slot[dev];
    if (pk.device != dev) {
        void *h = nullptr, *d = nullptr;
        if (hipHostMalloc(&h, 64, hipHostMallocMapped | hipHostMallocPortable) != hipSuccess ||
            hipHostGetDevicePointer(&d, h, 0) != hipSuccess ||
            hipEventCreateWithFlags(&pk.ev, hipEventDisableTiming) != hipSuccess) {
            (void)hipGetLastError();
            if (h) (void)hipHostFree(h);
            pk.ev = nullptr;
            return nullptr;
        }
        pk.host = (unsigned long long *)h; pk.dev = (unsigned long long *)d; pk.device = dev;
    }
    // a look armed by an earlier call that returned early (an error between arm and wait): its kernel may still be about
    // to write the mailbox -- wait it out before the word is reused
    if (pk.armed) { (void)hipEventSynchronize(pk.ev); (void)hipGetLastError(); }
    pk.armed = false;
    return &pk;
}

// after level L-1's classification: ask for level L's size and level L-1's task counts
static void msb_peek_arm(MsbPeek *pk, const MsbWs &ws, int L, hipStream_t s)
{
    if (!pk) return;
    hipLaunchKernelGGL(msb_peek_kernel, dim3(1), dim3(1), 0, s, ws, L, pk->dev);
    pk->armed = hipEventRecord(pk->ev, s) == hipSuccess;
    if (!pk->armed) (void)hipGetLastError();
}
struct MsbLook { bool ok = false; uint32_t buckets = 0, tiles = 0, tasks[MSB_NCLASS] = {0, 0, 0, 0}; };
// to be called once plenty of work has been enqueued behind the classification (the device is busy while the host waits)
static MsbLook msb_peek_wait(MsbPeek *pk)
{
    MsbLook lk;
    if (!pk || !pk->armed) return lk;
    pk->armed = false;
    if (hipEventSynchronize(pk->ev) != hipSuccess) { (void)hipGetLastError(); return lk; }
    const volatile unsigned long long *m = pk->host;
    const unsigned long long a = m[0], b = m[1], c = m[2];
    lk.ok = true;
    lk.buckets = (uint32_t)(a >> 32); lk.tiles = (uint32_t)a;
    lk.tasks[0] = (uint32_t)b; lk.tasks[1] = (uint32_t)(b >> 32); lk.tasks[2] = (uint32_t)c; lk.tasks[3] = (uint32_t)(c >> 32);
    return lk;
}

// `stop_level` (test access, gs_msb_classify_upto): return right after that level's classification; `allow_pivot` = false
// keeps the heavy-hitter path off whatever the environment says.
static void msb_run_levels(const MsbWs &ws, uint64_t num_items, bool pairs, uint32_t npieces, uint32_t *const buf_k[2],
                           uint32_t *const buf_v[2], const PassParams &tw, hipStream_t s, int stop_level = 99,
                           bool allow_pivot = true)
{
    const int nclass = msb_num_classes(pairs);
    const uint32_t tiles_all = (uint32_t)((num_items + MSB_TILE - 1) / MSB_TILE);
    const uint32_t max_tasks_lvl = ws.max_tasks;
    uint32_t *d_keys = buf_k[0], *d_vals = buf_v[0];
    // (not in the multi-GPU finish, npieces != 0: that path stays free of host waits between its collectives)
    MsbPeek *peek = (stop_level == 99 && (npieces == 0 || getenv("GS_MSB_PEEK_FINISH"))) ? msb_peek_get(s) : nullptr;   // env: diagnosis only
    uint32_t known_b = 0, known_tiles = 0;     // level L's exact bucket / tile counts when the look succeeded
    bool known = false;
    for (int L = 1; L <= 3; ++L) {
        const int shift = 24 - 8 * L;
        const DigitSel dsel{shift, nullptr, 0, 0, 0u, 8, 0};
        uint32_t *sk = buf_k[L & 1], *dk = buf_k[(L + 1) & 1];
        uint32_t *sv = buf_v[L & 1], *dv = buf_v[(L + 1) & 1];
        const bool in_pieces = npieces != 0 && L == 1;
        // heavy-hitter path: keys only, buckets in one piece, and not at the last byte (a level-2 bucket's strangers
        // already cover it)
        const bool pivot = allow_pivot && msb_pivot_enabled() && !in_pieces && L <= 2;
        // buckets at level L: <= 256 at level 1, else bounded by size; tiles: n/T + one ragged tile per bucket (piece)
        uint32_t max_b = (L == 1) ? (uint32_t)RADIX : ws.max_buckets;
        uint32_t max_tiles = tiles_all + (in_pieces ? 2u * npieces : max_b + tiles_all / MSB_ALIGN_MIN_TILES + 1u);
        if (known && L >= 2) {                 // exact (never more than the bounds above)
            if (known_b < max_b) max_b = known_b;
            if (known_tiles < max_tiles) max_tiles = known_tiles;
            known = false;
        }
        // one tile per block, dispatched in order (blocks that own a long run of tiles march in
        // lockstep and lose a third of the bandwidth, like the LSB downsweep); surplus blocks exit
        const bool last = (L == 3);
        { KernelTimer kt(GS_K_MSB_HISTOGRAM, s);
          if (in_pieces) {
              hipLaunchKernelGGL(msb_expand_pieces_kernel, dim3(npieces < 4096u ? npieces : 4096u), dim3(256), 0, s, ws, npieces);
          } else {
              const uint32_t eg = max_b < 4096u ? max_b : 4096u;
              hipLaunchKernelGGL(msb_expand_kernel, dim3(eg), dim3(256), 0, s, ws, L, pivot ? (const uint32_t *)sk : (const uint32_t *)nullptr);
          }
          const uint32_t hg_ub = max_tiles / MSB_WAVES + 1;                 // one block per chunk
          const uint32_t hg = hg_ub < MSB_MAX_GRID ? hg_ub : MSB_MAX_GRID;
          if (pivot) hipLaunchKernelGGL((msb_upsweep_kernel<false, true>), dim3(hg), dim3(MSB_THREADS), 0, s, ws, L, (const uint32_t *)sk, dsel);
          else hipLaunchKernelGGL((msb_upsweep_kernel<false, false>), dim3(hg), dim3(MSB_THREADS), 0, s, ws, L, (const uint32_t *)sk, dsel);
          hipLaunchKernelGGL(msb_scan_kernel, dim3(RADIX), dim3(1024), 0, s, ws, L); }
        { KernelTimer kt(GS_K_MSB_CLASSIFY, s);
          const uint32_t cg = max_b < 4096u ? max_b : 4096u;
          if (last) hipLaunchKernelGGL((msb_classify_kernel<true, false>), dim3(cg), dim3(256), 0, s, ws, L, (const uint32_t *)nullptr, nclass);
          else if (pivot) hipLaunchKernelGGL((msb_classify_kernel<false, true>), dim3(cg), dim3(256), 0, s, ws, L, (const uint32_t *)nullptr, nclass);
          else hipLaunchKernelGGL((msb_classify_kernel<false, false>), dim3(cg), dim3(256), 0, s, ws, L, (const uint32_t *)nullptr, nclass); }
        if (L == stop_level) return;
        if (!last) msb_peek_arm(peek, ws, L + 1, s);
        { KernelTimer kt(GS_K_MSB_PARTITION, s);
          const bool big = num_items > (1ull << 30);
          const uint32_t *svc = pairs ? (const uint32_t *)sv : (const uint32_t *)nullptr;
          uint32_t *dvc = pairs ? dv : (uint32_t *)nullptr;
#define GS_SC(HV, TW) launch_scatter<HV, false, TW>(ws, L, max_tiles, max_b, big, (const uint32_t *)sk, dk, svc, dvc, dsel, tw.f32_out, tw.xor_out, s, in_pieces, \
                                                  pivot ? d_keys : (uint32_t *)nullptr, tw.f32_out, tw.xor_out, pivot && pairs ? d_vals : (uint32_t *)nullptr)
          if (pairs) { if (last) GS_SC(true, true); else GS_SC(true, false); }
          else { if (last) GS_SC(false, true); else GS_SC(false, false); }
#undef GS_SC
        }
        if (!last) {
            // the scatter is enqueued: now the host may wait for the look (the device is busy with the scatter)
            const MsbLook lk = msb_peek_wait(peek);
            // a bucket emits at most 256 tasks
            const uint32_t tb = (uint64_t)max_b * RADIX < (uint64_t)max_tasks_lvl ? max_b * (uint32_t)RADIX : max_tasks_lvl;
            const uint32_t *kt_ = lk.ok ? lk.tasks : nullptr;
            const bool skew = !(lk.ok && lk.buckets == 0);   // the host has looked: no next level = nothing for the sample look to find
            if (pairs) launch_local_sorts<true>(ws, L, tb, dk, d_keys, dv, d_vals, 0, 0u, tw.f32_out, tw.xor_out, s, 24 - 8 * L, num_items, kt_, skew);
            else launch_local_sorts<false>(ws, L, tb, dk, d_keys, nullptr, nullptr, 0, 0u, tw.f32_out, tw.xor_out, s, 24 - 8 * L, num_items, kt_, skew);
            if (lk.ok) {
                if (lk.buckets == 0) return;   // nothing left for the levels below
                known = true; known_b = lk.buckets; known_tiles = lk.tiles;
            }
        }
    }
}

// ---- single-workgroup path of the LSB sort (CUB's single-tile path, dispatch_radix_sort.cuh:1182-1187):
// arrays that fit one workgroup are sorted by ONE stable local sort instead of 3 launches per pass
__global__ void msb_small_task_kernel(MsbLevel *level, MsbTask *task, uint32_t n, int cls, uint32_t bits, uint32_t shift0)
{
    if (threadIdx.x == 0) {
        MsbLevel z{};
        z.task_count[cls] = 1;
        level[0] = z;
        task[0] = MsbTask{0u, n, bits, shift0};
    }
}

uint32_t small_sort_capacity(bool pairs) { return msb_class_cap(msb_num_classes(pairs) - 1); }

int small_stable_sort(void *scratch, size_t scratch_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
                      uint32_t n, int begin_bit, int end_bit, int f32_in, uint32_t xor_in, int f32_out, uint32_t xor_out,
                      hipStream_t s)
{
    const bool pairs = vin != nullptr;
    if (n == 0 || n > small_sort_capacity(pairs) || scratch_bytes < 256 + sizeof(MsbTask)) return hipErrorInvalidValue;
    int cls = 0;
    while (msb_class_cap(cls) < n) ++cls;
    MsbWs ws{};
    ws.level = (MsbLevel *)scratch;
    for (int c = 0; c < MSB_NCLASS; ++c) ws.tasks[c] = (MsbTask *)((char *)scratch + 256);
    ws.max_tasks = 1;
    KernelTimer kt(GS_K_MSB_LOCAL_SORT, s);
    hipLaunchKernelGGL(msb_small_task_kernel, dim3(1), dim3(64), 0, s, ws.level, ws.tasks[cls], n, cls,
                       (uint32_t)(end_bit - begin_bit), (uint32_t)begin_bit);
#define GS_SM(C, HV) hipLaunchKernelGGL((msb_local_sort_kernel<msb_class_threads(C), msb_class_kpt(C), HV, true, LS_ALL>), dim3(1), \
                                        dim3(msb_class_threads(C)), 0, s, ws, 0, C, kin, kout, vin, vout, f32_in, xor_in, f32_out, xor_out)
    if (pairs) { if (cls == 0) GS_SM(0, true); else if (cls == 1) GS_SM(1, true); else if (cls == 2 || GS_PAIR_CLASSES < 4) GS_SM(2, true); else GS_SM(3, true); }
    else { if (cls == 0) GS_SM(0, false); else if (cls == 1) GS_SM(1, false); else if (cls == 2) GS_SM(2, false); else GS_SM(3, false); }
#undef GS_SM
    return (int)hipGetLastError();
}


// ---- segmented sort (cub::DeviceSegmentedRadixSort, dispatch_radix_sort.cuh:321-432)
// Segments that fit a workgroup become stable local-sort tasks; the others become the buckets of ONE level
// that is partitioned once per 8-bit digit, least significant first, with the level machinery above.
// `tiny_cap` != 0: segments of up to tiny_cap (256 / 512 / 1024) elements go to three lists of their own (one WAVE sorts such
// a segment with 4 / 8 / 16 elements per lane, seg_wave_sort_kernel) and those of up to 64 elements to a fourth (a wave sorts
// four of them at a time, seg_wave4_sort_kernel): list q grows downwards from the end of the class-q task array and is
// counted in level[2].task_count[q] (all lists together hold at most one task per segment, which is what the arrays are
// sized for).
constexpr uint32_t SEG_TINY = 256;           // the smallest of the three; the largest is 4 * SEG_TINY
__global__ __launch_bounds__(256) void seg_classify_kernel(MsbWs ws, const int *__restrict__ seg_begin,
                                                           const int *__restrict__ seg_end, uint32_t nseg, int nclass,
                                                           uint32_t sort_bits, uint32_t shift0, uint32_t num_items,
                                                           uint32_t tiny_cap = 0)
{
    // 4 segments per thread, and ONE global atomic per workgroup step and list (1024 segments): the per-wave atomics on the list
    // counters were the kernel's time for many segments (2^20 segments: 190 us)
    constexpr int NW = 4, SPT = 4, NLIST = NW + MSB_NCLASS;   // lists 0-2: one wave per segment, 3: four segments per wave; 4..: the classes
    __shared__ uint32_t s_cnt[NLIST], s_base[NLIST];
    const uint32_t cap_max = ws.caps[nclass - 1];
    const int tid = threadIdx.x;
    for (uint32_t base = blockIdx.x * (256u * SPT); base < nseg; base += gridDim.x * (256u * SPT)) {
        if (tid < NLIST) s_cnt[tid] = 0;
        __syncthreads();
        uint32_t b[SPT], size[SPT], off[NLIST], cnt[NLIST];
        int list[SPT];
#pragma unroll
        for (int q = 0; q < NLIST; ++q) cnt[q] = 0;
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
            const uint32_t sg = base + (uint32_t)tid * SPT + j;
            b[j] = 0; size[j] = 0; list[j] = -1;
            if (sg < nseg) {
                int lo = seg_begin[sg], hi = seg_end[sg];
                // offsets outside [0, num_items] are the caller's error; clamp them so that they cannot become
                // out-of-bounds accesses
                if (lo < 0) lo = 0;
                if (hi > (int)num_items) hi = (int)num_items;
                if (hi > lo) { b[j] = (uint32_t)lo; size[j] = (uint32_t)(hi - lo); }
            }
            const uint32_t sz = size[j];
            if (sz != 0 && tiny_cap != 0 && sz <= 4u * tiny_cap) list[j] = sz <= (uint32_t)WAVE ? 3 : sz <= tiny_cap ? 0 : sz <= 2u * tiny_cap ? 1 : 2;
            else if (sz != 0 && sz <= cap_max) { int c = 0; while (ws.caps[c] < sz) ++c; list[j] = NW + c; }
            else if (sz > cap_max) {          // a bucket of the level (rare: one atomic each)
                const uint32_t tiles = ws_tiles_of_at(ws, b[j], sz);
                const unsigned long long old = atomicAdd(&ws.level[1].packed, (1ull << 32) | tiles);
                if ((uint32_t)(old >> 32) < ws.max_buckets) ws.buckets[1][(uint32_t)(old >> 32)] = MsbBucket{b[j], sz, (uint32_t)old, tiles};
                else MSB_OVERFLOW(ws);
            }
#pragma unroll
            for (int q = 0; q < NLIST; ++q) cnt[q] += list[j] == q ? 1u : 0u;
        }
#pragma unroll
        for (int q = 0; q < NLIST; ++q) off[q] = cnt[q] ? atomicAdd(&s_cnt[q], cnt[q]) : 0u;       // LDS
        __syncthreads();
        if (tid < NLIST && s_cnt[tid] != 0u)
            s_base[tid] = atomicAdd(tid < NW ? &ws.level[2].task_count[tid] : &ws.level[1].task_count[tid - NW], s_cnt[tid]);
        __syncthreads();
#pragma unroll
        for (int j = 0; j < SPT; ++j) {
#pragma unroll
            for (int q = 0; q < NLIST; ++q) {
                if (list[j] != q) continue;
                const uint32_t at = s_base[q] + off[q]++;
                if (at < ws.max_tasks) {
                    if (q < NW) ws.tasks[q][ws.max_tasks - 1u - at] = MsbTask{b[j], size[j], sort_bits, shift0};
                    else ws.tasks[q - NW][at] = MsbTask{b[j], size[j], sort_bits, shift0};
                } else {
                    MSB_OVERFLOW(ws);
                }
            }
        }
        __syncthreads();
    }
}


// Segments of up to 256 elements: ONE WAVE sorts a segment (4 elements per lane), four segments at a time per workgroup and no
// workgroup barrier anywhere -- a 512-thread workgroup per 256-element segment spent its time in barriers (2^28 keys in 2^20
// segments: 12.8 ms).  Stable LSD passes of 8 bits: wave64 ballot match + wave-private counters for the ranks, the 256 counters
// scanned by the wave (4 per lane), elements exchanged through the wave's own 1-2 KiB of LDS.  LDS serves one wave's operations
// in order, so a lane reads what another lane of its wave wrote earlier without any wait beyond the compiler fence.
template <bool HAS_VALUES, int WKPT /* 4, 8, 16: list 0, 1, 2 */>
__global__ __launch_bounds__(256) void seg_wave_sort_kernel(MsbWs ws, const uint32_t *__restrict__ src_k, uint32_t *__restrict__ dst_k,
                                                            const uint32_t *__restrict__ src_v, uint32_t *__restrict__ dst_v, int f32_in,
                                                            uint32_t xor_in, int f32_out, uint32_t xor_out)
{
    constexpr int LIST = WKPT == 4 ? 0 : WKPT == 8 ? 1 : 2, CAP = WKPT * WAVE;
    __shared__ __attribute__((aligned(16))) uint32_t hist[4][RADIX];
    __shared__ uint32_t stage_k[4][CAP];
    __shared__ uint32_t stage_v[HAS_VALUES ? 4 : 1][HAS_VALUES ? CAP : 1];
    const int w = wave_id(), lane = lane_id();
    uint32_t ntasks = ws.level[2].task_count[LIST];
    if (ntasks > ws.max_tasks) ntasks = ws.max_tasks;
    uint32_t *my = hist[w];
    auto fence = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    for (uint32_t t = blockIdx.x * 4u + (uint32_t)w; t < ntasks; t += gridDim.x * 4u) {
        const MsbTask Tv = ws.tasks[LIST][ws.max_tasks - 1u - t];
        const uint32_t off = __builtin_amdgcn_readfirstlane(Tv.offset), size = __builtin_amdgcn_readfirstlane(Tv.size);
        const uint32_t B = __builtin_amdgcn_readfirstlane(Tv.sort_bits), shift0 = __builtin_amdgcn_readfirstlane(Tv.pad);
        uint32_t key[WKPT], val[HAS_VALUES ? WKPT : 1], pos[WKPT];
        const uint32_t last = size - 1u;
#pragma unroll
        for (int i = 0; i < WKPT; ++i) {
            const uint32_t idx = (uint32_t)(i * WAVE + lane), at = off + (idx < last ? idx : last);
            key[i] = src_k[at];
            if (HAS_VALUES) val[i] = src_v[at];
        }
#pragma unroll
        for (int i = 0; i < WKPT; ++i) {
            const uint32_t k = twiddle_in(key[i], f32_in, xor_in);
            key[i] = ((uint32_t)(i * WAVE + lane) < size) ? k : 0xffffffffu;   // pads: last in position, largest in every digit
        }
        for (uint32_t done = 0; done < B; done += RADIX_BITS) {
            const uint32_t bw = B - done < (uint32_t)RADIX_BITS ? B - done : (uint32_t)RADIX_BITS, sh = shift0 + done;
            reinterpret_cast<uint4 *>(my)[lane] = make_uint4(0u, 0u, 0u, 0u);
            fence();
#pragma unroll
            for (int i = 0; i < WKPT; ++i) {
                const uint32_t d = __builtin_amdgcn_ubfe(key[i], sh, bw);
                uint32_t lo, hi;
                match_digit(d, lo, hi);
                const uint32_t lower = count_lower(lo, hi);
                pos[i] = my[d] + lower;
                if (lower == 0) my[d] += (uint32_t)(__popc(lo) + __popc(hi));   // one lane per digit: no two writers of a word
                fence();
            }
            {   // exclusive scan of the 256 counters, 4 per lane
                const uint4 c = reinterpret_cast<const uint4 *>(my)[lane];
                const uint32_t sum = c.x + c.y + c.z + c.w;
                const uint32_t ex = wave_inclusive_scan(sum) - sum;
                reinterpret_cast<uint4 *>(my)[lane] = make_uint4(ex, ex + c.x, ex + c.x + c.y, ex + c.x + c.y + c.z);
            }
            fence();
#pragma unroll
            for (int i = 0; i < WKPT; ++i) {
                const uint32_t at = pos[i] + my[__builtin_amdgcn_ubfe(key[i], sh, bw)];
                stage_k[w][at] = key[i];
                if (HAS_VALUES) stage_v[w][at] = val[i];
            }
            fence();
#pragma unroll
            for (int i = 0; i < WKPT; ++i) {
                key[i] = stage_k[w][i * WAVE + lane];
                if (HAS_VALUES) val[i] = stage_v[w][i * WAVE + lane];
            }
            fence();
        }
#pragma unroll
        for (int i = 0; i < WKPT; ++i) {
            const uint32_t idx = (uint32_t)(i * WAVE + lane);
            if (idx < size) {
                dst_k[off + idx] = twiddle_out(key[i], f32_out, xor_out);
                if (HAS_VALUES) dst_v[off + idx] = val[i];
            }
        }
    }
}

// Segments of up to 64 elements: a wave sorts FOUR of them at a time, one per round of its four elements per lane -- every
// segment has its own 256 counters and its own 64 slots of the wave's LDS, and a segment's elements sit in one round, so the
// rank inside a digit is just the count of equal digits in the lower lanes.  (One such segment per wave left three quarters of
// the wave idle: 2^23 segments of 32 keys took 11 ms.)
template <bool HAS_VALUES>
__global__ __launch_bounds__(256) void seg_wave4_sort_kernel(MsbWs ws, const uint32_t *__restrict__ src_k, uint32_t *__restrict__ dst_k,
                                                             const uint32_t *__restrict__ src_v, uint32_t *__restrict__ dst_v, int f32_in,
                                                             uint32_t xor_in, int f32_out, uint32_t xor_out)
{
    constexpr int NS = 4;                                          // segments per wave and step
    __shared__ __attribute__((aligned(16))) uint32_t hist[4][NS][RADIX];
    __shared__ uint32_t stage_k[4][NS * WAVE];
    __shared__ uint32_t stage_v[HAS_VALUES ? 4 : 1][HAS_VALUES ? NS * WAVE : 1];
    const int w = wave_id(), lane = lane_id();
    uint32_t ntasks = ws.level[2].task_count[3];
    if (ntasks > ws.max_tasks) ntasks = ws.max_tasks;
    auto fence = [] { __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront"); __builtin_amdgcn_wave_barrier(); };
    for (uint32_t t0 = (blockIdx.x * 4u + (uint32_t)w) * NS; t0 < ntasks; t0 += gridDim.x * 4u * NS) {
        uint32_t off[NS], size[NS], key[NS], val[HAS_VALUES ? NS : 1], pos[NS];
        uint32_t B = 0, shift0 = 0;
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            off[i] = 0; size[i] = 0;
            if (t0 + i < ntasks) {                                 // wave-uniform
                const MsbTask Tv = ws.tasks[3][ws.max_tasks - 1u - (t0 + i)];
                off[i] = __builtin_amdgcn_readfirstlane(Tv.offset); size[i] = __builtin_amdgcn_readfirstlane(Tv.size);
                B = __builtin_amdgcn_readfirstlane(Tv.sort_bits); shift0 = __builtin_amdgcn_readfirstlane(Tv.pad);   // the same for every segment of a call
            }
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            key[i] = 0xffffffffu;
            if (HAS_VALUES) val[i] = 0;
            if ((uint32_t)lane < size[i]) {
                key[i] = twiddle_in(src_k[off[i] + lane], f32_in, xor_in);
                if (HAS_VALUES) val[i] = src_v[off[i] + lane];
            } else {
                key[i] = 0xffffffffu;                              // pads: behind the segment's elements, largest in every digit
            }
        }
        for (uint32_t done = 0; done < B; done += RADIX_BITS) {
            const uint32_t bw = B - done < (uint32_t)RADIX_BITS ? B - done : (uint32_t)RADIX_BITS, sh = shift0 + done;
#pragma unroll
            for (int i = 0; i < NS; ++i) reinterpret_cast<uint4 *>(hist[w][i])[lane] = make_uint4(0u, 0u, 0u, 0u);
            fence();
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const uint32_t d = __builtin_amdgcn_ubfe(key[i], sh, bw);
                uint32_t lo, hi;
                match_digit(d, lo, hi);
                pos[i] = count_lower(lo, hi);
                if (pos[i] == 0) hist[w][i][d] = (uint32_t)(__popc(lo) + __popc(hi));
            }
            fence();
#pragma unroll
            for (int i = 0; i < NS; ++i) {   // exclusive scan of segment i's 256 counters, 4 per lane
                const uint4 c = reinterpret_cast<const uint4 *>(hist[w][i])[lane];
                const uint32_t sum = c.x + c.y + c.z + c.w;
                const uint32_t ex = wave_inclusive_scan(sum) - sum;
                reinterpret_cast<uint4 *>(hist[w][i])[lane] = make_uint4(ex, ex + c.x, ex + c.x + c.y, ex + c.x + c.y + c.z);
            }
            fence();
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                const uint32_t at = (uint32_t)(i * WAVE) + pos[i] + hist[w][i][__builtin_amdgcn_ubfe(key[i], sh, bw)];
                stage_k[w][at] = key[i];
                if (HAS_VALUES) stage_v[w][at] = val[i];
            }
            fence();
#pragma unroll
            for (int i = 0; i < NS; ++i) {
                key[i] = stage_k[w][i * WAVE + lane];
                if (HAS_VALUES) val[i] = stage_v[w][i * WAVE + lane];
            }
            fence();
        }
#pragma unroll
        for (int i = 0; i < NS; ++i) {
            if ((uint32_t)lane < size[i]) {
                dst_k[off[i] + lane] = twiddle_out(key[i], f32_out, xor_out);
                if (HAS_VALUES) dst_v[off[i] + lane] = val[i];
            }
        }
    }
}

// ================================================================ wide MSB ==
// rdxsrt_unstable_sort for 64-bit keys and / or 64-bit values: the reference instantiates its hybrid sort for 8-byte keys
// and values too (RadixSortConfig<8,*>, msb/src/sort/gpu_sort_config.h:179-198; msb/tests/test_sort_keys.cu:154-195,
// test_sort_pairs.cu:223-281).  Same structure as the 32-bit path above -- top byte with one stable pass (the wide LSB
// pass of gs_wide.hip), then per level: expand / upsweep / scan / classify / scatter on the bucket lists, buckets that fit a
// workgroup finished by an LSD local sort in LDS on their remaining bits -- with kernels of their own for the element
// types: tiles of 4096 elements (8 per thread), keys and values staged one after the other through one LDS buffer, local
// sorts of up to 2048 / 8192 elements.  expand, scan and classify are the kernels above (geometry from MsbWs).  Keys stay in
// the caller's representation in memory; the order-preserving transform is applied where a digit is taken.
constexpr int MW_THREADS = 512, MW_WAVES = MW_THREADS / WAVE, MW_KPT = 8, MW_TILE = MW_THREADS * MW_KPT;
constexpr uint32_t MW_CAP = 8192;                 // largest local sort (elements)
struct MwNoVal {};

template <typename K> __device__ __forceinline__ K mw_tw_in(K k, int f, uint64_t x)
{
    if constexpr (sizeof(K) == 8) { if (f) k ^= (uint64_t)((int64_t)k >> 63) | 0x8000000000000000ull; return k ^ x; }
    else return twiddle_in((uint32_t)k, f, (uint32_t)x);
}
template <typename K> __device__ __forceinline__ K mw_tw_out(K k, int f, uint64_t x)
{
    if constexpr (sizeof(K) == 8) { k ^= x; if (f) k ^= ~(uint64_t)((int64_t)k >> 63) | 0x8000000000000000ull; return k; }
    else return twiddle_out((uint32_t)k, f, (uint32_t)x);
}

template <typename K>
__global__ __launch_bounds__(MW_THREADS) void mw_upsweep_kernel(MsbWs ws, int L, const K *__restrict__ src, uint32_t shift, int f_in,
                                                                uint64_t xor_in, uint32_t mask = 0xffu)
{
    __shared__ uint32_t lh[MW_WAVES][RADIX];
    uint32_t ntiles = (uint32_t)ws.level[L].packed;
    if (ntiles > ws.max_tiles - MW_WAVES) ntiles = ws.max_tiles - MW_WAVES;
    const uint32_t nchunks = ntiles / MW_WAVES + 1;
    const int tid = threadIdx.x, w = wave_id(), lane = lane_id();
    uint32_t *my = lh[w];
    constexpr int BATCH = 16;
    for (uint32_t c = blockIdx.x; c < nchunks; c += gridDim.x) {
        for (int i = lane; i < RADIX; i += WAVE) my[i] = 0;
        const uint32_t g = c * MW_WAVES + (uint32_t)w;
        if (g < ntiles) {
            const MsbTile T = ws.tiles[g];
            const K *p = src + T.lo;
            const uint32_t last = T.valid - 1u;
#pragma unroll 1
            for (uint32_t j = 0; j < T.valid; j += BATCH * WAVE) {
                K v[BATCH];
#pragma unroll
                for (int u = 0; u < BATCH; ++u) {
                    const uint32_t idx = j + u * WAVE + lane;
                    v[u] = __builtin_nontemporal_load(&p[idx < last ? idx : last]);
                }
#pragma unroll
                for (int u = 0; u < BATCH; ++u)
                    if (j + u * WAVE + lane < T.valid) hist_add(my, (uint32_t)(mw_tw_in<K>(v[u], f_in, xor_in) >> shift) & mask);
            }
        }
        __syncthreads();
        if (tid < RADIX) {
            uint32_t run = 0;
#pragma unroll
            for (int j = 0; j < MW_WAVES; ++j) {
                ws.prefix16[(size_t)(c * MW_WAVES + j) * RADIX + tid] = (uint16_t)run;
                run += lh[j][tid];
            }
            ws.spine[(size_t)tid * ws.stride + c] = run;
        }
        __syncthreads();
    }
}

// ranks of the wave's KPT rounds by ballot match + wave-private counters (as in the scatter above); `dig(i)` = digit of round i
template <int KPT, typename F>
__device__ __forceinline__ void mw_rank(uint32_t *my, uint32_t (&pos)[KPT], F dig)
{
#pragma unroll
    for (int i = 0; i < KPT; ++i) {
        const uint32_t d = dig(i);
        uint32_t lo, hi;
        match_digit(d, lo, hi);
        const uint32_t lower = count_lower(lo, hi);
        pos[i] = my[d] + lower;
        if (lower == 0)
            __hip_atomic_fetch_add(&my[d], (uint32_t)(__popc(lo) + __popc(hi)), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WAVEFRONT);
    }
#pragma unroll
    for (int i = 0; i < KPT; ++i) asm volatile("" : "+v"(pos[i]));
}
// wave 0: the WAVES rows of wave-private counts -> element bases per (wave, digit) in place; returns nothing, writes
// ex4 (exclusive prefix of the block's digit totals, digits 4l..4l+3 of lane l) for the caller
template <int WAVES>
__device__ __forceinline__ void mw_scan_rows(uint32_t (*whist)[RADIX], uint32_t (&ex)[4])
{
    const int lane = lane_id();
    uint32_t run[4] = {0, 0, 0, 0};
#pragma unroll
    for (int j = 0; j < WAVES; ++j) {
        const uint4 x = reinterpret_cast<const uint4 *>(whist[j])[lane];
        run[0] += x.x; run[1] += x.y; run[2] += x.z; run[3] += x.w;
    }
    const uint32_t lane_sum = run[0] + run[1] + run[2] + run[3];
    ex[0] = wave_inclusive_scan(lane_sum) - lane_sum;
    ex[1] = ex[0] + run[0];
    ex[2] = ex[1] + run[1];
    ex[3] = ex[2] + run[2];
    uint4 e4 = make_uint4(ex[0], ex[1], ex[2], ex[3]);
#pragma unroll
    for (int j = 0; j < WAVES; ++j) {
        const uint4 x = reinterpret_cast<const uint4 *>(whist[j])[lane];
        reinterpret_cast<uint4 *>(whist[j])[lane] = e4;
        e4.x += x.x; e4.y += x.y; e4.z += x.z; e4.w += x.w;
    }
}

// one level tile: stable counting-sort scatter on byte `shift / 8` (ragged tiles included: pads rank last and are not stored)
template <typename K, typename V>
__global__ __launch_bounds__(MW_THREADS, 4) void mw_scatter_kernel(MsbWs ws, int L, const K *__restrict__ src_k, K *__restrict__ dst_k,
                                                                   const V *__restrict__ src_v, V *__restrict__ dst_v, uint32_t shift,
                                                                   int f_in, uint64_t xor_in, uint32_t mask = 0xffu)
{
    constexpr bool HAS_VALUES = !std::is_same<V, MwNoVal>::value;
    constexpr size_t ELEM = sizeof(K) > (HAS_VALUES ? sizeof(V) : 1) ? sizeof(K) : sizeof(V);
    __shared__ __attribute__((aligned(16))) uint32_t whist[MW_WAVES][RADIX];
    __shared__ __attribute__((aligned(16))) uint32_t gbase[RADIX];
    __shared__ __attribute__((aligned(16))) unsigned char stage_raw[MW_TILE * ELEM];
    K *stage_k = reinterpret_cast<K *>(stage_raw);
    const uint32_t ntiles = (uint32_t)ws.level[L].packed;
    if (blockIdx.x >= ntiles || blockIdx.x >= ws.max_tiles) return;
    const uint32_t g = tile_of_item(blockIdx.x, ntiles);
    const MsbTile T = ws.tiles[g];
    const int lane = lane_id(), w = wave_id();
    uint32_t *my = whist[w];
    const uint32_t wbase = (uint32_t)w * (WAVE * MW_KPT) + lane, valid = T.valid;
    uint32_t tbase[4] = {0, 0, 0, 0};
    if (w == 0) {
        const uint4 cur = reinterpret_cast<const uint4 *>(ws.cursors + (size_t)T.bucket * RADIX)[lane];
        const uint32_t *sp = ws.spine + g / MW_WAVES + (size_t)(4 * lane) * ws.stride;
        const uint2 pf = reinterpret_cast<const uint2 *>(ws.prefix16 + (size_t)g * RADIX)[lane];
        tbase[0] = cur.x + sp[0] + (pf.x & 0xffffu);
        tbase[1] = cur.y + sp[ws.stride] + (pf.x >> 16);
        tbase[2] = cur.z + sp[2 * (size_t)ws.stride] + (pf.y & 0xffffu);
        tbase[3] = cur.w + sp[3 * (size_t)ws.stride] + (pf.y >> 16);
    }
    K key[MW_KPT];
    uint32_t pos[MW_KPT];
    const K *pk = src_k + T.lo;
#pragma unroll
    for (int i = 0; i < MW_KPT; ++i) {
        const uint32_t idx = wbase + i * WAVE;
        key[i] = pk[idx < valid ? idx : valid - 1u];
    }
    auto digit_of = [&](K k) { return (uint32_t)(mw_tw_in<K>(k, f_in, xor_in) >> shift) & mask; };
#pragma unroll
    for (int i = lane; i < RADIX; i += WAVE) my[i] = 0;
    mw_rank<MW_KPT>(my, pos, [&](int i) { return (wbase + i * WAVE < valid) ? digit_of(key[i]) : 255u; });
    __syncthreads();
    if (w == 0) {
        uint32_t ex[4];
        mw_scan_rows<MW_WAVES>(whist, ex);
        reinterpret_cast<uint4 *>(gbase)[lane] = make_uint4(tbase[0] - ex[0], tbase[1] - ex[1], tbase[2] - ex[2], tbase[3] - ex[3]);
    }
    __syncthreads();
#pragma unroll
    for (int i = 0; i < MW_KPT; ++i) {
        pos[i] += my[(wbase + i * WAVE < valid) ? digit_of(key[i]) : 255u];
        stage_k[pos[i]] = key[i];
    }
    __syncthreads();
    uint32_t dst[MW_KPT];
#pragma unroll
    for (int i = 0; i < MW_KPT; ++i) {
        const uint32_t slot = wbase + i * WAVE;          // wave-contiguous slots (see lsb_downsweep_kernel)
        const K k = stage_k[slot];
        dst[i] = gbase[digit_of(k)] + slot;
        if (slot < valid) dst_k[dst[i]] = k;
    }
    if constexpr (HAS_VALUES) {
        V *stage_v = reinterpret_cast<V *>(stage_raw);
        V val[MW_KPT];
        const V *pv = src_v + T.lo;
#pragma unroll
        for (int i = 0; i < MW_KPT; ++i) {
            const uint32_t idx = wbase + i * WAVE;
            val[i] = pv[idx < valid ? idx : valid - 1u];
        }
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MW_KPT; ++i) stage_v[pos[i]] = val[i];
        __syncthreads();
#pragma unroll
        for (int i = 0; i < MW_KPT; ++i) {
            const uint32_t slot = wbase + i * WAVE;
            if (slot < valid) dst_v[dst[i]] = stage_v[slot];
        }
    }
}

// local sort of one class: LSD passes of 8 bits over the task's low `sort_bits` bits, all stable (ballot match), keys
// (and values) exchanged through LDS after every pass; pads are all-ones keys, which every pass ranks last.
// `unstable_ok` (the MSB sort; round 3), 64-bit keys with more than 16 bits left: ONE order-free counting pass on the task's top 11
// bits (2048 bins for <= 8192 keys: a handful of keys per bin on anything like uniform keys), then every thread finishes four
// neighbouring bins by insertion on the full key -- two trips through LDS instead of six LSD passes for the 48 bits a level-1 task
// of a 64-bit sort has left (2^28 uniform u64 keys: local sorts 5.5 -> see DESIGN.md).  With values the element's index rides in
// the key's top 16 bits (which every key of a task shares: <= 48 bits to sort) and the values are fetched through it at the end.
// A bin of MW_BIN_LIMIT or more keys (skew): the task falls back to the LSD passes.
constexpr uint32_t MW_BINS = 2048, MW_BIN_BITS = 11, MW_BIN_LIMIT = 24;
template <typename K, typename V, int KPT>
__global__ __launch_bounds__(MW_THREADS, KPT > 8 ? 2 : 4) void mw_local_sort_kernel(MsbWs ws, int L, int cls, const K *__restrict__ src_k,
                                                                                    K *__restrict__ dst_k, const V *__restrict__ src_v,
                                                                                    V *__restrict__ dst_v, int f, uint64_t x, int unstable_ok = 0)
{
    constexpr bool HAS_VALUES = !std::is_same<V, MwNoVal>::value;
    constexpr size_t ELEM = sizeof(K) > (HAS_VALUES ? sizeof(V) : 1) ? sizeof(K) : sizeof(V);
    constexpr int CAP = MW_THREADS * KPT;
    __shared__ __attribute__((aligned(16))) uint32_t whist[MW_WAVES][RADIX];
    __shared__ __attribute__((aligned(16))) unsigned char stage_raw[CAP * ELEM];
    __shared__ uint32_t mw_wtot[MW_WAVES];
    __shared__ K mw_prefix;
    static_assert(MW_WAVES * RADIX == MW_BINS && MW_BINS == 4 * MW_THREADS, "the bins of the order-free pass live in the wave counters");
    K *stage_k = reinterpret_cast<K *>(stage_raw);
    V *stage_v = reinterpret_cast<V *>(stage_raw);
    uint32_t ntasks = ws.level[L].task_count[cls];
    if (ntasks > ws.max_tasks) ntasks = ws.max_tasks;
    const int lane = lane_id(), w = wave_id();
    uint32_t *my = whist[w];
    const uint32_t wbase = (uint32_t)w * (WAVE * KPT) + lane;
    for (uint32_t t = blockIdx.x; t < ntasks; t += gridDim.x) {
        const MsbTask T = ws.tasks[cls][t];
        const uint32_t size = T.size < (uint32_t)CAP ? T.size : (uint32_t)CAP, last = size - 1u;
        K key[KPT];
        V val[HAS_VALUES ? KPT : 1];
        uint32_t pos[KPT];
        // the order-free plan (below) deals the elements out round-robin over the workgroup, so that a task smaller than the class
        // keeps every wave busy (wave-contiguous: a 4096-key task of the 8192 class left four of the eight waves idle)
        const bool fast = sizeof(K) == 8 && unstable_ok && T.pad == 0u && T.sort_bits > 16u && (!HAS_VALUES || T.sort_bits <= 48u);   // uniform
        const uint32_t ibase = fast ? (uint32_t)threadIdx.x : wbase, istep = fast ? (uint32_t)MW_THREADS : (uint32_t)WAVE;
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t idx = ibase + i * istep;
            const K k = mw_tw_in<K>(src_k[T.offset + (idx < size ? idx : last)], f, x);
            key[i] = idx < size ? k : (K) ~(K)0;
            if constexpr (HAS_VALUES) val[i] = src_v[T.offset + (idx < size ? idx : last)];
        }
        bool finished = false;
        if constexpr (sizeof(K) == 8) {
            if (fast) {
                constexpr K LOW48 = (K)0x0000ffffffffffffull;
                uint32_t *hist = &whist[0][0];
                const int tid = (int)threadIdx.x;
                const uint32_t bshift = T.sort_bits - MW_BIN_BITS;
                reinterpret_cast<uint4 *>(hist)[tid] = make_uint4(0u, 0u, 0u, 0u);
                if (tid == 0) mw_prefix = key[0];          // element 0 is never a pad
                __syncthreads();
                uint32_t over = 0;
#pragma unroll
                for (int i = 0; i < KPT; ++i) {
                    pos[i] = 0;
                    if (ibase + i * istep < size) {
                        pos[i] = atomicAdd(&hist[(uint32_t)(key[i] >> bshift) & (MW_BINS - 1u)], 1u);
                        over |= pos[i] >= MW_BIN_LIMIT - 1u ? 1u : 0u;
                    }
                }
                if (!__syncthreads_or((int)over)) {
                    // exclusive scan of the bins in place: four neighbouring bins per thread
                    const uint4 c4 = reinterpret_cast<const uint4 *>(hist)[tid];
                    const uint32_t sum = c4.x + c4.y + c4.z + c4.w, inc = wave_inclusive_scan(sum);
                    if (lane == 63) mw_wtot[w] = inc;
                    __syncthreads();
                    uint32_t run = inc - sum;
#pragma unroll
                    for (int j = 0; j < MW_WAVES; ++j) run += j < w ? mw_wtot[j] : 0u;
                    const uint32_t b0 = run, b1 = b0 + c4.x, b2 = b1 + c4.y, b3 = b2 + c4.z;
                    reinterpret_cast<uint4 *>(hist)[tid] = make_uint4(b0, b1, b2, b3);
                    __syncthreads();
                    // into the buffer, bin by bin; with values an element is {low 48 key bits : 48, its index : 16}, so elements are
                    // distinct and their unsigned order is the order of the keys
                    K elem[KPT];
#pragma unroll
                    for (int i = 0; i < KPT; ++i) {
                        const uint32_t idx = ibase + i * istep;
                        elem[i] = HAS_VALUES ? (K)(((key[i] & LOW48) << 16) | (K)idx) : key[i];
                        if (idx < size) {
                            pos[i] += hist[(uint32_t)(key[i] >> bshift) & (MW_BINS - 1u)];
                            stage_k[pos[i]] = elem[i];
                        }
                    }
                    __syncthreads();
                    // rank inside the bin by counting: every key reads its bin (a handful of elements) -- a thread-per-bin insertion
                    // sort was a chain of dependent LDS round trips (3.0 of the kernel's 4.6 ms at 2^28 uniform u64 keys; counting:
                    // 2.1 of 3.7 ms, and 1.0 of 2.6 ms once the elements are dealt out round-robin); all keys of a thread advancing
                    // in lockstep rounds instead of key by key: slower (4.2 against 3.75 ms).  Equal keys (no values) are
                    // ordered by their slot
                    uint32_t fin[KPT];
#pragma unroll
                    for (int i = 0; i < KPT; ++i) {
                        fin[i] = 0;
                        if (ibase + i * istep < size) {
                            const K kk = HAS_VALUES ? (K)(elem[i] >> 16) : elem[i];
                            const uint32_t bin = (uint32_t)(kk >> bshift) & (MW_BINS - 1u);
                            const uint32_t lo = hist[bin], hi = bin == MW_BINS - 1u ? size : hist[bin + 1u];
                            uint32_t c = lo;
                            for (uint32_t q = lo; q < hi; ++q) {
                                const K e = stage_k[q];
                                c += (e < elem[i] || (!HAS_VALUES && e == elem[i] && q < pos[i])) ? 1u : 0u;
                            }
                            fin[i] = c;
                        }
                    }
                    __syncthreads();                       // every bin has been read: the elements move to their final slots
#pragma unroll
                    for (int i = 0; i < KPT; ++i)
                        if (ibase + i * istep < size) stage_k[fin[i]] = elem[i];
                    __syncthreads();
                    const K prefix = mw_prefix & ~LOW48;
#pragma unroll
                    for (int i = 0; i < KPT; ++i) {
                        const uint32_t slot = ibase + i * istep;
                        if (slot < size) {
                            K k = stage_k[slot];
                            if constexpr (HAS_VALUES) { pos[i] = (uint32_t)k & 0xffffu; k = (K)((k >> 16) | prefix); }
                            dst_k[T.offset + slot] = mw_tw_out<K>(k, f, x);
                        }
                    }
                    if constexpr (HAS_VALUES) {
                        __syncthreads();                   // every key has been read: the buffer takes the values, by original index
#pragma unroll
                        for (int i = 0; i < KPT; ++i) stage_v[ibase + i * istep] = val[i];
                        __syncthreads();
#pragma unroll
                        for (int i = 0; i < KPT; ++i) {
                            const uint32_t slot = ibase + i * istep;
                            if (slot < size) dst_v[T.offset + slot] = stage_v[pos[i]];
                        }
                    }
                    finished = true;
                }
                __syncthreads();
            }
        }
        if (finished) continue;
        if (fast) {
            // the attempt was abandoned (a crowded bin): the LSD passes need the wave-contiguous order -- a pad (all ones) and a real
            // key whose sorted bits are all ones are told apart only by the pads coming LAST in the order the stable passes keep
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                const uint32_t idx = wbase + i * WAVE;
                const K k = mw_tw_in<K>(src_k[T.offset + (idx < size ? idx : last)], f, x);
                key[i] = idx < size ? k : (K) ~(K)0;
                if constexpr (HAS_VALUES) val[i] = src_v[T.offset + (idx < size ? idx : last)];
            }
        }
        // the task's bits start at bit `pad` of the key (0 in the MSB sort, begin_bit in a segmented sort); the last digit
        // may be narrower than 8 bits.  Pads are all-ones keys: the widest digit value in every pass, so they rank last.
#pragma unroll 1
        for (uint32_t done = 0; done < T.sort_bits; done += 8) {
            const uint32_t shift = T.pad + done, dm = (T.sort_bits - done < 8u) ? ((1u << (T.sort_bits - done)) - 1u) : 0xffu;
#pragma unroll
            for (int i = lane; i < RADIX; i += WAVE) my[i] = 0;
            mw_rank<KPT>(my, pos, [&](int i) { return (uint32_t)(key[i] >> shift) & dm; });
            __syncthreads();
            if (w == 0) { uint32_t ex[4]; mw_scan_rows<MW_WAVES>(whist, ex); }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < KPT; ++i) {
                pos[i] += my[(uint32_t)(key[i] >> shift) & dm];
                stage_k[pos[i]] = key[i];
            }
            __syncthreads();
#pragma unroll
            for (int i = 0; i < KPT; ++i) key[i] = stage_k[wbase + i * WAVE];
            if constexpr (HAS_VALUES) {
                __syncthreads();
#pragma unroll
                for (int i = 0; i < KPT; ++i) stage_v[pos[i]] = val[i];
                __syncthreads();
#pragma unroll
                for (int i = 0; i < KPT; ++i) val[i] = stage_v[wbase + i * WAVE];
            }
            __syncthreads();
        }
#pragma unroll
        for (int i = 0; i < KPT; ++i) {
            const uint32_t idx = wbase + i * WAVE;
            if (idx < size) {
                dst_k[T.offset + idx] = mw_tw_out<K>(key[i], f, x);
                if constexpr (HAS_VALUES) dst_v[T.offset + idx] = val[i];
            }
        }
        __syncthreads();
    }
}

__global__ void mw_single_task_kernel(MsbWs ws, uint32_t n, int cls, uint32_t bits)
{
    if (threadIdx.x == 0) { ws.tasks[cls][0] = MsbTask{0u, n, bits, 0u}; ws.level[0].task_count[cls] = 1; }
}

template <typename K, typename V>
static void mw_launch_local_sorts(const MsbWs &ws, int L, const K *sk, K *dk, const V *sv, V *dv, int f, uint64_t x, hipStream_t s,
                                  bool unstable_ok = false)
{
    KernelTimer kt(GS_K_MSB_LOCAL_SORT, s);
    const uint32_t g = ws.max_tasks < MSB_MAX_GRID ? ws.max_tasks : MSB_MAX_GRID;
    static const bool fast_on = [] { const char *e = getenv("GS_MSB_WIDE_FAST"); return !(e && e[0] == '0'); }();   // A/B switch
    const int u = unstable_ok && fast_on ? 1 : 0;
    hipLaunchKernelGGL((mw_local_sort_kernel<K, V, 4>), dim3(g), dim3(MW_THREADS), 0, s, ws, L, 0, sk, dk, sv, dv, f, x, u);
    hipLaunchKernelGGL((mw_local_sort_kernel<K, V, 16>), dim3(g), dim3(MW_THREADS), 0, s, ws, L, 1, sk, dk, sv, dv, f, x, u);
}

static size_t mw_lsb_bytes(uint64_t n, int kb, int vb) { return align256(gs_lsb_wide_temp_bytes(n, kb, vb)); }

template <typename K, typename V>
static int msb_wide_sort(void *d_temp, K *keys, V *vals, uint64_t num_items, K *keys_alt, V *vals_alt, int key_type, hipStream_t s)
{
    constexpr bool pairs = !std::is_same<V, MwNoVal>::value;
    constexpr int KB = (int)sizeof(K), VB = pairs ? (int)sizeof(V) : 0, key_bits = 8 * KB, nclass = 2;
    const uint32_t n = (uint32_t)num_items;
    const MsbWs ws = msb_carve((char *)d_temp + mw_lsb_bytes(num_items, KB, VB), num_items, pairs, 0, 0, MW_CAP, (uint32_t)key_bits);
    const bool is_float = key_type == GS_KEY_F32 || key_type == GS_KEY_F64;
    const bool is_signed = key_type == GS_KEY_I32 || key_type == GS_KEY_I64;
    const int f = is_float ? 1 : 0;
    const uint64_t x = is_signed ? (KB == 8 ? 0x8000000000000000ull : 0x80000000ull) : 0ull;
    { KernelTimer kt(GS_K_OTHER, s); hipLaunchKernelGGL(msb_init_kernel, dim3(1), dim3(64), 0, s, ws, n); }
    if (n <= MW_CAP) {
        hipLaunchKernelGGL(mw_single_task_kernel, dim3(1), dim3(64), 0, s, ws, n, n <= 2048u ? 0 : 1, (uint32_t)key_bits);
        mw_launch_local_sorts<K, V>(ws, 0, keys, keys, vals, vals, f, x, s, true);
        return (int)hipGetLastError();
    }
    // level 0: the top byte with one stable wide LSB pass, IN -> ALT (keys keep the caller's representation)
    void *k2[2] = {keys, keys_alt}, *v2[2] = {(void *)vals, (void *)vals_alt};
    int sel = 0;
    int e = gs_lsb_sort_wide(d_temp, mw_lsb_bytes(num_items, KB, VB), k2, pairs ? v2 : nullptr, &sel, num_items, KB, VB, key_bits - 8,
                             key_bits, 0, key_type, s);
    if (e) return e;
    { KernelTimer kt(GS_K_MSB_CLASSIFY, s);
      hipLaunchKernelGGL((msb_classify_kernel<false, false>), dim3(1), dim3(256), 0, s, ws, 0, wide_totals_ptr(d_temp, num_items), nclass); }
    mw_launch_local_sorts<K, V>(ws, 0, keys_alt, keys, vals_alt, vals, f, x, s, true);
    K *buf_k[2] = {keys, keys_alt};
    V *buf_v[2] = {vals, vals_alt};
    const uint32_t tiles_all = (uint32_t)((num_items + MW_TILE - 1) / MW_TILE);
    MsbPeek *peek = msb_peek_get(s);           // a 64-bit key has 7 levels below the first; most of them are empty
    uint32_t known_b = 0, known_tiles = 0;
    bool known = false;
    for (int L = 1; L < KB; ++L) {
        if (peek && peek->armed) {             // see msb_run_levels
            peek->armed = false;
            if (hipEventSynchronize(peek->ev) == hipSuccess) {
                const unsigned long long pkd = *(volatile unsigned long long *)peek->host;
                known_b = (uint32_t)(pkd >> 32); known_tiles = (uint32_t)pkd; known = true;
                if (known_b == 0) break;
            } else {
                (void)hipGetLastError();
                known = false;
            }
        }
        const uint32_t shift = (uint32_t)(key_bits - 8 - 8 * L);
        const K *sk = buf_k[L & 1];
        K *dk = buf_k[(L + 1) & 1];
        const V *sv = buf_v[L & 1];
        V *dv = buf_v[(L + 1) & 1];
        const bool last = L == KB - 1;
        uint32_t max_b = (L == 1) ? (uint32_t)RADIX : ws.max_buckets;
        uint32_t max_tiles = tiles_all + max_b + tiles_all / MSB_ALIGN_MIN_TILES + 1u;
        if (known && L >= 2) {
            if (known_b < max_b) max_b = known_b;
            if (known_tiles < max_tiles) max_tiles = known_tiles;
            known = false;
        }
        { KernelTimer kt(GS_K_MSB_HISTOGRAM, s);
          hipLaunchKernelGGL(msb_expand_kernel, dim3(max_b < 4096u ? max_b : 4096u), dim3(256), 0, s, ws, L, (const uint32_t *)nullptr);
          const uint32_t hg_ub = max_tiles / MW_WAVES + 1;
          hipLaunchKernelGGL((mw_upsweep_kernel<K>), dim3(hg_ub < MSB_MAX_GRID ? hg_ub : MSB_MAX_GRID), dim3(MW_THREADS), 0, s, ws, L, sk,
                             shift, f, x);
          hipLaunchKernelGGL(msb_scan_kernel, dim3(RADIX), dim3(1024), 0, s, ws, L); }
        { KernelTimer kt(GS_K_MSB_CLASSIFY, s);
          const uint32_t cg = max_b < 4096u ? max_b : 4096u;
          if (last) hipLaunchKernelGGL((msb_classify_kernel<true, false>), dim3(cg), dim3(256), 0, s, ws, L, (const uint32_t *)nullptr, nclass);
          else hipLaunchKernelGGL((msb_classify_kernel<false, false>), dim3(cg), dim3(256), 0, s, ws, L, (const uint32_t *)nullptr, nclass); }
        if (peek && !last) {
            hipLaunchKernelGGL(msb_peek_kernel, dim3(1), dim3(1), 0, s, ws, L + 1, peek->dev);
            peek->armed = hipEventRecord(peek->ev, s) == hipSuccess;
            if (!peek->armed) (void)hipGetLastError();
        }
        { KernelTimer kt(GS_K_MSB_PARTITION, s);
          hipLaunchKernelGGL((mw_scatter_kernel<K, V>), dim3(max_tiles), dim3(MW_THREADS), 0, s, ws, L, sk, dk, sv, dv, shift, f, x); }
        if (!last) mw_launch_local_sorts<K, V>(ws, L, (const K *)dk, buf_k[0], (const V *)dv, buf_v[0], f, x, s, true);
    }
    return (int)hipGetLastError();
}

// cub::DeviceSegmentedRadixSort for the wide element types (dispatch_radix_sort.cuh:321-432 is type-generic): the structure
// of gs_segmented_sort_u32 with the wide kernel set -- segments of <= 8192 elements become stable local-sort tasks, the larger
// ones the buckets of one level that is partitioned once per 8-bit digit from begin_bit up (keys stay in the caller's
// representation, so there is no first / last pass distinction).
template <typename K, typename V>
static int seg_wide_sort(void *d_temp, void *d_keys[2], void *d_vals[2], int *selector, uint64_t num_items, uint32_t num_segments,
                         const int32_t *d_begin_offsets, const int32_t *d_end_offsets, int begin_bit, int end_bit, int descending,
                         int key_type, hipStream_t s)
{
    constexpr bool pairs = !std::is_same<V, MwNoVal>::value;
    constexpr int KB = (int)sizeof(K), nclass = 2;
    const MsbWs ws = msb_carve(d_temp, num_items, pairs, 0, num_segments, MW_CAP, (uint32_t)(8 * KB));
    const int num_bits = end_bit - begin_bit, passes = (num_bits + RADIX_BITS - 1) / RADIX_BITS;
    const int sel = *selector, fin = sel ^ (passes & 1);
    const bool is_float = key_type == GS_KEY_F32 || key_type == GS_KEY_F64;
    const bool is_signed = key_type == GS_KEY_I32 || key_type == GS_KEY_I64;
    const int f = is_float ? 1 : 0;
    const uint64_t ones = KB == 8 ? ~0ull : 0xffffffffull;
    const uint64_t x = (is_signed ? (KB == 8 ? 0x8000000000000000ull : 0x80000000ull) : 0ull) ^ (descending ? ones : 0ull);
    hipError_t e = zero_async(ws.level, MSB_LEVELS * sizeof(MsbLevel), s);
    if (e != hipSuccess) return (int)e;
    { KernelTimer kt(GS_K_MSB_CLASSIFY, s);
      const uint32_t g = (num_segments + 1023u) / 1024u;
      hipLaunchKernelGGL(seg_classify_kernel, dim3(g < 4096u ? g : 4096u), dim3(256), 0, s, ws, d_begin_offsets, d_end_offsets,
                         num_segments, nclass, (uint32_t)num_bits, (uint32_t)begin_bit, (uint32_t)num_items); }
    mw_launch_local_sorts<K, V>(ws, 1, (const K *)d_keys[sel], (K *)d_keys[fin], pairs ? (const V *)d_vals[sel] : nullptr,
                                pairs ? (V *)d_vals[fin] : nullptr, f, x, s);
    const uint32_t max_b = ws.max_buckets;
    const uint32_t tiles_ub = (uint32_t)(num_items / MW_TILE) + max_b;
    { KernelTimer kt(GS_K_MSB_HISTOGRAM, s);
      hipLaunchKernelGGL(msb_expand_kernel, dim3(max_b < 4096u ? max_b : 4096u), dim3(256), 0, s, ws, 1, (const uint32_t *)nullptr); }
    for (int p = 0; p < passes; ++p) {
        const uint32_t shift = (uint32_t)(begin_bit + p * RADIX_BITS);
        const int bits = (end_bit - (int)shift < RADIX_BITS) ? end_bit - (int)shift : RADIX_BITS;
        const uint32_t mask = (1u << bits) - 1u;
        const K *sk = (const K *)d_keys[sel ^ (p & 1)];
        K *dk = (K *)d_keys[sel ^ ((p + 1) & 1)];
        const V *sv = pairs ? (const V *)d_vals[sel ^ (p & 1)] : nullptr;
        V *dv = pairs ? (V *)d_vals[sel ^ ((p + 1) & 1)] : nullptr;
        { KernelTimer kt(GS_K_MSB_HISTOGRAM, s);
          const uint32_t hg_ub = tiles_ub / MW_WAVES + 1;
          hipLaunchKernelGGL((mw_upsweep_kernel<K>), dim3(hg_ub < MSB_MAX_GRID ? hg_ub : MSB_MAX_GRID), dim3(MW_THREADS), 0, s, ws, 1, sk,
                             shift, f, x, mask);
          hipLaunchKernelGGL(msb_scan_kernel, dim3(RADIX), dim3(1024), 0, s, ws, 1); }
        { KernelTimer kt(GS_K_MSB_CLASSIFY, s);
          hipLaunchKernelGGL((msb_classify_kernel<true, false>), dim3(max_b < 4096u ? max_b : 4096u), dim3(256), 0, s, ws, 1,
                             (const uint32_t *)nullptr, nclass); }
        { KernelTimer kt(GS_K_MSB_PARTITION, s);
          hipLaunchKernelGGL((mw_scatter_kernel<K, V>), dim3(tiles_ub), dim3(MW_THREADS), 0, s, ws, 1, sk, dk, sv, dv, shift, f, x, mask); }
    }
    *selector = fin;
    return (int)hipGetLastError();
}

}  // namespace gs

using namespace gs;

extern "C" {

size_t gs_msb_temp_bytes(uint64_t num_items, int has_values)
{
    return align256(lsb_temp_bytes(num_items)) + msb_ws_bytes(num_items, has_values != 0);
}

// A synchronous sort ends by reading the sort's overflow word (MsbLevel::overflow): a clamped device-side append means a
// record was dropped and the result is wrong -- reported as hipErrorUnknown instead of a silent hipSuccess.  Asynchronous
// callers read the same word through gs_msb_census (`overflow`).
static int msb_sync_and_check(const MsbWs &ws, hipStream_t s)
{
    uint32_t ovf = 0;
    hipError_t e = hipMemcpyAsync(&ovf, &ws.level[0].overflow, sizeof(ovf), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return (int)e;
    return ovf ? (int)hipErrorUnknown : 0;
}

static int msb_sort_impl(void *d_temp, size_t temp_bytes, uint32_t *d_keys, uint32_t *d_vals, uint64_t num_items,
                         uint32_t *d_keys_alt, uint32_t *d_vals_alt, uint32_t **d_sorted_keys, uint32_t **d_sorted_vals,
                         int key_type, void *stream, int synchronize, int stop_level, bool allow_pivot)
{
    GS_CLEAR_STALE_ERROR();
    if (num_items >= (1ull << 32)) return hipErrorInvalidValue;
    if (key_type < GS_KEY_U32 || key_type > GS_KEY_F32) return hipErrorInvalidValue;
    if (d_sorted_keys) *d_sorted_keys = d_keys;       // 32-bit keys: result in the input arrays
    if (d_sorted_vals) *d_sorted_vals = d_vals;
    if (num_items == 0) return hipSuccess;
    const bool pairs = d_vals != nullptr;
    if (!d_keys || !d_keys_alt || (pairs && !d_vals_alt)) return hipErrorInvalidValue;
    if (!d_temp || temp_bytes < gs_msb_temp_bytes(num_items, pairs)) return hipErrorInvalidValue;

    hipStream_t s = (hipStream_t)stream;
    const uint32_t n = (uint32_t)num_items;
    const int nclass = msb_num_classes(pairs);
    const MsbWs ws = msb_carve((char *)d_temp + align256(lsb_temp_bytes(num_items)), num_items, pairs);
    const LsbWorkspace lw = lsb_carve(d_temp, num_items);

    // key twiddles: the first reader maps in, every final writer maps out
    PassParams tw{};
    lsb_twiddle_masks(key_type, 0, true, true, tw);

    { KernelTimer kt(GS_K_OTHER, s); hipLaunchKernelGGL(msb_init_kernel, dim3(1), dim3(64), 0, s, ws, n); }

    if (n <= msb_class_cap(nclass - 1)) {
        // fits one workgroup: one local sort on all 32 bits, in place
        int cls = 0;
        while (msb_class_cap(cls) < n) ++cls;
        hipLaunchKernelGGL(msb_single_task_kernel, dim3(1), dim3(64), 0, s, ws, n, cls);
        if (pairs) launch_local_sorts<true>(ws, 0, 1, d_keys, d_keys, d_vals, d_vals, tw.f32_in, tw.xor_in, tw.f32_out, tw.xor_out, s, 32);
        else launch_local_sorts<false>(ws, 0, 1, d_keys, d_keys, nullptr, nullptr, tw.f32_in, tw.xor_in, tw.f32_out, tw.xor_out, s, 32);
    } else {
        uint32_t *buf_k[2] = {d_keys, d_keys_alt};
        uint32_t *buf_v[2] = {d_vals, d_vals_alt};
        // level 0: the top byte with one stable LSB pass, IN -> ALT (keys stay twiddled)
        PassParams p0 = lsb_make_params(num_items, 24, 8);
        lsb_twiddle_masks(key_type, 0, true, false, p0);
        int e;
        if ((e = lsb_upsweep(d_keys, lw.spine, lw.prefix16, p0, s))) return e;
        if ((e = lsb_scan(lw.spine, lw.totals, p0.grid, s))) return e;
        { KernelTimer kt(GS_K_MSB_CLASSIFY, s);
          hipLaunchKernelGGL(msb_classify_kernel<false>, dim3(1), dim3(256), 0, s, ws, 0, (const uint32_t *)lw.totals, nclass); }
        if (stop_level == 0) { const int e0 = (int)hipGetLastError(); return e0 ? e0 : (synchronize ? (int)hipStreamSynchronize(s) : 0); }
        MsbPeek *peek = stop_level == 99 ? msb_peek_get(s) : nullptr;      // see msb_run_levels
        msb_peek_arm(peek, ws, 1, s);
        if ((e = lsb_downsweep(d_keys, d_keys_alt, d_vals, d_vals_alt, lw.spine, lw.prefix16, lw.totals, p0, s))) return e;
        const MsbLook lk = msb_peek_wait(peek);                             // the device is busy with the scatter of level 0
        // upper bounds of what a level can hold (surplus blocks exit immediately)
        const uint32_t max_tasks_lvl = ws.max_tasks;
        const uint32_t task_grid0 = max_tasks_lvl < 2u * RADIX ? max_tasks_lvl : 2u * RADIX;   // level 0 emits <= 256 tasks
        const uint32_t *kt0 = lk.ok ? lk.tasks : nullptr;
        if (pairs) launch_local_sorts<true>(ws, 0, task_grid0, d_keys_alt, d_keys, d_vals_alt, d_vals, 0, 0u, tw.f32_out, tw.xor_out, s, 24, num_items, kt0);
        else launch_local_sorts<false>(ws, 0, task_grid0, d_keys_alt, d_keys, nullptr, nullptr, 0, 0u, tw.f32_out, tw.xor_out, s, 24, num_items, kt0);

        if (!(lk.ok && lk.buckets == 0))                                    // (no top-byte bucket outgrew the local sorts: done)
            msb_run_levels(ws, num_items, pairs, /*pieces=*/0u, buf_k, buf_v, tw, s, stop_level, allow_pivot);
    }
    int err = (int)hipGetLastError();
    if (err) return err;
    if (synchronize) err = msb_sync_and_check(ws, s);
    return err;
}

int gs_msb_sort_u32(void *d_temp, size_t temp_bytes, uint32_t *d_keys, uint32_t *d_vals, uint64_t num_items,
                    uint32_t *d_keys_alt, uint32_t *d_vals_alt, uint32_t **d_sorted_keys, uint32_t **d_sorted_vals,
                    int key_type, void *stream, int synchronize)
{
    return msb_sort_impl(d_temp, temp_bytes, d_keys, d_vals, num_items, d_keys_alt, d_vals_alt, d_sorted_keys, d_sorted_vals, key_type,
                         stream, synchronize, 99, true);
}

int gs_msb_classify_upto(void *d_temp, size_t temp_bytes, uint32_t *d_keys, uint32_t *d_keys_alt, uint64_t num_items,
                         int stop_level, int flags, void *stream)
{
    if (stop_level < 0 || stop_level > 2) return hipErrorInvalidValue;
    if (num_items <= msb_class_cap(msb_num_classes(false) - 1)) return hipErrorInvalidValue;   // such arrays have no levels
    return msb_sort_impl(d_temp, temp_bytes, d_keys, nullptr, num_items, d_keys_alt, nullptr, nullptr, nullptr, GS_KEY_U32, stream, 1,
                         stop_level, (flags & 1) == 0);
}

#ifdef GS_EXP_LS_PHASES
// out == nullptr: clear the stamps; otherwise copy all of them out ([plan][class][task][16] words)
int gs_exp_ls_phases(uint32_t *out)
{
    void *p = nullptr;
    if (hipGetSymbolAddress(&p, HIP_SYMBOL(gs_ls_phase_buf)) != hipSuccess) return (int)hipGetLastError();
    if (!out) return (int)hipMemset(p, 0, sizeof(uint32_t) * 3 * MSB_NCLASS * LSP_TASKS * 16);
    return (int)hipMemcpy(out, p, sizeof(uint32_t) * 3 * MSB_NCLASS * LSP_TASKS * 16, hipMemcpyDeviceToHost);
}
#endif
#ifdef GS_EXP_CLS
int gs_exp_cls_stamps(unsigned long long *out) { return (int)hipMemcpyFromSymbol(out, HIP_SYMBOL(gs_cls_stamp), sizeof(unsigned long long) * 16 * 8); }
#endif
void gs_msb_capacities(uint64_t num_items, int has_values, uint32_t *max_buckets, uint32_t *max_tasks, uint32_t *max_tiles)
{
    if (max_buckets) *max_buckets = msb_max_buckets(num_items, has_values != 0);
    if (max_tasks) *max_tasks = msb_max_tasks(num_items, has_values != 0);
    if (max_tiles) *max_tiles = msb_max_tiles(num_items, has_values != 0);
}

int gs_msb_census(void *d_temp, uint64_t num_items, int has_values, gs_msb_level_census out[4], void *stream)
{
    GS_CLEAR_STALE_ERROR();
    if (!d_temp || !out || num_items >= (1ull << 32)) return hipErrorInvalidValue;
    const MsbWs ws = msb_carve((char *)d_temp + align256(lsb_temp_bytes(num_items)), num_items, has_values != 0);
    MsbLevel lv[4];
    std::vector<MsbCensusSlot> slots((size_t)4 * MSB_CLASSIFY_GRID);
    hipError_t e = hipMemcpyAsync(lv, ws.level, sizeof(lv), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess)
        e = hipMemcpyAsync(slots.data(), ws.census, slots.size() * sizeof(MsbCensusSlot), hipMemcpyDeviceToHost, (hipStream_t)stream);
    if (e == hipSuccess) e = hipStreamSynchronize((hipStream_t)stream);
    if (e != hipSuccess) return (int)e;
    MsbCensusSlot sum[4] = {};
    for (int L = 0; L < 4; ++L) {
        const uint32_t nbk = lv[L].census_blocks < MSB_CLASSIFY_GRID ? lv[L].census_blocks : MSB_CLASSIFY_GRID;
        for (uint32_t i = 0; i < nbk; ++i) {
            const MsbCensusSlot &q = slots[(size_t)L * MSB_CLASSIFY_GRID + i];
            sum[L].next_keys += q.next_keys; sum[L].task_keys += q.task_keys;
            sum[L].pivot_keys += q.pivot_keys; sum[L].pivot_buckets += q.pivot_buckets;
        }
    }
    for (int L = 0; L < 4; ++L) {
        gs_msb_level_census c{};
        c.buckets = lv[L].packed >> 32;
        c.tiles = (uint32_t)lv[L].packed;
        c.keys = L == 0 ? lv[0].keys : sum[L - 1].next_keys;
        c.pivot_buckets = (uint32_t)sum[L].pivot_buckets;
        c.pivot_keys = sum[L].pivot_keys;
        c.task_keys = sum[L].task_keys;
        for (int q = 0; q < MSB_NCLASS; ++q) c.tasks[q] = lv[L].task_count[q];
        c.flagged = lv[L].flagged;
        c.overflow = lv[0].overflow;          // the sort's one word, repeated in every level's record
        out[L] = c;
    }
    return hipSuccess;
}

int gs_msb_read_lists(void *d_temp, uint64_t num_items, int has_values, int level, uint32_t *h_buckets, uint32_t max_buckets,
                      uint32_t *n_buckets, uint32_t *h_tasks[4], uint32_t max_tasks, uint32_t n_tasks[4], void *stream)
{
    GS_CLEAR_STALE_ERROR();
    if (!d_temp || level < 0 || level > 2 || num_items >= (1ull << 32)) return hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    const MsbWs ws = msb_carve((char *)d_temp + align256(lsb_temp_bytes(num_items)), num_items, has_values != 0);
    MsbLevel lv[4];
    hipError_t e = hipMemcpyAsync(lv, ws.level, sizeof(lv), hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    if (e != hipSuccess) return (int)e;
    uint32_t nb = (uint32_t)(lv[level + 1].packed >> 32);
    if (nb > ws.max_buckets) nb = ws.max_buckets;
    if (n_buckets) *n_buckets = nb;
    if (h_buckets && nb) {
        const uint32_t k = nb < max_buckets ? nb : max_buckets;
        std::vector<MsbBucket> tmp(k);
        e = hipMemcpyAsync(tmp.data(), ws.buckets[(level + 1) & 1], (size_t)k * sizeof(MsbBucket), hipMemcpyDeviceToHost, s);
        if (e == hipSuccess) e = hipStreamSynchronize(s);
        if (e != hipSuccess) return (int)e;
        for (uint32_t i = 0; i < k; ++i) { h_buckets[2 * i] = tmp[i].offset; h_buckets[2 * i + 1] = tmp[i].size; }
    }
    for (int c = 0; c < MSB_NCLASS; ++c) {
        uint32_t nt = lv[level].task_count[c];
        if (nt > ws.max_tasks) nt = ws.max_tasks;
        if (n_tasks) n_tasks[c] = nt;
        if (h_tasks && h_tasks[c] && nt) {
            const uint32_t k = nt < max_tasks ? nt : max_tasks;
            std::vector<MsbTask> tmp(k);
            e = hipMemcpyAsync(tmp.data(), ws.tasks[c], (size_t)k * sizeof(MsbTask), hipMemcpyDeviceToHost, s);
            if (e == hipSuccess) e = hipStreamSynchronize(s);
            if (e != hipSuccess) return (int)e;
            for (uint32_t i = 0; i < k; ++i) {
                h_tasks[c][3 * i] = tmp[i].offset; h_tasks[c][3 * i + 1] = tmp[i].size; h_tasks[c][3 * i + 2] = tmp[i].sort_bits;
            }
        }
    }
    return hipSuccess;
}

size_t gs_segmented_temp_bytes_impl(uint64_t num_items, int has_values, uint32_t num_segments)
{
    return msb_ws_bytes(num_items, has_values != 0, 0, num_segments);
}

// ---- the MSB path cut at the exchange point of the multi-GPU sort (north_star:
// "a single RCCL all-to-all after the first digit pass")

__global__ void msb_counts64_kernel(const uint32_t *__restrict__ totals, unsigned long long *__restrict__ out)
{
    out[threadIdx.x] = totals[threadIdx.x];
}

int gs_msb_first_pass_u32(void *d_temp, size_t temp_bytes, const uint32_t *d_keys_in, uint32_t *d_keys_out,
                          const uint32_t *d_vals_in, uint32_t *d_vals_out, uint64_t num_items, int key_type,
                          uint64_t *d_bucket_counts, void *stream)
{
    GS_CLEAR_STALE_ERROR();
    if (num_items >= (1ull << 32) || !d_bucket_counts) return hipErrorInvalidValue;
    if (key_type < GS_KEY_U32 || key_type > GS_KEY_F32) return hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    if (num_items == 0) return (int)zero_async(d_bucket_counts, RADIX * sizeof(uint64_t), s);
    if ((d_vals_in == nullptr) != (d_vals_out == nullptr) || !d_keys_in || !d_keys_out) return hipErrorInvalidValue;
    if (!d_temp || temp_bytes < lsb_temp_bytes(num_items)) return hipErrorInvalidValue;
    const LsbWorkspace lw = lsb_carve(d_temp, num_items);
    PassParams p0 = lsb_make_params(num_items, 24, 8);
    lsb_twiddle_masks(key_type, 0, true, false, p0);          // keys leave in their order-preserving u32 form
    int e;
    if ((e = lsb_upsweep(d_keys_in, lw.spine, lw.prefix16, p0, s))) return e;
    if ((e = lsb_scan(lw.spine, lw.totals, p0.grid, s))) return e;
    if ((e = lsb_downsweep(d_keys_in, d_keys_out, d_vals_in, d_vals_out, lw.spine, lw.prefix16, lw.totals, p0, s))) return e;
    hipLaunchKernelGGL(msb_counts64_kernel, dim3(1), dim3(RADIX), 0, s, (const uint32_t *)lw.totals,
                       (unsigned long long *)d_bucket_counts);
    return (int)hipGetLastError();
}

size_t gs_msb_finish_temp_bytes(uint64_t num_items, int has_values, int num_src)
{
    const uint32_t extra = (uint32_t)(num_src > 0 ? num_src : 0) * RADIX;
    return align256(lsb_temp_bytes(num_items)) + msb_ws_bytes(num_items, has_values != 0, extra);
}

int gs_msb_finish_u32(void *d_temp, size_t temp_bytes, uint32_t *d_keys, uint32_t *d_vals, uint32_t *d_keys_out,
                      uint32_t *d_vals_out, uint64_t num_items, const uint64_t *h_piece_counts, int num_src, int key_type,
                      void *stream, int synchronize)
{
    GS_CLEAR_STALE_ERROR();
    if (num_items >= (1ull << 32) || num_src < 1 || num_src > RADIX || !h_piece_counts) return hipErrorInvalidValue;
    if (key_type < GS_KEY_U32 || key_type > GS_KEY_F32) return hipErrorInvalidValue;
    // bucket b = all pieces (s, b); the buffer holds source 0's pieces in byte order, then source 1's, ...
    uint64_t total = 0;
    for (int i = 0; i < num_src * RADIX; ++i) total += h_piece_counts[i];
    if (total != num_items) return hipErrorInvalidValue;
    if (num_items == 0) return hipSuccess;
    const bool pairs = d_vals != nullptr;
    if (!d_keys || !d_keys_out || (pairs && !d_vals_out)) return hipErrorInvalidValue;
    if (!d_temp || temp_bytes < gs_msb_finish_temp_bytes(num_items, pairs, num_src)) return hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    const uint32_t extra = (uint32_t)num_src * RADIX;
    const MsbWs ws = msb_carve((char *)d_temp + align256(lsb_temp_bytes(num_items)), num_items, pairs, extra);

    // host side of what the level-0 classification does on one GPU: bucket list + pieces of level 1.  The tables
    // live on the heap until a host callback behind the copies frees them, so the call never blocks the host: a
    // caller that pipelines several finishes behind their exchanges enqueues them all at once.
    struct HostTables {
        MsbLevel hl[5];
        MsbBucket hb[RADIX];
        MsbPiece *hp;
        ~HostTables() { delete[] hp; }
    };
    HostTables *ht = new (std::nothrow) HostTables{};
    if (ht) ht->hp = new (std::nothrow) MsbPiece[(size_t)num_src * RADIX];
    if (!ht || !ht->hp) { delete ht; return hipErrorOutOfMemory; }
    uint64_t *src_off = new (std::nothrow) uint64_t[(size_t)num_src * RADIX];
    if (!src_off) { delete ht; return hipErrorOutOfMemory; }
    {   // the buffer holds source 0's pieces in byte order, then source 1's, ...
        uint64_t run = 0;
        for (int i = 0; i < num_src * RADIX; ++i) { src_off[i] = run; run += h_piece_counts[i]; }
    }
    uint32_t nb = 0, np = 0, tile = 0;
    {
        uint64_t out_off = 0;
        for (int b = 0; b < RADIX; ++b) {
            uint64_t size = 0;
            for (int sidx = 0; sidx < num_src; ++sidx) size += h_piece_counts[sidx * RADIX + b];
            if (size == 0) continue;
            const uint32_t tile_start = tile;
            for (int sidx = 0; sidx < num_src; ++sidx) {
                const uint64_t c = h_piece_counts[sidx * RADIX + b];
                if (c == 0) continue;
                ht->hp[np++] = MsbPiece{(uint32_t)src_off[sidx * RADIX + b], (uint32_t)c, tile, nb};
                tile += msb_piece_tiles((uint32_t)src_off[sidx * RADIX + b], (uint32_t)c);
            }
            ht->hb[nb++] = MsbBucket{(uint32_t)out_off, (uint32_t)size, tile_start, tile - tile_start};
            out_off += size;
        }
    }
    delete[] src_off;
    ht->hl[1].packed = ((unsigned long long)nb << 32) | tile;
    hipError_t e = hipMemcpyAsync(ws.level, ht->hl, sizeof(ht->hl), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(ws.buckets[1], ht->hb, nb * sizeof(MsbBucket), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipMemcpyAsync(ws.pieces, ht->hp, np * sizeof(MsbPiece), hipMemcpyHostToDevice, s);
    if (e == hipSuccess) e = hipLaunchHostFunc(s, [](void *p) { delete static_cast<HostTables *>(p); }, ht);
    if (e != hipSuccess) {             // the callback was not enqueued: wait for what was, then free here
        (void)hipStreamSynchronize(s);
        delete ht;
        return (int)e;
    }

    PassParams tw{};
    lsb_twiddle_masks(key_type, 0, true, true, tw);
    uint32_t *buf_k[2] = {d_keys_out, d_keys};                  // level 1 reads the received buffer, writes the output
    uint32_t *buf_v[2] = {d_vals_out, d_vals};
    msb_run_levels(ws, num_items, pairs, np, buf_k, buf_v, tw, s);
    int err = (int)hipGetLastError();
    if (err) return err;
    if (synchronize) err = msb_sync_and_check(ws, s);
    return err;
}

size_t gs_segmented_temp_bytes(uint64_t num_items, int has_values, uint32_t num_segments)
{
    return gs_segmented_temp_bytes_impl(num_items, has_values, num_segments);
}

int gs_segmented_sort_u32(void *d_temp, size_t temp_bytes, uint32_t *d_keys[2], uint32_t *d_vals[2], int *selector,
                          uint64_t num_items, uint32_t num_segments, const int32_t *d_begin_offsets,
                          const int32_t *d_end_offsets, int begin_bit, int end_bit, int descending, int key_type, void *stream)
{
    GS_CLEAR_STALE_ERROR();
    if (!selector || (*selector != 0 && *selector != 1) || !d_keys) return hipErrorInvalidValue;
    if (begin_bit < 0 || end_bit > 32 || begin_bit > end_bit) return hipErrorInvalidValue;
    if (num_items >= (1ull << 31)) return hipErrorInvalidValue;                  // int offsets
    if (key_type < GS_KEY_U32 || key_type > GS_KEY_F32) return hipErrorInvalidValue;
    if (num_items == 0 || num_segments == 0 || begin_bit == end_bit) return hipSuccess;
    const bool pairs = d_vals != nullptr;
    if (!d_begin_offsets || !d_end_offsets) return hipErrorInvalidValue;
    if (!d_temp || temp_bytes < gs_segmented_temp_bytes(num_items, pairs, num_segments)) return hipErrorInvalidValue;
    if (!d_keys[0] || !d_keys[1] || (pairs && (!d_vals[0] || !d_vals[1]))) return hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    const MsbWs ws = msb_carve(d_temp, num_items, pairs, 0, num_segments);
    const int nclass = msb_num_classes(pairs);
    const int num_bits = end_bit - begin_bit, passes = (num_bits + RADIX_BITS - 1) / RADIX_BITS;
    const int sel = *selector, fin = sel ^ (passes & 1);                         // one flip per pass, like gs_lsb_sort_u32
    PassParams tw{};
    lsb_twiddle_masks(key_type, descending, true, true, tw);

    static_assert(sizeof(MsbLevel) % 8 == 0, "levels are zeroed 8 bytes at a time");
    hipError_t e = zero_async(ws.level, MSB_LEVELS * sizeof(MsbLevel), s);
    if (e != hipSuccess) return (int)e;
    { KernelTimer kt(GS_K_MSB_CLASSIFY, s);
      const uint32_t g = (num_segments + 1023u) / 1024u;
      hipLaunchKernelGGL(seg_classify_kernel, dim3(g < 4096u ? g : 4096u), dim3(256), 0, s, ws, d_begin_offsets, d_end_offsets,
                         num_segments, nclass, (uint32_t)num_bits, (uint32_t)begin_bit, (uint32_t)num_items, SEG_TINY); }
    {   // tiny segments: one wave each
        KernelTimer kt(GS_K_MSB_LOCAL_SORT, s);
        const uint32_t wg = (num_segments + 3u) / 4u;
        const dim3 grid(wg < MSB_MAX_GRID ? wg : MSB_MAX_GRID);
#define GS_WS(HV, K) hipLaunchKernelGGL((seg_wave_sort_kernel<HV, K>), grid, dim3(256), 0, s, ws, (const uint32_t *)d_keys[sel], d_keys[fin], \
                                      HV ? (const uint32_t *)d_vals[sel] : (const uint32_t *)nullptr, HV ? d_vals[fin] : (uint32_t *)nullptr, \
                                      tw.f32_in, tw.xor_in, tw.f32_out, tw.xor_out)
        if (pairs) { GS_WS(true, 4); GS_WS(true, 8); GS_WS(true, 16); }
        else { GS_WS(false, 4); GS_WS(false, 8); GS_WS(false, 16); }
#undef GS_WS
        const uint32_t wg4 = (num_segments + 15u) / 16u;
        const dim3 grid4(wg4 < MSB_MAX_GRID ? wg4 : MSB_MAX_GRID);
        if (pairs) hipLaunchKernelGGL(seg_wave4_sort_kernel<true>, grid4, dim3(256), 0, s, ws, (const uint32_t *)d_keys[sel], d_keys[fin],
                                      (const uint32_t *)d_vals[sel], d_vals[fin], tw.f32_in, tw.xor_in, tw.f32_out, tw.xor_out);
        else hipLaunchKernelGGL(seg_wave4_sort_kernel<false>, grid4, dim3(256), 0, s, ws, (const uint32_t *)d_keys[sel], d_keys[fin],
                                (const uint32_t *)nullptr, (uint32_t *)nullptr, tw.f32_in, tw.xor_in, tw.f32_out, tw.xor_out);
    }
    // small segments: one stable local sort each, straight into the final buffer
    if (pairs) launch_local_sorts<true, true>(ws, 1, num_segments, d_keys[sel], d_keys[fin], d_vals[sel], d_vals[fin], tw.f32_in,
                                              tw.xor_in, tw.f32_out, tw.xor_out, s);
    else launch_local_sorts<false, true>(ws, 1, num_segments, d_keys[sel], d_keys[fin], nullptr, nullptr, tw.f32_in, tw.xor_in,
                                         tw.f32_out, tw.xor_out, s);
    // large segments: `passes` stable partitions of the same bucket list, 8 bits at a time from begin_bit
    const uint32_t max_b = ws.max_buckets;
    const uint32_t tiles_ub = (uint32_t)(num_items / MSB_TILE) + max_b;
    { KernelTimer kt(GS_K_MSB_HISTOGRAM, s);
      hipLaunchKernelGGL(msb_expand_kernel, dim3(max_b < 4096u ? max_b : 4096u), dim3(256), 0, s, ws, 1); }
    const bool any_tw = tw.f32_in || tw.xor_in;
    for (int p = 0; p < passes; ++p) {
        const int shift = begin_bit + p * RADIX_BITS;
        const int bits = (end_bit - shift < RADIX_BITS) ? end_bit - shift : RADIX_BITS;
        const bool first = p == 0, last = p == passes - 1;
        const DigitSel dsel{shift, nullptr, 0, tw.f32_in, tw.xor_in, bits, (first && any_tw) ? 1 : 0};
        uint32_t *sk = d_keys[sel ^ (p & 1)], *dk = d_keys[sel ^ ((p + 1) & 1)];
        const uint32_t *sv = pairs ? d_vals[sel ^ (p & 1)] : nullptr;
        uint32_t *dv = pairs ? d_vals[sel ^ ((p + 1) & 1)] : nullptr;
        { KernelTimer kt(GS_K_MSB_HISTOGRAM, s);
          const uint32_t hg_ub = tiles_ub / MSB_WAVES + 1;
          hipLaunchKernelGGL(msb_upsweep_kernel<false>, dim3(hg_ub < MSB_MAX_GRID ? hg_ub : MSB_MAX_GRID), dim3(MSB_THREADS), 0, s, ws,
                             1, (const uint32_t *)sk, dsel);
          hipLaunchKernelGGL(msb_scan_kernel, dim3(RADIX), dim3(1024), 0, s, ws, 1); }
        { KernelTimer kt(GS_K_MSB_CLASSIFY, s);
          hipLaunchKernelGGL(msb_classify_kernel<true>, dim3(max_b < 4096u ? max_b : 4096u), dim3(256), 0, s, ws, 1,
                             (const uint32_t *)nullptr, nclass); }
        { KernelTimer kt(GS_K_MSB_PARTITION, s);
          if (pairs) { if (last) launch_scatter<true, false, true>(ws, 1, tiles_ub, max_b, false, sk, dk, sv, dv, dsel, tw.f32_out, tw.xor_out, s);
                       else launch_scatter<true, false, false>(ws, 1, tiles_ub, max_b, false, sk, dk, sv, dv, dsel, 0, 0u, s); }
          else { if (last) launch_scatter<false, false, true>(ws, 1, tiles_ub, max_b, false, sk, dk, sv, dv, dsel, tw.f32_out, tw.xor_out, s);
                 else launch_scatter<false, false, false>(ws, 1, tiles_ub, max_b, false, sk, dk, sv, dv, dsel, 0, 0u, s); }
        }
    }
    *selector = fin;
    return (int)hipGetLastError();
}

int gs_shard_histogram_u32(const uint32_t *d_keys, uint64_t num_items, int bits, uint64_t *d_hist, int key_type,
                           void *stream)
{
    GS_CLEAR_STALE_ERROR();
    if (bits < 1 || bits > SHARD_MAX_BITS || !d_hist) return hipErrorInvalidValue;
    if (key_type < GS_KEY_U32 || key_type > GS_KEY_F32) return hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e = zero_async(d_hist, sizeof(uint64_t) << bits, s);
    if (e != hipSuccess) return (int)e;
    if (num_items == 0) return hipSuccess;
    if (!d_keys) return hipErrorInvalidValue;
    PassParams tw{};
    lsb_twiddle_masks(key_type, 0, true, true, tw);
    const uint64_t blocks = (num_items + MSB_TILE - 1) / MSB_TILE;
    KernelTimer kt(GS_K_SHARD, s);
    const dim3 grid((uint32_t)(blocks < 2048 ? blocks : 2048)), block(MSB_THREADS);
    if (((uintptr_t)d_keys & 15u) == 0)
        hipLaunchKernelGGL(shard_hist_kernel<true>, grid, block, 0, s, d_keys, num_items, bits, (unsigned long long *)d_hist,
                           tw.f32_in, tw.xor_in);
    else
        hipLaunchKernelGGL(shard_hist_kernel<false>, grid, block, 0, s, d_keys, num_items, bits,
                           (unsigned long long *)d_hist, tw.f32_in, tw.xor_in);
    return (int)hipGetLastError();
}

int gs_shard_partition_u32(void *d_temp, size_t temp_bytes, const uint32_t *d_keys_in, uint32_t *d_keys_out,
                           const uint32_t *d_vals_in, uint32_t *d_vals_out, uint64_t num_items, int bits,
                           const uint8_t *d_dest_of_bin, int num_ranks, const uint64_t *d_bin_hist, uint64_t *d_counts,
                           int key_type, void *stream)
{
    GS_CLEAR_STALE_ERROR();
    if (bits < 1 || bits > SHARD_MAX_BITS || num_ranks < 1 || num_ranks > RADIX || !d_dest_of_bin || !d_counts)
        return hipErrorInvalidValue;
    if (num_items >= (1ull << 32)) return hipErrorInvalidValue;
    if (key_type < GS_KEY_U32 || key_type > GS_KEY_F32) return hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    hipError_t e0 = zero_async(d_counts, sizeof(uint64_t) * num_ranks, s);
    if (e0 != hipSuccess) return (int)e0;
    if (num_items == 0) return hipSuccess;
    if ((d_vals_in == nullptr) != (d_vals_out == nullptr) || !d_keys_in || !d_keys_out) return hipErrorInvalidValue;
    const bool pairs = d_vals_in != nullptr;
    if (!d_temp || temp_bytes < gs_msb_temp_bytes(num_items, pairs)) return hipErrorInvalidValue;
    const uint32_t n = (uint32_t)num_items;
    const MsbWs ws = msb_carve((char *)d_temp + align256(lsb_temp_bytes(num_items)), num_items, pairs);
    PassParams tw{};
    lsb_twiddle_masks(key_type, 0, true, true, tw);
    const DigitSel dsel{0, d_dest_of_bin, 32 - bits, tw.f32_in, tw.xor_in, 8, 0};
    const uint32_t tiles = msb_tiles_of(n);
    const uint32_t grid = tiles;   // the tile count is known here: one tile per block, dispatched in order
    (void)d_bin_hist;              // per-tile counts are needed now: the keys are always read once more
    KernelTimer kt(GS_K_SHARD, s);
    // one bucket = the whole shard, "digit" = destination rank: upsweep -> scan -> cursors -> scatter
    hipLaunchKernelGGL(msb_init_kernel, dim3(1), dim3(64), 0, s, ws, n);
    hipLaunchKernelGGL(msb_expand_kernel, dim3(1), dim3(256), 0, s, ws, 0);
    hipLaunchKernelGGL(msb_upsweep_kernel<true>, dim3(tiles / MSB_WAVES + 1), dim3(MSB_THREADS), 0, s, ws, 0, d_keys_in, dsel);
    hipLaunchKernelGGL(msb_scan_kernel, dim3(RADIX), dim3(1024), 0, s, ws, 0);
    hipLaunchKernelGGL(msb_classify_kernel<true>, dim3(1), dim3(256), 0, s, ws, 0 /* cursors only */, (const uint32_t *)nullptr, 0);
    hipLaunchKernelGGL(shard_counts_kernel, dim3(1), dim3(RADIX), 0, s, (const uint32_t *)ws.cursors, n, num_ranks,
                       (unsigned long long *)d_counts);
    const bool big = num_items > (1ull << 30);
    if (pairs) launch_scatter<true, true, false>(ws, 0, grid, 1, big, d_keys_in, d_keys_out, d_vals_in, d_vals_out, dsel, 0, 0u, s);
    else launch_scatter<false, true, false>(ws, 0, grid, 1, big, d_keys_in, d_keys_out, nullptr, nullptr, dsel, 0, 0u, s);
    return (int)hipGetLastError();
}


size_t gs_segmented_wide_temp_bytes(uint64_t num_items, int /*key_bytes*/, int val_bytes, uint32_t num_segments)
{
    return msb_ws_bytes(num_items, val_bytes != 0, 0, num_segments, MW_CAP);
}

int gs_segmented_sort_wide(void *d_temp, size_t temp_bytes, void *d_keys[2], void *d_vals[2], int *selector, uint64_t num_items,
                           uint32_t num_segments, const int32_t *d_begin_offsets, const int32_t *d_end_offsets, int key_bytes,
                           int val_bytes, int begin_bit, int end_bit, int descending, int key_type, void *stream)
{
    GS_CLEAR_STALE_ERROR();
    if (!selector || (*selector != 0 && *selector != 1) || !d_keys) return hipErrorInvalidValue;
    if (key_bytes != 4 && key_bytes != 8) return hipErrorInvalidValue;
    if (val_bytes != 0 && val_bytes != 4 && val_bytes != 8) return hipErrorInvalidValue;
    if (key_bytes == 4 && val_bytes != 8) return hipErrorInvalidValue;          // (u32, none | u32) is gs_segmented_sort_u32's
    if ((val_bytes != 0) != (d_vals != nullptr)) return hipErrorInvalidValue;
    if (begin_bit < 0 || end_bit > 8 * key_bytes || begin_bit > end_bit) return hipErrorInvalidValue;
    if (num_items >= (1ull << 31)) return hipErrorInvalidValue;                  // int offsets
    const bool k64 = key_bytes == 8;
    if (k64 ? (key_type < GS_KEY_U64 || key_type > GS_KEY_F64) : (key_type < GS_KEY_U32 || key_type > GS_KEY_F32)) return hipErrorInvalidValue;
    if (num_items == 0 || num_segments == 0 || begin_bit == end_bit) return hipSuccess;
    if (!d_begin_offsets || !d_end_offsets) return hipErrorInvalidValue;
    if (!d_temp || temp_bytes < gs_segmented_wide_temp_bytes(num_items, key_bytes, val_bytes, num_segments)) return hipErrorInvalidValue;
    if (!d_keys[0] || !d_keys[1] || (d_vals && (!d_vals[0] || !d_vals[1]))) return hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
#define GS_SW(K, V) return seg_wide_sort<K, V>(d_temp, d_keys, d_vals, selector, num_items, num_segments, d_begin_offsets, d_end_offsets, \
                                                begin_bit, end_bit, descending, key_type, s)
    if (k64) {
        if (val_bytes == 0) GS_SW(uint64_t, MwNoVal);
        if (val_bytes == 4) GS_SW(uint64_t, uint32_t);
        GS_SW(uint64_t, uint64_t);
    }
    GS_SW(uint32_t, uint64_t);
#undef GS_SW
}

size_t gs_msb_wide_temp_bytes(uint64_t num_items, int key_bytes, int val_bytes)
{
    return mw_lsb_bytes(num_items, key_bytes, val_bytes) + msb_ws_bytes(num_items, val_bytes != 0, 0, 0, MW_CAP);
}

int gs_msb_sort_wide(void *d_temp, size_t temp_bytes, void *d_keys, void *d_vals, uint64_t num_items, void *d_keys_alt,
                     void *d_vals_alt, int key_bytes, int val_bytes, void **d_sorted_keys, void **d_sorted_vals, int key_type,
                     void *stream, int synchronize)
{
    GS_CLEAR_STALE_ERROR();
    if (key_bytes != 4 && key_bytes != 8) return hipErrorInvalidValue;
    if (val_bytes != 0 && val_bytes != 4 && val_bytes != 8) return hipErrorInvalidValue;
    if (key_bytes == 4 && val_bytes != 8) return hipErrorInvalidValue;          // (u32, none | u32) is gs_msb_sort_u32's
    if (num_items >= (1ull << 32)) return hipErrorInvalidValue;
    const bool k64 = key_bytes == 8;
    if (k64 ? (key_type < GS_KEY_U64 || key_type > GS_KEY_F64) : (key_type < GS_KEY_U32 || key_type > GS_KEY_F32)) return hipErrorInvalidValue;
    if (d_sorted_keys) *d_sorted_keys = d_keys;         // an even number of byte levels: the result is in the input arrays
    if (d_sorted_vals) *d_sorted_vals = d_vals;
    if (num_items == 0) return hipSuccess;
    if ((val_bytes != 0) != (d_vals != nullptr)) return hipErrorInvalidValue;
    if (!d_keys || !d_keys_alt || (d_vals && !d_vals_alt)) return hipErrorInvalidValue;
    if (!d_temp || temp_bytes < gs_msb_wide_temp_bytes(num_items, key_bytes, val_bytes)) return hipErrorInvalidValue;
    hipStream_t s = (hipStream_t)stream;
    int e;
#define GS_MW(K, V) e = msb_wide_sort<K, V>(d_temp, (K *)d_keys, (V *)d_vals, num_items, (K *)d_keys_alt, (V *)d_vals_alt, key_type, s)
    if (k64) {
        if (val_bytes == 0) GS_MW(uint64_t, MwNoVal);
        else if (val_bytes == 4) GS_MW(uint64_t, uint32_t);
        else GS_MW(uint64_t, uint64_t);
    } else {
        GS_MW(uint32_t, uint64_t);
    }
#undef GS_MW
    if (e) return e;
    if (synchronize) e = (int)hipStreamSynchronize(s);
    return e;
}

}  // extern "C"
