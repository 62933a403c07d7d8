// gs_lsb.hpp -- geometry, pass parameters and host entry points of the LSB
// three-kernel pass, shared with the MSB driver (its top-byte partition is one
// LSB pass at shift 24).
#pragma once

#include "gs_host.hpp"

namespace gs {

constexpr int LSB_THREADS = 512;                     // 8 waves
constexpr int LSB_WAVES = LSB_THREADS / WAVE;
constexpr int LSB_KPT = 16;                          // keys per thread per tile
constexpr int LSB_TILE = LSB_THREADS * LSB_KPT;      // 8192 keys = 32 KiB
constexpr int LSB_CHUNK = LSB_WAVES;                 // tiles per chunk = waves per upsweep block
constexpr int MI355X_CUS = 256;
constexpr int MI355X_XCDS = 8;
// Tiles are handed out in groups of LSB_GROUP consecutive tiles; inside a group the blocks of one
// XCD take a contiguous slice (tile_of_item), so neighbouring digit runs meet in the same L2.
#ifndef LSB_GROUP
#define LSB_GROUP 512
#endif
constexpr uint32_t LSB_RESIDENT = LSB_GROUP;

struct PassParams {
    uint32_t n;          // number of keys
    uint32_t num_tiles;  // ceil(n / LSB_TILE)
    uint32_t grid;       // chunks = upsweep blocks = spine row length
    uint32_t ds_grid;    // downsweep blocks = full tiles
    uint32_t shift;      // digit = (key >> shift) & mask
    uint32_t bits;       // digit width (<= 8)
    uint32_t mask;
    int f32_in, f32_out;          // float twiddle on read / undo on write
    uint32_t xor_in, xor_out;     // uniform xor on read / write (sign flip, descending)
};

// Block i -> tile.  Blocks are dispatched in order, so the resident blocks work on
// consecutive tiles; inside every group of
// LSB_RESIDENT items the blocks of one XCD (same b % 8 under round-robin dispatch) take
// a contiguous slice.  Speed only: any bijection gives the same result.
__device__ __forceinline__ uint32_t tile_of_item(uint32_t i, uint32_t full_tiles)
{
    const uint32_t base = (i / LSB_RESIDENT) * LSB_RESIDENT;
    if (base + LSB_RESIDENT > full_tiles) return i;   // ragged last group: identity
    const uint32_t r = i - base;
    return base + (r % MI355X_XCDS) * (LSB_RESIDENT / MI355X_XCDS) + r / MI355X_XCDS;
}

// The same with groups as large as the array allows (the three-launch LSB downsweep, round 3): groups of 2^k items, k <= 13, the
// remainder cut into smaller powers of two the same way, so that arrays of any size keep XCD-contiguous slices.  In-process A/B
// on the same buffers at 2^30 keys (tools/ab_inproc.py): groups of 256 items 1.810 ms, 512 1.766-1.775, 1024 1.781, 2048 1.792,
// 4096 1.772, 8192 1.758, 32768 1.764, all items 1.743-1.756 ms per launch: slices of >= 1024 consecutive tiles per XCD are worth 1 %.
constexpr uint32_t LSB_WIDE_GROUP = 8192;
__device__ __forceinline__ uint32_t tile_of_item_wide(uint32_t i, uint32_t full_tiles)
{
    uint32_t base = 0, rem = full_tiles;
    for (;;) {
        const uint32_t G = rem >= LSB_WIDE_GROUP ? LSB_WIDE_GROUP : (1u << (31u - (uint32_t)__builtin_clz(rem | 1u)));
        if (G < 2u * MI355X_XCDS) return i;                 // a handful of tiles: identity
        const uint32_t span = (rem / G) * G;                // whole groups of this size
        if (i < base + span) {
            const uint32_t r = (i - base) % G;
            return i - r + (r % MI355X_XCDS) * (G / MI355X_XCDS) + r / MI355X_XCDS;
        }
        base += span; rem -= span;
        if (rem == 0) return i;                             // (i >= full_tiles: not a tile; callers guard)
    }
}

// Upsweep block i -> chunk: inside every group of LSB_UPSWEEP_GROUP blocks the blocks of one XCD take consecutive
// chunks (their totals are neighbours in the spine rows).  Speed only.
#ifndef GS_EXP_UPS_GROUP
#define GS_EXP_UPS_GROUP 256
#endif
constexpr uint32_t LSB_UPSWEEP_GROUP = GS_EXP_UPS_GROUP;
__device__ __forceinline__ uint32_t chunk_of_block(uint32_t b, uint32_t grid)
{
    const uint32_t base = (b / LSB_UPSWEEP_GROUP) * LSB_UPSWEEP_GROUP;
    if (base + LSB_UPSWEEP_GROUP > grid) return b;   // ragged last group: identity
    const uint32_t r = b - base;
    return base + (r % MI355X_XCDS) * (LSB_UPSWEEP_GROUP / MI355X_XCDS) + r / MI355X_XCDS;
}

// Pipelined pass (gs_lsb.hip, lsb_pipe_pass_kernel): the three steps of a pass run inside ONE launch, the upsweep a
// bounded distance ahead of the downsweep, so that the downsweep's re-read of the keys is served by the Infinity Cache.
struct PipeParams {
    uint32_t tag;                  // pass index + 1: marks the words this pass published (never 0)
    uint32_t lead_chunks;          // upsweep blocks dispatched before the first downsweep block (multiple of 8)
    uint32_t scan_rows;            // rows the scanner walks: the chunks, rounded up to whole scanner batches
    uint32_t next_shift, next_bits;   // digit of the NEXT pass, whose totals the upsweep gathers on the way (0 bits: none)
    uint32_t test_drop_chunk_plus1;   // test hook (GS_LSB_PIPE_TEST_DROP): the upsweep role of this chunk (+ 1) publishes nothing,
                                      // so the waits behind it give up -- tests/test_lsb_gpu.py::test_one_launch_pass_give_up_stores_nothing
};
constexpr uint32_t PIPE_LEAD_CHUNKS = 128;   // 1024 tiles = 32 MiB of keys ahead (re-read window of the Infinity Cache: < 2048 tiles)
constexpr uint32_t PIPE_SUB_BLOCKS = 72;     // 8 upsweep blocks (chunks) followed by the 64 downsweep blocks of 64 tiles
constexpr uint32_t PIPE_GROUP_BLOCKS = PIPE_SUB_BLOCKS * (LSB_GROUP / 64);

struct LsbWorkspace {
    uint32_t *spine;
    uint32_t *totals;
    uint16_t *prefix16;
    // pipelined passes: all zeroed once per sort
    uint32_t *cc;           // [chunks][256] chunk counts, tag in bits 31:28
    uint64_t *sc;           // [chunks][256] {tag, exclusive prefix over the earlier chunks}
    uint32_t *ptotals;      // [5][256] digit totals of pass q in row q (row 0 unused: the scan kernel writes `totals`)
    uint32_t *error_word;   // set if a bounded spin ever gave up
};

size_t lsb_temp_bytes(uint64_t n);
LsbWorkspace lsb_carve(void *temp, uint64_t n);
PassParams lsb_make_params(uint64_t n, int shift, int bits);
void lsb_twiddle_masks(int key_type, int descending, bool first, bool last, PassParams &p);
int lsb_upsweep(const uint32_t *keys, uint32_t *spine, uint16_t *prefix16, const PassParams &p, hipStream_t s);
int lsb_scan(uint32_t *spine, uint32_t *totals, uint32_t grid, hipStream_t s);
int lsb_downsweep(const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout, const uint32_t *spine,
                  const uint16_t *prefix16, const uint32_t *totals, const PassParams &p, hipStream_t s);

// gs_wide.hip: digit totals of the last wide pass inside its workspace
const uint32_t *wide_totals_ptr(void *d_temp, uint64_t n);

// single-workgroup stable sort of a small array (gs_msb.hip): n <= small_sort_capacity(pairs)
uint32_t small_sort_capacity(bool pairs);
int small_stable_sort(void *scratch, size_t scratch_bytes, const uint32_t *kin, uint32_t *kout, const uint32_t *vin, uint32_t *vout,
                      uint32_t n, int begin_bit, int end_bit, int f32_in, uint32_t xor_in, int f32_out, uint32_t xor_out,
                      hipStream_t s);

}  // namespace gs
