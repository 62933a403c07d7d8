/*
 * gpusort.h -- C ABI of the MI355X-native radix sort (libgpusort.so).
 *
 * This is the drop-in boundary for the two sort drivers of
 * anilshanbhag/gpu-sort.  The reference has no FFI of its own (both sorts are
 * header templates instantiated in the caller's translation unit, SURVEY.md
 * 8b); each entry point below names the reference call it replaces.  All
 * citations are relative to the reference tree.
 *
 * Conventions
 *   - every pointer named d_* is a DEVICE pointer owned by the caller;
 *   - return value: 0 (= hipSuccess) or a hipError_t value; nothing throws;
 *   - entry points are re-entrant given distinct streams and workspaces (tests/test_concurrency_gpu.py); `stream`
 *     is a hipStream_t passed as void*.  State outside the caller's buffers: a few switches read from the
 *     environment (most once per process; INTEGRATION.md lists them), and per host thread one pinned 64-byte
 *     mailbox + event per device the thread has run an MSB sort on (the "look" of DESIGN.md section 1), released
 *     when the thread ends; nothing else persists between calls;
 *   - LSB entry points only enqueue work on `stream` and return (like
 *     cub::DeviceRadixSort, dispatch_radix_sort.cuh:899-979);
 *   - counts are 64-bit in the signature; the current kernels index with
 *     32-bit offsets, so num_items must be < 2^32 (the reference caps at
 *     int / unsigned, device_radix_sort.cuh:599,606, gpu_radix_sort.h:526).
 */
#ifndef GPUSORT_H_
#define GPUSORT_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define GS_VERSION 100

/* Key categories: the order-preserving key -> u32 map applied on the first
 * read and undone on the last write (cub::Traits<K>::TwiddleIn/Out,
 * lsb/cub/cub/util_type.cuh:966-974, 1009-1017, 1079-1089). */
enum gs_key_type {
    GS_KEY_U32 = 0,   /* identity                                  */
    GS_KEY_I32 = 1,   /* flip the sign bit                         */
    GS_KEY_F32 = 2,   /* negative: flip all bits; else flip sign   */
    GS_KEY_U64 = 3,   /* 64-bit keys: gs_lsb_sort_wide only        */
    GS_KEY_I64 = 4,
    GS_KEY_F64 = 5,
    GS_KEY_U8 = 6,    /* 8- and 16-bit keys: gs_lsb_sort_any only   */
    GS_KEY_I8 = 7,    /* (bool / unsigned char: U8; char / signed   */
    GS_KEY_U16 = 8,   /*  char: I8; unsigned short / short)         */
    GS_KEY_I16 = 9
};

int         gs_version(void);
const char *gs_error_string(int err);

/* ------------------------------------------------------------------ LSB --
 * Stable least-significant-digit radix sort, 8-bit digits, three kernels per
 * pass (upsweep histogram -> spine scan -> downsweep scatter).
 * Replaces cub::DeviceRadixSort::SortKeys / SortPairs / SortKeysDescending /
 * SortPairsDescending, DoubleBuffer overloads
 * (lsb/cub/cub/device/device_radix_sort.cuh:248-272, 595-621, 754-780),
 * as called from lsb/sort.cu:36,42,59,65.                                   */

/* Bytes of temp storage for a sort of num_items (spine, digit totals and the
 * per-tile u16 prefixes: about 1.6 % of the key bytes).
 * Replaces the d_temp_storage==NULL size query
 * (dispatch_radix_sort.cuh:1094-1110).  has_values is accepted for symmetry;
 * like CUB with is_overwrite_okay the value path needs no extra scratch.    */
size_t gs_lsb_temp_bytes(uint64_t num_items, int has_values);

/* d_keys[2] / d_vals[2] are the two halves of a DoubleBuffer
 * (util_type.cuh:785-817); *selector says which half is current on entry and
 * which half holds the result on return.  d_vals == NULL sorts keys only.
 * Both halves may be overwritten.  Sorts on key bits [begin_bit, end_bit).
 * Errors: hipErrorInvalidValue for a NULL/too-small workspace, bad bit range
 * or num_items >= 2^32 (util_device.cuh:90-93 behaviour).                    */
int gs_lsb_sort_u32(void *d_temp, size_t temp_bytes,
                    uint32_t *d_keys[2], uint32_t *d_vals[2], int *selector,
                    uint64_t num_items, int begin_bit, int end_bit,
                    int descending, int key_type, void *stream);

/* Non-overwriting form (the plain-pointer overloads of cub::DeviceRadixSort,
 * lsb/cub/cub/device/device_radix_sort.cuh:156-180,503-527 with
 * is_overwrite_okay == false, dispatch_radix_sort.cuh:1099-1129): the input
 * arrays are left untouched, the result is written to d_*_out, and the extra
 * ping-pong buffers are part of the workspace.                              */
size_t gs_lsb_copy_temp_bytes(uint64_t num_items, int has_values);
int gs_lsb_sort_copy_u32(void *d_temp, size_t temp_bytes,
                         const uint32_t *d_keys_in, uint32_t *d_keys_out,
                         const uint32_t *d_vals_in, uint32_t *d_vals_out,
                         uint64_t num_items, int begin_bit, int end_bit,
                         int descending, int key_type, void *stream);

/* Wider element types of the DeviceRadixSort contract
 * (lsb/cub/test/test_device_radix_sort.cu:934-943,1244-1265): keys of
 * key_bytes = 4 or 8 (GS_KEY_U32..F32 / GS_KEY_U64..F64), values of val_bytes =
 * 0 (keys only), 4 or 8.  Same DoubleBuffer/selector semantics as
 * gs_lsb_sort_u32; bits [begin_bit, end_bit) with end_bit <= 8*key_bytes.
 * General kernels (gs_wide.hip); the u32 / (u32,u32) cases are served faster
 * by gs_lsb_sort_u32.                                                       */
size_t gs_lsb_wide_temp_bytes(uint64_t num_items, int key_bytes, int val_bytes);
int gs_lsb_sort_wide(void *d_temp, size_t temp_bytes, void *d_keys[2], void *d_vals[2],
                     int *selector, uint64_t num_items, int key_bytes, int val_bytes,
                     int begin_bit, int end_bit, int descending, int key_type, void *stream);

/* Bring-up / test access to the three kernels of one pass (SURVEY.md 8a rows
 * L4-L6), all working on the same d_temp workspace (gs_lsb_temp_bytes):
 *   upsweep   -> spine[digit*grid + chunk] (u32 counts per chunk of
 *                tiles_per_chunk tiles) and prefix16[tile*256 + digit] (u16:
 *                count of the digit in the chunk's earlier tiles);
 *   scan      -> spine rows exclusive-scanned in place + totals[256];
 *   downsweep -> stable scatter of one pass.
 * gs_lsb_geometry reports the decomposition used for num_items so the oracle
 * can mirror it; gs_lsb_workspace_layout reports where the three arrays live
 * inside d_temp.                                                            */
void gs_lsb_geometry(uint64_t num_items, int has_values, uint32_t *grid, uint32_t *tile,
                     uint32_t *tiles_per_chunk);
int  gs_lsb_workspace_layout(void *d_temp, uint64_t num_items, uint32_t **d_spine,
                             uint32_t **d_totals, uint16_t **d_prefix16);
/* Pipelined passes (one launch per pass, the three steps as roles of its workgroups): every wait
 * inside such a launch is bounded; a wait that gave up sets a bit of the workspace's status word
 * (1 = a downsweep tile, 2 = the scanner) and the sort's output is then invalid.  Copies the word of
 * the last sort that used d_temp to *h_status after synchronising `stream` (0 = clean, also for
 * sorts that ran no pipelined pass).  Diagnostic: tests and benches read it.                      */
int  gs_lsb_pipe_status(void *d_temp, uint64_t num_items, uint32_t *h_status, void *stream);
int  gs_lsb_upsweep_u32(void *d_temp, size_t temp_bytes, const uint32_t *d_keys_in,
                        uint64_t num_items, int shift, int bits, int descending,
                        int key_type_in, void *stream);
int  gs_lsb_scan_spine(void *d_temp, size_t temp_bytes, uint64_t num_items, void *stream);
int  gs_lsb_downsweep_u32(void *d_temp, size_t temp_bytes,
                          const uint32_t *d_keys_in, uint32_t *d_keys_out,
                          const uint32_t *d_vals_in, uint32_t *d_vals_out,
                          uint64_t num_items, int shift, int bits, int descending,
                          int key_type_in, int key_type_out, void *stream);

/* The rest of DeviceRadixSort's type contract (lsb/cub/test/test_device_radix_sort.cu:930-945,1244-1265): keys of 8, 16, 32
 * or 64 bits (key_type names width and category) with values of ANY size -- none (val_bytes 0), the 1- and 2-byte values of
 * TestBackend<KeyT, KeyT>, the 16-byte TestFoo (test_util.h:1004-1010), ... -- stable, ascending or descending, on the bits
 * [begin_bit, end_bit) of the key's own width.  Plain-pointer form: the input arrays are untouched, the result is written
 * to the output arrays (which must not alias the inputs; 16-byte values must be 16-byte aligned).  Built on the library's
 * stable sort of (order-preserving sort key, index) pairs and one gather (gs_any.hip); only enqueues work on `stream`.   */
size_t gs_lsb_any_temp_bytes(uint64_t num_items, int key_type, int val_bytes);
int gs_lsb_sort_any(void *d_temp, size_t temp_bytes, const void *d_keys_in, void *d_keys_out,
                    const void *d_vals_in, void *d_vals_out, uint64_t num_items, int key_type, int val_bytes,
                    int begin_bit, int end_bit, int descending, void *stream);

/* ------------------------------------------------------------------ MSB --
 * Unstable most-significant-digit hybrid radix sort, ascending.
 * Replaces rdxsrt_unstable_sort<K,V,IndexT>
 * (msb/src/sort/gpu_radix_sort.h:197-507) as called from
 * msb/src/test.cu:53,55 and gpu_radix_sort.h:529,570.                        */

/* Workspace bytes (replaces RDXSRT_GPUDataManager sizing,
 * gpu_radix_sort.h:50-166; one allocation instead of eight). */
size_t gs_msb_temp_bytes(uint64_t num_items, int has_values);

/* d_keys/d_vals: input arrays (d_vals NULL = keys only); d_keys_alt /
 * d_vals_alt: scratch of the same size.  On return *d_sorted_keys /
 * *d_sorted_vals point at the arrays holding the result -- for 32-bit keys
 * these are the caller's INPUT arrays, as in the reference
 * (gpu_radix_sort.h:359-360, 505-506).  Enqueues on `stream` and, when
 * `synchronize` is nonzero, waits for completion like the reference does
 * (gpu_radix_sort.h:489-491).  Unless `stream` is being captured into a HIP
 * graph (or GS_MSB_PEEK=0 is set), the call also waits -- with the level's
 * scatter already enqueued, so the device stays busy -- until each level's
 * classification has run, to size or skip the next level's launches
 * (DESIGN.md section 1); the result does not depend on it.  (A host wait is
 * not allowed while ANOTHER stream of the process is being captured in the
 * global capture mode: set GS_MSB_PEEK=0 there.)                             */
int gs_msb_sort_u32(void *d_temp, size_t temp_bytes,
                    uint32_t *d_keys, uint32_t *d_vals, uint64_t num_items,
                    uint32_t *d_keys_alt, uint32_t *d_vals_alt,
                    uint32_t **d_sorted_keys, uint32_t **d_sorted_vals,
                    int key_type, void *stream, int synchronize);

/* The same sort for the wider element types the reference instantiates (RadixSortConfig<8,*>,
 * msb/src/sort/gpu_sort_config.h:179-198; msb/tests/test_sort_keys.cu:154-195, test_sort_pairs.cu:223-281): 64-bit keys
 * (GS_KEY_U64 / I64 / F64) with no, 32-bit or 64-bit values, and 32-bit keys with 64-bit values.  MSD hybrid like
 * gs_msb_sort_u32 (top byte by one stable pass, then byte levels on bucket lists, buckets of <= 8192 elements finished
 * by an LSD local sort in LDS); ascending, unstable; the result is in the caller's input arrays.                        */
size_t gs_msb_wide_temp_bytes(uint64_t num_items, int key_bytes, int val_bytes);
int gs_msb_sort_wide(void *d_temp, size_t temp_bytes, void *d_keys, void *d_vals, uint64_t num_items,
                     void *d_keys_alt, void *d_vals_alt, int key_bytes, int val_bytes,
                     void **d_sorted_keys, void **d_sorted_vals, int key_type, void *stream, int synchronize);

/* Census of the last gs_msb_sort_u32 that used d_temp (read back after synchronising `stream`): what every level
 * partitioned and what it handed to local sorts.  SURVEY.md 8d: the MSB path's algorithmic bytes are data-dependent --
 * "the harness must log the per-pass census and compute bytes from it": level 0 moves every key once (12 B/key), a level
 * L >= 1 reads its `keys` once for the histogram (4 B/key) and moves the ones outside heavy-hitter buckets (8 B/key;
 * 16 B/pair), heavy-hitter buckets are read once more (4 B/key), and every key is finished by exactly one local sort
 * (`task_keys`, 8 B/key; 16 B/pair), by the last level's scatter, or inside a heavy-hitter bucket.                     */
typedef struct gs_msb_level_census {
    uint64_t buckets;         /* buckets partitioned at this level (level 0: the whole array)            */
    uint64_t tiles;           /* their 8192-key tiles                                                     */
    uint64_t keys;            /* keys in them                                                             */
    uint64_t pivot_buckets;   /* of those, buckets finished by the heavy-hitter path                      */
    uint64_t pivot_keys;      /* keys in them                                                             */
    uint64_t task_keys;       /* keys handed to local sorts by this level's classification                */
    uint32_t tasks[4];        /* local-sort tasks per size class (2048 / 4608 / 9216 / 17408)             */
    uint32_t flagged;         /* != 0: some tasks needed the general local-sort plan                      */
    uint32_t overflow;        /* != 0: a device-side list of the sort overflowed and a record was dropped:  */
                              /* the result is WRONG (same word in all four records).  A synchronous        */
                              /* gs_msb_sort_u32 returns hipErrorUnknown in that case.                      */
} gs_msb_level_census;
int gs_msb_census(void *d_temp, uint64_t num_items, int has_values, gs_msb_level_census out[4], void *stream);
/* Capacities of the workspace's device-side lists for num_items (records): buckets a level may hold, local-sort tasks per
 * size class, tile records.  The sizing argument (gs_msb.hip, msb_max_*): a level's buckets are > the largest local sort
 * each; a task is >= 3000 keys or followed by something that did not merge with it (<= 2n / 3000, + 256 per bucket's
 * ragged end).  Exposed so that tests can hold the bounds against adversarial size patterns (tests/test_oracle.py).      */
void gs_msb_capacities(uint64_t num_items, int has_values, uint32_t *max_buckets, uint32_t *max_tasks, uint32_t *max_tiles);

/* Test access to the classification (cuda_radix_sort.h:1084-1087,1241-1247,
 * cuda_radix_sort_config.h:9).  Runs the sort of gs_msb_sort_u32 up to and including the classification of
 * `stop_level` (0..2; that level's scatter and everything after it do not run, so the arrays hold an intermediate
 * state) and leaves its lists in the workspace; `flags` bit 0 switches the heavy-hitter path off.
 * gs_msb_read_lists then copies them out: the buckets the classification passed to level stop_level + 1 as
 * {offset, size} pairs and the local-sort tasks of every class as {offset, size, sort_bits} triples (in device order,
 * which is not deterministic: compare as sets).  Counts are returned through n_buckets / n_tasks[4]; at most
 * max_buckets / max_tasks entries are copied per list.                                                              */
int gs_msb_classify_upto(void *d_temp, size_t temp_bytes, uint32_t *d_keys, uint32_t *d_keys_alt, uint64_t num_items,
                         int stop_level, int flags, void *stream);
int gs_msb_read_lists(void *d_temp, uint64_t num_items, int has_values, int level,
                      uint32_t *h_buckets, uint32_t max_buckets, uint32_t *n_buckets,
                      uint32_t *h_tasks[4], uint32_t max_tasks, uint32_t n_tasks[4], void *stream);

/* ------------------------------------------------ multi-GPU shard helpers --
 * One process per GPU; the exchange itself (one all-to-all over RCCL/xGMI)
 * is issued by the host between these calls (no reference
 * counterpart -- the reference is single-GPU).                               */

/* cub::DeviceSegmentedRadixSort (lsb/cub/cub/device/device_segmented_radix_sort.cuh:77-861;
 * DeviceSegmentedRadixSortKernel, dispatch_radix_sort.cuh:321-432): every segment
 * [d_begin_offsets[i], d_end_offsets[i]) of the keys (and values) is sorted on its own, stably,
 * on bits [begin_bit, end_bit); segments must not overlap, empty ones are fine, and positions
 * outside every segment are not written (offsets outside [0, num_items] are clamped on the device).  DoubleBuffer semantics as gs_lsb_sort_u32 (both
 * halves may be clobbered, the result is d_keys[*selector] after the call).  num_items < 2^31.
 * Segments that fit one workgroup (<= 17408 keys or pairs) cost one read and one write; the
 * larger ones are partitioned together, one 8-bit digit per pass.                       */
size_t gs_segmented_temp_bytes(uint64_t num_items, int has_values, uint32_t num_segments);
int gs_segmented_sort_u32(void *d_temp, size_t temp_bytes, uint32_t *d_keys[2], uint32_t *d_vals[2],
                          int *selector, uint64_t num_items, uint32_t num_segments,
                          const int32_t *d_begin_offsets, const int32_t *d_end_offsets, int begin_bit,
                          int end_bit, int descending, int key_type, void *stream);

/* The same for the wider element types (64-bit keys: GS_KEY_U64 / I64 / F64, with no, 32-bit or 64-bit values; 32-bit keys
 * with 64-bit values) -- cub's segmented dispatch is type-generic (dispatch_radix_sort.cuh:321-432).  Stable; bits
 * [begin_bit, end_bit) of the key; segments of <= 8192 elements cost one read and one write.  num_items < 2^31.        */
size_t gs_segmented_wide_temp_bytes(uint64_t num_items, int key_bytes, int val_bytes, uint32_t num_segments);
int gs_segmented_sort_wide(void *d_temp, size_t temp_bytes, void *d_keys[2], void *d_vals[2], int *selector,
                           uint64_t num_items, uint32_t num_segments, const int32_t *d_begin_offsets,
                           const int32_t *d_end_offsets, int key_bytes, int val_bytes, int begin_bit, int end_bit,
                           int descending, int key_type, void *stream);

/* The MSB path cut at the exchange point (north_star: "a single RCCL all-to-all after the
 * first digit pass"): gs_msb_first_pass_u32 is the top-byte partition on its own -- keys (and
 * values) leave grouped by top byte in d_*_out, in their order-preserving u32 form, and
 * d_bucket_counts[256] (u64) gets the bucket sizes; d_temp sized by gs_lsb_temp_bytes.  The
 * host assigns contiguous bucket ranges to ranks, so every rank's share is one contiguous
 * slice of d_keys_out and goes out in ONE all-to-all.  gs_msb_finish_u32 completes the sort on
 * the receiving rank: d_keys holds, source after source, each source's slice (its buckets in
 * byte order); h_piece_counts[num_src][256] (HOST memory) are the piece sizes.  The buckets are
 * picked up where they lie -- no regrouping pass -- and the sorted keys (in key_type's own
 * representation) land in d_keys_out; d_keys / d_vals are clobbered.  d_temp sized by
 * gs_msb_finish_temp_bytes.  h_piece_counts is read before the call returns; the call itself
 * is asynchronous on `stream` unless `synchronize` is set, so a host that cuts the exchange
 * into several collectives can enqueue one finish per group of buckets behind its collective
 * (pointers offset to the group's slice of the receive and output buffers, one d_temp per
 * finish that may be in flight -- calls on one stream may share it).              */
int gs_msb_first_pass_u32(void *d_temp, size_t temp_bytes, const uint32_t *d_keys_in,
                          uint32_t *d_keys_out, const uint32_t *d_vals_in, uint32_t *d_vals_out,
                          uint64_t num_items, int key_type, uint64_t *d_bucket_counts, void *stream);
size_t gs_msb_finish_temp_bytes(uint64_t num_items, int has_values, int num_src);
int gs_msb_finish_u32(void *d_temp, size_t temp_bytes, uint32_t *d_keys, uint32_t *d_vals,
                      uint32_t *d_keys_out, uint32_t *d_vals_out, uint64_t num_items,
                      const uint64_t *h_piece_counts, int num_src, int key_type, void *stream,
                      int synchronize);

/* 2^bits-bin histogram (u64 counts) of the top `bits` bits of each key. */
int gs_shard_histogram_u32(const uint32_t *d_keys, uint64_t num_items, int bits,
                           uint64_t *d_hist, int key_type, void *stream);
/* Partition by destination rank:
 * d_dest_of_bin[top `bits` bits] gives the rank (monotone non-decreasing,
 * < num_ranks <= 256).  Writes keys (and values) grouped by rank into d_*_out
 * and the per-rank counts into d_counts[num_ranks] (u64).  The order inside a
 * rank's group is the input order (deterministic).  d_bin_hist: accepted for
 * compatibility and ignored (the partition counts per tile itself), may be
 * NULL.  d_temp sized by gs_msb_temp_bytes.                                   */
int gs_shard_partition_u32(void *d_temp, size_t temp_bytes,
                           const uint32_t *d_keys_in, uint32_t *d_keys_out,
                           const uint32_t *d_vals_in, uint32_t *d_vals_out,
                           uint64_t num_items, int bits, const uint8_t *d_dest_of_bin,
                           int num_ranks, const uint64_t *d_bin_hist, uint64_t *d_counts,
                           int key_type, void *stream);

/* -------------------------------------------------- on-device test inputs --
 * Counter-based generators identical to oracle/oracle.c; they
 * replace the cuRAND fills of lsb/sort.cu:125-131, msb/src/test.cu:38-43 and
 * msb/tests/data_gen.h:33-84.                                                */
enum gs_gen_kind { GS_GEN_UNIFORM = 0, GS_GEN_ZIPF = 1, GS_GEN_ENTROPY_AND = 2, GS_GEN_ENUMERATED = 3 };
int gs_generate_u32(uint32_t *d_out, uint64_t num_items, int kind, uint64_t seed,
                    uint64_t start_index, int level, void *stream);

/* Size-independent result checks run on the device (used at full BASELINE
 * sizes where a host oracle would take minutes): d_result[0] = number of
 * adjacent inversions, [1] = sum of splitmix64(key), [2] = xor of the same. */
int gs_check_sorted_u32(const uint32_t *d_keys, uint64_t num_items, int descending,
                        uint64_t *d_result, void *stream);
/* d_result[0] = number of i with d_keys_in[d_vals[i]] != d_keys_sorted[i] or
 * d_vals[i] >= num_items, [1] = sum of d_vals (msb/tests/test_sort_pairs.cu
 * :141-146,166-176 on the device).                                           */
int gs_check_pairs_enumerated_u32(const uint32_t *d_keys_in, const uint32_t *d_keys_sorted,
                                  const uint32_t *d_vals, uint64_t num_items,
                                  uint64_t *d_result, void *stream);

/* ------------------------------------------------------ kernel timing hook --
 * Optional per-kernel device timing with hipEvents recorded on the SAME stream
 * the kernels are launched on (replaces the reference's cudaEvent pairs,
 * lsb/gpu_utils.h:3-11, and its gated per-pass BM_* events,
 * msb/src/sort/gpu_radix_sort.h:266-269).  While a profile is bound to the
 * calling thread (gs_profile_begin .. gs_profile_end) every kernel the library
 * launches from that thread is bracketed by an event pair.  gs_profile_read
 * waits for the recorded events and accumulates milliseconds and launch counts
 * per kernel id; it may be called once the work has been enqueued.            */
enum gs_kernel_id {
    GS_K_LSB_UPSWEEP = 0, GS_K_LSB_SCAN = 1, GS_K_LSB_DOWNSWEEP = 2,
    GS_K_MSB_HISTOGRAM = 3, GS_K_MSB_CLASSIFY = 4, GS_K_MSB_PARTITION = 5, GS_K_MSB_LOCAL_SORT = 6,
    GS_K_SHARD = 7, GS_K_OTHER = 8,
    GS_K_LSB_PASS = 9,   /* a whole pass in one launch (upsweep / scan / downsweep as roles of one kernel) */
    GS_K_COUNT = 10
};
typedef struct gs_profile gs_profile;
gs_profile *gs_profile_create(void);
void        gs_profile_destroy(gs_profile *p);
void        gs_profile_begin(gs_profile *p);   /* bind to this thread */
void        gs_profile_end(void);              /* unbind              */
int         gs_profile_read(gs_profile *p, double total_ms[GS_K_COUNT], uint64_t launches[GS_K_COUNT]);
const char *gs_kernel_name(int kernel_id);

#ifdef __cplusplus
}
#endif
#endif /* GPUSORT_H_ */
