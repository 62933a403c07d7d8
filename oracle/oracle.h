/*
 * oracle.h -- CPU restatement of the reference's sort semantics.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load this library, and only as the checker.  The product path lives in
 * gpu-sort_amd/csrc and fails loudly when its HIP library is missing.
 *
 * Parity pin (see DESIGN.md "Oracle"): the reference holds NO golden vectors
 * (SURVEY.md 8c); every reference test computes its expectation at run time
 * with a stable sort.  An ascending stable sort of a u32 key (pair) array is
 * unique, so this restatement is pinned by (1) the MT19937 known-answer
 * stream produced by the reference's own lsb/cub/test/mersenne.h compiled
 * into oracle/_ref, (2) an independent numpy stable sort over the committed
 * fixtures in tests/golden.
 *
 * All citations are relative to /root/reference.
 */
#ifndef GS_ORACLE_H_
#define GS_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- synthetic inputs (reference inputs are cuRAND XORWOW,
 *      lsb/sort.cu:126-131, msb/tests/data_gen.h:33-41, not reproducible
 *      off-NVIDIA, so a counter-based generator replaces them) ------------ */
uint64_t orc_splitmix64(uint64_t x);
/* key[i] = hi32(splitmix64(seed*GOLDEN + start + i)) */
void orc_gen_uniform(uint32_t *out, uint64_t n, uint64_t seed, uint64_t start);
/* integer-only Zipf(s=1) over 2^24 ranks, key = rank * 0x9E3779B1 */
void orc_gen_zipf(uint32_t *out, uint64_t n, uint64_t seed, uint64_t start);
/* Thearling-Smith entropy reduction, msb/tests/data_gen.h:55-68:
 * level<=0 -> zeros; else AND of `level` uniform draws seeded seed+17*i */
void orc_gen_entropy_and(uint32_t *out, uint64_t n, uint64_t seed, uint64_t start, int level);
/* v[i] = start+i, msb/tests/data_gen.h:78-84 */
void orc_gen_enumerated(uint32_t *out, uint64_t n, uint64_t start);

/* ---- MT19937 + RandomBits, lsb/cub/test/mersenne.h:59-150 and
 *      lsb/cub/test/test_util.h:96-97,408-458 ---------------------------- */
void     orc_mt_init_genrand(uint32_t s);
void     orc_mt_init_by_array(const uint32_t *key, int key_length);
uint32_t orc_mt_genrand_int32(void);
/* CommandLineArgs ctor seed {0x123,0x234,0x345,0x456}, test_util.h:96-97 */
void     orc_mt_init_cub_default(void);
/* RandomBits<unsigned int> for n keys in sequence */
void     orc_random_bits_u32(uint32_t *keys, uint64_t n, int entropy_reduction,
                             int begin_bit, int end_bit);

/* ---- key twiddles, lsb/cub/cub/util_type.cuh:966-974,1009-1017,1079-1089 - */
uint32_t orc_twiddle_in_u32(uint32_t k);
uint32_t orc_twiddle_in_i32(uint32_t k);
uint32_t orc_twiddle_in_f32(uint32_t k);
uint32_t orc_twiddle_out_f32(uint32_t k);

/* ---- LSB oracle: InitializeSolution,
 *      lsb/cub/test/test_device_radix_sort.cu:634-693 (+ :888-889 values) -- */
/* ranks[i] = original index of the i-th output element */
/* 64-bit keys: key_type 3 = unsigned, 4 = signed, 5 = double (gs_key_type) */
uint64_t orc_twiddle_in_u64(uint64_t k);
uint64_t orc_twiddle_in_i64(uint64_t k);
uint64_t orc_twiddle_in_f64(uint64_t k);
uint64_t orc_twiddle_out_f64(uint64_t k);
void orc_lsb_reference_ranks_u64(const uint64_t *keys, uint64_t n, int key_type, int begin_bit,
                                 int end_bit, int descending, uint32_t *ranks);
void orc_lsb_reference_ranks(const uint32_t *keys, uint64_t n, int begin_bit, int end_bit,
                             int descending, uint32_t *ranks);
void orc_lsb_sort_keys(const uint32_t *keys_in, uint32_t *keys_out, uint64_t n,
                       int begin_bit, int end_bit, int descending);
void orc_lsb_sort_pairs(const uint32_t *keys_in, const uint32_t *vals_in,
                        uint32_t *keys_out, uint32_t *vals_out, uint64_t n,
                        int begin_bit, int end_bit, int descending);

/* ---- per-kernel goldens for the three-kernel pass (derived; semantics of
 *      agent_radix_sort_upsweep.cuh:215-229, dispatch_radix_sort.cuh:102-148,
 *      agent_radix_sort_downsweep.cuh:560-566) --------------------------- */
/* tile->chunk split (the role of GridEvenShare, grid_even_share.cuh:103-139:
 * every block owns one contiguous run of tiles): chunk c = tiles
 * [c*tiles_per_chunk, (c+1)*tiles_per_chunk) clipped to num_tiles */
void orc_chunk_tiles(uint64_t num_tiles, uint32_t tiles_per_chunk, uint32_t c,
                     uint64_t *tile_begin, uint64_t *tile_end);
/* spine[d*grid + c] = count of digit d in chunk c's tiles */
void orc_upsweep(const uint32_t *keys, uint64_t n, int shift, int bits, int descending,
                 uint32_t tile, uint32_t tiles_per_chunk, uint32_t grid, uint32_t *spine);
/* in-place exclusive prefix sum over len ints */
void orc_exclusive_scan(uint32_t *spine, uint64_t len);
/* one stable counting pass on digit (key>>shift)&((1<<bits)-1) */
void orc_downsweep(const uint32_t *keys_in, const uint32_t *vals_in,
                   uint32_t *keys_out, uint32_t *vals_out, uint64_t n,
                   int shift, int bits, int descending);
/* full LSD radix restatement, 8-bit digits (north_star formulation) */
void orc_lsd_radix_sort(uint32_t *keys, uint32_t *vals, uint32_t *keys_tmp, uint32_t *vals_tmp,
                        uint64_t n, int begin_bit, int end_bit, int descending,
                        int *result_in_tmp);

/* ---- MSB checkers, msb/tests/test_sort_keys.cu:50-80,
 *      msb/tests/test_sort_pairs.cu:67-118,141-146,166-176 --------------- */
/* 0 = ok; else 1 + index of first mismatch */
uint64_t orc_msb_check_keys(const uint32_t *keys_in, const uint32_t *keys_sorted, uint64_t n);
/* values compared after sorting inside each equal-key run (unstable contract) */
uint64_t orc_msb_check_pairs(const uint32_t *keys_in, const uint32_t *vals_in,
                             const uint32_t *keys_sorted, uint32_t *vals_sorted, uint64_t n);
/* fast check for enumerated values: key_of[v]==k, v<n, sum v = n(n-1)/2 */
uint64_t orc_msb_check_pairs_enumerated(const uint32_t *keys_in, const uint32_t *keys_sorted,
                                        const uint32_t *vals_sorted, uint64_t n);

/* ---- size-independent properties used at full BASELINE sizes ------------ */
/* order-independent multiset checksum: sum and xor of splitmix64(key) */
void orc_multiset_checksum(const uint32_t *keys, uint64_t n, uint64_t *sum, uint64_t *xr);
/* number of i with a[i] > a[i+1] (descending: a[i] < a[i+1]) */
uint64_t orc_count_inversions_adjacent(const uint32_t *a, uint64_t n, int descending);

#ifdef __cplusplus
}
#endif
#endif /* GS_ORACLE_H_ */
