/*
 * oracle.c -- CPU restatement of the reference's sort semantics (plain C).
 * TEST INFRASTRUCTURE ONLY -- see oracle.h for the rules and the parity pin.
 * Citations are relative to /root/reference.
 */
#include "oracle.h"

#include <stdlib.h>
#include <string.h>

#define GOLDEN64 0x9E3779B97F4A7C15ull

/* ------------------------------------------------------------------ inputs */

uint64_t orc_splitmix64(uint64_t x)
{
    uint64_t z = x + GOLDEN64;
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}

static inline uint32_t uniform_at(uint64_t seed, uint64_t idx)
{
    return (uint32_t)(orc_splitmix64(seed * GOLDEN64 + idx) >> 32);
}

void orc_gen_uniform(uint32_t *out, uint64_t n, uint64_t seed, uint64_t start)
{
    for (uint64_t i = 0; i < n; ++i) out[i] = uniform_at(seed, start + i);
}

void orc_gen_zipf(uint32_t *out, uint64_t n, uint64_t seed, uint64_t start)
{
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t h = orc_splitmix64((seed + 0x2545F491ull) * GOLDEN64 + start + i);
        uint32_t e = (uint32_t)(((h >> 32) * 24ull) >> 32);      /* uniform in [0,24) */
        uint32_t m = (uint32_t)h & ((1u << e) - 1u);               /* uniform in [0,2^e) */
        uint32_t r = (1u << e) + m;                                /* P(r) ~ 1/r */
        out[i] = r * 0x9E3779B1u;                                  /* odd => bijection */
    }
}

/* msb/tests/data_gen.h:55-68 */
void orc_gen_entropy_and(uint32_t *out, uint64_t n, uint64_t seed, uint64_t start, int level)
{
    if (level < 1) { memset(out, 0, n * sizeof(uint32_t)); return; }
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t k = uniform_at(seed, start + i);
        for (int l = 1; l < level; ++l) k &= uniform_at(seed + 17ull * (uint64_t)l, start + i);
        out[i] = k;
    }
}

/* msb/tests/data_gen.h:78-84 */
void orc_gen_enumerated(uint32_t *out, uint64_t n, uint64_t start)
{
    for (uint64_t i = 0; i < n; ++i) out[i] = (uint32_t)(start + i);
}

/* ----------------------------------------------------------------- MT19937 */
/* lsb/cub/test/mersenne.h:48-150 (the canonical Matsumoto-Nishimura
 * generator; constants are the published MT19937 parameters) */
#define MT_N 624
#define MT_M 397
static uint32_t mt_state[MT_N];
static int mt_index = MT_N + 1;

void orc_mt_init_genrand(uint32_t s)
{
    mt_state[0] = s;
    for (mt_index = 1; mt_index < MT_N; ++mt_index)
        mt_state[mt_index] =
            1812433253u * (mt_state[mt_index - 1] ^ (mt_state[mt_index - 1] >> 30)) + (uint32_t)mt_index;
}

void orc_mt_init_by_array(const uint32_t *key, int key_length)
{
    int i = 1, j = 0, k;
    orc_mt_init_genrand(19650218u);
    k = (MT_N > key_length) ? MT_N : key_length;
    for (; k; --k) {
        mt_state[i] = (mt_state[i] ^ ((mt_state[i - 1] ^ (mt_state[i - 1] >> 30)) * 1664525u)) + key[j] + (uint32_t)j;
        ++i; ++j;
        if (i >= MT_N) { mt_state[0] = mt_state[MT_N - 1]; i = 1; }
        if (j >= key_length) j = 0;
    }
    for (k = MT_N - 1; k; --k) {
        mt_state[i] = (mt_state[i] ^ ((mt_state[i - 1] ^ (mt_state[i - 1] >> 30)) * 1566083941u)) - (uint32_t)i;
        ++i;
        if (i >= MT_N) { mt_state[0] = mt_state[MT_N - 1]; i = 1; }
    }
    mt_state[0] = 0x80000000u;
}

uint32_t orc_mt_genrand_int32(void)
{
    static const uint32_t mag01[2] = {0u, 0x9908b0dfu};
    uint32_t y;
    if (mt_index >= MT_N) {
        int kk;
        if (mt_index == MT_N + 1) orc_mt_init_genrand(5489u);
        for (kk = 0; kk < MT_N - MT_M; ++kk) {
            y = (mt_state[kk] & 0x80000000u) | (mt_state[kk + 1] & 0x7fffffffu);
            mt_state[kk] = mt_state[kk + MT_M] ^ (y >> 1) ^ mag01[y & 1u];
        }
        for (; kk < MT_N - 1; ++kk) {
            y = (mt_state[kk] & 0x80000000u) | (mt_state[kk + 1] & 0x7fffffffu);
            mt_state[kk] = mt_state[kk + (MT_M - MT_N)] ^ (y >> 1) ^ mag01[y & 1u];
        }
        y = (mt_state[MT_N - 1] & 0x80000000u) | (mt_state[0] & 0x7fffffffu);
        mt_state[MT_N - 1] = mt_state[MT_M - 1] ^ (y >> 1) ^ mag01[y & 1u];
        mt_index = 0;
    }
    y = mt_state[mt_index++];
    y ^= (y >> 11);
    y ^= (y << 7) & 0x9d2c5680u;
    y ^= (y << 15) & 0xefc60000u;
    y ^= (y >> 18);
    return y;
}

/* lsb/cub/test/test_util.h:96-97 */
void orc_mt_init_cub_default(void)
{
    static const uint32_t k[4] = {0x123, 0x234, 0x345, 0x456};
    orc_mt_init_by_array(k, 4);
}

/* lsb/cub/test/test_util.h:408-458, K = unsigned int (one word, never NaN) */
void orc_random_bits_u32(uint32_t *keys, uint64_t n, int entropy_reduction, int begin_bit, int end_bit)
{
    for (uint64_t i = 0; i < n; ++i) {
        if (entropy_reduction == -1) { keys[i] = 0; continue; }
        int eb = end_bit < 0 ? 32 : end_bit;
        uint32_t word = 0xffffffffu;
        int lo = begin_bit > 0 ? begin_bit : 0;
        int hi = (32 - eb) > 0 ? (32 - eb) : 0;
        word &= (lo >= 32) ? 0u : (0xffffffffu << lo);
        word &= (hi >= 32) ? 0u : (0xffffffffu >> hi);
        for (int e = 0; e <= entropy_reduction; ++e) word &= orc_mt_genrand_int32();
        keys[i] = word;
    }
}

/* ---------------------------------------------------------------- twiddles */
/* util_type.cuh:966-974 */
uint32_t orc_twiddle_in_u32(uint32_t k) { return k; }
/* util_type.cuh:1009-1017 */
uint32_t orc_twiddle_in_i32(uint32_t k) { return k ^ 0x80000000u; }
/* util_type.cuh:1079-1089 */
uint32_t orc_twiddle_in_f32(uint32_t k) { return k ^ ((k & 0x80000000u) ? 0xffffffffu : 0x80000000u); }
uint32_t orc_twiddle_out_f32(uint32_t k) { return k ^ ((k & 0x80000000u) ? 0x80000000u : 0xffffffffu); }

/* -------------------------------------------------------------- LSB oracle */

typedef struct { uint32_t key; uint32_t value; } orc_pair_t;

/* std::stable_sort restated as a bottom-up merge sort; compares .key only,
 * as Pair::operator< does (test_device_radix_sort.cu:590-612, integer keys) */
static void stable_sort_pairs(orc_pair_t *a, uint64_t n)
{
    if (n < 2) return;
    orc_pair_t *tmp = (orc_pair_t *)malloc(n * sizeof(orc_pair_t));
    orc_pair_t *src = a, *dst = tmp;
    for (uint64_t w = 1; w < n; w *= 2) {
        for (uint64_t lo = 0; lo < n; lo += 2 * w) {
            uint64_t mid = lo + w < n ? lo + w : n;
            uint64_t hi = lo + 2 * w < n ? lo + 2 * w : n;
            uint64_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) {
                if (src[j].key < src[i].key) dst[k++] = src[j++];   /* strict: left wins ties */
                else dst[k++] = src[i++];
            }
            while (i < mid) dst[k++] = src[i++];
            while (j < hi) dst[k++] = src[j++];
        }
        orc_pair_t *t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, n * sizeof(orc_pair_t));
    free(tmp);
}

static void reverse_pairs(orc_pair_t *a, uint64_t n)
{
    for (uint64_t i = 0, j = n; i + 1 < j; ++i) { --j; orc_pair_t t = a[i]; a[i] = a[j]; a[j] = t; }
}

/* test_device_radix_sort.cu:634-693 */
void orc_lsb_reference_ranks(const uint32_t *keys, uint64_t n, int begin_bit, int end_bit,
                             int descending, uint32_t *ranks)
{
    orc_pair_t *p = (orc_pair_t *)malloc((n ? n : 1) * sizeof(orc_pair_t));
    int num_bits = end_bit - begin_bit;
    for (uint64_t i = 0; i < n; ++i) {
        if (num_bits < 32) {                                   /* :650-656 */
            uint64_t base = keys[i];
            base &= ((1ull << num_bits) - 1) << begin_bit;
            p[i].key = (uint32_t)base;
        } else {
            p[i].key = keys[i];
        }
        p[i].value = (uint32_t)i;
    }
    if (descending) reverse_pairs(p, n);                       /* :673 */
    stable_sort_pairs(p, n);                                   /* :674 */
    if (descending) reverse_pairs(p, n);                       /* :675 */
    for (uint64_t i = 0; i < n; ++i) ranks[i] = p[i].value;    /* :685-689 */
    free(p);
}

void orc_lsb_sort_keys(const uint32_t *keys_in, uint32_t *keys_out, uint64_t n,
                       int begin_bit, int end_bit, int descending)
{
    uint32_t *ranks = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    orc_lsb_reference_ranks(keys_in, n, begin_bit, end_bit, descending, ranks);
    for (uint64_t i = 0; i < n; ++i) keys_out[i] = keys_in[ranks[i]];
    free(ranks);
}

void orc_lsb_sort_pairs(const uint32_t *keys_in, const uint32_t *vals_in,
                        uint32_t *keys_out, uint32_t *vals_out, uint64_t n,
                        int begin_bit, int end_bit, int descending)
{
    uint32_t *ranks = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    orc_lsb_reference_ranks(keys_in, n, begin_bit, end_bit, descending, ranks);
    for (uint64_t i = 0; i < n; ++i) {
        keys_out[i] = keys_in[ranks[i]];
        vals_out[i] = vals_in[ranks[i]];                       /* :888-889 */
    }
    free(ranks);
}

/* ---- 64-bit keys: the same InitializeSolution logic for
 * KeyT = unsigned long long / long long / double
 * (test_device_radix_sort.cu:1244-1265 instantiates them).                    */
/* util_type.cuh:966-974, 1009-1017, 1079-1089 with UnsignedBits = 64 bits */
uint64_t orc_twiddle_in_u64(uint64_t k) { return k; }
uint64_t orc_twiddle_in_i64(uint64_t k) { return k ^ 0x8000000000000000ull; }
uint64_t orc_twiddle_in_f64(uint64_t k) { return k ^ ((k >> 63) ? ~0ull : 0x8000000000000000ull); }
uint64_t orc_twiddle_out_f64(uint64_t k) { return k ^ ((k >> 63) ? 0x8000000000000000ull : ~0ull); }

typedef struct { uint64_t key; uint32_t value; } orc_pair64_t;

/* Pair::operator< : typed '<' for integers (test_device_radix_sort.cu:569-583),
 * and for floating point '<' plus "-0 sorts before +0" (:588-612).  NaNs are
 * never generated (test_util.h RandomBits rejects them).  key_type: 3 = u64,
 * 4 = i64, 5 = f64 (gs_key_type in include/gpusort.h).                       */
static int pair64_less(const orc_pair64_t *a, const orc_pair64_t *b, int key_type)
{
    if (key_type == 3) return a->key < b->key;
    if (key_type == 4) return (int64_t)a->key < (int64_t)b->key;
    double x, y;
    memcpy(&x, &a->key, 8);
    memcpy(&y, &b->key, 8);
    if (x < y) return 1;
    if (x > y) return 0;
    return (a->key >> 63) != 0 && (b->key >> 63) == 0;
}

static void stable_sort_pairs64(orc_pair64_t *a, uint64_t n, int key_type)
{
    if (n < 2) return;
    orc_pair64_t *tmp = (orc_pair64_t *)malloc(n * sizeof(orc_pair64_t));
    orc_pair64_t *src = a, *dst = tmp;
    for (uint64_t w = 1; w < n; w *= 2) {
        for (uint64_t lo = 0; lo < n; lo += 2 * w) {
            uint64_t mid = lo + w < n ? lo + w : n, hi = lo + 2 * w < n ? lo + 2 * w : n;
            uint64_t i = lo, j = mid, k = lo;
            while (i < mid && j < hi) dst[k++] = pair64_less(&src[j], &src[i], key_type) ? src[j++] : src[i++];
            while (i < mid) dst[k++] = src[i++];
            while (j < hi) dst[k++] = src[j++];
        }
        orc_pair64_t *t = src; src = dst; dst = t;
    }
    if (src != a) memcpy(a, src, n * sizeof(orc_pair64_t));
    free(tmp);
}

/* test_device_radix_sort.cu:634-693, KeyT 64 bits wide */
void orc_lsb_reference_ranks_u64(const uint64_t *keys, uint64_t n, int key_type, int begin_bit,
                                 int end_bit, int descending, uint32_t *ranks)
{
    orc_pair64_t *p = (orc_pair64_t *)malloc((n ? n : 1) * sizeof(orc_pair64_t));
    int num_bits = end_bit - begin_bit;
    for (uint64_t i = 0; i < n; ++i) {
        p[i].key = (num_bits < 64) ? (keys[i] & (((1ull << num_bits) - 1) << begin_bit)) : keys[i];   /* :650-661 */
        p[i].value = (uint32_t)i;
    }
    if (descending) for (uint64_t i = 0, j = n; i + 1 < j; ++i) { --j; orc_pair64_t t = p[i]; p[i] = p[j]; p[j] = t; }
    stable_sort_pairs64(p, n, key_type);
    if (descending) for (uint64_t i = 0, j = n; i + 1 < j; ++i) { --j; orc_pair64_t t = p[i]; p[i] = p[j]; p[j] = t; }
    for (uint64_t i = 0; i < n; ++i) ranks[i] = p[i].value;
    free(p);
}

/* ------------------------------------------------------ per-kernel goldens */

void orc_chunk_tiles(uint64_t num_tiles, uint32_t tiles_per_chunk, uint32_t c,
                     uint64_t *tile_begin, uint64_t *tile_end)
{
    uint64_t lo = (uint64_t)c * tiles_per_chunk, hi = lo + tiles_per_chunk;
    if (lo > num_tiles) lo = num_tiles;
    if (hi > num_tiles) hi = num_tiles;
    *tile_begin = lo;
    *tile_end = hi;
}

static inline uint32_t digit_of(uint32_t key, int shift, int bits, int descending)
{
    uint32_t k = descending ? ~key : key;
    return (k >> shift) & ((1u << bits) - 1u);
}

void orc_upsweep(const uint32_t *keys, uint64_t n, int shift, int bits, int descending,
                 uint32_t tile, uint32_t tiles_per_chunk, uint32_t grid, uint32_t *spine)
{
    uint64_t num_tiles = (n + tile - 1) / tile;
    uint32_t radix = 1u << bits;
    memset(spine, 0, (size_t)radix * grid * sizeof(uint32_t));
    for (uint32_t b = 0; b < grid; ++b) {
        uint64_t t0, t1;
        orc_chunk_tiles(num_tiles, tiles_per_chunk, b, &t0, &t1);
        uint64_t lo = t0 * tile, hi = t1 * tile;
        if (hi > n) hi = n;
        for (uint64_t i = lo; i < hi; ++i)
            spine[(uint64_t)digit_of(keys[i], shift, bits, descending) * grid + b]++;
    }
}

void orc_exclusive_scan(uint32_t *spine, uint64_t len)
{
    uint32_t run = 0;
    for (uint64_t i = 0; i < len; ++i) { uint32_t c = spine[i]; spine[i] = run; run += c; }
}

void orc_downsweep(const uint32_t *keys_in, const uint32_t *vals_in,
                   uint32_t *keys_out, uint32_t *vals_out, uint64_t n,
                   int shift, int bits, int descending)
{
    uint32_t radix = 1u << bits;
    uint64_t *off = (uint64_t *)calloc(radix + 1, sizeof(uint64_t));
    for (uint64_t i = 0; i < n; ++i) off[digit_of(keys_in[i], shift, bits, descending) + 1]++;
    for (uint32_t d = 0; d < radix; ++d) off[d + 1] += off[d];
    for (uint64_t i = 0; i < n; ++i) {
        uint64_t dst = off[digit_of(keys_in[i], shift, bits, descending)]++;
        keys_out[dst] = keys_in[i];
        if (vals_in) vals_out[dst] = vals_in[i];
    }
    free(off);
}

void orc_lsd_radix_sort(uint32_t *keys, uint32_t *vals, uint32_t *keys_tmp, uint32_t *vals_tmp,
                        uint64_t n, int begin_bit, int end_bit, int descending, int *result_in_tmp)
{
    int sel = 0;
    for (int shift = begin_bit; shift < end_bit; shift += 8) {
        int bits = end_bit - shift < 8 ? end_bit - shift : 8;
        if (sel == 0) orc_downsweep(keys, vals, keys_tmp, vals_tmp, n, shift, bits, descending);
        else orc_downsweep(keys_tmp, vals ? vals_tmp : NULL, keys, vals, n, shift, bits, descending);
        sel ^= 1;
    }
    *result_in_tmp = sel;
}

/* ------------------------------------------------------------ MSB checkers */

static int cmp_u32(const void *a, const void *b)
{
    uint32_t x = *(const uint32_t *)a, y = *(const uint32_t *)b;
    return (x > y) - (x < y);
}

/* msb/tests/test_sort_keys.cu:50-80: sorted reference + memcmp */
uint64_t orc_msb_check_keys(const uint32_t *keys_in, const uint32_t *keys_sorted, uint64_t n)
{
    uint32_t *ref = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    orc_lsb_sort_keys(keys_in, ref, n, 0, 32, 0);
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n; ++i)
        if (ref[i] != keys_sorted[i]) { bad = i + 1; break; }
    free(ref);
    return bad;
}

/* msb/tests/test_sort_pairs.cu:67-118: reference = values-then-keys double
 * stable sort (i.e. sort by (key, value)); candidate values are sorted
 * inside each equal-key run (:80-103) and then memcmp'ed (:106) */
uint64_t orc_msb_check_pairs(const uint32_t *keys_in, const uint32_t *vals_in,
                             const uint32_t *keys_sorted, uint32_t *vals_sorted, uint64_t n)
{
    uint32_t *k1 = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    uint32_t *v1 = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    uint32_t *k2 = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    uint32_t *v2 = (uint32_t *)malloc((n ? n : 1) * sizeof(uint32_t));
    /* SortPairs(d_values, d_keys) then SortPairs(d_keys, d_values), :38-41 */
    orc_lsb_sort_pairs(vals_in, keys_in, v1, k1, n, 0, 32, 0);
    orc_lsb_sort_pairs(k1, v1, k2, v2, n, 0, 32, 0);
    uint64_t bad = 0;
    for (uint64_t i = 0; i < n && !bad; ++i)
        if (k2[i] != keys_sorted[i]) bad = i + 1;
    if (!bad && n > 0) {
        uint64_t start = 0;
        for (uint64_t i = 1; i <= n; ++i) {
            if (i == n || keys_sorted[i] != keys_sorted[start]) {
                if (i - start > 1) qsort(vals_sorted + start, i - start, sizeof(uint32_t), cmp_u32);
                start = i;
            }
        }
        for (uint64_t i = 0; i < n && !bad; ++i)
            if (v2[i] != vals_sorted[i]) bad = i + 1;
    }
    free(k1); free(v1); free(k2); free(v2);
    return bad;
}

/* msb/tests/test_sort_pairs.cu:141-146,166-176 */
uint64_t orc_msb_check_pairs_enumerated(const uint32_t *keys_in, const uint32_t *keys_sorted,
                                        const uint32_t *vals_sorted, uint64_t n)
{
    uint64_t bad = orc_msb_check_keys(keys_in, keys_sorted, n);
    if (bad) return bad;
    uint64_t total = 0;
    for (uint64_t i = 0; i < n; ++i) {
        uint32_t v = vals_sorted[i];
        if (v >= n || keys_in[v] != keys_sorted[i]) return i + 1;
        total += v;
    }
    if (n && total != n * (n - 1) / 2) return n + 1;
    return 0;
}

/* -------------------------------------------------------------- properties */

void orc_multiset_checksum(const uint32_t *keys, uint64_t n, uint64_t *sum, uint64_t *xr)
{
    uint64_t s = 0, x = 0;
    for (uint64_t i = 0; i < n; ++i) { uint64_t h = orc_splitmix64(keys[i]); s += h; x ^= h; }
    *sum = s; *xr = x;
}

uint64_t orc_count_inversions_adjacent(const uint32_t *a, uint64_t n, int descending)
{
    uint64_t c = 0;
    for (uint64_t i = 0; i + 1 < n; ++i)
        c += descending ? (a[i] < a[i + 1]) : (a[i] > a[i + 1]);
    return c;
}
