// msb_harness.cpp -- counterpart of the reference's MSB test/benchmark binary
// (msb/tests/main.cu, test_sort_keys.cu, test_sort_pairs.cu) on libgpusort.so.
//   CLI        -r <repeats> -k <keys> -p <pairs> -s <MB>   (msb/tests/main.cu:41-54)
//              --gtest_filter=<substring>                  (selects tests by name)
//   tests      Sort_Keys.Entropy_{UINT,UINT64,DOUBLE}[_NUMKEYS], Sort_Pairs.{UINT,UINT64}_{UINT,UINT64}
//              [_NUMKEYS] (test_sort_keys.cu:154-195, test_sort_pairs.cu:223-281); 64-bit keys or
//              values go through rdxsrt_unstable_sort's wide (LSB) path
//   per run    12 entropy levels {1..11,0} x repeats (test_sort_keys.cu:126), the
//              "--- SORTKEYS.ENTROPIES ..." line (:139), host arrays through
//              rdxsrt_unstable_sort_keys/_pairs (gpu_radix_sort.h:511-587)
//   check      u32 keys: memcmp against the LSB sort of the same input on the same GPU, as the
//              reference checks against CUB (test_sort_keys.cu:50-80); 64-bit keys: memcmp against
//              std::sort on the host in radix order (bit patterns; doubles through the sign
//              transform, so NaN patterns are placed like CUB places them); pairs: the enumerated
//              value -> key map and the value sum (test_sort_pairs.cu:141-146,166-176)
//   output     the tab-separated profile table at exit (msb/tests/main.cu:65-69)
// Inputs: the counter-based entropy-AND generator of gs_generate_u32 (msb/tests/data_gen.h:43-75).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cmath>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string>
#include <type_traits>
#include <unistd.h>
#include <vector>

#include "gpusort.hpp"

typedef unsigned int uint;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(2); } } while (0)

static unsigned sort_repeat_count = 3, sort_keys_prob_size_in_mb = 0, sort_keys_default_prob_size = 200000,
                sort_pairs_default_prob_size = 100000;

struct Row { std::string profile; unsigned num_keys; int entr_and; double bit_entr, rand_ms, sort_ms, check_sort_ms, verif_ms; };
static std::vector<Row> g_rows;

static double now_ms() { return std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now().time_since_epoch()).count(); }

static void generate_random_keys(uint *h_keys, size_t n, unsigned long long seed, int entropy_level)
{
    uint *d = nullptr;
    CHECK(hipMalloc(&d, sizeof(uint) * (n ? n : 1)));
    CHECK((hipError_t)gs_generate_u32(d, n, GS_GEN_ENTROPY_AND, seed, 0, entropy_level, 0));
    CHECK(hipMemcpy(h_keys, d, sizeof(uint) * n, hipMemcpyDeviceToHost));
    CHECK(hipFree(d));
}

// the reference sorts the check copy with CUB on the same GPU; here: the LSB path
static double check_sort_keys(const uint *h_in, size_t n, uint *h_out)
{
    gpusort::DoubleBuffer<uint> d_keys;
    CHECK(hipMalloc(&d_keys.d_buffers[0], sizeof(uint) * (n ? n : 1)));
    CHECK(hipMalloc(&d_keys.d_buffers[1], sizeof(uint) * (n ? n : 1)));
    CHECK(hipMemcpy(d_keys.d_buffers[0], h_in, sizeof(uint) * n, hipMemcpyHostToDevice));
    size_t bytes = 0; void *tmp = nullptr;
    CHECK(gpusort::DeviceRadixSort::SortKeys(tmp, bytes, d_keys, (int)n));
    CHECK(hipMalloc(&tmp, bytes));
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a, 0));
    CHECK(gpusort::DeviceRadixSort::SortKeys(tmp, bytes, d_keys, (int)n));
    CHECK(hipEventRecord(b, 0)); CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b));
    CHECK(hipMemcpy(h_out, d_keys.Current(), sizeof(uint) * n, hipMemcpyDeviceToHost));
    CHECK(hipFree(d_keys.d_buffers[0])); CHECK(hipFree(d_keys.d_buffers[1])); CHECK(hipFree(tmp));
    CHECK(hipEventDestroy(a)); CHECK(hipEventDestroy(b));
    return ms;
}

// 64-bit keys are 2 x 32 random bits, as the reference draws them (data_gen.h:33-41: num_keys * sizeof(KeyT) / 4 words)
template <typename K>
static void generate_keys_of(K *h_keys, size_t n, unsigned long long seed, int entropy_level)
{
    generate_random_keys(reinterpret_cast<uint *>(h_keys), n * (sizeof(K) / sizeof(uint)), seed, entropy_level);
}

typedef unsigned long long u64;
static inline u64 radix_order_bits(u64 bits, bool is_double)
{
    return is_double ? (bits ^ ((bits >> 63) ? ~0ull : 0x8000000000000000ull)) : bits;
}

// expected order of the keys: GPU LSB sort for 32-bit keys (the reference checks against CUB),
// host std::sort on the radix-order bit patterns for 64-bit keys
template <typename K>
static double check_sort_keys_of(const K *h_in, size_t n, K *h_out)
{
    if constexpr (sizeof(K) == 4) {
        return check_sort_keys(reinterpret_cast<const uint *>(h_in), n, reinterpret_cast<uint *>(h_out));
    } else {
        const double t0 = now_ms();
        constexpr bool dbl = std::is_floating_point<K>::value;
        std::vector<u64> b(n);
        std::memcpy(b.data(), h_in, n * 8);
        for (auto &x : b) x = radix_order_bits(x, dbl);
        std::sort(b.begin(), b.end());
        for (auto &x : b) x = dbl ? (x ^ ((x >> 63) ? 0x8000000000000000ull : ~0ull)) : x;
        std::memcpy(h_out, b.data(), n * 8);
        return now_ms() - t0;
    }
}

template <typename K>
static bool test_sort_keys(unsigned num_keys, int entropy_level, Row &row)
{
    std::vector<K> in(num_keys), cpy(num_keys), sorted(num_keys), ref(num_keys);
    double t0 = now_ms();
    generate_keys_of<K>(in.data(), num_keys, 0, entropy_level);
    row.rand_ms = now_ms() - t0;
    cpy = in;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a, 0));
    rdxsrt_unstable_sort_keys<K>(in.data(), num_keys, sorted.data());
    CHECK(hipEventRecord(b, 0)); CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b)); row.sort_ms = ms;
    t0 = now_ms();
    row.check_sort_ms = check_sort_keys_of<K>(cpy.data(), num_keys, ref.data());
    const bool ok = num_keys == 0 || std::memcmp(ref.data(), sorted.data(), sizeof(K) * num_keys) == 0;   // binary: NaN != NaN
    row.verif_ms = now_ms() - t0;
    if (!ok)
        for (unsigned i = 0; i < num_keys; ++i)
            if (std::memcmp(&ref[i], &sorted[i], sizeof(K)) != 0) { printf("Mismatch at index %u\n", i); break; }
    return ok;
}

template <typename K, typename V>
static bool test_sort_pairs(unsigned num_pairs, int entropy_level, Row &row)
{
    std::vector<K> kin(num_pairs), ks(num_pairs), ref(num_pairs), v2k(num_pairs);
    std::vector<V> vin(num_pairs), vs(num_pairs);
    double t0 = now_ms();
    generate_keys_of<K>(kin.data(), num_pairs, 0, entropy_level);
    row.rand_ms = now_ms() - t0;
    for (unsigned i = 0; i < num_pairs; ++i) { vin[i] = (V)i; v2k[i] = kin[i]; }     // generate_enumerated_values
    std::vector<K> kcpy = kin;
    std::vector<V> vcpy = vin;
    hipEvent_t a, b; CHECK(hipEventCreate(&a)); CHECK(hipEventCreate(&b));
    CHECK(hipEventRecord(a, 0));
    rdxsrt_unstable_sort_pairs<K, V>(kcpy.data(), vcpy.data(), num_pairs, ks.data(), vs.data());
    CHECK(hipEventRecord(b, 0)); CHECK(hipEventSynchronize(b));
    float ms = 0; CHECK(hipEventElapsedTime(&ms, a, b)); row.sort_ms = ms;
    t0 = now_ms();
    row.check_sort_ms = check_sort_keys_of<K>(kin.data(), num_pairs, ref.data());
    bool ok = num_pairs == 0 || std::memcmp(ref.data(), ks.data(), sizeof(K) * num_pairs) == 0;
    size_t total = 0;
    for (unsigned i = 0; i < num_pairs && ok; ++i) {
        total += (size_t)vs[i];
        ok = (size_t)vs[i] < num_pairs && std::memcmp(&v2k[(size_t)vs[i]], &ks[i], sizeof(K)) == 0;
    }
    ok = ok && total == (size_t)num_pairs * ((size_t)num_pairs - (num_pairs ? 1 : 0)) / 2;
    row.verif_ms = now_ms() - t0;
    return ok;
}

typedef bool (*OneTest)(unsigned, int, Row &);

static bool run_test_over_entropies(unsigned n, bool pairs, const char *profile, OneTest one, int key_bits = 32)
{
    static const int levels[] = {1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 0};
    bool all_ok = true;
    for (int level : levels) {
        const double p = std::pow(0.5, (double)level);
        const double bit_e = level <= 0 ? 0.0 : (-p * std::log2(p) - (1 - p) * std::log2(1 - p));
        const double entropy = (double)key_bits * bit_e;
        for (unsigned i = 0; i < sort_repeat_count; ++i) {
            printf(" --- %s.ENTROPIES (Entropy Lev. (bit entropy): %2d (%6.3f), Iteration: %2u - key_count: %u)---\n",
                   pairs ? "SORTPAIRS" : "SORTKEYS", level, entropy, i, n);
            Row row{profile, n, level, level <= 0 ? 0.0 : entropy, 0, 0, 0, 0};
            const bool ok = one(n, level, row);
            g_rows.push_back(row);
            if (!ok) { printf("Test: \n - key count: %u\n - entropy level: %d\n", n, level); all_ok = false; }
        }
    }
    return all_ok;
}

static bool numkeys_sweep(unsigned nmax, bool pairs, const char *profile, OneTest one, int key_bits = 32)
{
    bool ok = true;
    for (long double i = 100000; i < nmax; i *= 1.25892541179416721042395410639580060609361740946L)
        ok = run_test_over_entropies((unsigned)i, pairs, profile, one, key_bits) && ok;
    return run_test_over_entropies(nmax, pairs, profile, one, key_bits) && ok;
}

struct Test { const char *name; bool (*fn)(); bool built; };
static unsigned keys_size(size_t kb = sizeof(uint)) { return sort_keys_prob_size_in_mb ? sort_keys_prob_size_in_mb * (1000000 / kb) : sort_keys_default_prob_size; }
static unsigned pairs_size(size_t kb = sizeof(uint)) { return sort_keys_prob_size_in_mb ? sort_keys_prob_size_in_mb * (1000000 / kb) : sort_pairs_default_prob_size; }
static bool t_keys_uint() { return run_test_over_entropies(keys_size(), false, "sort_keys_UINT", test_sort_keys<uint>); }
static bool t_keys_uint64() { return run_test_over_entropies(keys_size(8), false, "sort_keys_UINT64", test_sort_keys<u64>, 64); }
static bool t_keys_double() { return run_test_over_entropies(keys_size(8), false, "sort_keys_DOUBLE", test_sort_keys<double>, 64); }
static bool t_keys_uint_numkeys() { return numkeys_sweep(keys_size(), false, "sort_keys_numkeys_UINT", test_sort_keys<uint>); }
static bool t_keys_uint64_numkeys() { return numkeys_sweep(keys_size(8), false, "sort_keys_numkeys_UINT64", test_sort_keys<u64>, 64); }
static bool t_pairs_uint_uint() { return run_test_over_entropies(pairs_size(), true, "sort_pairs_UINT_UINT", test_sort_pairs<uint, uint>); }
static bool t_pairs_uint_uint64() { return run_test_over_entropies(pairs_size(), true, "sort_pairs_UINT_UINT64", test_sort_pairs<uint, u64>); }
static bool t_pairs_uint64_uint() { return run_test_over_entropies(pairs_size(8), true, "sort_pairs_UINT64_UINT", test_sort_pairs<u64, uint>, 64); }
static bool t_pairs_uint64_uint64() { return run_test_over_entropies(pairs_size(8), true, "sort_pairs_UINT64_UINT64", test_sort_pairs<u64, u64>, 64); }
static bool t_pairs_uint_uint_numkeys() { return numkeys_sweep(pairs_size(), true, "sort_pairs_numkeys_UINT_UINT", test_sort_pairs<uint, uint>); }
static bool t_pairs_uint64_uint64_numkeys() { return numkeys_sweep(pairs_size(8), true, "sort_pairs_numkeys_UINT64_UINT64", test_sort_pairs<u64, u64>, 64); }

int main(int argc, char **argv)
{
    int deviceCount = 0;
    CHECK(hipGetDeviceCount(&deviceCount));
    printf("\n --- DEVICES ---\n");
    for (int d = 0; d < deviceCount; ++d) {
        hipDeviceProp_t pr; CHECK(hipGetDeviceProperties(&pr, d));
        printf(" -> device %d (%s) has %d CUs (LDS: %zu B) @%d MHz, memory-bus: %d bit @%d MHz.\n", d, pr.name,
               pr.multiProcessorCount, pr.sharedMemPerBlock, pr.clockRate / 1000, pr.memoryBusWidth, pr.memoryClockRate / 1000);
    }
    printf(" --- DEVICES ---\n");
    CHECK(hipSetDevice(0));                       // msb/tests/main.cu:30-31 hard-selects device 0

    std::string filter;
    std::vector<char *> rest;
    for (int i = 0; i < argc; ++i) {
        if (std::strncmp(argv[i], "--gtest_filter=", 15) == 0) filter = argv[i] + 15; else rest.push_back(argv[i]);
    }
    opterr = 0;
    int c, rc = (int)rest.size();
    while ((c = getopt(rc, rest.data(), "r:k:p:s:")) != -1) {
        switch (c) {
            case 'r': sort_repeat_count = atoi(optarg); break;
            case 'k': sort_keys_default_prob_size = atoi(optarg); break;
            case 'p': sort_pairs_default_prob_size = atoi(optarg); break;
            case 's': sort_keys_prob_size_in_mb = atoi(optarg); break;
            default: break;
        }
    }
    const Test tests[] = {
        {"Sort_Keys.Entropy_UINT", t_keys_uint, true},
        {"Sort_Keys.Entropy_UINT64", t_keys_uint64, true},
        {"Sort_Keys.Entropy_DOUBLE", t_keys_double, true},
        {"Sort_Keys.Entropy_UINT_NUMKEYS", t_keys_uint_numkeys, true},
        {"Sort_Keys.Entropy_UINT64_NUMKEYS", t_keys_uint64_numkeys, true},
        {"Sort_Pairs.UINT_UINT", t_pairs_uint_uint, true},
        {"Sort_Pairs.UINT_UINT64", t_pairs_uint_uint64, true},
        {"Sort_Pairs.UINT64_UINT", t_pairs_uint64_uint, true},
        {"Sort_Pairs.UINT64_UINT64", t_pairs_uint64_uint64, true},
        {"Sort_Pairs.UINT_UINT_NUMKEYS", t_pairs_uint_uint_numkeys, true},
        {"Sort_Pairs.UINT64_UINT64_NUMKEYS", t_pairs_uint64_uint64_numkeys, true},
    };
    int ran = 0, failed = 0;
    for (const Test &t : tests) {
        std::string f = filter;
        while (!f.empty() && f.back() == '*') f.pop_back();
        if (!f.empty() && std::string(t.name).find(f) == std::string::npos) continue;
        if (!t.built) { printf("[  SKIPPED ] %s (64-bit keys/values are not built: 32-bit library)\n", t.name); continue; }
        printf("[ RUN      ] %s\n", t.name);
        ++ran;
        const bool ok = t.fn();
        printf(ok ? "[       OK ] %s\n" : "[  FAILED  ] %s\n", t.name);
        failed += ok ? 0 : 1;
    }
    if (ran > 0) {
        printf("\n --- PROFILE OVERVIEW ---\n");
        printf("run\tnum_keys\tentr_and\tbit_entr\trand_data_ms\tsort_total_ms\tcheck_sort_ms\tcpu_verif_ms\n");
        for (const Row &r : g_rows)
            printf("%s\t%u\t%d\t%.3f\t%.3f\t%.3f\t%.3f\t%.3f\n", r.profile.c_str(), r.num_keys, r.entr_and, r.bit_entr,
                   r.rand_ms, r.sort_ms, r.check_sort_ms, r.verif_ms);
        printf(" --- PROFILE OVERVIEW ---\n");
    }
    printf("[==========] %d tests ran, %d failed.\n", ran, failed);
    return failed ? 1 : 0;
}
