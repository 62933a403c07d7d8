// lsb_types.cpp -- typed DeviceRadixSort driver in the shape of the reference's CUB test
// (lsb/cub/test/test_device_radix_sort.cu:1244-1265: one Test<KeyT, ValueT> per type pair,
// ascending and descending, sizes shrinking to 1), compiled against gpusort.hpp.
// Host check: std::stable_sort over (key, index) pairs, as InitializeSolution does (:634-693).
//   usage: lsb_types [num_items]     exit code 0 iff every case matches
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <functional>
#include <random>
#include <vector>

#include "gpusort.hpp"

#define HIP_OK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("HIP error %d at %s:%d\n", (int)e_, __FILE__, __LINE__); exit(2); } } while (0)

template <typename T> static const char *name_of();
template <> const char *name_of<unsigned int>() { return "u32"; }
template <> const char *name_of<int>() { return "i32"; }
template <> const char *name_of<float>() { return "f32"; }
template <> const char *name_of<unsigned long long>() { return "u64"; }
template <> const char *name_of<long long>() { return "i64"; }
template <> const char *name_of<double>() { return "f64"; }
template <> const char *name_of<gpusort::NullType>() { return "-"; }
template <> const char *name_of<bool>() { return "bool"; }
template <> const char *name_of<char>() { return "char"; }
template <> const char *name_of<signed char>() { return "i8"; }
template <> const char *name_of<unsigned char>() { return "u8"; }
template <> const char *name_of<short>() { return "i16"; }
template <> const char *name_of<unsigned short>() { return "u16"; }
// the reference's 16-byte value type (lsb/cub/test/test_util.h:1004-1010), restated: assignable from an int, comparable
struct TestFoo {
    long long x; int y; short z; char w;
    TestFoo() : x(0), y(0), z(0), w(0) {}
    explicit TestFoo(int b) : x(b), y(b), z((short)b), w((char)b) {}
    bool operator!=(const TestFoo &o) const { return x != o.x || y != o.y || z != o.z || w != o.w; }
};
static_assert(sizeof(TestFoo) == 16, "TestFoo is a 16-byte value");
template <> const char *name_of<TestFoo>() { return "TestFoo"; }

// host copies of bool keys live in unsigned chars (std::vector<bool> is a bit set)
template <typename T> using HostT = typename std::conditional<std::is_same<T, bool>::value, unsigned char, T>::type;

template <typename KeyT> static void fill_keys(std::vector<HostT<KeyT>> &k, std::mt19937_64 &rng)
{
    for (auto &x : k) {
        do {                                              // test_util.h RandomBits: no NaNs
            unsigned long long bits = rng() & rng();      // a little entropy reduction -> duplicates
            if (std::is_same<KeyT, bool>::value) bits &= 1ull;    // a bool holds 0 or 1
            memcpy(&x, &bits, sizeof(KeyT));
        } while (x != x);
    }
}

template <typename KeyT, typename ValueT>
static int test_case(int n, bool descending)
{
    constexpr bool PAIRS = !std::is_same<ValueT, gpusort::NullType>::value;
    using StoreV = typename std::conditional<PAIRS, ValueT, unsigned int>::type;
    std::mt19937_64 rng(1234 + n);
    std::vector<HostT<KeyT>> h_keys(n);
    fill_keys<KeyT>(h_keys, rng);
    std::vector<StoreV> h_vals(n);
    for (int i = 0; i < n; ++i) h_vals[i] = (StoreV)i;

    std::vector<int> ranks(n);
    for (int i = 0; i < n; ++i) ranks[i] = i;
    auto less = [&](int a, int b) {
        if (h_keys[a] < h_keys[b]) return true;
        if (h_keys[a] > h_keys[b]) return false;
        if (std::is_floating_point<KeyT>::value) {        // -0 before +0 (:588-612)
            unsigned long long x = 0, y = 0;
            memcpy(&x, &h_keys[a], sizeof(KeyT)); memcpy(&y, &h_keys[b], sizeof(KeyT));
            const int sb = sizeof(KeyT) * 8 - 1;
            return ((x >> sb) & 1) && !((y >> sb) & 1);
        }
        return false;
    };
    if (descending) std::reverse(ranks.begin(), ranks.end());
    std::stable_sort(ranks.begin(), ranks.end(), less);
    if (descending) std::reverse(ranks.begin(), ranks.end());

    KeyT *d_k[2];
    StoreV *d_v[2] = {nullptr, nullptr};
    const size_t kb = (size_t)(n ? n : 1) * sizeof(KeyT), vb = (size_t)(n ? n : 1) * sizeof(StoreV);
    HIP_OK(hipMalloc(&d_k[0], kb)); HIP_OK(hipMalloc(&d_k[1], kb));
    HIP_OK(hipMemcpy(d_k[0], h_keys.data(), (size_t)n * sizeof(KeyT), hipMemcpyHostToDevice));
    if (PAIRS) {
        HIP_OK(hipMalloc(&d_v[0], vb)); HIP_OK(hipMalloc(&d_v[1], vb));
        HIP_OK(hipMemcpy(d_v[0], h_vals.data(), (size_t)n * sizeof(StoreV), hipMemcpyHostToDevice));
    }
    gpusort::DoubleBuffer<KeyT> keys(d_k[0], d_k[1]);
    gpusort::DoubleBuffer<StoreV> vals(d_v[0], d_v[1]);
    void *d_temp = nullptr;
    size_t temp_bytes = 0;
    auto run = [&]() -> hipError_t {
        if constexpr (PAIRS)
            return descending ? gpusort::DeviceRadixSort::SortPairsDescending(d_temp, temp_bytes, keys, vals, n)
                              : gpusort::DeviceRadixSort::SortPairs(d_temp, temp_bytes, keys, vals, n);
        else
            return descending ? gpusort::DeviceRadixSort::SortKeysDescending(d_temp, temp_bytes, keys, n)
                              : gpusort::DeviceRadixSort::SortKeys(d_temp, temp_bytes, keys, n);
    };
    HIP_OK(run());
    HIP_OK(hipMalloc(&d_temp, temp_bytes ? temp_bytes : 1));
    HIP_OK(run());
    HIP_OK(hipDeviceSynchronize());
    std::vector<HostT<KeyT>> out_k(n);
    std::vector<StoreV> out_v(n);
    HIP_OK(hipMemcpy(out_k.data(), keys.Current(), (size_t)n * sizeof(KeyT), hipMemcpyDeviceToHost));
    if (PAIRS) HIP_OK(hipMemcpy(out_v.data(), vals.Current(), (size_t)n * sizeof(StoreV), hipMemcpyDeviceToHost));
    int bad = 0;
    for (int i = 0; i < n && !bad; ++i) {
        if (memcmp(&out_k[i], &h_keys[ranks[i]], sizeof(KeyT)) != 0) bad = 1;
        if (PAIRS && out_v[i] != (StoreV)ranks[i]) bad = 1;
    }
    printf("%s keys, %s values, n=%d, %s: %s\n", name_of<KeyT>(), name_of<ValueT>(), n,
           descending ? "descending" : "ascending", bad ? "FAIL" : "CORRECT");
    HIP_OK(hipFree(d_k[0])); HIP_OK(hipFree(d_k[1])); HIP_OK(hipFree(d_temp));
    if (PAIRS) { HIP_OK(hipFree(d_v[0])); HIP_OK(hipFree(d_v[1])); }
    return bad;
}

template <typename KeyT, typename ValueT>
static int test_type(int max_items)
{
    int bad = 0;
    for (int n = max_items; ; n = (n + 31) / 32) {                 // :1034-1046
        bad += test_case<KeyT, ValueT>(n, false);
        bad += test_case<KeyT, ValueT>(n, true);
        if (n <= 1) break;
    }
    bad += test_case<KeyT, ValueT>(0, false);
    return bad;
}

// segmented: random cut points (test_device_radix_sort.cu:1092-1100), std::stable_sort per segment on the host
template <typename KeyT>
static int test_segmented(int n, int num_segments, bool descending, const char *name)
{
    std::mt19937_64 rng(99 + n + num_segments);
    std::vector<KeyT> h_keys(n);
    for (auto &x : h_keys) x = (KeyT)(rng() & rng());
    std::vector<int> offs(num_segments + 1);
    offs[0] = 0; offs[num_segments] = n;
    for (int i = 1; i < num_segments; ++i) offs[i] = (int)(rng() % (unsigned long long)(n + 1));
    std::sort(offs.begin(), offs.end());
    std::vector<KeyT> want(h_keys);
    for (int i = 0; i < num_segments; ++i) {
        if (descending) std::stable_sort(want.begin() + offs[i], want.begin() + offs[i + 1], std::greater<KeyT>());
        else std::stable_sort(want.begin() + offs[i], want.begin() + offs[i + 1]);
    }
    KeyT *d_k[2]; int *d_offs;
    HIP_OK(hipMalloc(&d_k[0], (size_t)n * sizeof(KeyT))); HIP_OK(hipMalloc(&d_k[1], (size_t)n * sizeof(KeyT)));
    HIP_OK(hipMalloc(&d_offs, (size_t)(num_segments + 1) * 4));
    HIP_OK(hipMemcpy(d_k[0], h_keys.data(), (size_t)n * sizeof(KeyT), hipMemcpyHostToDevice));
    HIP_OK(hipMemcpy(d_offs, offs.data(), (size_t)(num_segments + 1) * 4, hipMemcpyHostToDevice));
    gpusort::DoubleBuffer<KeyT> keys(d_k[0], d_k[1]);
    void *d_temp = nullptr; size_t temp_bytes = 0;
    auto run = [&]() {
        return descending ? gpusort::DeviceSegmentedRadixSort::SortKeysDescending(d_temp, temp_bytes, keys, n, num_segments, d_offs, d_offs + 1)
                          : gpusort::DeviceSegmentedRadixSort::SortKeys(d_temp, temp_bytes, keys, n, num_segments, d_offs, d_offs + 1);
    };
    HIP_OK(run());
    HIP_OK(hipMalloc(&d_temp, temp_bytes ? temp_bytes : 1));
    HIP_OK(run());
    HIP_OK(hipDeviceSynchronize());
    std::vector<KeyT> out(n);
    HIP_OK(hipMemcpy(out.data(), keys.Current(), (size_t)n * sizeof(KeyT), hipMemcpyDeviceToHost));
    const int bad = memcmp(out.data(), want.data(), (size_t)n * sizeof(KeyT)) != 0;
    printf("segmented %s keys, n=%d, %d segments, %s: %s\n", name, n, num_segments, descending ? "descending" : "ascending",
           bad ? "FAIL" : "CORRECT");
    HIP_OK(hipFree(d_k[0])); HIP_OK(hipFree(d_k[1])); HIP_OK(hipFree(d_offs)); HIP_OK(hipFree(d_temp));
    return bad;
}

int main(int argc, char **argv)
{
    const int n = argc > 1 ? atoi(argv[1]) : 1000003;
    int bad = 0;
    bad += test_type<unsigned int, gpusort::NullType>(n);
    bad += test_type<int, unsigned int>(n);
    bad += test_type<float, unsigned int>(n);
    bad += test_type<unsigned int, unsigned long long>(n);
    bad += test_type<unsigned long long, gpusort::NullType>(n);
    bad += test_type<long long, unsigned int>(n);
    bad += test_type<double, unsigned long long>(n);
    // 8- and 16-bit keys, 1- / 2- / 16-byte values (test_device_radix_sort.cu:930-945,1244-1250)
    const int ns = n < 300007 ? n : 300007;
    bad += test_type<bool, gpusort::NullType>(ns);
    bad += test_type<char, char>(ns);
    bad += test_type<signed char, unsigned int>(ns);
    bad += test_type<unsigned char, unsigned long long>(ns);
    bad += test_type<short, short>(ns);
    bad += test_type<unsigned short, TestFoo>(ns);
    bad += test_type<unsigned short, gpusort::NullType>(ns);
    bad += test_type<unsigned int, TestFoo>(ns);
    bad += test_type<long long, TestFoo>(ns);
    bad += test_type<double, unsigned short>(ns);
    bad += test_segmented<unsigned int>(n, 1, false, "u32");
    bad += test_segmented<unsigned int>(n, 37, true, "u32");
    bad += test_segmented<unsigned int>(n, 5000, false, "u32");
    bad += test_segmented<unsigned long long>(n, 1, true, "u64");
    bad += test_segmented<unsigned long long>(n, 37, false, "u64");
    bad += test_segmented<long long>(n, 5000, true, "i64");
    printf("%s\n", bad ? "SOME CASES FAILED" : "ALL CORRECT");
    return bad ? 1 : 0;
}
