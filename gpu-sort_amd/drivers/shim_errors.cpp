// shim_errors -- the drop-in shims of include/gpusort.hpp must not hand back unsorted data when something failed
// (the reference's rdxsrt_unstable_sort has no error channel: msb/src/sort/gpu_radix_sort.h:197-507).
//   shim_errors nodevice   no GPU needed: the scratch allocation fails -> {nullptr, nullptr} for 32- and 64-bit keys
//   shim_errors gpu        a data manager sized for fewer keys -> {nullptr, nullptr}; a fitting one -> sorted input arrays
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <vector>
#include "gpusort.hpp"

static int fails = 0;
#define EXPECT(c) do { if (!(c)) { std::printf("FAIL line %d: %s\n", __LINE__, #c); ++fails; } } while (0)

int main(int argc, char **argv)
{
    const bool gpu = argc > 1 && std::strcmp(argv[1], "gpu") == 0;
    if (!gpu) {
        // pointers that are never dereferenced: every path fails before a kernel is launched
        unsigned int *k = reinterpret_cast<unsigned int *>(0x1000), *ko = reinterpret_cast<unsigned int *>(0x2000);
        RDXSRT_GPUDataManager dm(1000, false);
        EXPECT(!dm.ok());
        auto a = rdxsrt_unstable_sort<unsigned int, gpusort::NullType, unsigned int>(k, nullptr, 1000u, ko, nullptr, nullptr, &dm);
        EXPECT(a.sorted_keys == nullptr && a.sorted_values == nullptr);
        auto b = rdxsrt_unstable_sort<unsigned int, gpusort::NullType, unsigned int>(k, nullptr, 1000u, ko, nullptr);
        EXPECT(b.sorted_keys == nullptr);
        unsigned long long *k8 = reinterpret_cast<unsigned long long *>(0x1000), *k8o = reinterpret_cast<unsigned long long *>(0x3000);
        auto c = rdxsrt_unstable_sort<unsigned long long, gpusort::NullType, unsigned int>(k8, nullptr, 1000u, k8o, nullptr);
        EXPECT(c.sorted_keys == nullptr);
        // the host-pointer wrappers: every runtime call is checked; on failure the outputs stay untouched and stderr says why
        std::vector<unsigned int> hk(1000, 7u), hv(1000, 9u), ok(1000, 0xabababab), ov(1000, 0xcdcdcdcd);
        rdxsrt_unstable_sort_keys<unsigned int>(hk.data(), 1000ull, ok.data());
        bool untouched = true;
        for (unsigned int x : ok) untouched = untouched && x == 0xabababab;
        EXPECT(untouched);
        rdxsrt_unstable_sort_pairs<unsigned int, unsigned int>(hk.data(), hv.data(), 1000ull, ok.data(), ov.data());
        for (unsigned int x : ok) untouched = untouched && x == 0xabababab;
        for (unsigned int x : ov) untouched = untouched && x == 0xcdcdcdcd;
        EXPECT(untouched);
    } else {
        const unsigned int n = 300000;
        std::vector<unsigned int> h(n);
        for (unsigned int i = 0; i < n; ++i) h[i] = (i * 2654435761u) ^ (i >> 3);
        unsigned int *k = nullptr, *ko = nullptr;
        EXPECT(hipMalloc(&k, n * 4) == hipSuccess && hipMalloc(&ko, n * 4) == hipSuccess);
        EXPECT(hipMemcpy(k, h.data(), n * 4, hipMemcpyHostToDevice) == hipSuccess);
        RDXSRT_GPUDataManager small(1000, false);      // sized for 1000 keys, used for 300000
        EXPECT(small.ok());
        auto a = rdxsrt_unstable_sort<unsigned int, gpusort::NullType, unsigned int>(k, nullptr, n, ko, nullptr, nullptr, &small);
        EXPECT(a.sorted_keys == nullptr && a.sorted_values == nullptr);
        RDXSRT_GPUDataManager fits(n, false);
        auto b = rdxsrt_unstable_sort<unsigned int, gpusort::NullType, unsigned int>(k, nullptr, n, ko, nullptr, nullptr, &fits);
        EXPECT(b.sorted_keys == k);                    // 32-bit keys: the result is in the input array
        std::vector<unsigned int> r(n);
        EXPECT(hipMemcpy(r.data(), b.sorted_keys ? b.sorted_keys : k, n * 4, hipMemcpyDeviceToHost) == hipSuccess);
        bool sorted = true;
        for (unsigned int i = 1; i < n; ++i) sorted = sorted && r[i - 1] <= r[i];
        EXPECT(sorted);
        (void)hipFree(k); (void)hipFree(ko);
    }
    std::printf(fails ? "FAILED\n" : "OK\n");
    return fails ? 1 : 0;
}
