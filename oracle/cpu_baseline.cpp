// cpu_baseline.cpp -- the "reference's CPU std::sort" leg.
// TEST/BENCH INFRASTRUCTURE ONLY (see oracle.h).  The reference has no CPU
// sort path; its only CPU sorts are its test oracles: std::stable_sort
// (lsb/cub/test/test_device_radix_sort.cu:674) and std::sort on value runs
// (msb/tests/test_sort_pairs.cu:89).  These are timed here on host cores.
#include <algorithm>
#include <chrono>
#include <cstdint>
#include <cstring>
#include <vector>
#include <thread>

namespace {
struct PairKV { uint32_t key, value; };
inline bool operator<(const PairKV &a, const PairKV &b) { return a.key < b.key; }
double now_s() {
    return std::chrono::duration<double>(std::chrono::steady_clock::now().time_since_epoch()).count();
}
}  // namespace

extern "C" {

// in-place std::sort of n u32 keys on one thread; returns seconds
double orc_std_sort_u32(uint32_t *keys, uint64_t n) {
    double t0 = now_s();
    std::sort(keys, keys + n);
    return now_s() - t0;
}

// std::stable_sort of (key,value) structs by key on one thread; returns seconds
double orc_std_stable_sort_pairs(uint32_t *keys, uint32_t *vals, uint64_t n) {
    std::vector<PairKV> p(n);
    for (uint64_t i = 0; i < n; ++i) p[i] = PairKV{keys[i], vals[i]};
    double t0 = now_s();
    std::stable_sort(p.begin(), p.end());
    double dt = now_s() - t0;
    for (uint64_t i = 0; i < n; ++i) { keys[i] = p[i].key; vals[i] = p[i].value; }
    return dt;
}

// all-cores variant: T threads std::sort equal slices, then a log2(T) tree of
// std::inplace_merge; returns seconds.  threads<=0 -> hardware_concurrency.
double orc_std_sort_u32_mt(uint32_t *keys, uint64_t n, int threads, int *threads_used) {
    int T = threads > 0 ? threads : (int)std::thread::hardware_concurrency();
    if (T < 1) T = 1;
    while ((uint64_t)T > n && T > 1) T /= 2;
    int P = 1; while (P * 2 <= T) P *= 2;   // power of two slices
    if (threads_used) *threads_used = P;
    double t0 = now_s();
    std::vector<uint64_t> cut(P + 1);
    for (int i = 0; i <= P; ++i) cut[i] = n * (uint64_t)i / P;
    {
        std::vector<std::thread> th;
        for (int i = 0; i < P; ++i)
            th.emplace_back([&, i] { std::sort(keys + cut[i], keys + cut[i + 1]); });
        for (auto &t : th) t.join();
    }
    for (int w = 1; w < P; w *= 2) {
        std::vector<std::thread> th;
        for (int i = 0; i + w < P; i += 2 * w) {
            int hi = std::min(i + 2 * w, P);
            th.emplace_back([&, i, w, hi] {
                std::inplace_merge(keys + cut[i], keys + cut[i + w], keys + cut[hi]);
            });
        }
        for (auto &t : th) t.join();
    }
    return now_s() - t0;
}

int orc_hardware_threads() { return (int)std::thread::hardware_concurrency(); }

}  // extern "C"
