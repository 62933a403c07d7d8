// msb_test.cpp -- drop-in counterpart of the reference's MSB stand-alone bench
// (msb/src/test.cu) on libgpusort.so: 2^28 u32 keys + u32 values (seed 0), one trial,
// keys-only then pairs through rdxsrt_unstable_sort, prints "Time Sort K" / "Time Sort KV"
// (msb/src/test.cu:49-57).  Optional first argument: log2 of the key count.
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <iostream>

#include "gpusort.hpp"

typedef unsigned int uint;
using namespace std;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)
#define SETUP_TIMING() hipEvent_t start, stop; CHECK(hipEventCreate(&start)); CHECK(hipEventCreate(&stop));
#define TIME_FUNC(f, t) { CHECK(hipEventRecord(start, 0)); f; CHECK(hipEventRecord(stop, 0)); \
    CHECK(hipEventSynchronize(stop)); CHECK(hipEventElapsedTime(&t, start, stop)); }

void test_sort_keys(unsigned int num_keys)
{
    uint *d_key_buf, *d_val_buf, *d_key_backup, *d_val_backup, *d_key_alt_buf, *d_val_alt_buf;
    const size_t bytes = sizeof(uint) * (size_t)num_keys;
    CHECK(hipMalloc(&d_key_buf, bytes)); CHECK(hipMalloc(&d_key_backup, bytes)); CHECK(hipMalloc(&d_key_alt_buf, bytes));
    CHECK(hipMalloc(&d_val_buf, bytes)); CHECK(hipMalloc(&d_val_backup, bytes)); CHECK(hipMalloc(&d_val_alt_buf, bytes));
    const int seed = 0;
    CHECK((hipError_t)gs_generate_u32(d_key_buf, num_keys, GS_GEN_UNIFORM, seed, 0, 1, 0));
    CHECK((hipError_t)gs_generate_u32(d_val_buf, num_keys, GS_GEN_UNIFORM, seed + 1, 0, 1, 0));
    CHECK(hipMemcpy(d_key_backup, d_key_buf, bytes, hipMemcpyDeviceToDevice));
    CHECK(hipMemcpy(d_val_backup, d_val_buf, bytes, hipMemcpyDeviceToDevice));

    SETUP_TIMING();
    int num_trials = 1;
    for (int i = 0; i < num_trials; i++) {
        float time_sort_k, time_sort_kv;
        TIME_FUNC((rdxsrt_unstable_sort<uint, gpusort::NullType, unsigned int>(d_key_buf, NULL, num_keys, d_key_alt_buf, NULL)),
                  time_sort_k);
        cout << "Time Sort K: " << time_sort_k << endl;
        CHECK(hipMemcpy(d_key_buf, d_key_backup, bytes, hipMemcpyDeviceToDevice));
        TIME_FUNC((rdxsrt_unstable_sort<uint, uint, unsigned int>(d_key_buf, d_val_buf, num_keys, d_key_alt_buf, d_val_alt_buf)),
                  time_sort_kv);
        cout << "Time Sort KV: " << time_sort_kv << endl;
        uint64_t *d_res; CHECK(hipMalloc(&d_res, 3 * sizeof(uint64_t)));
        uint64_t h_res[3];
        CHECK((hipError_t)gs_check_sorted_u32(d_key_buf, num_keys, 0, d_res, 0));
        CHECK(hipMemcpy(h_res, d_res, sizeof(h_res), hipMemcpyDeviceToHost));
        cout << "Adjacent inversions in result: " << h_res[0] << endl;
        CHECK(hipFree(d_res));
        CHECK(hipMemcpy(d_key_buf, d_key_backup, bytes, hipMemcpyDeviceToDevice));
        CHECK(hipMemcpy(d_val_buf, d_val_backup, bytes, hipMemcpyDeviceToDevice));
    }
    CHECK(hipFree(d_key_buf)); CHECK(hipFree(d_key_backup)); CHECK(hipFree(d_key_alt_buf));
    CHECK(hipFree(d_val_buf)); CHECK(hipFree(d_val_backup)); CHECK(hipFree(d_val_alt_buf));
}

int main(int argc, char **argv)
{
    const int lg = argc > 1 ? atoi(argv[1]) : 28;
    test_sort_keys(1u << lg);
    return 0;
}
