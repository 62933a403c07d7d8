// lsb_sort.cpp -- drop-in counterpart of the reference's LSB benchmark driver (lsb/sort.cu)
// on libgpusort.so: same CLI (--n --t --device, lsb/sort.cu:91-105), same two timed
// wrappers (sortPairsGPU ascending, :25-47; sortKeysGPU sized with SortKeys but run with
// SortKeysDescending and printing the first 32 keys, :49-76), float keys uniform in (0,1]
// and uint values (:125-131), input restored device-to-device between sorts (:141-146),
// one JSON line per trial (:148-151).  cuRAND XORWOW is replaced by the counter-based
// generator of gs_generate_u32: the u32 draw is mapped to a float in (0,1]
// the way curandGenerateUniform does ((x + 1) * 2^-32).
#include <hip/hip_runtime.h>

#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <iostream>
#include <string>
#include <vector>

#include "gpusort.hpp"

using gpusort::DeviceRadixSort;
using gpusort::DoubleBuffer;
typedef unsigned int uint;

#define CHECK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { \
    fprintf(stderr, "%s:%d %s\n", __FILE__, __LINE__, hipGetErrorString(e_)); exit(1); } } while (0)

// lsb/gpu_utils.h:3-11
#define SETUP_TIMING() hipEvent_t start, stop; CHECK(hipEventCreate(&start)); CHECK(hipEventCreate(&stop));
#define TIME_FUNC(f, t) { CHECK(hipEventRecord(start, 0)); f; CHECK(hipEventRecord(stop, 0)); \
    CHECK(hipEventSynchronize(stop)); CHECK(hipEventElapsedTime(&t, start, stop)); }

__global__ void u32_to_unit_float(float *keys, int n)
{
    const int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i < n) {
        const unsigned int x = reinterpret_cast<unsigned int *>(keys)[i];
        keys[i] = ((float)x + 1.0f) * 2.3283064365386963e-10f;   // (0, 1]
    }
}

float sortPairsGPU(float *d_key_buf, float *d_key_alt_buf, uint *d_value_buf, uint *d_value_alt_buf, int num_items)
{
    SETUP_TIMING();
    float time_sort_kv;
    DoubleBuffer<float> d_keys(d_key_buf, d_key_alt_buf);
    DoubleBuffer<uint> d_values(d_value_buf, d_value_alt_buf);
    void *d_temp_storage = NULL;
    size_t temp_storage_bytes = 0;
    CHECK(DeviceRadixSort::SortPairs(d_temp_storage, temp_storage_bytes, d_keys, d_values, num_items));
    CHECK(hipMalloc(&d_temp_storage, temp_storage_bytes));
    TIME_FUNC(CHECK(DeviceRadixSort::SortPairs(d_temp_storage, temp_storage_bytes, d_keys, d_values, num_items)),
              time_sort_kv);
    CHECK(hipFree(d_temp_storage));
    return time_sort_kv;
}

float sortKeysGPU(float *d_key_buf, float *d_key_alt_buf, int num_items)
{
    SETUP_TIMING();
    float time_sort_k;
    DoubleBuffer<float> d_keys(d_key_buf, d_key_alt_buf);
    void *d_temp_storage = NULL;
    size_t temp_storage_bytes = 0;
    CHECK(DeviceRadixSort::SortKeys(d_temp_storage, temp_storage_bytes, d_keys, num_items));
    CHECK(hipMalloc(&d_temp_storage, temp_storage_bytes));
    TIME_FUNC(CHECK(DeviceRadixSort::SortKeysDescending(d_temp_storage, temp_storage_bytes, d_keys, num_items)),
              time_sort_k);
    const int show = num_items < 32 ? num_items : 32;
    std::vector<float> res_vec(32);
    CHECK(hipMemcpy(res_vec.data(), d_keys.Current(), show * sizeof(float), hipMemcpyDeviceToHost));
    for (int i = 0; i < show; i++) std::cout << res_vec[i] << " ";
    std::cout << std::endl;
    CHECK(hipFree(d_temp_storage));
    return time_sort_k;
}

static bool get_arg(int argc, char **argv, const char *name, long &out)
{
    const std::string key = std::string("--") + name + "=";
    for (int i = 1; i < argc; ++i)
        if (std::strncmp(argv[i], key.c_str(), key.size()) == 0) { out = std::atol(argv[i] + key.size()); return true; }
    return false;
}

int main(int argc, char **argv)
{
    long num_items_l = 1 << 28, num_trials_l = 3, device = 0;
    get_arg(argc, argv, "n", num_items_l);
    get_arg(argc, argv, "t", num_trials_l);
    get_arg(argc, argv, "device", device);
    for (int i = 1; i < argc; ++i)
        if (std::strcmp(argv[i], "--help") == 0) {
            printf("%s [--n=<input items>] [--t=<num trials>] [--device=<device-id>] [--v] \n", argv[0]);
            return 0;
        }
    const int num_items = (int)num_items_l, num_trials = (int)num_trials_l;
    CHECK(hipSetDevice((int)device));

    float *d_key_buf, *d_key_alt_buf, *d_key_backup;
    uint *d_value_buf, *d_value_alt_buf, *d_value_backup;
    const size_t kb = sizeof(float) * (size_t)(num_items > 0 ? num_items : 1);
    CHECK(hipMalloc(&d_key_buf, kb)); CHECK(hipMalloc(&d_key_alt_buf, kb)); CHECK(hipMalloc(&d_key_backup, kb));
    CHECK(hipMalloc(&d_value_buf, kb)); CHECK(hipMalloc(&d_value_alt_buf, kb)); CHECK(hipMalloc(&d_value_backup, kb));

    const int seed = 0;
    CHECK((hipError_t)gs_generate_u32(reinterpret_cast<uint32_t *>(d_key_buf), num_items, GS_GEN_UNIFORM, seed, 0, 1, 0));
    if (num_items > 0) u32_to_unit_float<<<(num_items + 255) / 256, 256>>>(d_key_buf, num_items);
    CHECK((hipError_t)gs_generate_u32(d_value_buf, num_items, GS_GEN_UNIFORM, seed + 1, 0, 1, 0));
    CHECK(hipMemcpy(d_key_backup, d_key_buf, sizeof(float) * num_items, hipMemcpyDeviceToDevice));
    CHECK(hipMemcpy(d_value_backup, d_value_buf, sizeof(uint) * num_items, hipMemcpyDeviceToDevice));

    float time_sort_kv_gpu, time_sort_k_gpu;
    for (int t = 0; t < num_trials; t++) {
        time_sort_kv_gpu = sortPairsGPU(d_key_buf, d_key_alt_buf, d_value_buf, d_value_alt_buf, num_items);
        CHECK(hipMemcpy(d_key_buf, d_key_backup, sizeof(float) * num_items, hipMemcpyDeviceToDevice));
        CHECK(hipMemcpy(d_value_buf, d_value_backup, sizeof(uint) * num_items, hipMemcpyDeviceToDevice));
        time_sort_k_gpu = sortKeysGPU(d_key_buf, d_key_alt_buf, num_items);
        CHECK(hipMemcpy(d_key_buf, d_key_backup, sizeof(float) * num_items, hipMemcpyDeviceToDevice));
        std::cout << "{" << "\"time_sort_kv_gpu\":" << time_sort_kv_gpu << ",\"time_sort_k_gpu\":" << time_sort_k_gpu
                  << "}" << std::endl;
    }
    CHECK(hipFree(d_key_buf)); CHECK(hipFree(d_key_alt_buf)); CHECK(hipFree(d_key_backup));
    CHECK(hipFree(d_value_buf)); CHECK(hipFree(d_value_alt_buf)); CHECK(hipFree(d_value_backup));
    return 0;
}
