// Comparison only (the ROCm image ships rocPRIM/hipCUB device_radix_sort; "may be used only
// as an extra on-device cross-check / comparison line, never as the implementation").  Times
// rocprim::radix_sort_keys / radix_sort_pairs on the same workload as bench.py and checks libgpusort's result
// against it.   usage: rocprim_sort [log2n] [pairs]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <string.h>
#include <vector>
#include <rocprim/device/device_radix_sort.hpp>
#include "gpusort.h"

#define OK(x) do { hipError_t e_ = (hipError_t)(x); if (e_ != hipSuccess) { printf("error %d at line %d\n", (int)e_, __LINE__); return 2; } } while (0)

int main(int argc, char **argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 30;
    const bool pairs = argc > 2 && !strcmp(argv[2], "pairs");
    const size_t n = (size_t)1 << log2n;
    uint32_t *src, *a, *b, *va = nullptr, *vb = nullptr, *ours;
    OK(hipMalloc(&src, n * 4)); OK(hipMalloc(&a, n * 4)); OK(hipMalloc(&b, n * 4)); OK(hipMalloc(&ours, n * 4));
    if (pairs) { OK(hipMalloc(&va, n * 4)); OK(hipMalloc(&vb, n * 4)); }
    OK(gs_generate_u32(src, n, GS_GEN_UNIFORM, 0, 0, 1, nullptr));
    hipEvent_t e0, e1; OK(hipEventCreate(&e0)); OK(hipEventCreate(&e1));

    size_t tb = 0;
    if (pairs) OK(rocprim::radix_sort_pairs(nullptr, tb, a, b, va, vb, n, 0, 32));
    else OK(rocprim::radix_sort_keys(nullptr, tb, a, b, n, 0, 32));
    void *temp; OK(hipMalloc(&temp, tb));
    float best = 1e9f, best_db = 1e9f;
    for (int it = 0; it < 4; ++it) {
        OK(hipMemcpy(a, src, n * 4, hipMemcpyDeviceToDevice));
        if (pairs) OK(gs_generate_u32(va, n, GS_GEN_ENUMERATED, 0, 0, 1, nullptr));
        OK(hipDeviceSynchronize());
        OK(hipEventRecord(e0));
        if (pairs) OK(rocprim::radix_sort_pairs(temp, tb, a, b, va, vb, n, 0, 32));
        else OK(rocprim::radix_sort_keys(temp, tb, a, b, n, 0, 32));
        OK(hipEventRecord(e1)); OK(hipEventSynchronize(e1));
        float ms; OK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < best) best = ms;
    }
    printf("rocprim::radix_sort_%s  2^%d u32: %.3f ms  %.2f G%s/s  (temp %.0f MiB)\n", pairs ? "pairs" : "keys", log2n, best,
           n / best / 1e6, pairs ? "pairs" : "keys", tb / 1048576.0);

    // like for like with gs_lsb_sort_u32's DoubleBuffer form: rocprim::double_buffer (both halves may be overwritten, no
    // internal ping-pong copy of the input)
    {
        size_t tb2 = 0;
        rocprim::double_buffer<uint32_t> dk(a, b), dv(va, vb);
        if (pairs) OK(rocprim::radix_sort_pairs(nullptr, tb2, dk, dv, n, 0, 32));
        else OK(rocprim::radix_sort_keys(nullptr, tb2, dk, n, 0, 32));
        void *temp2; OK(hipMalloc(&temp2, tb2 ? tb2 : 1));
        float best2 = 1e9f;
        for (int it = 0; it < 4; ++it) {
            rocprim::double_buffer<uint32_t> k2(a, b), v2(va, vb);
            OK(hipMemcpy(a, src, n * 4, hipMemcpyDeviceToDevice));
            if (pairs) OK(gs_generate_u32(va, n, GS_GEN_ENUMERATED, 0, 0, 1, nullptr));
            OK(hipDeviceSynchronize());
            OK(hipEventRecord(e0));
            if (pairs) OK(rocprim::radix_sort_pairs(temp2, tb2, k2, v2, n, 0, 32));
            else OK(rocprim::radix_sort_keys(temp2, tb2, k2, n, 0, 32));
            OK(hipEventRecord(e1)); OK(hipEventSynchronize(e1));
            float ms; OK(hipEventElapsedTime(&ms, e0, e1));
            if (it > 0 && ms < best2) best2 = ms;
        }
        printf("rocprim::radix_sort_%s (double_buffer form) 2^%d u32: %.3f ms  %.2f G%s/s  (temp %.0f MiB)\n", pairs ? "pairs" : "keys",
               log2n, best2, n / best2 / 1e6, pairs ? "pairs" : "keys", tb2 / 1048576.0);
        best_db = best2;
        OK(hipFree(temp2));
        // leave the non-overwriting form's output in b / vb for the cross-check below
        OK(hipMemcpy(a, src, n * 4, hipMemcpyDeviceToDevice));
        if (pairs) OK(gs_generate_u32(va, n, GS_GEN_ENUMERATED, 0, 0, 1, nullptr));
        if (pairs) OK(rocprim::radix_sort_pairs(temp, tb, a, b, va, vb, n, 0, 32));
        else OK(rocprim::radix_sort_keys(temp, tb, a, b, n, 0, 32));
        OK(hipDeviceSynchronize());
    }

    // libgpusort on the same input
    const size_t gtb = gs_lsb_temp_bytes(n, pairs);
    void *gtemp; OK(hipMalloc(&gtemp, gtb));
    uint32_t *alt; OK(hipMalloc(&alt, n * 4));
    uint32_t *valt = nullptr, *vin = nullptr;
    if (pairs) { OK(hipMalloc(&valt, n * 4)); OK(hipMalloc(&vin, n * 4)); }
    float gbest = 1e9f; int sel = 0;
    uint32_t *kk[2] = {ours, alt}, *vv[2] = {vin, valt};
    for (int it = 0; it < 4; ++it) {
        OK(hipMemcpy(ours, src, n * 4, hipMemcpyDeviceToDevice));
        if (pairs) OK(gs_generate_u32(vin, n, GS_GEN_ENUMERATED, 0, 0, 1, nullptr));
        sel = 0;
        OK(hipDeviceSynchronize());
        OK(hipEventRecord(e0));
        OK(gs_lsb_sort_u32(gtemp, gtb, kk, pairs ? vv : nullptr, &sel, n, 0, 32, 0, GS_KEY_U32, nullptr));
        OK(hipEventRecord(e1)); OK(hipEventSynchronize(e1));
        float ms; OK(hipEventElapsedTime(&ms, e0, e1));
        if (it > 0 && ms < gbest) gbest = ms;
    }
    printf("gs_lsb_sort_u32 (DoubleBuffer form) 2^%d u32: %.3f ms  %.2f G%s/s  -> %.2fx rocprim's double_buffer form, %.2fx its "
           "non-overwriting form\n", log2n, gbest, n / gbest / 1e6, pairs ? "pairs" : "keys", best_db / gbest, best / gbest);
    // bit-exact cross-check on a sample (whole arrays up to 2^28)
    const size_t m = n <= ((size_t)1 << 28) ? n : ((size_t)1 << 28);
    std::vector<uint32_t> x(m), y(m);
    OK(hipMemcpy(x.data(), b, m * 4, hipMemcpyDeviceToHost));     // rocprim output (keys_output)
    OK(hipMemcpy(y.data(), kk[sel], m * 4, hipMemcpyDeviceToHost));
    printf("first %zu keys identical: %s\n", m, memcmp(x.data(), y.data(), m * 4) == 0 ? "yes" : "NO");
    if (pairs) {
        OK(hipMemcpy(x.data(), vb, m * 4, hipMemcpyDeviceToHost));
        OK(hipMemcpy(y.data(), vv[sel], m * 4, hipMemcpyDeviceToHost));
        printf("first %zu values identical (both stable): %s\n", m, memcmp(x.data(), y.data(), m * 4) == 0 ? "yes" : "NO");
    }
    return 0;
}
