// pairs_alloc.hip -- does the PHYSICAL placement of the four arrays explain the spread of the LSB pairs sort (3.3 - 4.3 ms per
// downsweep on the same GPU, profiles/r03_pairs_alloc_exp.txt)?  Per trial: fresh allocations of the four arrays, once with
// hipMalloc and once with hipExtMallocWithFlags(hipDeviceMallocContiguous) (physically contiguous), the same pairs sort on both.
// build: hipcc --offload-arch=gfx950 -O2 -std=c++17 -Iinclude tools/micro/pairs_alloc.hip -o tools/micro/pairs_alloc \
//        -Lgpu-sort_amd/lib -lgpusort -Wl,-rpath,'$ORIGIN/../../gpu-sort_amd/lib'
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include "gpusort.h"
#define CK(x) do { int e_ = (int)(x); if (e_) { fprintf(stderr, "%s:%d: %s -> %d\n", __FILE__, __LINE__, #x, e_); exit(1); } } while (0)

int main(int argc, char **argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 30, trials = argc > 2 ? atoi(argv[2]) : 5;
    const bool keys_only = argc > 3 && argv[3][0] == 'k';
    // argv[1] > 64: the number of keys itself (a count that is no power of two puts the digit streams at an odd spacing)
    const uint64_t n = log2n > 64 ? (uint64_t)atoll(argv[1]) : 1ull << log2n;
    const size_t tb = gs_lsb_temp_bytes(n, 1);
    void *temp;
    CK(hipMalloc(&temp, tb));
    gs_profile *prof = gs_profile_create();
    void *hold[8] = {nullptr};
    for (int t = 0; t < trials; ++t) {
        for (int mode = 0; mode < 2; ++mode) {
            uint32_t *k[2], *v[2];
            for (int i = 0; i < 2; ++i) {
                if (mode == 0) { CK(hipMalloc(&k[i], n * 4)); CK(hipMalloc(&v[i], n * 4)); }
                else {
                    if (hipExtMallocWithFlags((void **)&k[i], n * 4, hipDeviceMallocContiguous) != hipSuccess ||
                        hipExtMallocWithFlags((void **)&v[i], n * 4, hipDeviceMallocContiguous) != hipSuccess) {
                        printf("trial %d: contiguous allocation refused\n", t); (void)hipGetLastError(); k[i] = v[i] = nullptr; goto next;
                    }
                }
            }
            {
                double ms[GS_K_COUNT]; uint64_t cnt[GS_K_COUNT];
                for (int r = 0; r < 4; ++r) {
                    CK(gs_generate_u32(k[0], n, GS_GEN_UNIFORM, 0, 0, 0, nullptr));
                    CK(gs_generate_u32(v[0], n, GS_GEN_ENUMERATED, 0, 0, 0, nullptr));
                    if (r == 1) gs_profile_begin(prof);
                    int sel = 0;
                    CK(gs_lsb_sort_u32(temp, tb, k, keys_only ? nullptr : v, &sel, n, 0, 32, 0, GS_KEY_U32, nullptr));
                }
                gs_profile_end();
                CK(hipDeviceSynchronize());
                CK(gs_profile_read(prof, ms, cnt));
                static double last_ds = 0, last_us = 0; static uint64_t last_dc = 0, last_uc = 0;      // the profile accumulates
                const double ds = (ms[GS_K_LSB_DOWNSWEEP] - last_ds) / (double)(cnt[GS_K_LSB_DOWNSWEEP] - last_dc);
                const double us = (ms[GS_K_LSB_UPSWEEP] - last_us) / (double)(cnt[GS_K_LSB_UPSWEEP] - last_uc);
                last_ds = ms[GS_K_LSB_DOWNSWEEP]; last_dc = cnt[GS_K_LSB_DOWNSWEEP]; last_us = ms[GS_K_LSB_UPSWEEP]; last_uc = cnt[GS_K_LSB_UPSWEEP];
                printf("trial %d  %-10s  %s downsweep %.3f ms  upsweep %.3f ms   (%.4f / %.4f ns per Ki keys)\n", t, mode ? "contiguous" : "hipMalloc", keys_only ? "keys " : "pairs", ds, us, ds * 1e6 / (double)n * 1024, us * 1e6 / (double)n * 1024);
                fflush(stdout);
            }
        next:
            for (int i = 0; i < 2; ++i) { if (k[i]) (void)hipFree(k[i]); if (v[i]) (void)hipFree(v[i]); }
        }
        // perturb what the next trial's allocations get
        if (t < 8) CK(hipMalloc(&hold[t], (size_t)(3 + 2 * t) << 28));
    }
    return 0;
}
