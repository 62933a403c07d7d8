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

// mode 4: the four arrays built with the virtual-memory API from physical chunks of `chunk` bytes mapped in a SHUFFLED order
#include <vector>
#include <algorithm>
#include <random>
struct Vmm { std::vector<hipMemGenericAllocationHandle_t> h; void *va = nullptr; size_t bytes = 0; };
static bool vmm_alloc(Vmm &m, size_t bytes_each, int arrays, size_t chunk, unsigned seed, void **out)
{
    hipMemAllocationProp prop = {};
    prop.type = hipMemAllocationTypePinned; prop.location.type = hipMemLocationTypeDevice; prop.location.id = 0;
    size_t gran = 0;
    if (hipMemGetAllocationGranularity(&gran, &prop, hipMemAllocationGranularityRecommended) != hipSuccess) return false;
    if (chunk < gran) chunk = gran;
    chunk = (chunk + gran - 1) / gran * gran;
    const size_t per = (bytes_each + chunk - 1) / chunk, total = per * (size_t)arrays;
    m.bytes = total * chunk;
    if (hipMemAddressReserve(&m.va, m.bytes, 0, nullptr, 0) != hipSuccess) return false;
    m.h.resize(total);
    for (size_t i = 0; i < total; ++i) if (hipMemCreate(&m.h[i], chunk, &prop, 0) != hipSuccess) { printf("hipMemCreate failed at %zu\n", i); return false; }
    std::vector<size_t> perm(total);
    for (size_t i = 0; i < total; ++i) perm[i] = i;
    if (seed) { std::mt19937 g(seed); std::shuffle(perm.begin(), perm.end(), g); }
    for (size_t i = 0; i < total; ++i) if (hipMemMap((char *)m.va + i * chunk, chunk, 0, m.h[perm[i]], 0) != hipSuccess) return false;
    hipMemAccessDesc acc = {};
    acc.location = prop.location; acc.flags = hipMemAccessFlagsProtReadWrite;
    if (hipMemSetAccess(m.va, m.bytes, &acc, 1) != hipSuccess) return false;
    for (int a = 0; a < arrays; ++a) out[a] = (char *)m.va + (size_t)a * per * chunk;
    printf("   (vmm: granularity %zu, chunk %zu, %zu chunks)\n", gran, chunk, total);
    return true;
}
static void vmm_free(Vmm &m)
{
    if (m.va) { (void)hipMemUnmap(m.va, m.bytes); (void)hipMemAddressFree(m.va, m.bytes); }
    for (auto &h : m.h) (void)hipMemRelease(h);
    m.h.clear(); m.va = nullptr;
}

int main(int argc, char **argv)
{
    const int log2n = argc > 1 ? atoi(argv[1]) : 30, trials = argc > 2 ? atoi(argv[2]) : 5;
    const bool keys_only = argc > 3 && argv[3][0] == 'k';
    const size_t pad_arg = argc > 4 ? (size_t)atoll(argv[4]) : ((2u << 20) + 4096 + 256);   // mode 3: stagger of the carved arrays (bytes)
    const int only_mode = argc > 5 ? atoi(argv[5]) : -1;
    const bool no_shuffle = argc > 6;
    // argv[1] > 64: the number of keys itself (a count that is no power of two puts the digit streams at an odd spacing)
    const uint64_t n = log2n > 64 ? (uint64_t)atoll(argv[1]) : 1ull << log2n;
    const size_t tb = gs_lsb_temp_bytes(n, 1);
    void *temp;
    CK(hipMalloc(&temp, tb));
    gs_profile *prof = gs_profile_create();
    void *hold[8] = {nullptr};
    for (int t = 0; t < trials; ++t) {
        for (int mode = 0; mode < 5; ++mode) {
            if (only_mode >= 0 && mode != only_mode) continue;
            uint32_t *k[2], *v[2];
            char *slab = nullptr;
            Vmm vm;
            if (mode == 4) {
                void *p4[4];
                if (!vmm_alloc(vm, n * 4, 4, pad_arg, no_shuffle ? 0u : 12345u + (unsigned)t, p4)) { printf("vmm allocation failed\n"); (void)hipGetLastError(); vmm_free(vm); continue; }
                k[0] = (uint32_t *)p4[0]; v[0] = (uint32_t *)p4[1]; k[1] = (uint32_t *)p4[2]; v[1] = (uint32_t *)p4[3];
            }
            if (mode == 2 || mode == 3) {
                // ONE allocation carved into the four arrays (mode 3: each array a further 2 MiB + 4 KiB + 256 B along)
                const size_t pad = mode == 3 ? pad_arg : 0, each = ((n * 4 + (2u << 20) - 1) >> 21 << 21) + pad;
                CK(hipMalloc(&slab, 4 * each + 4 * pad));
                k[0] = (uint32_t *)(slab); v[0] = (uint32_t *)(slab + each); k[1] = (uint32_t *)(slab + 2 * each); v[1] = (uint32_t *)(slab + 3 * each);
            }
            for (int i = 0; i < 2 && mode < 2; ++i) {
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
                printf("pad %zu trial %d  %-10s  %s downsweep %.3f ms  upsweep %.3f ms   (%.4f / %.4f ns per Ki keys)\n", pad_arg, t, mode == 0 ? "hipMalloc" : mode == 1 ? "contiguous" : mode == 2 ? "one slab" : mode == 3 ? "slab+pads" : "vmm shuffled", keys_only ? "keys " : "pairs", ds, us, ds * 1e6 / (double)n * 1024, us * 1e6 / (double)n * 1024);
                fflush(stdout);
            }
        next:
            if (mode == 4) vmm_free(vm);
            else if (slab) (void)hipFree(slab);
            else for (int i = 0; i < 2; ++i) { if (k[i]) (void)hipFree(k[i]); if (v[i]) (void)hipFree(v[i]); }
        }
        // perturb what the next trial's allocations get
        if (t < 8) CK(hipMalloc(&hold[t], (size_t)(3 + 2 * t) << 28));
    }
    return 0;
}
