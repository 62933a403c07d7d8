// gs_host.hpp -- host-side declarations shared by the translation units of
// libgpusort.so.  The public C ABI is include/gpusort.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "gpusort.h"

#include <vector>

// Per-kernel event timing (see gs_profile_* in gpusort.h).  KernelTimer brackets
// one launch with an event pair when a profile is bound to this thread; it is a
// no-op (two pointer compares) otherwise.
struct gs_profile {
    struct Span { int id; hipEvent_t a, b; };
    std::vector<Span> spans;
    std::vector<hipEvent_t> pool;
    double ms[GS_K_COUNT] = {0};
    uint64_t launches[GS_K_COUNT] = {0};
};

// Entry points report launch failures through hipGetLastError(), which is per host thread and also holds whatever
// an EARLIER HIP call of the caller (another library, the framework) left behind: drop that first, so that only
// this call's own errors are reported.
#define GS_CLEAR_STALE_ERROR() ((void)hipGetLastError())

namespace gs {
extern thread_local gs_profile *tl_profile;

struct KernelTimer {
    gs_profile *p;
    hipStream_t s;
    gs_profile::Span span;
    KernelTimer(int id, hipStream_t stream);
    ~KernelTimer();
};
// Zero `bytes` (a multiple of 8, at an 8-byte aligned address) with a kernel.  Used instead of hipMemsetAsync
// everywhere a sort may be captured in a HIP graph: the memset node of a captured graph was seen not to take
// effect from the second replay on (counters kept growing until the task lists overflowed); a kernel node does.
hipError_t zero_async(void *p, size_t bytes, hipStream_t s);
}  // namespace gs
