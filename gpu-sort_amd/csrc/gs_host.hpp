// gs_host.hpp -- host-side declarations shared by the translation units of
// libgpusort.so.  The public C ABI is include/gpusort.h.
#pragma once

#include <hip/hip_runtime.h>
#include <stddef.h>
#include <stdint.h>

#include "gpusort.h"
