"""ctypes binding of the C ABI in include/gpusort.h (libgpusort.so).

There is deliberately NO fallback: if the HIP library has not been built, or
does not export a declared symbol, importing this module raises.
"""
import ctypes as C
import os

# torch BEFORE the library: the ROCm wheel of torch carries its own HIP runtime (torch/lib/libamdhip64.so) and loads
# it into the global symbol scope; libgpusort.so, loaded afterwards, then binds its HIP calls to that same runtime,
# which is what makes torch's streams, graphs and allocations meaningful to it.  Loaded first, the library would bind
# to /opt/rocm's copy instead: a second runtime in the process, which finds no device once torch's owns it
# (every call fails with hipError 100).  tests/test_concurrency_gpu.py and tests/test_abi.py pin this.
import torch  # noqa: F401,E402

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("GS_LIB_PATH", os.path.join(_HERE, "lib", "libgpusort.so"))   # override: experiments only
CSRC_DIR = os.path.join(_HERE, "csrc")
INCLUDE_DIR = os.path.join(os.path.dirname(_HERE), "include")

GS_KEY_U32, GS_KEY_I32, GS_KEY_F32, GS_KEY_U64, GS_KEY_I64, GS_KEY_F64 = 0, 1, 2, 3, 4, 5
GS_KEY_U8, GS_KEY_I8, GS_KEY_U16, GS_KEY_I16 = 6, 7, 8, 9
GS_GEN_UNIFORM, GS_GEN_ZIPF, GS_GEN_ENTROPY_AND, GS_GEN_ENUMERATED = 0, 1, 2, 3

u64, i32, vp, sz = C.c_uint64, C.c_int, C.c_void_p, C.c_size_t
pp = C.POINTER(C.c_void_p)

# name -> (restype, argtypes); must list every symbol include/gpusort.h declares
SIGNATURES = {
    "gs_version": (i32, []),
    "gs_error_string": (C.c_char_p, [i32]),
    "gs_lsb_temp_bytes": (sz, [u64, i32]),
    "gs_lsb_sort_u32": (i32, [vp, sz, pp, pp, C.POINTER(i32), u64, i32, i32, i32, i32, vp]),
    "gs_lsb_copy_temp_bytes": (sz, [u64, i32]),
    "gs_lsb_sort_copy_u32": (i32, [vp, sz, vp, vp, vp, vp, u64, i32, i32, i32, i32, vp]),
    "gs_lsb_wide_temp_bytes": (sz, [u64, i32, i32]),
    "gs_lsb_sort_wide": (i32, [vp, sz, pp, pp, C.POINTER(i32), u64, i32, i32, i32, i32, i32, i32, vp]),
    "gs_lsb_any_temp_bytes": (sz, [u64, i32, i32]),
    "gs_lsb_sort_any": (i32, [vp, sz, vp, vp, vp, vp, u64, i32, i32, i32, i32, i32, vp]),
    "gs_lsb_geometry": (None, [u64, i32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "gs_lsb_workspace_layout": (i32, [vp, u64, pp, pp, pp]),
    "gs_lsb_pipe_status": (i32, [vp, u64, C.POINTER(C.c_uint32), vp]),
    "gs_lsb_upsweep_u32": (i32, [vp, sz, vp, u64, i32, i32, i32, i32, vp]),
    "gs_lsb_scan_spine": (i32, [vp, sz, u64, vp]),
    "gs_lsb_downsweep_u32": (i32, [vp, sz, vp, vp, vp, vp, u64, i32, i32, i32, i32, i32, vp]),
    "gs_msb_temp_bytes": (sz, [u64, i32]),
    "gs_msb_sort_u32": (i32, [vp, sz, vp, vp, u64, vp, vp, pp, pp, i32, vp, i32]),
    "gs_msb_wide_temp_bytes": (sz, [u64, i32, i32]),
    "gs_msb_sort_wide": (i32, [vp, sz, vp, vp, u64, vp, vp, i32, i32, pp, pp, i32, vp, i32]),
    "gs_msb_census": (i32, [vp, u64, i32, vp, vp]),
    "gs_msb_capacities": (None, [u64, i32, C.POINTER(C.c_uint32), C.POINTER(C.c_uint32), C.POINTER(C.c_uint32)]),
    "gs_msb_classify_upto": (i32, [vp, sz, vp, vp, u64, i32, i32, vp]),
    "gs_msb_read_lists": (i32, [vp, u64, i32, i32, vp, C.c_uint32, C.POINTER(C.c_uint32), C.POINTER(C.POINTER(C.c_uint32)), C.c_uint32,
                                C.POINTER(C.c_uint32), vp]),
    "gs_segmented_temp_bytes": (sz, [u64, i32, C.c_uint32]),
    "gs_segmented_sort_u32": (i32, [vp, sz, pp, pp, C.POINTER(i32), u64, C.c_uint32, vp, vp, i32, i32, i32, i32, vp]),
    "gs_segmented_wide_temp_bytes": (sz, [u64, i32, i32, C.c_uint32]),
    "gs_segmented_sort_wide": (i32, [vp, sz, pp, pp, C.POINTER(i32), u64, C.c_uint32, vp, vp, i32, i32, i32, i32, i32, i32, vp]),
    "gs_msb_first_pass_u32": (i32, [vp, sz, vp, vp, vp, vp, u64, i32, vp, vp]),
    "gs_msb_finish_temp_bytes": (sz, [u64, i32, i32]),
    "gs_msb_finish_u32": (i32, [vp, sz, vp, vp, vp, vp, u64, vp, i32, i32, vp, i32]),
    "gs_shard_histogram_u32": (i32, [vp, u64, i32, vp, i32, vp]),
    "gs_shard_partition_u32": (i32, [vp, sz, vp, vp, vp, vp, u64, i32, vp, i32, vp, vp, i32, vp]),
    "gs_generate_u32": (i32, [vp, u64, i32, u64, u64, i32, vp]),
    "gs_check_sorted_u32": (i32, [vp, u64, i32, vp, vp]),
    "gs_check_pairs_enumerated_u32": (i32, [vp, vp, vp, u64, vp, vp]),
    "gs_profile_create": (vp, []),
    "gs_profile_destroy": (None, [vp]),
    "gs_profile_begin": (None, [vp]),
    "gs_profile_end": (None, []),
    "gs_profile_read": (i32, [vp, C.POINTER(C.c_double), C.POINTER(u64)]),
    "gs_kernel_name": (C.c_char_p, [i32]),
}
GS_K_COUNT = 10


class GpuSortError(RuntimeError):
    def __init__(self, code, what):
        self.code = code
        super().__init__(f"{what}: hipError {code} ({error_string(code)})")


def _load():
    if not os.path.exists(LIB_PATH):
        raise ImportError(
            f"{LIB_PATH} is missing: the HIP extension is not built. "
            "Run `python -c 'import __graft_entry__ as g; g.build()'` (or make -C gpu-sort_amd/csrc). "
            "There is no CPU fallback."
        )
    lib = C.CDLL(LIB_PATH)
    for name, (res, args) in SIGNATURES.items():
        fn = getattr(lib, name)  # AttributeError if the library lacks a declared symbol
        fn.restype = res
        fn.argtypes = args
    return lib


lib = _load()


def error_string(code):
    s = lib.gs_error_string(int(code))
    return s.decode() if s else "?"


def check(code, what):
    if code != 0:
        raise GpuSortError(code, what)


class KernelProfile:
    """Per-kernel hipEvent timing on the launch stream (gs_profile_* in gpusort.h).

    with KernelProfile() as prof: ...sorts...; prof.read() -> {name: (total_ms, launches)}"""

    def __init__(self):
        self._p = lib.gs_profile_create()

    def __enter__(self):
        lib.gs_profile_begin(self._p)
        return self

    def __exit__(self, *exc):
        lib.gs_profile_end()
        return False

    def read(self):
        ms = (C.c_double * GS_K_COUNT)()
        cnt = (u64 * GS_K_COUNT)()
        check(lib.gs_profile_read(self._p, ms, cnt), "gs_profile_read")
        return {lib.gs_kernel_name(i).decode(): (ms[i], int(cnt[i])) for i in range(GS_K_COUNT) if cnt[i]}

    def close(self):
        if self._p and lib is not None:
            lib.gs_profile_destroy(self._p)
        self._p = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
