"""ctypes/numpy front-end of the CPU oracle (oracle.c, cpu_baseline.cpp).

TEST INFRASTRUCTURE ONLY.  Importable only from tests/, from
__graft_entry__.smoke() and from bench.py's cpu_baseline leg -- never from
the product package (gpu-sort_amd/), which must fail loudly without its HIP
library instead of falling back to anything here.
"""
import ctypes as C
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB_PATH = os.path.join(_HERE, "liboracle.so")
_REF_PATH = os.path.join(_HERE, "_ref", "libref_mersenne.so")


def build(force=False):
    """Compile liboracle.so (and oracle/_ref when /root/reference is present)."""
    if force or not os.path.exists(_LIB_PATH) or any(
        os.path.getmtime(os.path.join(_HERE, f)) > os.path.getmtime(_LIB_PATH)
        for f in ("oracle.c", "oracle.h", "cpu_baseline.cpp")
    ):
        subprocess.check_call(["make", "-s", "-C", _HERE, "liboracle.so"])
    if os.path.isdir("/root/reference") and (force or not os.path.exists(_REF_PATH)):
        subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])


_lib = None
_u32p = np.ctypeslib.ndpointer(dtype=np.uint32, flags="C_CONTIGUOUS")


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_LIB_PATH)
        u64, i32, p = C.c_uint64, C.c_int, C.c_void_p
        L.orc_splitmix64.restype = u64
        L.orc_splitmix64.argtypes = [u64]
        for name in ("orc_gen_uniform", "orc_gen_zipf"):
            getattr(L, name).argtypes = [_u32p, u64, u64, u64]
        L.orc_gen_entropy_and.argtypes = [_u32p, u64, u64, u64, i32]
        L.orc_gen_enumerated.argtypes = [_u32p, u64, u64]
        L.orc_mt_genrand_int32.restype = C.c_uint32
        L.orc_mt_init_genrand.argtypes = [C.c_uint32]
        L.orc_random_bits_u32.argtypes = [_u32p, u64, i32, i32, i32]
        for name in ("orc_twiddle_in_u32", "orc_twiddle_in_i32", "orc_twiddle_in_f32", "orc_twiddle_out_f32"):
            getattr(L, name).restype = C.c_uint32
            getattr(L, name).argtypes = [C.c_uint32]
        L.orc_lsb_reference_ranks.argtypes = [_u32p, u64, i32, i32, i32, _u32p]
        for name in ("orc_twiddle_in_u64", "orc_twiddle_in_i64", "orc_twiddle_in_f64", "orc_twiddle_out_f64"):
            getattr(L, name).restype = u64
            getattr(L, name).argtypes = [u64]
        L.orc_lsb_reference_ranks_u64.argtypes = [np.ctypeslib.ndpointer(np.uint64, flags="C_CONTIGUOUS"), u64, i32, i32, i32,
                                                  i32, _u32p]
        L.orc_lsb_sort_keys.argtypes = [_u32p, _u32p, u64, i32, i32, i32]
        L.orc_lsb_sort_pairs.argtypes = [_u32p, _u32p, _u32p, _u32p, u64, i32, i32, i32]
        L.orc_chunk_tiles.argtypes = [u64, C.c_uint32, C.c_uint32, C.POINTER(u64), C.POINTER(u64)]
        L.orc_upsweep.argtypes = [_u32p, u64, i32, i32, i32, C.c_uint32, C.c_uint32, C.c_uint32, _u32p]
        L.orc_exclusive_scan.argtypes = [_u32p, u64]
        L.orc_downsweep.argtypes = [_u32p, p, _u32p, p, u64, i32, i32, i32]
        L.orc_lsd_radix_sort.argtypes = [_u32p, p, _u32p, p, u64, i32, i32, i32, C.POINTER(i32)]
        L.orc_msb_check_keys.restype = u64
        L.orc_msb_check_keys.argtypes = [_u32p, _u32p, u64]
        L.orc_msb_check_pairs.restype = u64
        L.orc_msb_check_pairs.argtypes = [_u32p, _u32p, _u32p, _u32p, u64]
        L.orc_msb_check_pairs_enumerated.restype = u64
        L.orc_msb_check_pairs_enumerated.argtypes = [_u32p, _u32p, _u32p, u64]
        L.orc_multiset_checksum.argtypes = [_u32p, u64, C.POINTER(u64), C.POINTER(u64)]
        L.orc_count_inversions_adjacent.restype = u64
        L.orc_count_inversions_adjacent.argtypes = [_u32p, u64, i32]
        L.orc_std_sort_u32.restype = C.c_double
        L.orc_std_sort_u32.argtypes = [_u32p, u64]
        L.orc_std_stable_sort_pairs.restype = C.c_double
        L.orc_std_stable_sort_pairs.argtypes = [_u32p, _u32p, u64]
        L.orc_std_sort_u32_mt.restype = C.c_double
        L.orc_std_sort_u32_mt.argtypes = [_u32p, u64, i32, C.POINTER(i32)]
        L.orc_hardware_threads.restype = i32
        _lib = L
    return _lib


def ref_mersenne():
    """The reference's own MT19937 (oracle/_ref), or None when not built."""
    if not os.path.exists(_REF_PATH):
        return None
    R = C.CDLL(_REF_PATH)
    R.ref_mt_genrand_int32.restype = C.c_uint32
    R.ref_mt_init_by_array.argtypes = [C.POINTER(C.c_uint32), C.c_int]
    R.ref_mt_init_genrand.argtypes = [C.c_uint32]
    return R


def _c(a):
    return np.ascontiguousarray(a, dtype=np.uint32)


# ---- generators ---------------------------------------------------------
def gen_uniform(n, seed=0, start=0):
    out = np.empty(n, np.uint32); lib().orc_gen_uniform(out, n, seed, start); return out


def gen_zipf(n, seed=0, start=0):
    out = np.empty(n, np.uint32); lib().orc_gen_zipf(out, n, seed, start); return out


def gen_entropy_and(n, level, seed=0, start=0):
    out = np.empty(n, np.uint32); lib().orc_gen_entropy_and(out, n, seed, start, level); return out


def gen_enumerated(n, start=0):
    out = np.empty(n, np.uint32); lib().orc_gen_enumerated(out, n, start); return out


def cub_random_keys(n, entropy_reduction=0, begin_bit=0, end_bit=32, reseed=True):
    """RANDOM keys exactly as lsb/cub/test/test_device_radix_sort.cu generates them."""
    if reseed:
        lib().orc_mt_init_cub_default()
    out = np.empty(n, np.uint32)
    lib().orc_random_bits_u32(out, n, entropy_reduction, begin_bit, end_bit)
    return out


# ---- LSB oracle -----------------------------------------------------------
def lsb_sort_keys(keys, begin_bit=0, end_bit=32, descending=False):
    keys = _c(keys); out = np.empty_like(keys)
    lib().orc_lsb_sort_keys(keys, out, keys.size, begin_bit, end_bit, int(descending)); return out


def lsb_sort_pairs(keys, vals, begin_bit=0, end_bit=32, descending=False):
    keys, vals = _c(keys), _c(vals); ko, vo = np.empty_like(keys), np.empty_like(vals)
    lib().orc_lsb_sort_pairs(keys, vals, ko, vo, keys.size, begin_bit, end_bit, int(descending))
    return ko, vo


def lsb_reference_ranks(keys, begin_bit=0, end_bit=32, descending=False):
    keys = _c(keys); r = np.empty(keys.size, np.uint32)
    lib().orc_lsb_reference_ranks(keys, keys.size, begin_bit, end_bit, int(descending), r); return r


def lsb_reference_ranks_u64(keys, key_type=3, begin_bit=0, end_bit=64, descending=False):
    """keys: uint64 bit patterns; key_type 3/4/5 = unsigned / signed / double."""
    keys = np.ascontiguousarray(keys, dtype=np.uint64); r = np.empty(keys.size, np.uint32)
    lib().orc_lsb_reference_ranks_u64(keys, keys.size, key_type, begin_bit, end_bit, int(descending), r); return r


def chunk_tiles(num_tiles, tiles_per_chunk, c):
    lo, hi = C.c_uint64(), C.c_uint64()
    lib().orc_chunk_tiles(num_tiles, tiles_per_chunk, c, C.byref(lo), C.byref(hi)); return lo.value, hi.value


def upsweep(keys, shift, bits, tile, tiles_per_chunk, grid, descending=False):
    keys = _c(keys); spine = np.zeros((1 << bits) * grid, np.uint32)
    lib().orc_upsweep(keys, keys.size, shift, bits, int(descending), tile, tiles_per_chunk, grid, spine); return spine


def exclusive_scan(spine):
    s = _c(spine).copy(); lib().orc_exclusive_scan(s, s.size); return s


def downsweep(keys, vals, shift, bits, descending=False):
    keys = _c(keys); ko = np.empty_like(keys)
    if vals is None:
        lib().orc_downsweep(keys, None, ko, None, keys.size, shift, bits, int(descending)); return ko, None
    vals = _c(vals); vo = np.empty_like(vals)
    lib().orc_downsweep(keys, vals.ctypes.data, ko, vo.ctypes.data, keys.size, shift, bits, int(descending))
    return ko, vo


def lsd_radix_sort(keys, vals=None, begin_bit=0, end_bit=32, descending=False):
    k = _c(keys).copy(); kt = np.empty_like(k); sel = C.c_int(0)
    if vals is None:
        lib().orc_lsd_radix_sort(k, None, kt, None, k.size, begin_bit, end_bit, int(descending), C.byref(sel))
        return (kt if sel.value else k), None
    v = _c(vals).copy(); vt = np.empty_like(v)
    lib().orc_lsd_radix_sort(k, v.ctypes.data, kt, vt.ctypes.data, k.size, begin_bit, end_bit,
                             int(descending), C.byref(sel))
    return (kt, vt) if sel.value else (k, v)


# ---- MSB checkers ---------------------------------------------------------
def msb_check_keys(keys_in, keys_sorted):
    a, b = _c(keys_in), _c(keys_sorted)
    return 0 if a.size != b.size and False else lib().orc_msb_check_keys(a, b, a.size)


def msb_check_pairs(keys_in, vals_in, keys_sorted, vals_sorted):
    return lib().orc_msb_check_pairs(_c(keys_in), _c(vals_in), _c(keys_sorted), _c(vals_sorted).copy(),
                                     _c(keys_in).size)


def msb_check_pairs_enumerated(keys_in, keys_sorted, vals_sorted):
    return lib().orc_msb_check_pairs_enumerated(_c(keys_in), _c(keys_sorted), _c(vals_sorted), _c(keys_in).size)


# ---- properties -----------------------------------------------------------
def multiset_checksum(keys):
    keys = _c(keys); s, x = C.c_uint64(), C.c_uint64()
    lib().orc_multiset_checksum(keys, keys.size, C.byref(s), C.byref(x)); return s.value, x.value


def count_inversions_adjacent(a, descending=False):
    a = _c(a); return lib().orc_count_inversions_adjacent(a, a.size, int(descending))


# ---- CPU baseline ---------------------------------------------------------
def time_std_sort(keys):
    k = _c(keys).copy(); return lib().orc_std_sort_u32(k, k.size), k


def time_std_stable_sort_pairs(keys, vals):
    k, v = _c(keys).copy(), _c(vals).copy(); return lib().orc_std_stable_sort_pairs(k, v, k.size), k, v


def time_std_sort_mt(keys, threads=0):
    k = _c(keys).copy(); used = C.c_int(0)
    return lib().orc_std_sort_u32_mt(k, k.size, threads, C.byref(used)), used.value, k


def hardware_threads():
    return lib().orc_hardware_threads()


# ---- M4: classification of the MSB levels, restated on the CPU with numpy.
# What it follows in the reference: a sub-bucket is EMPTY (skipped), NON-LOCAL (more keys than the largest local-sort
# configuration: it becomes a bucket of the next pass, gpu_radix_sort.h:426-468) or LOCAL; adjacent local sub-buckets are
# merged while the sum stays below RDXSRT_CFG_MERGE_LOCREC_THRESH = 3000 (cuda_radix_sort.h:1084-1087,
# cuda_radix_sort_config.h:9) and a merged range is sorted on one more byte (cuda_radix_sort.h:1601: byte - is_merged);
# every local range goes to the first configuration whose capacity holds it (cuda_radix_sort.h:1241-1247).
# What differs by design in the build under test and is restated here as built: the local-sort capacities
# (2048 / 4608 / 9216 / 17408 keys; reference 7-9 configurations up to 9216) and the heavy-hitter rule (no
# reference counterpart: a bucket in which one value holds at least half of the keys is finished where it stands).
MSB_CLASS_CAPS = (2048, 4608, 9216, 17408)
MSB_MERGE = 3000


def msb_class_of(size):
    for c, cap in enumerate(MSB_CLASS_CAPS):
        if size <= cap:
            return c
    raise ValueError(size)


def msb_classify_counts(counts, offset, level):
    """One bucket at `level` (0 = top byte) with its 256 sub-bucket counts, starting at `offset`:
    returns (next-level buckets [(offset, size)], tasks [(class, offset, size, sort_bits)])."""
    cap_max, rb = MSB_CLASS_CAPS[-1], 24 - 8 * level
    buckets, tasks = [], []
    starts = offset + np.concatenate(([0], np.cumsum(counts[:-1]))).astype(np.int64)
    run = None                      # [start offset, sum, non-empty sub-buckets]

    def flush():
        nonlocal run
        if run is not None:
            tasks.append((msb_class_of(run[1]), int(run[0]), int(run[1]), rb + (8 if run[2] > 1 else 0)))
            run = None

    for d in range(256):
        c = int(counts[d])
        if c == 0:
            continue
        if c > cap_max:
            flush()
            buckets.append((int(starts[d]), c))
            continue
        if run is not None and run[1] + c < MSB_MERGE:
            run[1] += c
            run[2] += 1
        else:
            flush()
            run = [starts[d], c, 1]
    flush()
    return buckets, tasks


def msb_heavy_hitter(content, cap_max=MSB_CLASS_CAPS[-1]):
    """The heavy-hitter rule on a bucket's keys in the order they lie on the device: (cand, less, eq) or None."""
    size = content.size
    q = size // 4
    a0, a1, a2 = content[q], content[2 * q], content[3 * q]
    cand = a1 if a1 == a2 else a0
    smp = content[[(size * (2 * i + 1)) >> 5 for i in range(16)]]
    if int((smp == cand).sum()) < 3:
        return None
    eq, less = int((content == cand).sum()), int((content < cand).sum())
    greater = size - eq - less
    if eq >= size - eq and less <= cap_max and greater <= cap_max:
        return int(cand), less, eq
    return None


def msb_level_lists(keys, stop_level, pivot=True):
    """Buckets passed to level stop_level + 1 and local-sort tasks emitted by the classification of `stop_level`, for
    u32 keys: (set of (offset, size), {class: set of (offset, size, sort_bits)}).  Every partition is stable, so a
    bucket's keys lie on the device in input order; their offsets are their positions in the sorted array."""
    keys = np.ascontiguousarray(keys, dtype=np.uint32)
    level_buckets = [(0, keys.size, keys)]                 # (offset, size, content in device order)
    for level in range(stop_level + 1):
        nxt, out_b, out_t = [], set(), {c: set() for c in range(4)}
        rb = 24 - 8 * level
        for off, size, content in level_buckets:
            hh = msb_heavy_hitter(content) if (pivot and level >= 1) else None
            if hh is not None:
                cand, less, eq = hh
                for o, s in ((off, less), (off + less + eq, size - less - eq)):
                    if s:
                        out_t[msb_class_of(s)].add((o, s, rb + 8))
                continue
            digit = (content >> np.uint32(rb)) & np.uint32(0xff)
            counts = np.bincount(digit, minlength=256)
            b, t = msb_classify_counts(counts, off, level)
            for c, o, s, bits in t:
                out_t[c].add((o, s, bits))
            if b:
                order = np.argsort(digit, kind="stable")
                by_digit = content[order]
                starts = np.concatenate(([0], np.cumsum(counts)))
                for (o, s) in b:
                    d = int(np.searchsorted(starts, o - off, side="right") - 1)
                    while counts[d] == 0 or starts[d] != o - off:
                        d += 1
                    nxt.append((o, s, by_digit[starts[d]:starts[d] + s]))
                    out_b.add((o, s))
        level_buckets = nxt
    return out_b, out_t
