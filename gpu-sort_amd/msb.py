"""MSB driver mirror: rdxsrt_unstable_sort over the C ABI (gs_msb_sort_u32).

Mirrors msb/src/sort/gpu_radix_sort.h: RDXSRT_SortedSequence (:31-34),
rdxsrt_unstable_sort (:197-507, device pointers; the result lands in the
caller's INPUT arrays for 32-bit keys, :359-360) and the host-pointer
conveniences rdxsrt_unstable_sort_keys / _pairs (:511-587).
"""
import collections
import ctypes as C

import numpy as np
import torch

from . import _lib
from ._lib import lib, check
from .lsb import _stream_ptr, _check_buf

RDXSRT_SortedSequence = collections.namedtuple("RDXSRT_SortedSequence", ["sorted_keys", "sorted_values"])


def rdxsrt_unstable_sort(dev_keys, dev_values, key_count, dev_sorted_keys_out, dev_sorted_values_out,
                         pre_allocated_dm=None, stream=None, key_type=_lib.GS_KEY_U32, synchronize=True):
    """Ascending, UNSTABLE sort.  dev_values/dev_sorted_values_out None = keys only
    (msb/src/test.cu:53).  pre_allocated_dm: optional uint8 workspace tensor
    (the reference's RDXSRT_GPUDataManager argument); allocated per call when None."""
    _check_buf(dev_keys, key_count, "dev_keys")
    _check_buf(dev_sorted_keys_out, key_count, "dev_sorted_keys_out")
    has_values = dev_values is not None
    if has_values:
        _check_buf(dev_values, key_count, "dev_values")
        _check_buf(dev_sorted_values_out, key_count, "dev_sorted_values_out")
    need = lib.gs_msb_temp_bytes(key_count, int(has_values))
    dm = pre_allocated_dm
    if dm is None:
        dm = torch.empty(max(need, 1), dtype=torch.uint8, device=dev_keys.device)
        if stream is not None and not synchronize:
            # allocated on torch's CURRENT stream, used on `stream`, dropped when this function returns: tell the
            # caching allocator, or it may hand the block to the next allocation while the sort still runs in it
            dm.record_stream(stream)
    sk, sv = C.c_void_p(), C.c_void_p()
    err = lib.gs_msb_sort_u32(dm.data_ptr(), dm.numel(), dev_keys.data_ptr(),
                              dev_values.data_ptr() if has_values else None, key_count,
                              dev_sorted_keys_out.data_ptr(),
                              dev_sorted_values_out.data_ptr() if has_values else None,
                              C.byref(sk), C.byref(sv), key_type, _stream_ptr(stream), int(synchronize))
    check(err, "gs_msb_sort_u32")

    def which(ptr, a, b):
        if ptr is None or ptr == 0 or ptr == a.data_ptr():   # empty tensors have a null data_ptr
            return a
        return b

    return RDXSRT_SortedSequence(which(sk.value, dev_keys, dev_sorted_keys_out),
                                 which(sv.value, dev_values, dev_sorted_values_out) if has_values else None)


def rdxsrt_unstable_sort_keys(keys, key_count=None, device="cuda"):
    """Host array in, sorted host array out (gpu_radix_sort.h:511-541)."""
    keys = np.ascontiguousarray(keys)
    n = keys.size if key_count is None else key_count
    dev_keys = torch.from_numpy(keys.view(np.int32)[:n].copy()).to(device)
    dev_keys_out = torch.empty_like(dev_keys)
    seq = rdxsrt_unstable_sort(dev_keys, None, n, dev_keys_out, None)
    return seq.sorted_keys.cpu().numpy().view(keys.dtype)


def rdxsrt_unstable_sort_pairs(keys, values, key_count=None, device="cuda"):
    """Host arrays in, sorted host arrays out (gpu_radix_sort.h:543-587)."""
    keys, values = np.ascontiguousarray(keys), np.ascontiguousarray(values)
    n = keys.size if key_count is None else key_count
    dev_keys = torch.from_numpy(keys.view(np.int32)[:n].copy()).to(device)
    dev_values = torch.from_numpy(values.view(np.int32)[:n].copy()).to(device)
    dev_keys_out, dev_values_out = torch.empty_like(dev_keys), torch.empty_like(dev_values)
    seq = rdxsrt_unstable_sort(dev_keys, dev_values, n, dev_keys_out, dev_values_out)
    return seq.sorted_keys.cpu().numpy().view(keys.dtype), seq.sorted_values.cpu().numpy().view(values.dtype)


# ---- census and test access to the classification (include/gpusort.h: gs_msb_census, gs_msb_classify_upto, gs_msb_read_lists)
class _LevelCensus(C.Structure):
    _fields_ = [("buckets", C.c_uint64), ("tiles", C.c_uint64), ("keys", C.c_uint64), ("pivot_buckets", C.c_uint64),
                ("pivot_keys", C.c_uint64), ("task_keys", C.c_uint64), ("tasks", C.c_uint32 * 4), ("flagged", C.c_uint32),
                ("overflow", C.c_uint32)]


def msb_census(dm, key_count, has_values=False):
    """Per-level census of the last rdxsrt_unstable_sort that used the workspace tensor `dm`: a list of 4 dicts."""
    out = (_LevelCensus * 4)()
    check(lib.gs_msb_census(dm.data_ptr(), key_count, int(has_values), C.cast(out, C.c_void_p), None), "gs_msb_census")
    return [{"buckets": int(c.buckets), "tiles": int(c.tiles), "keys": int(c.keys), "pivot_buckets": int(c.pivot_buckets),
             "pivot_keys": int(c.pivot_keys), "task_keys": int(c.task_keys), "tasks": [int(x) for x in c.tasks],
             "flagged": int(c.flagged), "overflow": int(c.overflow)} for c in out]


def msb_algorithmic_bytes(census, key_count, has_values=False):
    """SURVEY.md 8d for the MSB path, from the census: bytes per kernel group of ONE sort (level 0 = one LSB pass)."""
    kb, mv = 4, (8 if has_values else 4)            # a key read; a key (+ value) moved = read + write of mv each
    lv = census[1:4]
    part = sum(c["keys"] - c["pivot_keys"] for c in lv)
    return {"lsb_upsweep": kb * key_count, "lsb_downsweep": 2 * mv * key_count,
            "msb_histogram": kb * sum(c["keys"] for c in lv),
            # a heavy-hitter bucket is read once; keys only: strangers are the only writes; pairs: every pair is written once
            "msb_partition": 2 * mv * part + (2 * mv if has_values else kb) * sum(c["pivot_keys"] for c in lv),
            "msb_local_sort": 2 * mv * sum(c["task_keys"] for c in census)}


def msb_classify_upto(dev_keys, dev_keys_alt, key_count, stop_level, pivot=True, dm=None):
    """Run the sort up to and including the classification of `stop_level` and read its lists back:
    returns (set of (offset, size) next-level buckets, {class: set of (offset, size, sort_bits)}, workspace)."""
    need = lib.gs_msb_temp_bytes(key_count, 0)
    if dm is None:
        dm = torch.empty(max(need, 1), dtype=torch.uint8, device=dev_keys.device)
    check(lib.gs_msb_classify_upto(dm.data_ptr(), dm.numel(), dev_keys.data_ptr(), dev_keys_alt.data_ptr(), key_count, stop_level,
                                   0 if pivot else 1, None), "gs_msb_classify_upto")
    nb = C.c_uint32(0)
    nt = (C.c_uint32 * 4)()
    check(lib.gs_msb_read_lists(dm.data_ptr(), key_count, 0, stop_level, None, 0, C.byref(nb), None, 0, nt, None), "gs_msb_read_lists")
    hb = np.zeros(2 * max(nb.value, 1), dtype=np.uint32)
    ht = [np.zeros(3 * max(nt[c], 1), dtype=np.uint32) for c in range(4)]
    ptrs = (C.POINTER(C.c_uint32) * 4)(*[a.ctypes.data_as(C.POINTER(C.c_uint32)) for a in ht])
    check(lib.gs_msb_read_lists(dm.data_ptr(), key_count, 0, stop_level, hb.ctypes.data_as(C.c_void_p), nb.value, C.byref(nb), ptrs,
                                max(max(nt), 1), nt, None), "gs_msb_read_lists")
    buckets = {(int(hb[2 * i]), int(hb[2 * i + 1])) for i in range(nb.value)}
    tasks = {c: {(int(ht[c][3 * i]), int(ht[c][3 * i + 1]), int(ht[c][3 * i + 2])) for i in range(nt[c])} for c in range(4)}
    return buckets, tasks, dm


def rdxsrt_unstable_sort_wide(dev_keys, dev_values, key_count, dev_sorted_keys_out, dev_sorted_values_out, key_type=None,
                              dm=None, stream=None):
    """The wide element types of rdxsrt_unstable_sort (gs_msb_sort_wide): int64 / float64 / uint64-as-int64 key tensors with no,
    int32 or int64 values, and int32 keys with int64 values.  Ascending, unstable; the result is in the input tensors."""
    kb = dev_keys.element_size()
    vb = dev_values.element_size() if dev_values is not None else 0
    if key_type is None:
        key_type = {torch.int64: _lib.GS_KEY_I64, torch.float64: _lib.GS_KEY_F64, torch.int32: _lib.GS_KEY_I32,
                    torch.float32: _lib.GS_KEY_F32}[dev_keys.dtype]
    need = lib.gs_msb_wide_temp_bytes(key_count, kb, vb)
    if dm is None:
        dm = torch.empty(max(need, 1), dtype=torch.uint8, device=dev_keys.device)
    sk, sv = C.c_void_p(), C.c_void_p()
    check(lib.gs_msb_sort_wide(dm.data_ptr(), dm.numel(), dev_keys.data_ptr(), dev_values.data_ptr() if vb else None, key_count,
                               dev_sorted_keys_out.data_ptr(), dev_sorted_values_out.data_ptr() if vb else None, kb, vb,
                               C.byref(sk), C.byref(sv), key_type, _stream_ptr(stream), 1), "gs_msb_sort_wide")
    return RDXSRT_SortedSequence(dev_keys, dev_values), dm
