"""LSB driver mirror: DoubleBuffer + DeviceRadixSort over the C ABI.

Mirrors cub::DoubleBuffer (lsb/cub/cub/util_type.cuh:785-817) and the
DoubleBuffer overloads of cub::DeviceRadixSort
(lsb/cub/cub/device/device_radix_sort.cuh:248-272, 399-423, 595-621, 754-780),
and the two timed wrappers of the reference driver (lsb/sort.cu:25-76).
"""
import ctypes as C

import torch

from . import _lib
from ._lib import lib, check

_KEY_TYPES = {torch.int32: _lib.GS_KEY_I32, torch.float32: _lib.GS_KEY_F32,
              torch.int64: _lib.GS_KEY_I64, torch.float64: _lib.GS_KEY_F64}
_KEY_TYPES.update({torch.uint8: _lib.GS_KEY_U8, torch.bool: _lib.GS_KEY_U8, torch.int8: _lib.GS_KEY_I8, torch.int16: _lib.GS_KEY_I16})
if hasattr(torch, "uint16"):
    _KEY_TYPES[torch.uint16] = _lib.GS_KEY_U16
if hasattr(torch, "uint32"):
    _KEY_TYPES[torch.uint32] = _lib.GS_KEY_U32
if hasattr(torch, "uint64"):
    _KEY_TYPES[torch.uint64] = _lib.GS_KEY_U64


def _stream_ptr(stream):
    if stream is None:
        stream = torch.cuda.current_stream()
    return C.c_void_p(stream.cuda_stream)


def _check_buf(t, n, what, elem=4):
    if not isinstance(t, torch.Tensor) or not t.is_cuda or not t.is_contiguous():
        raise ValueError(f"{what}: expected a contiguous device tensor")
    if t.element_size() != elem or t.numel() < n:
        raise ValueError(f"{what}: need >= {n} {elem}-byte elements")


class DoubleBuffer:
    """Pair of device buffers + selector (util_type.cuh:785-817)."""

    def __init__(self, d_current=None, d_alternate=None):
        self.d_buffers = [d_current, d_alternate]
        self.selector = 0

    def Current(self):
        return self.d_buffers[self.selector]

    def Alternate(self):
        return self.d_buffers[self.selector ^ 1]


class DeviceRadixSort:
    """Two-phase API like CUB: call with d_temp_storage=None to get the size."""

    @staticmethod
    def _sort(d_temp_storage, temp_storage_bytes, d_keys, d_values, num_items, begin_bit, end_bit, descending,
              stream, key_type):
        has_values = d_values is not None
        kb = d_keys.d_buffers[0].element_size()
        vb = 0
        if has_values:      # a value is one element of a 1-D tensor or one row of a 2-D one (records: 16-byte rows ...)
            v0 = d_values.d_buffers[0]
            row = 1
            for d in v0.shape[1:]:
                row *= d
            vb = v0.element_size() * row
        if kb < 4 or vb not in (0, 4, 8):
            return DeviceRadixSort._sort_any(d_temp_storage, temp_storage_bytes, d_keys, d_values, num_items, begin_bit,
                                             end_bit, descending, stream, key_type, kb, vb)
        if kb == 8 or vb == 8:
            return DeviceRadixSort._sort_wide(d_temp_storage, temp_storage_bytes, d_keys, d_values, num_items, begin_bit,
                                              end_bit, descending, stream, key_type, kb, vb)
        need = lib.gs_lsb_temp_bytes(num_items, int(has_values))
        if d_temp_storage is None:                      # dispatch_radix_sort.cuh:1110
            return need
        if end_bit is None:
            end_bit = 32
        if key_type is None:
            key_type = _KEY_TYPES.get(d_keys.d_buffers[0].dtype, _lib.GS_KEY_U32)
        for b in d_keys.d_buffers:
            _check_buf(b, num_items, "d_keys")
        keys = (C.c_void_p * 2)(d_keys.d_buffers[0].data_ptr(), d_keys.d_buffers[1].data_ptr())
        vals = None
        if has_values:
            for b in d_values.d_buffers:
                _check_buf(b, num_items, "d_values")
            if d_values.selector != d_keys.selector:
                raise ValueError("d_keys and d_values selectors differ")
            vals = (C.c_void_p * 2)(d_values.d_buffers[0].data_ptr(), d_values.d_buffers[1].data_ptr())
        sel = C.c_int(d_keys.selector)
        err = lib.gs_lsb_sort_u32(C.c_void_p(d_temp_storage.data_ptr()), min(temp_storage_bytes, d_temp_storage.numel() * d_temp_storage.element_size()),
                                  keys, vals, C.byref(sel), num_items, begin_bit, end_bit, int(descending),
                                  key_type, _stream_ptr(stream))
        check(err, "gs_lsb_sort_u32")
        d_keys.selector = sel.value
        if has_values:
            d_values.selector = sel.value
        return need

    @staticmethod
    def _sort_any(d_temp_storage, temp_storage_bytes, d_keys, d_values, num_items, begin_bit, end_bit, descending,
                  stream, key_type, kb, vb):
        """8- and 16-bit keys (torch.bool / uint8 / int8 / int16 [/ uint16]) and values of any size (1- and 2-byte
        elements, or rows of a 2-D tensor: 16-byte records like the reference's TestFoo): gs_lsb_sort_any.  It is the
        plain-pointer form underneath (input untouched, result in the other buffer), so the sorted data ALWAYS ends in
        the alternate buffer and the selector flips once -- a DoubleBuffer contract as good as any other."""
        if key_type is None:
            key_type = _KEY_TYPES.get(d_keys.d_buffers[0].dtype)
            if key_type is None:
                raise TypeError(f"no key category for dtype {d_keys.d_buffers[0].dtype}: pass key_type")
        need = lib.gs_lsb_any_temp_bytes(num_items, key_type, vb)
        if d_temp_storage is None:
            return need
        if end_bit is None:
            end_bit = 8 * kb
        for b in d_keys.d_buffers:
            _check_buf(b, num_items, "d_keys", kb)
        if vb:
            for b in d_values.d_buffers:
                _check_buf(b, num_items * (vb // b.element_size()), "d_values", b.element_size())
            if d_values.selector != d_keys.selector:
                raise ValueError("d_keys and d_values selectors differ")
        sel = d_keys.selector
        err = lib.gs_lsb_sort_any(C.c_void_p(d_temp_storage.data_ptr()),
                                  min(temp_storage_bytes, d_temp_storage.numel() * d_temp_storage.element_size()),
                                  d_keys.d_buffers[sel].data_ptr(), d_keys.d_buffers[sel ^ 1].data_ptr(),
                                  d_values.d_buffers[sel].data_ptr() if vb else None,
                                  d_values.d_buffers[sel ^ 1].data_ptr() if vb else None, num_items, key_type, vb,
                                  begin_bit, end_bit, int(descending), _stream_ptr(stream))
        check(err, "gs_lsb_sort_any")
        d_keys.selector = sel ^ 1
        if vb:
            d_values.selector = sel ^ 1
        return need

    @staticmethod
    def _sort_wide(d_temp_storage, temp_storage_bytes, d_keys, d_values, num_items, begin_bit, end_bit, descending,
                   stream, key_type, kb, vb):
        """64-bit keys and/or 64-bit values: the general kernels behind gs_lsb_sort_wide."""
        need = lib.gs_lsb_wide_temp_bytes(num_items, kb, vb)
        if d_temp_storage is None:
            return need
        if end_bit is None:
            end_bit = 8 * kb
        if key_type is None:
            key_type = _KEY_TYPES.get(d_keys.d_buffers[0].dtype, _lib.GS_KEY_U64 if kb == 8 else _lib.GS_KEY_U32)
        for b in d_keys.d_buffers:
            _check_buf(b, num_items, "d_keys", kb)
        keys = (C.c_void_p * 2)(d_keys.d_buffers[0].data_ptr(), d_keys.d_buffers[1].data_ptr())
        vals = None
        if vb:
            for b in d_values.d_buffers:
                _check_buf(b, num_items, "d_values", vb)
            vals = (C.c_void_p * 2)(d_values.d_buffers[0].data_ptr(), d_values.d_buffers[1].data_ptr())
        sel = C.c_int(d_keys.selector)
        err = lib.gs_lsb_sort_wide(C.c_void_p(d_temp_storage.data_ptr()),
                                   min(temp_storage_bytes, d_temp_storage.numel() * d_temp_storage.element_size()),
                                   keys, vals, C.byref(sel), num_items, kb, vb, begin_bit, end_bit, int(descending),
                                   key_type, _stream_ptr(stream))
        check(err, "gs_lsb_sort_wide")
        d_keys.selector = sel.value
        if vb:
            d_values.selector = sel.value
        return need

    @staticmethod
    def _sort_copy(d_temp_storage, temp_storage_bytes, d_keys_in, d_keys_out, d_values_in, d_values_out, num_items,
                   begin_bit, end_bit, descending, stream, key_type):
        """Plain-pointer overloads (device_radix_sort.cuh:156-180, 503-527): input untouched, result in *_out."""
        has_values = d_values_in is not None
        need = lib.gs_lsb_copy_temp_bytes(num_items, int(has_values))
        if d_temp_storage is None:
            return need
        if end_bit is None:
            end_bit = 32
        if key_type is None:
            key_type = _KEY_TYPES.get(d_keys_in.dtype, _lib.GS_KEY_U32)
        _check_buf(d_keys_in, num_items, "d_keys_in")
        _check_buf(d_keys_out, num_items, "d_keys_out")
        if has_values:
            _check_buf(d_values_in, num_items, "d_values_in")
            _check_buf(d_values_out, num_items, "d_values_out")
        err = lib.gs_lsb_sort_copy_u32(C.c_void_p(d_temp_storage.data_ptr()),
                                       min(temp_storage_bytes, d_temp_storage.numel() * d_temp_storage.element_size()),
                                       d_keys_in.data_ptr(), d_keys_out.data_ptr(),
                                       d_values_in.data_ptr() if has_values else None,
                                       d_values_out.data_ptr() if has_values else None, num_items, begin_bit, end_bit,
                                       int(descending), key_type, _stream_ptr(stream))
        check(err, "gs_lsb_sort_copy_u32")
        return need

    @staticmethod
    def SortKeysCopy(d_temp_storage, temp_storage_bytes, d_keys_in, d_keys_out, num_items, begin_bit=0, end_bit=None,
                     stream=None, key_type=None, descending=False):
        return DeviceRadixSort._sort_copy(d_temp_storage, temp_storage_bytes, d_keys_in, d_keys_out, None, None,
                                          num_items, begin_bit, end_bit, descending, stream, key_type)

    @staticmethod
    def SortPairsCopy(d_temp_storage, temp_storage_bytes, d_keys_in, d_keys_out, d_values_in, d_values_out, num_items,
                      begin_bit=0, end_bit=None, stream=None, key_type=None, descending=False):
        return DeviceRadixSort._sort_copy(d_temp_storage, temp_storage_bytes, d_keys_in, d_keys_out, d_values_in,
                                          d_values_out, num_items, begin_bit, end_bit, descending, stream, key_type)

    @staticmethod
    def SortKeys(d_temp_storage, temp_storage_bytes, d_keys, num_items, begin_bit=0, end_bit=None, stream=None,
                 key_type=None):
        return DeviceRadixSort._sort(d_temp_storage, temp_storage_bytes, d_keys, None, num_items, begin_bit, end_bit,
                                     False, stream, key_type)

    @staticmethod
    def SortKeysDescending(d_temp_storage, temp_storage_bytes, d_keys, num_items, begin_bit=0, end_bit=None,
                           stream=None, key_type=None):
        return DeviceRadixSort._sort(d_temp_storage, temp_storage_bytes, d_keys, None, num_items, begin_bit, end_bit,
                                     True, stream, key_type)

    @staticmethod
    def SortPairs(d_temp_storage, temp_storage_bytes, d_keys, d_values, num_items, begin_bit=0, end_bit=None,
                  stream=None, key_type=None):
        return DeviceRadixSort._sort(d_temp_storage, temp_storage_bytes, d_keys, d_values, num_items, begin_bit,
                                     end_bit, False, stream, key_type)

    @staticmethod
    def SortPairsDescending(d_temp_storage, temp_storage_bytes, d_keys, d_values, num_items, begin_bit=0,
                            end_bit=None, stream=None, key_type=None):
        return DeviceRadixSort._sort(d_temp_storage, temp_storage_bytes, d_keys, d_values, num_items, begin_bit,
                                     end_bit, True, stream, key_type)


class DeviceSegmentedRadixSort:
    """cub::DeviceSegmentedRadixSort (lsb/cub/cub/device/device_segmented_radix_sort.cuh), DoubleBuffer form:
    segment i is [d_begin_offsets[i], d_end_offsets[i]) (int32 device tensors; one offsets tensor of
    num_segments + 1 entries can serve as both, the end offsets being offsets[1:]).  Two-phase like CUB."""

    @staticmethod
    def _sort(d_temp_storage, temp_storage_bytes, d_keys, d_values, num_items, num_segments, d_begin_offsets, d_end_offsets,
              begin_bit, end_bit, descending, stream, key_type):
        has_values = d_values is not None
        kb = d_keys.d_buffers[0].element_size()
        vb = d_values.d_buffers[0].element_size() if has_values else 0
        wide = kb == 8 or vb == 8            # the wide element types: gs_segmented_sort_wide
        need = (lib.gs_segmented_wide_temp_bytes(num_items, kb, vb, num_segments) if wide
                else lib.gs_segmented_temp_bytes(num_items, int(has_values), num_segments))
        if d_temp_storage is None:
            return need
        if end_bit is None:
            end_bit = 8 * kb
        if key_type is None:
            key_type = _KEY_TYPES.get(d_keys.d_buffers[0].dtype, _lib.GS_KEY_U32)
        for b in d_keys.d_buffers:
            _check_buf(b, num_items, "d_keys", kb)
        for o in (d_begin_offsets, d_end_offsets):
            if not isinstance(o, torch.Tensor) or o.dtype != torch.int32 or not o.is_cuda or o.numel() < num_segments:
                raise ValueError("segment offsets: expected int32 device tensors of >= num_segments entries")
        keys = (C.c_void_p * 2)(d_keys.d_buffers[0].data_ptr(), d_keys.d_buffers[1].data_ptr())
        vals = None
        if has_values:
            for b in d_values.d_buffers:
                _check_buf(b, num_items, "d_values", vb)
            vals = (C.c_void_p * 2)(d_values.d_buffers[0].data_ptr(), d_values.d_buffers[1].data_ptr())
        sel = C.c_int(d_keys.selector)
        if wide:
            err = lib.gs_segmented_sort_wide(C.c_void_p(d_temp_storage.data_ptr()),
                                             min(temp_storage_bytes, d_temp_storage.numel() * d_temp_storage.element_size()),
                                             keys, vals, C.byref(sel), num_items, num_segments,
                                             C.c_void_p(d_begin_offsets.data_ptr()), C.c_void_p(d_end_offsets.data_ptr()),
                                             kb, vb, begin_bit, end_bit, int(descending), key_type, _stream_ptr(stream))
            check(err, "gs_segmented_sort_wide")
            d_keys.selector = sel.value
            if has_values:
                d_values.selector = sel.value
            return need
        err = lib.gs_segmented_sort_u32(C.c_void_p(d_temp_storage.data_ptr()),
                                        min(temp_storage_bytes, d_temp_storage.numel() * d_temp_storage.element_size()),
                                        keys, vals, C.byref(sel), num_items, num_segments,
                                        C.c_void_p(d_begin_offsets.data_ptr()), C.c_void_p(d_end_offsets.data_ptr()),
                                        begin_bit, end_bit, int(descending), key_type, _stream_ptr(stream))
        check(err, "gs_segmented_sort_u32")
        d_keys.selector = sel.value
        if has_values:
            d_values.selector = sel.value
        return need

    @staticmethod
    def SortKeys(d_temp_storage, temp_storage_bytes, d_keys, num_items, num_segments, d_begin_offsets, d_end_offsets,
                 begin_bit=0, end_bit=None, stream=None, key_type=None):
        return DeviceSegmentedRadixSort._sort(d_temp_storage, temp_storage_bytes, d_keys, None, num_items, num_segments,
                                              d_begin_offsets, d_end_offsets, begin_bit, end_bit, False, stream, key_type)

    @staticmethod
    def SortKeysDescending(d_temp_storage, temp_storage_bytes, d_keys, num_items, num_segments, d_begin_offsets,
                           d_end_offsets, begin_bit=0, end_bit=None, stream=None, key_type=None):
        return DeviceSegmentedRadixSort._sort(d_temp_storage, temp_storage_bytes, d_keys, None, num_items, num_segments,
                                              d_begin_offsets, d_end_offsets, begin_bit, end_bit, True, stream, key_type)

    @staticmethod
    def SortPairs(d_temp_storage, temp_storage_bytes, d_keys, d_values, num_items, num_segments, d_begin_offsets,
                  d_end_offsets, begin_bit=0, end_bit=None, stream=None, key_type=None):
        return DeviceSegmentedRadixSort._sort(d_temp_storage, temp_storage_bytes, d_keys, d_values, num_items, num_segments,
                                              d_begin_offsets, d_end_offsets, begin_bit, end_bit, False, stream, key_type)

    @staticmethod
    def SortPairsDescending(d_temp_storage, temp_storage_bytes, d_keys, d_values, num_items, num_segments, d_begin_offsets,
                            d_end_offsets, begin_bit=0, end_bit=None, stream=None, key_type=None):
        return DeviceSegmentedRadixSort._sort(d_temp_storage, temp_storage_bytes, d_keys, d_values, num_items, num_segments,
                                              d_begin_offsets, d_end_offsets, begin_bit, end_bit, True, stream, key_type)


def _timed(fn):
    start, stop = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    start.record()
    fn()
    stop.record()
    stop.synchronize()
    return start.elapsed_time(stop)


def sortPairsGPU(d_key_buf, d_key_alt_buf, d_value_buf, d_value_alt_buf, num_items, key_type=None):
    """lsb/sort.cu:25-47: ascending stable pair sort; returns (ms, d_keys, d_values)."""
    d_keys, d_values = DoubleBuffer(d_key_buf, d_key_alt_buf), DoubleBuffer(d_value_buf, d_value_alt_buf)
    nbytes = DeviceRadixSort.SortPairs(None, 0, d_keys, d_values, num_items)
    temp = torch.empty(nbytes, dtype=torch.uint8, device=d_key_buf.device)
    ms = _timed(lambda: DeviceRadixSort.SortPairs(temp, nbytes, d_keys, d_values, num_items, key_type=key_type))
    return ms, d_keys, d_values


def sortKeysGPU(d_key_buf, d_key_alt_buf, num_items, key_type=None):
    """lsb/sort.cu:49-76: sizes with SortKeys, runs SortKeysDescending (:65); returns (ms, d_keys)."""
    d_keys = DoubleBuffer(d_key_buf, d_key_alt_buf)
    nbytes = DeviceRadixSort.SortKeys(None, 0, d_keys, num_items)
    temp = torch.empty(nbytes, dtype=torch.uint8, device=d_key_buf.device)
    ms = _timed(lambda: DeviceRadixSort.SortKeysDescending(temp, nbytes, d_keys, num_items, key_type=key_type))
    return ms, d_keys


def lsb_pass_kernels(keys_in, vals_in, shift, bits, descending=False, stream=None):
    """Run the three kernels of ONE pass separately (bring-up / parity of SURVEY.md rows L4-L6).

    Returns dict(spine_counts, prefix16, spine_scanned, totals, keys_out, vals_out, grid, tile, tiles_per_chunk)."""
    import numpy as np
    n = keys_in.numel()
    g, t, c = C.c_uint32(), C.c_uint32(), C.c_uint32()
    lib.gs_lsb_geometry(n, int(vals_in is not None), C.byref(g), C.byref(t), C.byref(c))
    dev = keys_in.device
    nbytes = lib.gs_lsb_temp_bytes(n, int(vals_in is not None))
    temp = torch.zeros(nbytes, dtype=torch.uint8, device=dev)
    sp, tot, pf = C.c_void_p(), C.c_void_p(), C.c_void_p()
    lib.gs_lsb_workspace_layout(temp.data_ptr(), n, C.byref(sp), C.byref(tot), C.byref(pf))
    base = temp.data_ptr()
    num_tiles = (n + t.value - 1) // t.value

    def view(ptr, count, dtype):
        off = ptr.value - base
        return temp[off: off + count * dtype.itemsize].cpu().numpy().view(dtype).copy()

    keys_out = torch.empty_like(keys_in)
    vals_out = torch.empty_like(vals_in) if vals_in is not None else None
    s = _stream_ptr(stream)
    check(lib.gs_lsb_upsweep_u32(temp.data_ptr(), nbytes, keys_in.data_ptr(), n, shift, bits, int(descending),
                                 _lib.GS_KEY_U32, s), "gs_lsb_upsweep_u32")
    torch.cuda.synchronize()
    counts = view(sp, 256 * g.value, np.dtype(np.uint32))
    prefix16 = view(pf, num_tiles * 256, np.dtype(np.uint16))
    check(lib.gs_lsb_scan_spine(temp.data_ptr(), nbytes, n, s), "gs_lsb_scan_spine")
    torch.cuda.synchronize()
    scanned = view(sp, 256 * g.value, np.dtype(np.uint32))
    totals = view(tot, 256, np.dtype(np.uint32))
    check(lib.gs_lsb_downsweep_u32(temp.data_ptr(), nbytes, keys_in.data_ptr(), keys_out.data_ptr(),
                                   vals_in.data_ptr() if vals_in is not None else None,
                                   vals_out.data_ptr() if vals_out is not None else None,
                                   n, shift, bits, int(descending), _lib.GS_KEY_U32, _lib.GS_KEY_U32, s),
          "gs_lsb_downsweep_u32")
    return dict(spine_counts=counts, prefix16=prefix16, spine_scanned=scanned, totals=totals, keys_out=keys_out,
                vals_out=vals_out, grid=g.value, tile=t.value, tiles_per_chunk=c.value)
