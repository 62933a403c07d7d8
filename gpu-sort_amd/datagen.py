"""On-device synthetic inputs and result checks (C ABI: gs_generate_u32, gs_check_*).

generate_random_keys mirrors msb/tests/data_gen.h:43-75 (entropy levels via
repeated AND; level <= 0 gives all zeros); generate_enumerated_values mirrors
:78-84.  The stream is the counter-based splitmix64 of SURVEY.md 8d instead of
cuRAND XORWOW, identical on CPU (oracle/oracle.c) and GPU.
"""
import ctypes as C

import torch

from . import _lib
from ._lib import lib, check
from .lsb import _stream_ptr


def _gen(n, kind, seed, start, level, device, out, stream):
    if out is None:
        out = torch.empty(n, dtype=torch.int32, device=device)
    check(lib.gs_generate_u32(out.data_ptr(), n, kind, seed, start, level, _stream_ptr(stream)), "gs_generate_u32")
    return out


def generate_uniform_keys(num_keys, seed=0, start=0, device="cuda", out=None, stream=None):
    return _gen(num_keys, _lib.GS_GEN_UNIFORM, seed, start, 1, device, out, stream)


def generate_zipf_keys(num_keys, seed=0, start=0, device="cuda", out=None, stream=None):
    return _gen(num_keys, _lib.GS_GEN_ZIPF, seed, start, 1, device, out, stream)


def generate_random_keys(num_keys, seed=0, entropy_level=1, start=0, device="cuda", out=None, stream=None):
    return _gen(num_keys, _lib.GS_GEN_ENTROPY_AND, seed, start, entropy_level, device, out, stream)


def generate_enumerated_values(num_values, start=0, device="cuda", out=None, stream=None):
    return _gen(num_values, _lib.GS_GEN_ENUMERATED, 0, start, 1, device, out, stream)


def check_sorted(d_keys, num_items=None, descending=False, stream=None):
    """-> (adjacent inversions, multiset sum, multiset xor) computed on the device."""
    n = d_keys.numel() if num_items is None else num_items
    res = torch.zeros(3, dtype=torch.int64, device=d_keys.device)
    check(lib.gs_check_sorted_u32(d_keys.data_ptr(), n, int(descending), res.data_ptr(), _stream_ptr(stream)),
          "gs_check_sorted_u32")
    r = res.cpu().numpy().astype("uint64")
    return int(r[0]), int(r[1]), int(r[2])


def check_pairs_enumerated(d_keys_in, d_keys_sorted, d_vals, num_items=None, stream=None):
    """-> (mismatches, sum of values): msb/tests/test_sort_pairs.cu:141-146,166-176 on the device."""
    n = d_keys_in.numel() if num_items is None else num_items
    res = torch.zeros(2, dtype=torch.int64, device=d_keys_in.device)
    check(lib.gs_check_pairs_enumerated_u32(d_keys_in.data_ptr(), d_keys_sorted.data_ptr(), d_vals.data_ptr(), n,
                                            res.data_ptr(), _stream_ptr(stream)), "gs_check_pairs_enumerated_u32")
    r = res.cpu().numpy().astype("uint64")
    return int(r[0]), int(r[1])
