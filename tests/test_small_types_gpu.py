"""GPU parity tests for the rest of DeviceRadixSort's type contract (gs_lsb_sort_any through the C ABI and the Python
mirror): 8- and 16-bit keys (bool / char / signed char / unsigned char / short / unsigned short) and values of any size
(none, the key's own 1- or 2-byte type, 32- and 64-bit, 16-byte records like TestFoo).

Model: lsb/cub/test/test_device_radix_sort.cu:930-945 (value types per key type), :1244-1250 (key types), :634-693
(InitializeSolution: mask to [begin_bit, end_bit), reverse / stable_sort / reverse for descending; values follow the
ranks, :888-889).  The expectation comes from the oracle's reference ranks on the keys' order-preserving 32-bit images
(the same rule the 32-bit tests are held to), so every comparison is bit-exact."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KEY_KINDS = {               # name -> (numpy dtype, torch dtype, gs key type name, bits)
    "bool": (np.uint8, torch.bool, "GS_KEY_U8", 8),
    "u8": (np.uint8, torch.uint8, "GS_KEY_U8", 8),
    "i8": (np.int8, torch.int8, "GS_KEY_I8", 8),
    "u16": (np.uint16, torch.int16, "GS_KEY_U16", 16),       # torch's uint16 support is partial: carried as int16 bit patterns
    "i16": (np.int16, torch.int16, "GS_KEY_I16", 16),
}


def _image(keys_np, bits, signed):
    """order-preserving u32 image of a narrow key: zero-extended, sign bit of the narrow type flipped for signed keys"""
    u = keys_np.view(np.uint8 if bits == 8 else np.uint16).astype(np.uint32)
    return u ^ np.uint32(1 << (bits - 1)) if signed else u


def _gen_keys(kind, n, seed):
    npdt, _, _, bits = KEY_KINDS[kind]
    rng = np.random.default_rng(seed)
    raw = rng.integers(0, 1 << bits, size=n, dtype=np.uint32)
    raw &= rng.integers(0, 1 << bits, size=n, dtype=np.uint32)            # some entropy reduction: duplicates (test_util.h RandomBits)
    if kind == "bool":
        raw &= 1
    return raw.astype(np.uint8 if bits == 8 else np.uint16).view(npdt)


def _values(n, vkind, keys_np, rng):
    if vkind == "none":
        return None
    if vkind == "key":
        return rng.integers(0, 256, size=n).astype(keys_np.dtype) if keys_np.dtype.itemsize == 1 else rng.integers(0, 65536, size=n).astype(np.uint16).view(keys_np.dtype)
    if vkind == "u32":
        return np.arange(n, dtype=np.uint32).view(np.int32)
    if vkind == "u64":
        return (np.arange(n, dtype=np.uint64) * np.uint64(0x100000001)).view(np.int64)
    if vkind == "foo16":                                                   # 16-byte records: rows of four int32
        return rng.integers(-2**31, 2**31 - 1, size=(n, 4), dtype=np.int64).astype(np.int32)
    raise ValueError(vkind)


@pytest.mark.parametrize("kind", list(KEY_KINDS))
@pytest.mark.parametrize("vkind", ["none", "key", "u32", "u64", "foo16"])
def test_small_key_types_and_value_sizes(gs, cuda, oracle, kind, vkind):
    npdt, tdt, ktname, bits = KEY_KINDS[kind]
    key_type = getattr(gs, ktname)
    signed = ktname in ("GS_KEY_I8", "GS_KEY_I16")
    rng = np.random.default_rng(7)
    for n, desc, bb, eb in ((0, False, 0, bits), (1, True, 0, bits), (777, False, 0, bits), (17409, True, 1, bits - 1),
                            (100003, False, 0, bits), (100003, True, 0, bits), ((1 << 20) + 7, False, 2, bits)):
        keys = _gen_keys(kind, n, seed=n + 1)
        vals = _values(n, vkind, keys, rng)
        ranks = oracle.lsb_reference_ranks(_image(keys, bits, signed), bb, eb, desc)
        tk = torch.from_numpy(keys.view(np.uint8 if bits == 8 else np.int16).copy()).to(cuda)
        if kind == "bool":
            tk = tk.view(torch.bool)
        elif kind in ("u8",):
            tk = tk.view(torch.uint8)
        elif kind == "i8":
            tk = tk.view(torch.int8)
        dk = gs.DoubleBuffer(tk, torch.empty_like(tk))
        dv = None
        if vals is not None:
            tv = torch.from_numpy(vals.copy()).to(cuda)
            dv = gs.DoubleBuffer(tv, torch.empty_like(tv))
        fn = (gs.DeviceRadixSort.SortPairsDescending if desc else gs.DeviceRadixSort.SortPairs) if dv is not None else \
             (gs.DeviceRadixSort.SortKeysDescending if desc else gs.DeviceRadixSort.SortKeys)
        args = (dk, dv, n) if dv is not None else (dk, n)
        nb = fn(None, 0, *args, key_type=key_type)
        temp = torch.empty(max(nb, 1), dtype=torch.uint8, device=cuda)
        fn(temp, nb, *args, bb, eb, key_type=key_type)
        torch.cuda.synchronize()
        got_k = dk.Current().cpu().numpy().view(np.uint8 if bits == 8 else np.uint16)
        exp_k = keys.view(np.uint8 if bits == 8 else np.uint16)[ranks]
        assert np.array_equal(got_k[:n], exp_k), (kind, vkind, n, desc, bb, eb)
        if dv is not None:
            assert np.array_equal(dv.Current().cpu().numpy()[:n], vals[ranks]), (kind, vkind, n, desc, bb, eb)
        if n:      # the input buffers are untouched (plain-pointer form underneath) and the selector flipped once
            assert dk.selector == 1 and np.array_equal(dk.Alternate().cpu().numpy().view(got_k.dtype)[:n], keys.view(got_k.dtype))


@pytest.mark.parametrize("key_dtype,ktname", [(np.uint32, "GS_KEY_U32"), (np.int32, "GS_KEY_I32"), (np.float32, "GS_KEY_F32"),
                                               (np.uint64, "GS_KEY_U64"), (np.int64, "GS_KEY_I64"), (np.float64, "GS_KEY_F64")])
@pytest.mark.parametrize("vbytes", [1, 2, 16])
def test_wide_keys_with_odd_value_sizes(gs, cuda, oracle, key_dtype, ktname, vbytes):
    """32- and 64-bit keys with 1-, 2- and 16-byte values (TestBackend<KeyT, TestFoo> for every key type)"""
    key_type = getattr(gs, ktname)
    rng = np.random.default_rng(11)
    n = 200003
    bits = 8 * np.dtype(key_dtype).itemsize
    raw = rng.integers(0, 2**63, size=n, dtype=np.uint64) & rng.integers(0, 2**63, size=n, dtype=np.uint64)
    keys = (raw >> np.uint64(64 - bits)).astype(np.uint64 if bits == 64 else np.uint32)
    if key_dtype in (np.float32, np.float64):       # no NaNs (test_util.h RandomBits)
        fk = keys.view(key_dtype)
        keys = np.where(np.isnan(fk), np.array(1.5, key_dtype), fk).view(keys.dtype)
    vals = {1: rng.integers(0, 256, size=n).astype(np.uint8), 2: rng.integers(-2**15, 2**15, size=n).astype(np.int16),
            16: rng.integers(-2**31, 2**31 - 1, size=(n, 4)).astype(np.int32)}[vbytes]
    for desc in (False, True):
        if bits == 32:
            img = keys.copy()
            if key_dtype == np.int32:
                img ^= np.uint32(0x80000000)
            elif key_dtype == np.float32:
                img = np.where(img >> 31, ~img, img | np.uint32(0x80000000)).astype(np.uint32)
            ranks = oracle.lsb_reference_ranks(img, 0, 32, desc)
        else:
            ranks = oracle.lsb_reference_ranks_u64(keys, {"GS_KEY_U64": 3, "GS_KEY_I64": 4, "GS_KEY_F64": 5}[ktname], 0, 64, desc)
        tk = torch.from_numpy(keys.view(np.int32 if bits == 32 else np.int64).copy()).to(cuda)
        tv = torch.from_numpy(vals.copy()).to(cuda)
        dk, dv = gs.DoubleBuffer(tk, torch.empty_like(tk)), gs.DoubleBuffer(tv, torch.empty_like(tv))
        fn = gs.DeviceRadixSort.SortPairsDescending if desc else gs.DeviceRadixSort.SortPairs
        nb = fn(None, 0, dk, dv, n, key_type=key_type)
        temp = torch.empty(nb, dtype=torch.uint8, device=cuda)
        fn(temp, nb, dk, dv, n, key_type=key_type)
        torch.cuda.synchronize()
        assert np.array_equal(dk.Current().cpu().numpy().view(keys.dtype), keys[ranks]), (ktname, vbytes, desc)
        assert np.array_equal(dv.Current().cpu().numpy(), vals[ranks]), (ktname, vbytes, desc)


def test_any_argument_checks(gs, cuda):
    import ctypes as C
    t = torch.zeros(64, dtype=torch.uint8, device=cuda)
    f = gs.lib.gs_lsb_sort_any
    assert gs.lib.gs_lsb_any_temp_bytes(1000, 99, 0) == 0                                      # unknown key type
    nb = gs.lib.gs_lsb_any_temp_bytes(16, gs.GS_KEY_U8, 0)
    temp = torch.empty(nb, dtype=torch.uint8, device=cuda)
    assert f(temp.data_ptr(), nb, t.data_ptr(), t.data_ptr(), None, None, 16, gs.GS_KEY_U8, 0, 0, 8, 0, None) == 1     # in == out
    o = torch.zeros(64, dtype=torch.uint8, device=cuda)
    assert f(temp.data_ptr(), nb, t.data_ptr(), o.data_ptr(), None, None, 16, gs.GS_KEY_U8, 0, 0, 9, 0, None) == 1     # end_bit beyond the key
    assert f(temp.data_ptr(), 8, t.data_ptr(), o.data_ptr(), None, None, 16, gs.GS_KEY_U8, 0, 0, 8, 0, None) == 1       # workspace too small
    assert f(temp.data_ptr(), nb, t.data_ptr(), o.data_ptr(), t.data_ptr(), None, 16, gs.GS_KEY_U8, 1, 0, 8, 0, None) == 1   # values out missing
    assert f(None, 0, None, None, None, None, 0, gs.GS_KEY_U8, 1, 0, 8, 0, None) == 0                                    # empty: nothing to do
