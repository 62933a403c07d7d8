"""GPU parity tests for the wider element types of the DeviceRadixSort contract
(gs_lsb_sort_wide): 64-bit unsigned / signed / double keys with no, 32-bit or
64-bit values, and 32-bit keys with 64-bit values.

Model: lsb/cub/test/test_device_radix_sort.cu:1244-1265 (key types
unsigned long long / long long / double, value types incl. 64-bit), bit ranges
full / [1, bits-1) / the two middle bits (:973-995), ascending + descending,
sizes shrinking to 1 and 0 (:1034-1046), NaNs never generated (test_util.h
RandomBits).  Expected ranks come from the CPU oracle
(oracle.lsb_reference_ranks_u64, the 64-bit InitializeSolution restatement).
"""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu

KT = {"u64": 3, "i64": 4, "f64": 5}


def _keys64(n, seed, kind, entropy_and=0):
    rng = np.random.default_rng(seed)
    k = rng.integers(0, 2**64, size=n, dtype=np.uint64)
    for _ in range(entropy_and):                            # test_util.h RandomBits entropy reduction
        k &= rng.integers(0, 2**64, size=n, dtype=np.uint64)
    if kind == "f64":                                       # replace NaN patterns, keep +-0 and infinities
        f = k.view(np.float64)
        k = np.where(np.isnan(f), np.uint64(0x8000000000000000), k)
        if n > 8:
            k[3] = 0; k[5] = np.uint64(0x8000000000000000); k[7] = np.uint64(0x7FF0000000000000)
    return np.ascontiguousarray(k)


def _dev64(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int64).copy()).to(dev)


def _dev32(a, dev):
    return torch.from_numpy(np.ascontiguousarray(a).view(np.int32).copy()).to(dev)


def _sort_wide(gs, keys, vals, key_type, begin_bit=0, end_bit=None, descending=False, dev="cuda:0"):
    n = keys.size
    kd = _dev64 if keys.dtype.itemsize == 8 else _dev32
    d_keys = gs.DoubleBuffer(kd(keys, dev), torch.empty_like(kd(keys, dev)))
    d_vals = None
    if vals is not None:
        vd = _dev64 if vals.dtype.itemsize == 8 else _dev32
        d_vals = gs.DoubleBuffer(vd(vals, dev), torch.empty_like(vd(vals, dev)))
    if d_vals is None:
        fn = gs.DeviceRadixSort.SortKeysDescending if descending else gs.DeviceRadixSort.SortKeys
        nb = fn(None, 0, d_keys, n)
        temp = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
        fn(temp, nb, d_keys, n, begin_bit, end_bit, key_type=key_type)
    else:
        fn = gs.DeviceRadixSort.SortPairsDescending if descending else gs.DeviceRadixSort.SortPairs
        nb = fn(None, 0, d_keys, d_vals, n)
        temp = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
        fn(temp, nb, d_keys, d_vals, n, begin_bit, end_bit, key_type=key_type)
    torch.cuda.synchronize()
    ko = d_keys.Current().cpu().numpy().view(keys.dtype)[:n]
    vo = d_vals.Current().cpu().numpy().view(vals.dtype)[:n] if d_vals is not None else None
    return ko, vo, d_keys


@pytest.mark.parametrize("kind", ["u64", "i64", "f64"])
@pytest.mark.parametrize("n", [0, 1, 63, 4095, 4096, 4097, 100003, (1 << 20) + 13])
def test_keys64_full_range(gs, cuda, oracle, kind, n):
    keys = _keys64(n, n + 1, kind)
    for desc in (False, True):
        ranks = oracle.lsb_reference_ranks_u64(keys, KT[kind], 0, 64, desc)
        ko, _, _ = _sort_wide(gs, keys, None, KT[kind], descending=desc)
        assert np.array_equal(ko, keys[ranks]), (kind, n, desc)
        vals = np.arange(n, dtype=np.uint32)
        ko, vo, _ = _sort_wide(gs, keys, vals, KT[kind], descending=desc)
        assert np.array_equal(vo, ranks), (kind, n, desc)             # stable: equals the reference ranks
        assert np.array_equal(ko, keys[ranks])


@pytest.mark.parametrize("begin_bit,end_bit", [(1, 63), (31, 33), (0, 8), (56, 64), (7, 24), (20, 20)])
@pytest.mark.parametrize("kind", ["u64", "i64"])
def test_keys64_bit_ranges(gs, cuda, oracle, kind, begin_bit, end_bit):
    n = 50021
    keys = _keys64(n, 77, kind, entropy_and=1)
    vals = (np.arange(n, dtype=np.uint64) << np.uint64(33)) | np.uint64(5)      # 64-bit values
    for desc in (False, True):
        ko, vo, dk = _sort_wide(gs, keys, vals, KT[kind], begin_bit, end_bit, desc)
        if begin_bit == end_bit:
            assert dk.selector == 0 and np.array_equal(ko, keys) and np.array_equal(vo, vals)
            continue
        ranks = oracle.lsb_reference_ranks_u64(keys, 3 if end_bit < 64 else KT[kind], begin_bit, end_bit, desc)
        assert np.array_equal(vo, vals[ranks]), (kind, begin_bit, end_bit, desc)
        assert np.array_equal(ko, keys[ranks])
        assert dk.selector == ((end_bit - begin_bit + 7) // 8) % 2


@pytest.mark.parametrize("entropy_and", [0, 3, 6])
def test_keys64_entropy_reduced_with_values64(gs, cuda, oracle, entropy_and):
    n = 300007
    keys = _keys64(n, 5 + entropy_and, "u64", entropy_and)
    vals = np.random.default_rng(9).integers(0, 2**64, size=n, dtype=np.uint64)
    ranks = oracle.lsb_reference_ranks_u64(keys, 3)
    ko, vo, _ = _sort_wide(gs, keys, vals, 3)
    assert np.array_equal(ko, keys[ranks]) and np.array_equal(vo, vals[ranks])


@pytest.mark.parametrize("n", [1, 4097, 250001])
def test_keys32_values64(gs, cuda, oracle, n):
    """(u32 key, 64-bit value) pairs: key types of the 32-bit family through the wide entry."""
    keys = oracle.gen_uniform(n, seed=3) & np.uint32(0xFFFF00FF)
    vals = np.random.default_rng(n).integers(0, 2**64, size=n, dtype=np.uint64)
    for desc in (False, True):
        ranks = oracle.lsb_reference_ranks(keys, 0, 32, desc)
        ko, vo, _ = _sort_wide(gs, keys, vals, gs.GS_KEY_U32, descending=desc)
        assert np.array_equal(ko, keys[ranks]) and np.array_equal(vo, vals[ranks])
    f = np.random.default_rng(4).standard_normal(n).astype(np.float32)
    ko, vo, _ = _sort_wide(gs, f.view(np.uint32), vals, gs.GS_KEY_F32)
    order = np.argsort(f, kind="stable")
    assert np.array_equal(ko.view(np.float32), f[order]) and np.array_equal(vo, vals[order])


def test_double_keys_special_values(gs, cuda, oracle):
    f = np.array([0.0, -0.0, 1.5, -1.5, np.inf, -np.inf, 5e-324, -5e-324, 1e308, -1e308, 0.0, -0.0], dtype=np.float64)
    keys = np.tile(f.view(np.uint64), 700)
    vals = np.arange(keys.size, dtype=np.uint32)
    ranks = oracle.lsb_reference_ranks_u64(keys, 5)
    ko, vo, _ = _sort_wide(gs, keys, vals, 5)
    assert np.array_equal(vo, ranks) and np.array_equal(ko, keys[ranks])
    kf = ko.view(np.float64)
    assert np.all(kf[1:] >= kf[:-1])
    zeros = ko[kf == 0.0]
    assert np.all(np.diff((zeros >> np.uint64(63)).astype(np.int64)) <= 0)     # every -0 before every +0


def test_wide_dtype_dispatch_and_errors(gs, cuda):
    n = 10000
    a = torch.randint(-2**62, 2**62, (n,), dtype=torch.int64, device=cuda)
    want = torch.sort(a).values
    dk = gs.DoubleBuffer(a.clone(), torch.empty_like(a))
    nb = gs.DeviceRadixSort.SortKeys(None, 0, dk, n)
    temp = torch.empty(nb, dtype=torch.uint8, device=cuda)
    gs.DeviceRadixSort.SortKeys(temp, nb, dk, n)                 # key type inferred from torch.int64
    assert torch.equal(dk.Current(), want)
    d = torch.randn(n, dtype=torch.float64, device=cuda)
    dk = gs.DoubleBuffer(d.clone(), torch.empty_like(d))
    gs.DeviceRadixSort.SortKeysDescending(temp, nb, dk, n)
    assert torch.equal(dk.Current(), torch.sort(d, descending=True).values)
    with pytest.raises(gs.GpuSortError):                         # temp too small
        gs.DeviceRadixSort.SortKeys(temp, 16, gs.DoubleBuffer(a.clone(), torch.empty_like(a)), n)
    with pytest.raises(gs.GpuSortError):                         # end_bit beyond the key width
        gs.DeviceRadixSort.SortKeys(temp, nb, gs.DoubleBuffer(a.clone(), torch.empty_like(a)), n, 0, 65)
    with pytest.raises(gs.GpuSortError):                         # 32-bit key type with 64-bit keys
        gs.DeviceRadixSort.SortKeys(temp, nb, gs.DoubleBuffer(a.clone(), torch.empty_like(a)), n, key_type=gs.GS_KEY_U32)


def test_large_int64_against_torch_sort(gs, cuda):
    """2^27 signed 64-bit keys with 64-bit payloads: sortedness + permutation checked on the device."""
    n = 1 << 27
    g = torch.Generator(device=cuda); g.manual_seed(11)
    a = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=cuda, generator=g)
    v = a ^ 0x5DEECE66D                                          # payload determined by the key
    dk = gs.DoubleBuffer(a.clone(), torch.empty_like(a))
    dv = gs.DoubleBuffer(v, torch.empty_like(v))
    nb = gs.DeviceRadixSort.SortPairs(None, 0, dk, dv, n)
    temp = torch.empty(nb, dtype=torch.uint8, device=cuda)
    gs.DeviceRadixSort.SortPairs(temp, nb, dk, dv, n)
    out, vout = dk.Current(), dv.Current()
    assert bool((out[1:] >= out[:-1]).all())
    assert bool(((out ^ 0x5DEECE66D) == vout).all())
    assert int(out.sum()) == int(a.sum()) and int(torch.bitwise_xor(out[::2], out[1::2]).sum()) != 0
    del dk, dv, vout
    assert torch.equal(out, torch.sort(a).values)


def test_wide_sort_is_capturable_in_a_hip_graph(gs, cuda):
    """64-bit keys: captured once, replayed three times on new data in the same buffers."""
    n = 150001
    sets = [_keys64(n, seed, "u64") for seed in (11, 12, 13)]
    src = _dev64(sets[0], cuda)
    a, b = src.clone(), torch.empty_like(src)
    dk = gs.DoubleBuffer(a, b)
    nb = gs.DeviceRadixSort.SortKeys(None, 0, dk, n)
    temp = torch.empty(max(nb, 1), dtype=torch.uint8, device=cuda)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, key_type=KT["u64"])
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    dk.selector = 0
    with torch.cuda.graph(g, stream=side):
        a.copy_(src)
        gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, key_type=KT["u64"])
    out = dk.Current()
    for keys in sets:
        src.copy_(_dev64(keys, cuda))
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(out.cpu().numpy().view(np.uint64)[:n], np.sort(keys))
