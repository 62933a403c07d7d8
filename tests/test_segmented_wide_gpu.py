"""Segmented sort for the wide element types (gs_segmented_sort_wide; cub's segmented dispatch is type-generic,
dispatch_radix_sort.cuh:321-432): 64-bit keys (u64 / i64 / f64) with no, 32-bit or 64-bit values and 32-bit keys with
64-bit values.  Expected results: the oracle's 64-bit reference ranks applied per segment (stable, so values are the ranks
exactly); bit sub-ranges, descending, empty segments and gaps between segments (positions outside every segment stay)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GS_KEY_U64, GS_KEY_I64, GS_KEY_F64 = 3, 4, 5


def _expected(oracle, keys_u64, key_type, begins, ends, begin_bit, end_bit, desc):
    """Full key width: the oracle's reference ranks (InitializeSolution).  Bit sub-ranges: the reference's own test masks
    the RAW key bits and therefore runs them for unsigned keys only (test_device_radix_sort.cu: "the bit-flipping techniques
    get in the way"); the device semantics -- cub's and ours -- are bits [begin, end) of the order-preserving image of the
    key (util_type.cuh:966-1089), restated here with numpy for signed and float keys."""
    ranks = np.arange(keys_u64.size, dtype=np.int64)
    full = begin_bit == 0 and end_bit == 64
    if not full:
        k = keys_u64
        if key_type == GS_KEY_I64:
            img = k ^ np.uint64(1 << 63)
        elif key_type == GS_KEY_F64:
            img = np.where((k >> np.uint64(63)).astype(bool), ~k, k ^ np.uint64(1 << 63))
        else:
            img = k
        mask = np.uint64((1 << (end_bit - begin_bit)) - 1)
        d = (img >> np.uint64(begin_bit)) & mask
        if desc:
            d = mask - d
    for lo, hi in zip(begins, ends):
        if hi > lo:
            if full:
                ranks[lo:hi] = lo + oracle.lsb_reference_ranks_u64(keys_u64[lo:hi], key_type, 0, 64, desc).astype(np.int64)
            else:
                ranks[lo:hi] = lo + np.argsort(d[lo:hi], kind="stable")
    return ranks


def _offsets(rng, n, nseg, gaps):
    cuts = np.sort(rng.integers(0, n + 1, size=2 * nseg if gaps else nseg + 1))
    if gaps:
        return cuts[0::2].astype(np.int64), cuts[1::2].astype(np.int64)
    cuts[0], cuts[-1] = 0, n
    return cuts[:-1].astype(np.int64), cuts[1:].astype(np.int64)


@pytest.mark.parametrize("kind", ["u64", "i64", "f64", "u64_dups"])
@pytest.mark.parametrize("n,nseg", [(1, 1), (5000, 3), (100003, 1), (100003, 700), (1200007, 5), (1200007, 30000)])
def test_wide_segments_keys_and_pairs(gs, oracle, cuda, kind, n, nseg):
    rng = np.random.default_rng(n * 7 + nseg)
    if kind == "f64":
        keys = (rng.standard_normal(n) * np.exp(rng.uniform(-100, 100, size=n))).view(np.uint64)
        kt = GS_KEY_F64
    elif kind == "i64":
        keys = rng.integers(-2**63, 2**63, size=n, dtype=np.int64).view(np.uint64)
        kt = GS_KEY_I64
    elif kind == "u64_dups":
        keys = rng.integers(0, 2**64, size=50, dtype=np.uint64)[rng.integers(0, 50, size=n)]
        kt = GS_KEY_U64
    else:
        keys = rng.integers(0, 2**64, size=n, dtype=np.uint64)
        kt = GS_KEY_U64
    S = gs.DeviceSegmentedRadixSort
    for desc, gaps, (bb, eb), vdt in ((False, False, (0, 64), np.int64), (True, True, (0, 64), np.int32), (False, True, (7, 53), None),
                                      (True, False, (40, 44), np.int64)):
        begins, ends = _offsets(rng, n, nseg, gaps)
        ranks = _expected(oracle, keys, kt, begins, ends, bb, eb, desc)
        ob = torch.from_numpy(begins.astype(np.int32)).to(cuda)
        oe = torch.from_numpy(ends.astype(np.int32)).to(cuda)
        dk = gs.DoubleBuffer(torch.from_numpy(keys.view(np.int64).copy()).to(cuda), torch.from_numpy(keys.view(np.int64).copy()).to(cuda))
        if vdt is None:
            fn = S.SortKeysDescending if desc else S.SortKeys
            nb = fn(None, 0, dk, n, nseg, ob, oe)
            temp = torch.empty(max(nb, 1), dtype=torch.uint8, device=cuda)
            fn(temp, nb, dk, n, nseg, ob, oe, bb, eb, key_type=kt)
        else:
            vals = np.arange(n, dtype=vdt)
            dv = gs.DoubleBuffer(torch.from_numpy(vals.copy()).to(cuda), torch.from_numpy(vals.copy()).to(cuda))
            fn = S.SortPairsDescending if desc else S.SortPairs
            nb = fn(None, 0, dk, dv, n, nseg, ob, oe)
            temp = torch.empty(max(nb, 1), dtype=torch.uint8, device=cuda)
            fn(temp, nb, dk, dv, n, nseg, ob, oe, bb, eb, key_type=kt)
            assert np.array_equal(dv.Current().cpu().numpy().astype(np.int64), ranks), (kind, n, nseg, desc, bb, eb)
        torch.cuda.synchronize()
        assert np.array_equal(dk.Current().cpu().numpy().view(np.uint64), keys[ranks]), (kind, n, nseg, desc, bb, eb)


def test_wide_segments_u32_keys_with_64bit_values(gs, oracle, cuda):
    n, nseg = 500009, 211
    rng = np.random.default_rng(3)
    keys = oracle.gen_uniform(n, seed=12)
    begins, ends = _offsets(rng, n, nseg, False)
    ranks = np.arange(n, dtype=np.int64)
    for lo, hi in zip(begins, ends):
        if hi > lo:
            ranks[lo:hi] = lo + oracle.lsb_reference_ranks(keys[lo:hi], 0, 32, False).astype(np.int64)
    vals = np.arange(n, dtype=np.int64) * 5
    dk = gs.DoubleBuffer(torch.from_numpy(keys.view(np.int32).copy()).to(cuda), torch.empty(n, dtype=torch.int32, device=cuda))
    dv = gs.DoubleBuffer(torch.from_numpy(vals.copy()).to(cuda), torch.empty(n, dtype=torch.int64, device=cuda))
    ob = torch.from_numpy(begins.astype(np.int32)).to(cuda)
    oe = torch.from_numpy(ends.astype(np.int32)).to(cuda)
    S = gs.DeviceSegmentedRadixSort
    nb = S.SortPairs(None, 0, dk, dv, n, nseg, ob, oe)
    temp = torch.empty(max(nb, 1), dtype=torch.uint8, device=cuda)
    S.SortPairs(temp, nb, dk, dv, n, nseg, ob, oe, key_type=gs.GS_KEY_U32)
    torch.cuda.synchronize()
    assert np.array_equal(dk.Current().cpu().numpy().view(np.uint32), keys[ranks])
    assert np.array_equal(dv.Current().cpu().numpy(), vals[ranks])
