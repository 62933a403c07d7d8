"""GPU parity tests of the segmented sort (gs_segmented_sort_u32) through the C ABI.

Model: lsb/cub/test/test_device_radix_sort.cu -- the CUB_SEGMENTED backends (:70, :386-470), segment
offsets from InitializeSegments (:1092-1100: random cut points, empty segments allowed), the reference
solution = reverse / std::stable_sort / reverse per segment on the masked bits (:634-693), bit ranges
full / [1,31) / the two middle bits (:973-995), keys-only and pairs, ascending and descending.
Expected results: the CPU oracle applied per segment.
"""
import numpy as np
import pytest
import torch

from conftest import to_dev, to_u32

pytestmark = pytest.mark.gpu


def _expected(oracle, keys, offsets, begin_bit, end_bit, desc):
    ranks = np.arange(keys.size, dtype=np.int64)
    for lo, hi in zip(offsets[:-1], offsets[1:]):
        if hi > lo:
            ranks[lo:hi] = lo + oracle.lsb_reference_ranks(keys[lo:hi], begin_bit, end_bit, desc).astype(np.int64)
    return ranks


def _run(gs, cuda, keys, vals, begin_offs, end_offs, begin_bit=0, end_bit=32, desc=False, key_type=None):
    n, nseg = keys.size, begin_offs.size
    dk = gs.DoubleBuffer(to_dev(keys, cuda), torch.full((max(n, 1),), -1, dtype=torch.int32, device=cuda))
    dv = gs.DoubleBuffer(to_dev(vals, cuda), torch.full((max(n, 1),), -1, dtype=torch.int32, device=cuda)) if vals is not None else None
    ob = torch.from_numpy(begin_offs.astype(np.int32)).to(cuda)
    oe = torch.from_numpy(end_offs.astype(np.int32)).to(cuda)
    S = gs.DeviceSegmentedRadixSort
    kt = gs.GS_KEY_U32 if key_type is None else key_type
    if dv is None:
        fn = S.SortKeysDescending if desc else S.SortKeys
        nb = fn(None, 0, dk, n, nseg, ob, oe)
        temp = torch.empty(max(nb, 1), dtype=torch.uint8, device=cuda)
        fn(temp, nb, dk, n, nseg, ob, oe, begin_bit, end_bit, key_type=kt)
    else:
        fn = S.SortPairsDescending if desc else S.SortPairs
        nb = fn(None, 0, dk, dv, n, nseg, ob, oe)
        temp = torch.empty(max(nb, 1), dtype=torch.uint8, device=cuda)
        fn(temp, nb, dk, dv, n, nseg, ob, oe, begin_bit, end_bit, key_type=kt)
    torch.cuda.synchronize()
    return to_u32(dk.Current())[:n], (to_u32(dv.Current())[:n] if dv is not None else None), dk


def _random_offsets(rng, n, nseg):
    cuts = np.sort(rng.integers(0, n + 1, size=nseg - 1)) if nseg > 1 else np.zeros(0, np.int64)
    return np.concatenate([[0], cuts, [n]]).astype(np.int64)


@pytest.mark.parametrize("n,nseg", [(1, 1), (1000, 1), (1000, 7), (100003, 1), (100003, 40), (100003, 5000),
                                    (3000017, 3), (3000017, 257), (3000017, 100000)])
def test_segments_full_range(gs, cuda, oracle, n, nseg):
    rng = np.random.default_rng(n + nseg)
    keys = oracle.gen_uniform(n, seed=nseg) & np.uint32(0xFFF0FFFF)
    vals = oracle.gen_enumerated(n)
    offs = _random_offsets(rng, n, nseg)
    for desc in (False, True):
        ranks = _expected(oracle, keys, offs, 0, 32, desc)
        ko, vo, _ = _run(gs, cuda, keys, vals, offs[:-1], offs[1:], desc=desc)
        assert np.array_equal(vo, ranks.astype(np.uint32)), (n, nseg, desc)       # stable: exactly the reference ranks
        assert np.array_equal(ko, keys[ranks])
        ko, _, _ = _run(gs, cuda, keys, None, offs[:-1], offs[1:], desc=desc)
        assert np.array_equal(ko, keys[ranks])


@pytest.mark.parametrize("begin_bit,end_bit", [(1, 31), (15, 17), (0, 8), (24, 32), (5, 20), (12, 12)])
def test_segments_bit_ranges_and_selector(gs, cuda, oracle, begin_bit, end_bit):
    n = 700001
    rng = np.random.default_rng(begin_bit * 37 + end_bit)
    keys = oracle.gen_uniform(n, seed=5)
    vals = oracle.gen_enumerated(n)
    offs = np.concatenate([[0, 0, 10, 3000, 3000, 30000, 400000], _random_offsets(rng, n - 400000, 50)[1:] + 400000])
    for desc in (False, True):
        ko, vo, dk = _run(gs, cuda, keys, vals, offs[:-1], offs[1:], begin_bit, end_bit, desc)
        if begin_bit == end_bit:
            assert dk.selector == 0 and np.array_equal(ko, keys) and np.array_equal(vo, vals)
            continue
        ranks = _expected(oracle, keys, offs, begin_bit, end_bit, desc)
        assert np.array_equal(vo, ranks.astype(np.uint32)), (begin_bit, end_bit, desc)
        assert np.array_equal(ko, keys[ranks])
        assert dk.selector == ((end_bit - begin_bit + 7) // 8) % 2


def test_segments_with_gaps_and_separate_offset_arrays(gs, cuda, oracle):
    """begin/end arrays need not be aliased: gaps between segments are not written (the alternate buffer
    keeps its fill where the result lands there)."""
    n = 200000
    keys = oracle.gen_uniform(n, seed=8)
    vals = oracle.gen_enumerated(n)
    begin = np.array([10, 5000, 5000, 60000, 150000], dtype=np.int64)
    end = np.array([4000, 5000, 50000, 140000, 199990], dtype=np.int64)
    ko, vo, dk = _run(gs, cuda, keys, vals, begin, end, 0, 24)           # 3 passes: result in the alternate buffer
    assert dk.selector == 1
    covered = np.zeros(n, bool)
    for lo, hi in zip(begin, end):
        r = lo + oracle.lsb_reference_ranks(keys[lo:hi], 0, 24, False).astype(np.int64)
        assert np.array_equal(ko[lo:hi], keys[r]) and np.array_equal(vo[lo:hi], r.astype(np.uint32))
        covered[lo:hi] = True
    assert np.all(ko[~covered] == np.uint32(0xFFFFFFFF))                  # untouched fill of the alternate buffer


def test_segments_signed_and_float_keys(gs, cuda, oracle):
    n = 300000
    rng = np.random.default_rng(3)
    offs = _random_offsets(rng, n, 30)
    f = rng.standard_normal(n).astype(np.float32)
    i = rng.integers(-2**31, 2**31, size=n, dtype=np.int64).astype(np.int32)
    kf, _, _ = _run(gs, cuda, f.view(np.uint32), None, offs[:-1], offs[1:], key_type=gs.GS_KEY_F32)
    ki, _, _ = _run(gs, cuda, i.view(np.uint32), None, offs[:-1], offs[1:], desc=True, key_type=gs.GS_KEY_I32)
    for lo, hi in zip(offs[:-1], offs[1:]):
        assert np.array_equal(kf[lo:hi].view(np.float32), np.sort(f[lo:hi]))
        assert np.array_equal(ki[lo:hi].view(np.int32), np.sort(i[lo:hi])[::-1])


@pytest.mark.parametrize("begin_bit,end_bit", [(0, 32), (3, 29), (8, 16), (31, 32)])
def test_tiny_segments_one_wave_each(gs, cuda, oracle, begin_bit, end_bit):
    """Segments of up to 256 / 512 / 1024 elements are sorted by one wave each (seg_wave_sort_kernel, 4 / 8 / 16 elements per
    lane), those of up to 64 elements four at a time by one wave (seg_wave4_sort_kernel): every size around the wave and the
    list edges, next to segments that take the workgroup paths, keys and pairs, both directions, u32 / i32 / f32."""
    sizes = [1, 2, 63, 64, 65, 127, 128, 129, 255, 256, 257, 300, 0, 1, 256, 511, 512, 513, 5000, 1023, 1024, 1025, 17, 9000, 3, 700] * 5
    offs = np.concatenate([[0], np.cumsum(sizes)]).astype(np.int64)
    n = int(offs[-1])
    rng = np.random.default_rng(begin_bit * 41 + end_bit)
    keys = oracle.gen_uniform(n, seed=11) & np.uint32(0xF0FFFF0F)        # duplicates inside the small segments
    vals = oracle.gen_enumerated(n)
    for desc in (False, True):
        ranks = _expected(oracle, keys, offs, begin_bit, end_bit, desc)
        ko, vo, _ = _run(gs, cuda, keys, vals, offs[:-1], offs[1:], begin_bit, end_bit, desc)
        assert np.array_equal(vo, ranks.astype(np.uint32)), (begin_bit, end_bit, desc)   # stable
        assert np.array_equal(ko, keys[ranks])
        ko, _, _ = _run(gs, cuda, keys, None, offs[:-1], offs[1:], begin_bit, end_bit, desc)
        assert np.array_equal(ko, keys[ranks])
    if (begin_bit, end_bit) == (0, 32):
        f = rng.standard_normal(n).astype(np.float32)
        i = rng.integers(-2**31, 2**31, size=n, dtype=np.int64).astype(np.int32)
        kf, _, _ = _run(gs, cuda, f.view(np.uint32), None, offs[:-1], offs[1:], desc=True, key_type=gs.GS_KEY_F32)
        ki, _, _ = _run(gs, cuda, i.view(np.uint32), None, offs[:-1], offs[1:], key_type=gs.GS_KEY_I32)
        for lo, hi in zip(offs[:-1], offs[1:]):
            assert np.array_equal(kf[lo:hi].view(np.float32), np.sort(f[lo:hi])[::-1])
            assert np.array_equal(ki[lo:hi].view(np.int32), np.sort(i[lo:hi]))


def test_segments_large_properties(gs, cuda):
    """2^27 keys in 1000 uneven segments + one of 2^26: every segment sorted, multiset preserved (device checks)."""
    n = 1 << 27
    keys = gs.generate_uniform_keys(n, seed=1, device=cuda)
    _, s0, x0 = gs.check_sorted(keys)
    rng = np.random.default_rng(0)
    offs = np.concatenate([[0, 1 << 26], (1 << 26) + _random_offsets(rng, n - (1 << 26), 1000)[1:]])
    ob = torch.from_numpy(offs.astype(np.int32)).to(cuda)
    dk = gs.DoubleBuffer(keys, torch.empty_like(keys))
    nseg = offs.size - 1
    nb = gs.DeviceSegmentedRadixSort.SortKeys(None, 0, dk, n, nseg, ob[:-1], ob[1:])
    temp = torch.empty(nb, dtype=torch.uint8, device=cuda)
    gs.DeviceSegmentedRadixSort.SortKeys(temp, nb, dk, n, nseg, ob[:-1], ob[1:], key_type=gs.GS_KEY_U32)
    out = dk.Current()
    _, s1, x1 = gs.check_sorted(out)
    assert (s1, x1) == (s0, x0)
    desc_pos = torch.nonzero(out[1:].view(torch.int32).to(torch.int64).bitwise_and(0xFFFFFFFF)
                             < out[:-1].view(torch.int32).to(torch.int64).bitwise_and(0xFFFFFFFF)).flatten().cpu().numpy() + 1
    assert np.all(np.isin(desc_pos, offs))                                # order only breaks at segment starts


def test_segmented_errors(gs, cuda):
    k = torch.zeros(100, dtype=torch.int32, device=cuda)
    dk = gs.DoubleBuffer(k, torch.empty_like(k))
    ob = torch.tensor([0, 50, 100], dtype=torch.int32, device=cuda)
    temp = torch.empty(1 << 20, dtype=torch.uint8, device=cuda)
    with pytest.raises(gs.GpuSortError):
        gs.DeviceSegmentedRadixSort.SortKeys(temp, 16, dk, 100, 2, ob[:-1], ob[1:])          # workspace too small
    with pytest.raises(gs.GpuSortError):
        gs.DeviceSegmentedRadixSort.SortKeys(temp, temp.numel(), dk, 100, 2, ob[:-1], ob[1:], 0, 33)
    with pytest.raises(ValueError):
        gs.DeviceSegmentedRadixSort.SortKeys(temp, temp.numel(), dk, 100, 2, ob[:-1].to(torch.int64), ob[1:])


def test_segment_offsets_out_of_range_are_clamped(gs, cuda, oracle):
    """Offsets outside [0, num_items] are a caller error (CUB: undefined); here they are clamped on the device, so
    the call sorts what lies inside the array and touches nothing else."""
    n = 5000
    keys = oracle.gen_uniform(n, seed=9)
    guard = 4096
    buf = torch.full((n + 2 * guard,), 0x5A5A5A5A, dtype=torch.int32, device=cuda)
    alt = torch.full_like(buf, 0x5A5A5A5A)
    buf[guard:guard + n] = to_dev(keys, cuda)
    dk = gs.DoubleBuffer(buf[guard:guard + n], alt[guard:guard + n])
    ob = torch.tensor([-700, 1000, 4000], dtype=torch.int32, device=cuda)
    oe = torch.tensor([1000, 4000, n + 900], dtype=torch.int32, device=cuda)
    S = gs.DeviceSegmentedRadixSort
    nb = S.SortKeys(None, 0, dk, n, 3, ob, oe)
    temp = torch.empty(max(nb, 1), dtype=torch.uint8, device=cuda)
    S.SortKeys(temp, nb, dk, n, 3, ob, oe, 0, 32, key_type=gs.GS_KEY_U32)
    torch.cuda.synchronize()
    got = to_u32(dk.Current())[:n]
    exp = np.concatenate([np.sort(keys[0:1000]), np.sort(keys[1000:4000]), np.sort(keys[4000:n])])
    assert np.array_equal(got, exp)
    for t in (buf, alt):
        g = t.cpu().numpy()
        assert np.all(g[:guard] == 0x5A5A5A5A) and np.all(g[guard + n:] == 0x5A5A5A5A)


def test_segmented_sort_is_capturable_in_a_hip_graph(gs, cuda, oracle):
    """No host-side decision depends on the segment sizes (they are classified on the device), so a call can be
    captured in a HIP graph and replayed on new keys AND new segment offsets of the same count."""
    n, nseg = 200003, 300
    rng = np.random.default_rng(5)
    S = gs.DeviceSegmentedRadixSort
    src = torch.empty(n, dtype=torch.int32, device=cuda)
    a, b = torch.empty_like(src), torch.empty_like(src)
    ob = torch.empty(nseg, dtype=torch.int32, device=cuda)
    oe = torch.empty(nseg, dtype=torch.int32, device=cuda)
    dk = gs.DoubleBuffer(a, b)
    nb = S.SortKeys(None, 0, dk, n, nseg, ob, oe)
    temp = torch.empty(max(nb, 1), dtype=torch.uint8, device=cuda)
    cases = []
    for seed in (1, 2):
        keys = oracle.gen_uniform(n, seed=seed)
        offs = _random_offsets(rng, n, nseg)
        cases.append((keys, offs))
    keys, offs = cases[0]
    src.copy_(to_dev(keys, cuda)); ob.copy_(torch.from_numpy(offs[:-1].astype(np.int32))); oe.copy_(torch.from_numpy(offs[1:].astype(np.int32)))
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        a.copy_(src)
        S.SortKeys(temp, nb, dk, n, nseg, ob, oe, 0, 32, key_type=gs.GS_KEY_U32)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    dk.selector = 0
    with torch.cuda.graph(g, stream=side):
        a.copy_(src)
        S.SortKeys(temp, nb, dk, n, nseg, ob, oe, 0, 32, key_type=gs.GS_KEY_U32)
    out = dk.Current()
    for keys, offs in cases + cases[::-1]:                 # four replays: state left by one must not leak into the next
        src.copy_(to_dev(keys, cuda)); ob.copy_(torch.from_numpy(offs[:-1].astype(np.int32))); oe.copy_(torch.from_numpy(offs[1:].astype(np.int32)))
        g.replay()
        torch.cuda.synchronize()
        assert np.array_equal(to_u32(out)[:n], keys[_expected(oracle, keys, offs, 0, 32, False)])
