"""GPU parity tests for the LSB path, through the C ABI (libgpusort.so).

Model: lsb/cub/test/test_device_radix_sort.cu -- key generators RANDOM (entropy
reduction 0/3/6), UNIFORM(=2), INTEGER_SEED (:1052-1084); sizes n -> ceil(n/32)
... 1, 0 (:1034-1046); bit ranges full / [1,31) / the two middle bits
(:973-995); ascending + descending; keys-only and pairs; bit-exact compare
(:790-804).  Expected results come from the CPU oracle (oracle/) and from the
committed golden fixtures.
"""
import numpy as np
import pytest
import torch

from conftest import to_dev, to_u32

pytestmark = pytest.mark.gpu


def _sort(gs, keys_np, vals_np=None, begin_bit=0, end_bit=32, descending=False, key_type=None, dev="cuda:0"):
    n = keys_np.size
    d_keys = gs.DoubleBuffer(to_dev(keys_np, dev), torch.empty(n, dtype=torch.int32, device=dev))
    d_vals = None
    if vals_np is not None:
        d_vals = gs.DoubleBuffer(to_dev(vals_np, dev), torch.empty(n, dtype=torch.int32, device=dev))
    kt = gs.GS_KEY_U32 if key_type is None else key_type
    if d_vals is None:
        fn = gs.DeviceRadixSort.SortKeysDescending if descending else gs.DeviceRadixSort.SortKeys
        nbytes = fn(None, 0, d_keys, n)
        temp = torch.empty(nbytes, dtype=torch.uint8, device=dev)
        fn(temp, nbytes, d_keys, n, begin_bit, end_bit, key_type=kt)
        torch.cuda.synchronize()
        return to_u32(d_keys.Current())[:n], None, d_keys
    fn = gs.DeviceRadixSort.SortPairsDescending if descending else gs.DeviceRadixSort.SortPairs
    nbytes = fn(None, 0, d_keys, d_vals, n)
    temp = torch.empty(nbytes, dtype=torch.uint8, device=dev)
    fn(temp, nbytes, d_keys, d_vals, n, begin_bit, end_bit, key_type=kt)
    torch.cuda.synchronize()
    return to_u32(d_keys.Current())[:n], to_u32(d_vals.Current())[:n], d_keys


def test_extension_loaded_is_in_tree(gs, cuda):
    import os
    maps = open("/proc/self/maps").read()
    assert os.path.realpath(gs.LIB_PATH) in maps


def test_golden_fixtures(gs, cuda, golden):
    for tag in golden["cases"]:
        tag = str(tag)
        er, n, b, e, d = tag.split("_")
        er, n, bb, eb, desc = int(er[2:]), int(n[1:]), int(b[1:]), int(e), int(d[1:])
        keys = golden[f"keys_er{er}"][:n]
        want = golden["v_" + tag]
        ko, vo, _ = _sort(gs, keys, np.arange(n, dtype=np.uint32), bb, eb, bool(desc))
        assert np.array_equal(vo, want), tag
        assert np.array_equal(ko, keys[want]), tag
        ko, _, _ = _sort(gs, keys, None, bb, eb, bool(desc))
        if bb == 0 and eb == 32:
            assert np.array_equal(ko, keys[want]), tag


@pytest.mark.parametrize("shift,bits", [(0, 8), (8, 8), (24, 8), (29, 3)])
@pytest.mark.parametrize("n", [1, 8191, 8192, 8193, 100003, 3 * 1024 * 8192 // 2 + 5, 5 * 1024 * 8192 + 77])
def test_three_kernels_of_one_pass(gs, cuda, oracle, n, shift, bits):
    """upsweep counts, spine scan and downsweep scatter each match their oracle."""
    keys = oracle.gen_uniform(n, seed=shift + 1)
    vals = oracle.gen_enumerated(n)
    r = gs.lsb_pass_kernels(to_dev(keys, cuda), to_dev(vals, cuda), shift, bits)
    torch.cuda.synchronize()
    counts = oracle.upsweep(keys, shift, bits, r["tile"], r["tiles_per_chunk"], r["grid"])
    got = r["spine_counts"].reshape(256, r["grid"])
    assert np.array_equal(got[: 1 << bits].reshape(-1), counts)
    assert got[1 << bits:].sum() == 0                      # device spine always has 256 rows
    # prefix16[tile][d] = count of digit d in the chunk's earlier tiles
    tile, tpc = r["tile"], r["tiles_per_chunk"]
    num_tiles = (n + tile - 1) // tile
    dig = (keys >> np.uint32(shift)) & np.uint32((1 << bits) - 1)
    per_tile = np.stack([np.bincount(dig[t * tile:(t + 1) * tile], minlength=256) for t in range(num_tiles)])
    want16 = np.zeros_like(per_tile)
    for t in range(num_tiles):
        c0 = (t // tpc) * tpc
        want16[t] = per_tile[c0:t].sum(0)
    assert np.array_equal(r["prefix16"].reshape(num_tiles, 256).astype(np.int64), want16)
    # scan: row-exclusive + totals; flattened exclusive scan == oracle's
    tot = r["totals"]
    rows = r["spine_scanned"].reshape(256, r["grid"])
    flat = (rows + (np.cumsum(tot, dtype=np.uint64) - tot).astype(np.uint32)[:, None])[: 1 << bits].reshape(-1)
    assert np.array_equal(flat, oracle.exclusive_scan(counts))
    ko, vo = oracle.downsweep(keys, vals, shift, bits)
    assert np.array_equal(to_u32(r["keys_out"]), ko)
    assert np.array_equal(to_u32(r["vals_out"]), vo)


def _cub_sizes(nmax):
    out, n = [], nmax
    while n > 1:
        out.append(n)
        n = (n + 31) // 32
    return out + [1, 0]


@pytest.mark.parametrize("entropy_reduction", [0, 3, 6])
def test_cub_random_keys_size_sweep(gs, cuda, oracle, entropy_reduction):
    nmax = 1000003
    base = oracle.cub_random_keys(nmax, entropy_reduction)
    for n in _cub_sizes(nmax):
        keys = base[:n]
        for desc in (False, True):
            ko, _, _ = _sort(gs, keys, None, descending=desc)
            assert np.array_equal(ko, oracle.lsb_sort_keys(keys, descending=desc)), (n, desc)
        vals = oracle.gen_enumerated(n)
        ko, vo, _ = _sort(gs, keys, vals)
        ek, ev = oracle.lsb_sort_pairs(keys, vals)
        assert np.array_equal(ko, ek) and np.array_equal(vo, ev), n


@pytest.mark.parametrize("gen", ["uniform2", "integer_seed"])
def test_cub_uniform_and_natural_keys(gs, cuda, oracle, gen):
    n = 300007
    keys = np.full(n, 2, np.uint32) if gen == "uniform2" else np.arange(n, dtype=np.uint32)
    vals = oracle.gen_uniform(n, seed=5)
    for desc in (False, True):
        ko, vo, _ = _sort(gs, keys, vals, descending=desc)
        ek, ev = oracle.lsb_sort_pairs(keys, vals, descending=desc)
        assert np.array_equal(ko, ek) and np.array_equal(vo, ev)


@pytest.mark.parametrize("begin_bit,end_bit", [(0, 32), (1, 31), (15, 17), (0, 8), (3, 12), (24, 32), (7, 7)])
@pytest.mark.parametrize("desc", [False, True])
def test_bit_ranges(gs, cuda, oracle, begin_bit, end_bit, desc):
    n = 70001
    keys = oracle.cub_random_keys(n, 0)
    vals = oracle.gen_enumerated(n)
    ko, vo, dk = _sort(gs, keys, vals, begin_bit, end_bit, desc)
    ek, ev = oracle.lsb_sort_pairs(keys, vals, begin_bit, end_bit, desc)
    assert np.array_equal(ko, ek) and np.array_equal(vo, ev)
    # selector: one flip per 8-bit pass (dispatch_radix_sort.cuh:1158 rule)
    assert dk.selector == (((end_bit - begin_bit) + 7) // 8) % 2


def test_signed_and_float_keys(gs, cuda, oracle):
    n = 200003
    raw = oracle.gen_uniform(n, seed=11)
    # int32
    ko, _, _ = _sort(gs, raw, None, key_type=gs.GS_KEY_I32)
    assert np.array_equal(ko.view(np.int32), np.sort(raw.view(np.int32)))
    ko, _, _ = _sort(gs, raw, None, descending=True, key_type=gs.GS_KEY_I32)
    assert np.array_equal(ko.view(np.int32), np.sort(raw.view(np.int32))[::-1])
    # float32 (the LSB driver's real key type, lsb/sort.cu:111,130): no NaNs, as RandomBits guarantees
    f = raw.view(np.float32).copy()
    f[np.isnan(f)] = 1.0
    f[:5] = [-0.0, 0.0, -np.inf, np.inf, -1.5]
    ko, _, _ = _sort(gs, f.view(np.uint32), None, key_type=gs.GS_KEY_F32)
    tw = np.array(f.view(np.uint32))
    order = np.argsort(np.where(tw >> 31 == 1, ~tw, tw | 0x80000000), kind="stable")
    assert np.array_equal(ko, f.view(np.uint32)[order])
    vals = oracle.gen_enumerated(n)
    ko, vo, _ = _sort(gs, f.view(np.uint32), vals, descending=True, key_type=gs.GS_KEY_F32)
    order = np.argsort(~np.where(tw >> 31 == 1, ~tw, tw | 0x80000000), kind="stable")
    assert np.array_equal(vo, vals[order])


def test_driver_wrappers_like_lsb_sort_cu(gs, cuda, oracle):
    """sortPairsGPU ascending; sortKeysGPU sorts DESCENDING (lsb/sort.cu:65)."""
    n = 1 << 20
    keys = oracle.gen_uniform(n)
    vals = oracle.gen_uniform(n, seed=1)
    kb, ka = to_dev(keys, cuda), torch.empty(n, dtype=torch.int32, device=cuda)
    vb, va = to_dev(vals, cuda), torch.empty(n, dtype=torch.int32, device=cuda)
    ms, dk, dv = gs.sortPairsGPU(kb, ka, vb, va, n, key_type=gs.GS_KEY_U32)
    ek, ev = oracle.lsb_sort_pairs(keys, vals)
    assert ms > 0 and np.array_equal(to_u32(dk.Current()), ek) and np.array_equal(to_u32(dv.Current()), ev)
    kb.copy_(to_dev(keys, cuda))
    ms, dk = gs.sortKeysGPU(kb, ka, n, key_type=gs.GS_KEY_U32)
    assert np.array_equal(to_u32(dk.Current()), np.sort(keys)[::-1])


def test_workspace_too_small_is_an_error(gs, cuda):
    n = 10000
    d_keys = gs.DoubleBuffer(torch.zeros(n, dtype=torch.int32, device=cuda), torch.zeros(n, dtype=torch.int32, device=cuda))
    temp = torch.empty(16, dtype=torch.uint8, device=cuda)
    with pytest.raises(gs.GpuSortError):
        gs.DeviceRadixSort.SortKeys(temp, 16, d_keys, n)


def test_unaligned_input_pointer(gs, cuda, oracle):
    n = 50001
    keys = oracle.gen_uniform(n + 3, seed=4)
    buf = to_dev(keys, cuda)
    alt = torch.empty(n + 3, dtype=torch.int32, device=cuda)
    d_keys = gs.DoubleBuffer(buf[3:], alt[1:n + 1])
    nb = gs.DeviceRadixSort.SortKeys(None, 0, d_keys, n)
    temp = torch.empty(nb, dtype=torch.uint8, device=cuda)
    gs.DeviceRadixSort.SortKeys(temp, nb, d_keys, n, key_type=gs.GS_KEY_U32)
    torch.cuda.synchronize()
    assert np.array_equal(to_u32(d_keys.Current())[:n], np.sort(keys[3:]))


@pytest.mark.parametrize("with_values", [False, True])
def test_large_properties(gs, cuda, with_values):
    """2^27 keys: device-side sortedness + multiset checksum + enumerated-value check
    (size-independent properties; the full 2^30 case runs in bench.py --verify)."""
    n = 1 << 27
    keys = gs.generate_uniform_keys(n, seed=0, device=cuda)
    _, s0, x0 = gs.check_sorted(keys)
    orig = keys.clone()
    d_keys = gs.DoubleBuffer(keys, torch.empty_like(keys))
    if with_values:
        d_vals = gs.DoubleBuffer(gs.generate_enumerated_values(n, device=cuda), torch.empty_like(keys))
        nb = gs.DeviceRadixSort.SortPairs(None, 0, d_keys, d_vals, n)
        temp = torch.empty(nb, dtype=torch.uint8, device=cuda)
        gs.DeviceRadixSort.SortPairs(temp, nb, d_keys, d_vals, n, key_type=gs.GS_KEY_U32)
    else:
        nb = gs.DeviceRadixSort.SortKeys(None, 0, d_keys, n)
        temp = torch.empty(nb, dtype=torch.uint8, device=cuda)
        gs.DeviceRadixSort.SortKeys(temp, nb, d_keys, n, key_type=gs.GS_KEY_U32)
    inv, s1, x1 = gs.check_sorted(d_keys.Current())
    assert inv == 0 and (s1, x1) == (s0, x0)
    if with_values:
        bad, vsum = gs.check_pairs_enumerated(orig, d_keys.Current(), d_vals.Current())
        assert bad == 0 and vsum == n * (n - 1) // 2
        # stability: inside equal-key runs the enumerated values must ascend
        k = d_keys.Current()[: 1 << 22].cpu().numpy().view(np.uint32)
        v = d_vals.Current()[: 1 << 22].cpu().numpy().view(np.uint32)
        same = k[1:] == k[:-1]
        assert np.all(v[1:][same] > v[:-1][same])


def test_device_generators_match_oracle(gs, cuda, oracle):
    n = 100000
    cases = [
        (gs.generate_uniform_keys(n, seed=3, start=5, device=cuda), oracle.gen_uniform(n, 3, 5)),
        (gs.generate_zipf_keys(n, seed=3, start=5, device=cuda), oracle.gen_zipf(n, 3, 5)),
        (gs.generate_random_keys(n, seed=3, entropy_level=3, start=5, device=cuda), oracle.gen_entropy_and(n, 3, 3, 5)),
        (gs.generate_random_keys(n, seed=3, entropy_level=0, device=cuda), oracle.gen_entropy_and(n, 0, 3, 0)),
        (gs.generate_enumerated_values(n, start=5, device=cuda), oracle.gen_enumerated(n, 5)),
    ]
    for got, want in cases:
        assert np.array_equal(to_u32(got), want)
    k = oracle.gen_uniform(n, 7)
    inv, s, x = gs.check_sorted(to_dev(np.sort(k), cuda))
    assert inv == 0 and (s, x) == oracle.multiset_checksum(k)
    assert gs.check_sorted(to_dev(k, cuda))[0] == oracle.count_inversions_adjacent(k)


@pytest.mark.parametrize("begin_bit,end_bit", [(0, 32), (0, 24), (5, 13), (9, 9)])
@pytest.mark.parametrize("desc", [False, True])
def test_no_overwrite_mode(gs, cuda, oracle, begin_bit, end_bit, desc):
    """CUB_NO_OVERWRITE backend of test_device_radix_sort.cu (:798-804): input must be untouched."""
    n = 123457
    keys = oracle.cub_random_keys(n, 3)
    vals = oracle.gen_enumerated(n)
    kin, vin = to_dev(keys, cuda), to_dev(vals, cuda)
    kout, vout = torch.empty_like(kin), torch.empty_like(vin)
    nb = gs.DeviceRadixSort.SortPairsCopy(None, 0, kin, kout, vin, vout, n)
    temp = torch.empty(nb, dtype=torch.uint8, device=cuda)
    gs.DeviceRadixSort.SortPairsCopy(temp, nb, kin, kout, vin, vout, n, begin_bit, end_bit, key_type=gs.GS_KEY_U32,
                                     descending=desc)
    torch.cuda.synchronize()
    ek, ev = oracle.lsb_sort_pairs(keys, vals, begin_bit, end_bit, desc)
    assert np.array_equal(to_u32(kout), ek) and np.array_equal(to_u32(vout), ev)
    assert np.array_equal(to_u32(kin), keys) and np.array_equal(to_u32(vin), vals)
    kout2 = torch.empty_like(kin)
    nb2 = gs.DeviceRadixSort.SortKeysCopy(None, 0, kin, kout2, n)
    temp2 = torch.empty(nb2, dtype=torch.uint8, device=cuda)
    gs.DeviceRadixSort.SortKeysCopy(temp2, nb2, kin, kout2, n, begin_bit, end_bit, key_type=gs.GS_KEY_U32, descending=desc)
    torch.cuda.synchronize()
    assert np.array_equal(to_u32(kout2), oracle.lsb_sort_keys(keys, begin_bit, end_bit, desc))
    assert np.array_equal(to_u32(kin), keys)


def test_one_launch_pass_mode_matches(cuda, oracle, tmp_path):
    """GS_LSB_MODE=pipe (every pass after the first as ONE launch: upsweep / scanner / downsweep roles that hand their
    results over inside the launch; opt-in) is bit-exact too, and none of its bounded waits
    gave up.  The mode and the size threshold are read once per process, so it runs in a child process."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, %r)
import gpu_sort_amd as gs
from oracle import oracle as O
dev = torch.device("cuda:0")
for n in (17409, 65536, 100003, 8192 * 64 * 8 + 5, (1 << 22) + 77):
    keys = O.cub_random_keys(n, 3); vals = O.gen_enumerated(n)
    for desc, bb, eb in ((False, 0, 32), (True, 0, 32), (False, 3, 29), (True, 8, 24)):
        dk = gs.DoubleBuffer(torch.from_numpy(keys.view(np.int32).copy()).to(dev), torch.empty(n, dtype=torch.int32, device=dev))
        dv = gs.DoubleBuffer(torch.from_numpy(vals.view(np.int32).copy()).to(dev), torch.empty(n, dtype=torch.int32, device=dev))
        fn = gs.DeviceRadixSort.SortPairsDescending if desc else gs.DeviceRadixSort.SortPairs
        nb = fn(None, 0, dk, dv, n); temp = torch.zeros(nb, dtype=torch.uint8, device=dev)
        fn(temp, nb, dk, dv, n, bb, eb, key_type=gs.GS_KEY_U32); torch.cuda.synchronize()
        st = C.c_uint32(7)
        assert gs.lib.gs_lsb_pipe_status(temp.data_ptr(), n, C.byref(st), None) == 0 and st.value == 0, st.value
        ek, ev = O.lsb_sort_pairs(keys, vals, bb, eb, desc)
        assert np.array_equal(dk.Current().cpu().numpy().view(np.uint32), ek)
        assert np.array_equal(dv.Current().cpu().numpy().view(np.uint32), ev)
        kk = gs.DoubleBuffer(torch.from_numpy(keys.view(np.int32).copy()).to(dev), torch.empty(n, dtype=torch.int32, device=dev))
        fk = gs.DeviceRadixSort.SortKeysDescending if desc else gs.DeviceRadixSort.SortKeys
        fk(temp, nb, kk, n, bb, eb, key_type=gs.GS_KEY_U32); torch.cuda.synchronize()
        assert np.array_equal(kk.Current().cpu().numpy().view(np.uint32), O.lsb_sort_keys(keys, bb, eb, desc))
print("pipe ok")
''' % root
    env = dict(os.environ, GS_LSB_MODE="pipe", GS_LSB_PIPE_MIN_TILES="0")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "pipe ok" in out.stdout, out.stderr[-2000:]


def test_one_launch_pass_give_up_stores_nothing(cuda, tmp_path):
    """when a bounded wait of the opt-in one-launch pass gives up, nothing may be scattered from
    untagged offsets (stale words of another pass can exceed n: an out-of-bounds write).  The test hook
    GS_LSB_PIPE_TEST_DROP makes ONE upsweep workgroup publish nothing, so the scanner's batch and every tile behind it
    time out: the status word must be non-zero, the call itself still returns, and the guard zones around all four
    buffers (keys / values, both halves of the double buffer) must be untouched."""
    import os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    code = r'''
import sys, ctypes as C, numpy as np, torch
sys.path.insert(0, %r)
import gpu_sort_amd as gs
from oracle import oracle as O
dev = torch.device("cuda:0")
n, G = (1 << 21) + 77, 1 << 16
keys = O.gen_uniform(n, 5); vals = O.gen_enumerated(n)
bufs = [torch.full((n + 2 * G,), 0x5a5a5a5a, dtype=torch.int32, device=dev) for _ in range(4)]
bufs[0][G:G + n] = torch.from_numpy(keys.view(np.int32).copy()).to(dev)
bufs[2][G:G + n] = torch.from_numpy(vals.view(np.int32).copy()).to(dev)
dk = gs.DoubleBuffer(bufs[0][G:G + n], bufs[1][G:G + n])
dv = gs.DoubleBuffer(bufs[2][G:G + n], bufs[3][G:G + n])
nb = gs.DeviceRadixSort.SortPairs(None, 0, dk, dv, n); temp = torch.zeros(nb, dtype=torch.uint8, device=dev)
gs.DeviceRadixSort.SortPairs(temp, nb, dk, dv, n, key_type=gs.GS_KEY_U32); torch.cuda.synchronize()
st = C.c_uint32(0)
assert gs.lib.gs_lsb_pipe_status(temp.data_ptr(), n, C.byref(st), None) == 0
assert st.value != 0, "a dropped publish must surface in the status word"
for b in bufs:
    assert bool((b[:G] == 0x5a5a5a5a).all()) and bool((b[G + n:] == 0x5a5a5a5a).all()), "write outside [0, n)"
print("give-up ok", st.value)
''' % root
    env = dict(os.environ, GS_LSB_MODE="pipe", GS_LSB_PIPE_MIN_TILES="0", GS_LSB_PIPE_TEST_DROP="5")
    out = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "give-up ok" in out.stdout, out.stdout[-500:] + out.stderr[-2000:]


@pytest.mark.parametrize("algo", ["lsb", "lsb_pairs", "lsb_f32_desc", "msb", "msb_pairs"])
def test_beyond_2p30_keys(gs, cuda, algo):
    """n > 2^30: byte offsets no longer fit 32 bits, so the kernels' 64-bit addressing variants run
    (device-side sortedness, multiset checksum and enumerated-value checks)."""
    n = (1 << 30) + 3 * 8192 + 5
    pairs = algo.endswith("pairs")
    keys = gs.generate_uniform_keys(n, seed=3, device=cuda)
    _, s0, x0 = gs.check_sorted(keys)
    orig = keys.clone() if pairs else None
    alt = torch.empty_like(keys)
    vals = gs.generate_enumerated_values(n, device=cuda) if pairs else None
    valt = torch.empty_like(keys) if pairs else None
    if algo.startswith("msb"):
        seq = gs.rdxsrt_unstable_sort(keys, vals, n, alt, valt)
        out_k, out_v = seq.sorted_keys, seq.sorted_values
    else:
        dk = gs.DoubleBuffer(keys, alt)
        if pairs:
            dv = gs.DoubleBuffer(vals, valt)
            nb = gs.DeviceRadixSort.SortPairs(None, 0, dk, dv, n)
            temp = torch.empty(nb, dtype=torch.uint8, device=cuda)
            gs.DeviceRadixSort.SortPairs(temp, nb, dk, dv, n, key_type=gs.GS_KEY_U32)
            out_v = dv.Current()
        elif algo == "lsb_f32_desc":
            nb = gs.DeviceRadixSort.SortKeysDescending(None, 0, dk, n)
            temp = torch.empty(nb, dtype=torch.uint8, device=cuda)
            # random bit patterns as floats (NaN patterns included: the order is that of the twiddled bits)
            gs.DeviceRadixSort.SortKeysDescending(temp, nb, dk, n, key_type=gs.GS_KEY_F32)
            out = dk.Current()
            _, s1, x1 = gs.check_sorted(out)
            assert (s1, x1) == (s0, x0)
            # descending in float order == ascending after mapping back through the complement twiddle
            h = out[:: 4097][:200000].cpu().numpy().view(np.uint32)
            tw = np.where(h >> 31, ~h, h | np.uint32(0x80000000))
            assert np.all(tw[1:] <= tw[:-1])
            return
        else:
            nb = gs.DeviceRadixSort.SortKeys(None, 0, dk, n)
            temp = torch.empty(nb, dtype=torch.uint8, device=cuda)
            gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, key_type=gs.GS_KEY_U32)
        out_k = dk.Current()
    torch.cuda.synchronize()
    inv, s1, x1 = gs.check_sorted(out_k)
    assert inv == 0 and (s1, x1) == (s0, x0)
    if pairs:
        bad, vsum = gs.check_pairs_enumerated(orig, out_k, out_v)
        assert bad == 0 and vsum == n * (n - 1) // 2


@pytest.mark.parametrize("n", [5000, 100003])
def test_sort_is_capturable_in_a_hip_graph(gs, cuda, oracle, n):
    """A sort call makes no host-side decisions and no synchronisation, so it can be captured in a HIP
    graph once and replayed on new data in the same buffers (small path and three-kernel path)."""
    keys1, keys2 = oracle.gen_uniform(n, seed=1), oracle.gen_zipf(n, seed=2)
    src = to_dev(keys1, cuda)
    a, b = src.clone(), torch.empty_like(src)
    nb = gs.lib.gs_lsb_temp_bytes(n, 0)
    temp = torch.empty(nb, dtype=torch.uint8, device=cuda)
    dk = gs.DoubleBuffer(a, b)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, key_type=gs.GS_KEY_U32)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    dk.selector = 0
    with torch.cuda.graph(g, stream=side):
        a.copy_(src)
        gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, key_type=gs.GS_KEY_U32)
    out = dk.Current()
    g.replay(); torch.cuda.synchronize()
    assert np.array_equal(to_u32(out)[:n], np.sort(keys1))
    src.copy_(to_dev(keys2, cuda))
    g.replay(); torch.cuda.synchronize()
    assert np.array_equal(to_u32(out)[:n], np.sort(keys2))


def test_index_type_limit(cuda):
    """num_items up to 2^32 - 1 (the ABI's limit: one 32-bit index type, as the reference's IndexT): 2^32 - 24583
    uniform keys through LSB and MSB, and 2^32 - 1 equal keys (one bucket of almost 2^32 keys at every MSB level;
    tile counts must not wrap).  Device-side property checks; run in a child process so that its ~50 GiB of
    buffers are gone when it returns."""
    import os, subprocess, sys
    free, _ = torch.cuda.mem_get_info()
    if free < 80 * (1 << 30):
        pytest.skip("needs 80 GiB of free device memory")
    tool = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tools", "near_2p32.py")
    for args in (["lsb", "msb"], ["--max-const", "lsb", "msb", "msb_pairs"]):
        out = subprocess.run([sys.executable, tool] + args, capture_output=True, text=True, timeout=600)
        assert out.returncode == 0, out.stdout[-2000:] + out.stderr[-2000:]
        assert out.stdout.count("-> OK") == len([a for a in args if not a.startswith("--")])
