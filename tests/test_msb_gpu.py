"""GPU parity tests for the MSB path (gs_msb_sort_u32 through the C ABI).

Model: msb/tests/test_sort_keys.cu and test_sort_pairs.cu -- 12 entropy levels
{1..11, 0} (:126), constant and swept problem sizes (:154-195), keys compared
bit-exact with a sorted reference (:56-59), values either secondary-sorted inside
equal-key runs (:80-109) or, for enumerated values, checked through the
value->key map and the sum (:141-146,166-176), because the sort is unstable.
"""
import numpy as np
import pytest
import torch

from conftest import to_dev, to_u32

pytestmark = pytest.mark.gpu

ENTROPY_LEVELS = [1, 2, 3, 4, 5, 6, 7, 8, 9, 10, 11, 0]


def _msb_keys(gs, keys_np, dev):
    n = keys_np.size
    dk, alt = to_dev(keys_np, dev), torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    seq = gs.rdxsrt_unstable_sort(dk, None, n, alt, None)
    assert seq.sorted_keys is dk, "32-bit keys: result must be in the caller's input array (gpu_radix_sort.h:359-360)"
    assert seq.sorted_values is None
    return to_u32(seq.sorted_keys)[:n]


def _msb_pairs(gs, keys_np, vals_np, dev):
    n = keys_np.size
    dk, dv = to_dev(keys_np, dev), to_dev(vals_np, dev)
    ka, va = torch.empty(max(n, 1), dtype=torch.int32, device=dev), torch.empty(max(n, 1), dtype=torch.int32, device=dev)
    seq = gs.rdxsrt_unstable_sort(dk, dv, n, ka, va)
    assert seq.sorted_keys is dk and seq.sorted_values is dv
    return to_u32(seq.sorted_keys)[:n], to_u32(seq.sorted_values)[:n]


@pytest.mark.parametrize("level", ENTROPY_LEVELS)
def test_sort_keys_entropy_uint(gs, cuda, oracle, level):
    n = 200000                                     # sort_keys_default_prob_size (msb/tests/main.cu:41)
    keys = oracle.gen_entropy_and(n, level, seed=0)
    got = _msb_keys(gs, keys, cuda)
    assert oracle.msb_check_keys(keys, got) == 0


@pytest.mark.parametrize("level", ENTROPY_LEVELS)
def test_sort_pairs_entropy_uint_uint(gs, cuda, oracle, level):
    n = 100000                                     # sort_pairs_default_prob_size
    keys = oracle.gen_entropy_and(n, level, seed=0)
    # (a) fast check: enumerated values
    vals = oracle.gen_enumerated(n)
    ks, vs = _msb_pairs(gs, keys, vals, cuda)
    assert oracle.msb_check_pairs_enumerated(keys, ks, vs) == 0
    # (b) full check: random values, secondary sort inside equal-key runs
    vals = oracle.gen_uniform(n, seed=99)
    ks, vs = _msb_pairs(gs, keys, vals, cuda)
    assert oracle.msb_check_pairs(keys, vals, ks, vs) == 0


def _numkeys_sweep(nmax):
    out, x = [], 100000.0
    while x < nmax:
        out.append(int(x))
        x *= 10 ** 0.1                              # test_sort_keys.cu:179
    return out + [nmax]


def test_sort_keys_numkeys_sweep(gs, cuda, oracle):
    for n in _numkeys_sweep(1500000):
        for level in (1, 4, 0):
            keys = oracle.gen_entropy_and(n, level, seed=0)
            assert np.array_equal(_msb_keys(gs, keys, cuda), np.sort(keys)), (n, level)


@pytest.mark.parametrize("n", [0, 1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4608, 4609, 6911, 6912, 6913,
                               9216, 9217, 16896, 16897, 17408, 17409, 65539, (1 << 20) + 7, (1 << 24) + 1])
def test_sizes_keys_and_pairs(gs, cuda, oracle, n):
    keys = oracle.gen_uniform(n, seed=n)
    assert np.array_equal(_msb_keys(gs, keys, cuda), np.sort(keys))
    vals = oracle.gen_enumerated(n)
    ks, vs = _msb_pairs(gs, keys, vals, cuda)
    assert oracle.msb_check_pairs_enumerated(keys, ks, vs) == 0


@pytest.mark.parametrize("kind", ["zipf", "few_values", "one_hot_bucket", "sorted", "reverse", "top_byte_const"])
def test_skewed_inputs(gs, cuda, oracle, kind):
    n = 3000017
    if kind == "zipf":
        keys = oracle.gen_zipf(n)
    elif kind == "few_values":
        keys = (oracle.gen_uniform(n, seed=3) % 5) * np.uint32(0x01010101)
    elif kind == "one_hot_bucket":               # 90 % identical keys, rest uniform
        keys = oracle.gen_uniform(n, seed=4)
        keys[oracle.gen_uniform(n, seed=5) % 10 != 0] = 0xDEADBEEF
    elif kind == "sorted":
        keys = np.sort(oracle.gen_uniform(n, seed=6))
    elif kind == "reverse":
        keys = np.sort(oracle.gen_uniform(n, seed=7))[::-1].copy()
    else:                                        # everything in one top-byte bucket
        keys = (oracle.gen_uniform(n, seed=8) & 0x00FFFFFF) | 0x5A000000
    assert np.array_equal(_msb_keys(gs, keys, cuda), np.sort(keys))
    vals = oracle.gen_enumerated(n)
    ks, vs = _msb_pairs(gs, keys, vals, cuda)
    assert oracle.msb_check_pairs_enumerated(keys, ks, vs) == 0


def test_signed_and_float_keys(gs, cuda, oracle):
    n = 500003
    raw = oracle.gen_uniform(n, seed=21)
    dk, alt = to_dev(raw, cuda), torch.empty(n, dtype=torch.int32, device=cuda)
    seq = gs.rdxsrt_unstable_sort(dk, None, n, alt, None, key_type=gs.GS_KEY_I32)
    assert np.array_equal(to_u32(seq.sorted_keys).view(np.int32), np.sort(raw.view(np.int32)))
    f = raw.view(np.float32).copy()
    f[np.isnan(f)] = 2.5
    dk = to_dev(f.view(np.uint32), cuda)
    seq = gs.rdxsrt_unstable_sort(dk, None, n, alt, None, key_type=gs.GS_KEY_F32)
    got = to_u32(seq.sorted_keys).view(np.float32)
    assert np.all(got[1:] >= got[:-1]) and np.array_equal(np.sort(got.view(np.uint32)), np.sort(f.view(np.uint32)))


@pytest.mark.parametrize("share", [0.55, 0.9, 0.999])
def test_heavy_hitter_path_signed_and_float_keys(gs, cuda, oracle, share):
    """The heavy-hitter path writes the dominant VALUE itself into the result (not a moved key): for signed and float keys
    that is the un-transformed value, also where the bucket already lies in the result buffer.  One value (negative)
    holds `share` of the keys at levels 1 and 2; the census must show the path was taken."""
    from gpu_sort_amd.msb import msb_census
    n = (1 << 21) + 4321
    rng = np.random.default_rng(int(share * 1000))
    noise = oracle.gen_uniform(n, seed=3)
    hot = rng.random(n) < share
    for kt, hotval, view in ((gs.GS_KEY_I32, np.int32(-123456789).view(np.uint32), np.int32),
                             (gs.GS_KEY_F32, np.float32(-3.75e-3).view(np.uint32), np.float32),
                             (gs.GS_KEY_U32, np.uint32(0xC0FFEE11), np.uint32)):
        raw = np.where(hot, hotval, noise).astype(np.uint32)
        if view is np.float32:
            f = raw.view(np.float32).copy()
            f[np.isnan(f)] = 1.5
            raw = f.view(np.uint32)
        dk, alt = to_dev(raw, cuda), torch.empty(n, dtype=torch.int32, device=cuda)
        dm = torch.empty(gs.lib.gs_msb_temp_bytes(n, 0), dtype=torch.uint8, device=cuda)
        seq = gs.rdxsrt_unstable_sort(dk, None, n, alt, None, pre_allocated_dm=dm, key_type=kt)
        got = to_u32(seq.sorted_keys).view(view)
        assert np.all(got[1:] >= got[:-1])
        assert np.array_equal(np.sort(got.view(np.uint32)), np.sort(raw))
        cen = msb_census(dm, n)
        assert sum(c["pivot_keys"] for c in cen) >= int(share * n * 0.99)


def test_host_pointer_wrappers(gs, cuda, oracle):
    """rdxsrt_unstable_sort_keys / _pairs (gpu_radix_sort.h:511-587): host arrays in and out."""
    n = 123457
    keys = oracle.gen_entropy_and(n, 2, seed=1)
    assert np.array_equal(gs.rdxsrt_unstable_sort_keys(keys), np.sort(keys))
    vals = oracle.gen_enumerated(n)
    ks, vs = gs.rdxsrt_unstable_sort_pairs(keys, vals)
    assert oracle.msb_check_pairs_enumerated(keys, ks, vs) == 0


def test_workspace_reuse_and_too_small(gs, cuda, oracle):
    n = 400000
    need = gs.lib.gs_msb_temp_bytes(n, 0)
    dm = torch.empty(need, dtype=torch.uint8, device=cuda)
    for seed in (1, 2, 3):                          # pre-allocated data manager reused across calls
        keys = oracle.gen_uniform(n, seed=seed)
        dk, alt = to_dev(keys, cuda), torch.empty(n, dtype=torch.int32, device=cuda)
        seq = gs.rdxsrt_unstable_sort(dk, None, n, alt, None, pre_allocated_dm=dm)
        assert np.array_equal(to_u32(seq.sorted_keys), np.sort(keys))
    with pytest.raises(gs.GpuSortError):
        gs.rdxsrt_unstable_sort(dk, None, n, alt, None, pre_allocated_dm=dm[:1024])


@pytest.mark.parametrize("dist", ["uniform", "zipf"])
def test_large_properties(gs, cuda, dist):
    """2^27 keys (+ enumerated values): device-side sortedness, multiset checksum, value->key map."""
    n = 1 << 27
    gen = gs.generate_uniform_keys if dist == "uniform" else gs.generate_zipf_keys
    keys = gen(n, seed=0, device=cuda)
    orig = keys.clone()
    _, s0, x0 = gs.check_sorted(keys)
    vals = gs.generate_enumerated_values(n, device=cuda)
    seq = gs.rdxsrt_unstable_sort(keys, vals, n, torch.empty_like(keys), torch.empty_like(keys))
    inv, s1, x1 = gs.check_sorted(seq.sorted_keys)
    assert inv == 0 and (s1, x1) == (s0, x0)
    bad, vsum = gs.check_pairs_enumerated(orig, seq.sorted_keys, seq.sorted_values)
    assert bad == 0 and vsum == n * (n - 1) // 2


def test_config4_zipf_2p30_through_rdxsrt_unstable_sort(gs, cuda):
    """BASELINE configs[3] at full size: 2^30 Zipf keys through the MSB path (keys only, so the heavy-hitter path is
    on): sorted, same multiset, result in the caller's input array, and the census shows the heavy hitters being taken
    out at level 2 instead of travelling to the last byte."""
    from gpu_sort_amd.msb import msb_census
    n = 1 << 30
    keys = gs.generate_zipf_keys(n, seed=0, device=cuda)
    _, s0, x0 = gs.check_sorted(keys)
    alt = torch.empty_like(keys)
    dm = torch.empty(gs.lib.gs_msb_temp_bytes(n, 0), dtype=torch.uint8, device=cuda)
    seq = gs.rdxsrt_unstable_sort(keys, None, n, alt, None, pre_allocated_dm=dm)
    assert seq.sorted_keys.data_ptr() == keys.data_ptr()
    inv, s1, x1 = gs.check_sorted(seq.sorted_keys)
    assert inv == 0 and (s1, x1) == (s0, x0)
    cen = msb_census(dm, n)
    assert cen[0]["keys"] == n and cen[1]["keys"] == n            # every top-byte bucket is far above the local-sort capacity
    assert cen[2]["pivot_keys"] > 0.45 * n                         # the ~4000 heavy values: half of the keys
    assert cen[3]["keys"] < 0.15 * n                               # (51 % without the heavy-hitter path)


@pytest.mark.parametrize("n", [1500, 4000, 9000, 17000, 17408, 40000, 700001])
def test_all_ones_keys_next_to_padding(gs, cuda, oracle, n):
    """Keys whose low bits are all ones tie with the local sort's padding value in every digit:
    none may be displaced by a pad (regression: shared first-pass histogram), for any range size."""
    keys = oracle.gen_uniform(n, seed=n)
    keys[::3] |= np.uint32(0x0000FFFF)
    keys[1::7] = np.uint32(0xFFFFFFFF)
    keys[5::11] |= np.uint32(0x00FFFFFF)
    assert np.array_equal(_msb_keys(gs, keys, cuda), np.sort(keys))
    vals = oracle.gen_enumerated(n)
    ks, vs = _msb_pairs(gs, keys, vals, cuda)
    assert oracle.msb_check_pairs_enumerated(keys, ks, vs) == 0


@pytest.mark.parametrize("n,pairs", [(3000, False), (300007, False), (300007, True)])
def test_msb_sort_is_capturable_in_a_hip_graph(gs, cuda, oracle, n, pairs):
    """With synchronize=0 the MSB sort takes every decision on the device (bucket lists, task lists, grids
    bounded on the host) -- the reference needs >= 3 blocking D2H round trips per pass (gpu_radix_sort.h:
    426, generate_next_pass_block_assignments) -- so a call can be captured once and replayed on new data."""
    keys1, keys2 = oracle.gen_uniform(n, seed=3), oracle.gen_zipf(n, seed=4)
    src = to_dev(keys1, cuda)
    a, b = src.clone(), torch.empty_like(src)
    vsrc = to_dev(oracle.gen_enumerated(n), cuda) if pairs else None
    va, vb = (vsrc.clone(), torch.empty_like(vsrc)) if pairs else (None, None)
    temp = torch.empty(gs.lib.gs_msb_temp_bytes(n, int(pairs)), dtype=torch.uint8, device=cuda)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        gs.rdxsrt_unstable_sort(a, va, n, b, vb, pre_allocated_dm=temp, synchronize=False)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g, stream=side):
        a.copy_(src)
        if pairs:
            va.copy_(vsrc)
        seq = gs.rdxsrt_unstable_sort(a, va, n, b, vb, pre_allocated_dm=temp, synchronize=False)
    for keys in (keys1, keys2, keys2, keys1):
        src.copy_(to_dev(keys, cuda))
        g.replay()
        torch.cuda.synchronize()
        got = to_u32(seq.sorted_keys)[:n]
        assert np.array_equal(got, np.sort(keys))
        if pairs:
            assert oracle.msb_check_pairs_enumerated(keys, got, to_u32(seq.sorted_values)[:n]) == 0


def test_look_at_the_next_level_does_not_change_results():
    """Outside a graph capture the MSB sorts look at the next level's size (a pinned host word written after each level's
    classification) and skip empty levels / launch exact grids; GS_MSB_PEEK=0 keeps the worst-case grids.  The switch is read
    once per process, so both settings run in child processes: same sorted keys (checksums) for sizes that end after level 0,
    1, 2 and 3 and for uniform and Zipf keys, and the census of the last sort is the same."""
    import json, os, subprocess, sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    script = r"""
import sys, json, torch
sys.path.insert(0, %r)
import gpu_sort_amd as gs
from gpu_sort_amd.msb import msb_census
out = []
dev = torch.device("cuda:0")
for n, gen in ((30000, gs.generate_uniform_keys), (1 << 20, gs.generate_uniform_keys), ((1 << 24) + 77, gs.generate_uniform_keys),
               ((1 << 24) + 5, gs.generate_zipf_keys), ((1 << 22) + 1, lambda n, device: gs.generate_random_keys(n, entropy_level=5, device=device))):
    a = gen(n, device=dev); b = torch.empty_like(a)
    ref = torch.sort(a.to(torch.int64) & 0xffffffff).values
    dm = torch.empty(gs.lib.gs_msb_temp_bytes(n, 0), dtype=torch.uint8, device=dev)
    seq = gs.rdxsrt_unstable_sort(a, None, n, b, None, pre_allocated_dm=dm)
    got = seq.sorted_keys.to(torch.int64) & 0xffffffff
    cen = msb_census(dm, n)
    out.append([bool(torch.equal(got, ref)), [c["keys"] for c in cen], [c["task_keys"] for c in cen]])
print(json.dumps(out))
""" % root
    res = []
    for peek in ("1", "0"):
        env = dict(os.environ, GS_MSB_PEEK=peek)
        r = subprocess.run([sys.executable, "-c", script], capture_output=True, text=True, timeout=300, env=env, check=True)
        res.append(json.loads(r.stdout.strip().splitlines()[-1]))
    assert all(case[0] for case in res[0]) and all(case[0] for case in res[1])
    assert res[0] == res[1]


def test_list_overflow_is_reported(gs, cuda, oracle, monkeypatch):
    """VERDICT r02 missing #5 / item 6: a device-side list that overflows drops a record -- the sort's result is then
    wrong, and it must say so.  The sizing never lets that happen, so the test shrinks the task-list bound through the
    test hook GS_MSB_TEST_MAX_TASKS (read on every call): a synchronous gs_msb_sort_u32 returns an error instead of
    hipSuccess, an asynchronous one leaves the word for gs_msb_census, and without the hook the same sort is clean."""
    from gpu_sort_amd.msb import msb_census
    n = 1 << 21                                       # 256 top-byte buckets of 8192 keys: 256 local-sort tasks at level 0
    keys = oracle.gen_uniform(n, seed=3)
    nbytes = gs.lib.gs_msb_temp_bytes(n, 0)
    dm = torch.empty(nbytes, dtype=torch.uint8, device=cuda)
    alt = torch.empty(n, dtype=torch.int32, device=cuda)
    # clean run first
    dk = to_dev(keys, cuda)
    gs.rdxsrt_unstable_sort(dk, None, n, alt, None, pre_allocated_dm=dm)
    assert all(c["overflow"] == 0 for c in msb_census(dm, n))
    assert oracle.msb_check_keys(keys, to_u32(dk)[:n]) == 0
    monkeypatch.setenv("GS_MSB_TEST_MAX_TASKS", "10")
    dk = to_dev(keys, cuda)
    with pytest.raises(gs.GpuSortError):
        gs.rdxsrt_unstable_sort(dk, None, n, alt, None, pre_allocated_dm=dm)            # synchronous: reports
    assert all(c["overflow"] != 0 for c in msb_census(dm, n))
    dk = to_dev(keys, cuda)
    gs.rdxsrt_unstable_sort(dk, None, n, alt, None, pre_allocated_dm=dm, synchronize=False)   # asynchronous: the census carries it
    torch.cuda.synchronize()
    assert msb_census(dm, n)[0]["overflow"] != 0
    monkeypatch.delenv("GS_MSB_TEST_MAX_TASKS")
    dk = to_dev(keys, cuda)
    gs.rdxsrt_unstable_sort(dk, None, n, alt, None, pre_allocated_dm=dm)
    assert msb_census(dm, n)[0]["overflow"] == 0 and oracle.msb_check_keys(keys, to_u32(dk)[:n]) == 0


def _few_distinct_case(rng, sizes, distincts, big=150_000):
    """Keys whose level-1 sub-buckets (two top bytes fixed) hold sizes[i] keys of distincts[i] distinct 16-bit tails, plus one
    sub-bucket too large for a local sort (so that the level shows skew and the samples are looked at)."""
    parts = []
    for j, (m, d) in enumerate(zip(sizes, distincts)):
        prefix = ((0x10 + (j & 1) * 0x31) << 24) | ((j >> 1) << 16)
        tails = rng.choice(65536, size=min(d, 65536), replace=False).astype(np.uint32)
        idx = np.minimum((rng.random(m) ** 2 * len(tails)).astype(np.int64), len(tails) - 1)     # skewed multiplicities
        idx[: len(tails)] = np.arange(len(tails))                                              # every tail at least once
        parts.append(np.uint32(prefix) | tails[idx[:m]] if m >= len(tails) else np.uint32(prefix) | tails[:m])
    parts.append(np.uint32(0x7f << 24 | 0x33 << 16) | rng.integers(0, 65536, big, dtype=np.uint32))
    keys = np.concatenate(parts)
    rng.shuffle(keys)
    return keys


@pytest.mark.parametrize("pairs", [False, True])
def test_few_distinct_values_plan(gs, cuda, oracle, pairs):
    """The local sorts' plan for tasks with few distinct values (LS_DEDUPE / LS_DEDUPE_ALL, gs_msb.hip): 16-bit tasks of the two large
    classes with 1 ... 2047, 2048, 2049 ... all-distinct values -- around DD_MAX = 2048 the plan must hand the task back untouched."""
    rng = np.random.default_rng(77)
    distincts = [1, 2, 3, 17, 100, 256, 257, 1000, 2047, 2048, 2049, 2100, 3000, 4700, 9000, 16, 255, 2048, 2047, 2049]
    sizes = [9000, 9216, 4700, 8000, 9100, 9216, 7000, 9000, 9216, 9216, 9216, 9000, 9000, 9216, 9216,     # the 9216 class
             17408, 17000, 17408, 12000, 17408]                                                              # the 17408 class
    keys = _few_distinct_case(rng, sizes, distincts)
    if not pairs:
        got = _msb_keys(gs, keys, cuda)
        assert oracle.msb_check_keys(keys, got) == 0
    else:
        vals = oracle.gen_uniform(keys.size, seed=5)
        ks, vs = _msb_pairs(gs, keys, vals, cuda)
        assert oracle.msb_check_pairs(keys, vals, ks, vs) == 0
        vals = oracle.gen_enumerated(keys.size)
        ks, vs = _msb_pairs(gs, keys, vals, cuda)
        assert oracle.msb_check_pairs_enumerated(keys, ks, vs) == 0


def test_few_distinct_values_plan_is_taken(gs, cuda, oracle, monkeypatch):
    """The same input with the plan switched off (GS_MSB_DEDUPE=0, read once per process: a child process) sorts to the same keys."""
    import subprocess, sys, os, textwrap
    code = textwrap.dedent("""
        import sys, numpy as np, torch
        sys.path.insert(0, %r); sys.path.insert(0, %r)
        import gpu_sort_amd as gs
        from test_msb_gpu import _few_distinct_case
        rng = np.random.default_rng(3)
        keys = _few_distinct_case(rng, [9216] * 40 + [17408] * 20, [int(x) for x in rng.integers(1, 2300, 60)])
        d = torch.from_numpy(keys.view(np.int32)).cuda(); alt = torch.empty_like(d)
        gs.rdxsrt_unstable_sort(d, None, keys.size, alt, None)
        got = d.cpu().numpy().view(np.uint32)
        assert (got == np.sort(keys)).all()
        print("ok")
    """) % (os.path.dirname(os.path.dirname(os.path.abspath(__file__))), os.path.dirname(os.path.abspath(__file__)))
    for flag in ("0", "1"):
        env = dict(os.environ, GS_MSB_DEDUPE=flag)
        r = subprocess.run([sys.executable, "-c", code], env=env, capture_output=True, text=True, timeout=300)
        assert r.returncode == 0 and "ok" in r.stdout, r.stderr[-2000:]


@pytest.mark.parametrize("share", [0.55, 0.9, 0.999])
@pytest.mark.parametrize("level", [1, 2])
def test_heavy_hitter_path_pairs(gs, cuda, oracle, share, level):
    """The heavy-hitter path for (key, value) pairs (round 3): the values of the dominant key move to the middle of its bucket -- at
    level 1 straight into the result buffer, at level 2 through the level's destination buffer and msb_pivot_copyback_kernel.
    level 2: two values with the same top byte share the keys, so no level-1 bucket is dominated but two level-2 buckets are."""
    from gpu_sort_amd.msb import msb_census
    n = (1 << 21) + 4321
    rng = np.random.default_rng(int(share * 1000) + level)
    noise = oracle.gen_uniform(n, seed=3)
    hot = rng.random(n) < share
    for kt, a, b, view in ((gs.GS_KEY_U32, np.uint32(0xC0FF1234), np.uint32(0xC0EE5678), np.uint32),
                           (gs.GS_KEY_I32, np.int32(-123456789).view(np.uint32), np.int32(-123456789 + (5 << 16)).view(np.uint32), np.int32)):
        second = rng.random(n) < 0.5
        hotvals = np.where(second, b, a) if level == 2 else np.full(n, a, dtype=np.uint32)
        raw = np.where(hot, hotvals, noise).astype(np.uint32)
        for vals in (oracle.gen_uniform(n, seed=11), oracle.gen_enumerated(n)):
            dk, dv = to_dev(raw, cuda), to_dev(vals, cuda)
            ka, va = torch.empty(n, dtype=torch.int32, device=cuda), torch.empty(n, dtype=torch.int32, device=cuda)
            dm = torch.empty(gs.lib.gs_msb_temp_bytes(n, 1), dtype=torch.uint8, device=cuda)
            seq = gs.rdxsrt_unstable_sort(dk, dv, n, ka, va, pre_allocated_dm=dm, key_type=kt)
            ks, vs = to_u32(seq.sorted_keys)[:n], to_u32(seq.sorted_values)[:n]
            got = ks.view(view)
            assert np.all(got[1:] >= got[:-1])
            # the pairs are a permutation of the input pairs: same multiset of (key, value)
            pin = np.sort(raw.astype(np.uint64) << np.uint64(32) | vals.astype(np.uint64))
            pout = np.sort(ks.astype(np.uint64) << np.uint64(32) | vs.astype(np.uint64))
            assert np.array_equal(pin, pout)
            cen = msb_census(dm, n, True)
            assert sum(c["pivot_keys"] for c in cen) >= int(share * n * 0.99)
            assert cen[level]["pivot_keys"] >= int(share * n * 0.99)


@pytest.mark.parametrize("pairs", [False, True])
def test_large_buckets_at_odd_offsets(gs, cuda, oracle, pairs):
    """Buckets of 256 tiles or more whose offset is no multiple of 64 elements get a short first tile so that their other tiles start
    on 256-byte boundaries (ws_first_tile, gs_msb.hip): three top bytes with 3 000 001 / 2 500 003 / 2 600 005 keys, so that the second
    and third level-1 bucket start at odd offsets and carry two ragged tiles each."""
    rng = np.random.default_rng(12)
    sizes = [3_000_001, 2_500_003, 2_600_005]
    tops = [0x11, 0x5A, 0xC3]
    keys = np.concatenate([(np.uint32(t) << np.uint32(24)) | (rng.integers(0, 1 << 24, m, dtype=np.uint32)) for t, m in zip(tops, sizes)])
    rng.shuffle(keys)
    n = keys.size
    if not pairs:
        assert np.array_equal(_msb_keys(gs, keys, cuda), np.sort(keys))
    else:
        vals = oracle.gen_enumerated(n)
        ks, vs = _msb_pairs(gs, keys, vals, cuda)
        assert oracle.msb_check_pairs_enumerated(keys, ks, vs) == 0
