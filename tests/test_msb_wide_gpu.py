"""The wide element types of the MSB path (msb/tests/test_sort_keys.cu:154-195 UINT64 / DOUBLE keys,
test_sort_pairs.cu:223-281 the (u32|u64, u32|u64) pair combinations) through gs_msb_sort_wide, the hybrid MSB kernel set for
64-bit keys and values.  Keys: bit-exact against the oracle's 64-bit reference ranks (orc_lsb_reference_ranks_u64 --
ascending order of u64 / i64 / f64 keys is unique); values: the reference's unstable-pair rule (the multiset of values
inside every run of equal keys, test_sort_pairs.cu:80-109)."""
import numpy as np
import pytest
import torch

pytestmark = pytest.mark.gpu
GS_KEY_U64, GS_KEY_I64, GS_KEY_F64 = 3, 4, 5


def _keys(kind, n, rng):
    if kind == "u64":
        return rng.integers(0, 2**64, size=n, dtype=np.uint64), GS_KEY_U64
    if kind == "u64_low_entropy":       # the reference's AND-of-draws family: many duplicates, skewed top bytes
        a = rng.integers(0, 2**64, size=(4, n), dtype=np.uint64)
        return a[0] & a[1] & a[2] & a[3], GS_KEY_U64
    if kind == "u64_few":               # heavy duplicates: long equal-key runs through every level
        return rng.integers(0, 2**64, size=37, dtype=np.uint64)[rng.integers(0, 37, size=n)], GS_KEY_U64
    if kind == "i64":
        return rng.integers(-2**63, 2**63, size=n, dtype=np.int64).view(np.uint64), GS_KEY_I64
    if kind == "f64":
        x = rng.standard_normal(n) * np.exp(rng.uniform(-200, 200, size=n))
        x[:min(5, n)] = [0.0, -0.0, np.inf, -np.inf, 1.0][:min(5, n)]
        return x.view(np.uint64), GS_KEY_F64
    raise ValueError(kind)


def _check(oracle, keys_u64, key_type, out_k, vals=None, out_v=None):
    ranks = oracle.lsb_reference_ranks_u64(keys_u64, key_type)
    exp = keys_u64[ranks]
    assert np.array_equal(out_k, exp)
    if vals is not None:
        # values may be permuted inside runs of equal keys only
        ev = vals[ranks]
        start = np.flatnonzero(np.concatenate(([True], exp[1:] != exp[:-1])))
        run_id = np.repeat(np.arange(start.size), np.diff(np.concatenate((start, [exp.size]))))
        a = np.lexsort((ev, run_id))
        b = np.lexsort((out_v, run_id))
        assert np.array_equal(ev[a], out_v[b])


@pytest.mark.parametrize("n", [1, 2000, 2048, 8192, 8193, 100003, (1 << 21) + 77])
@pytest.mark.parametrize("kind", ["u64", "u64_low_entropy", "u64_few", "i64", "f64"])
def test_wide_keys(gs, oracle, cuda, kind, n):
    from gpu_sort_amd.msb import rdxsrt_unstable_sort_wide
    rng = np.random.default_rng(n + len(kind))
    keys, kt = _keys(kind, n, rng)
    dk = torch.from_numpy(keys.view(np.int64).copy()).to(cuda)
    alt = torch.empty_like(dk)
    seq, _ = rdxsrt_unstable_sort_wide(dk, None, n, alt, None, key_type=kt)
    _check(oracle, keys, kt, seq.sorted_keys.cpu().numpy().view(np.uint64))


@pytest.mark.parametrize("n", [3, 5000, 8192, 70001, (1 << 20) + 5])
@pytest.mark.parametrize("combo", ["u64_u32", "u64_u64", "u32_u64"])
@pytest.mark.parametrize("kind", ["u64", "u64_few"])
def test_wide_pairs(gs, oracle, cuda, combo, kind, n):
    from gpu_sort_amd.msb import rdxsrt_unstable_sort_wide
    rng = np.random.default_rng(n + 3)
    keys, kt = _keys(kind, n, rng)
    if combo == "u32_u64":
        keys32 = (keys >> np.uint64(32)).astype(np.uint32)
        vals = rng.integers(0, 2**63, size=n, dtype=np.int64)
        dk = torch.from_numpy(keys32.view(np.int32).copy()).to(cuda)
        dv = torch.from_numpy(vals.copy()).to(cuda)
        seq, _ = rdxsrt_unstable_sort_wide(dk, dv, n, torch.empty_like(dk), torch.empty_like(dv), key_type=gs.GS_KEY_U32)
        out_k = seq.sorted_keys.cpu().numpy().view(np.uint32).astype(np.uint64)
        _check(oracle, keys32.astype(np.uint64), GS_KEY_U64, out_k, vals, seq.sorted_values.cpu().numpy())
        return
    vdt = np.int32 if combo == "u64_u32" else np.int64
    vals = np.arange(n, dtype=vdt)
    dk = torch.from_numpy(keys.view(np.int64).copy()).to(cuda)
    dv = torch.from_numpy(vals.copy()).to(cuda)
    seq, _ = rdxsrt_unstable_sort_wide(dk, dv, n, torch.empty_like(dk), torch.empty_like(dv), key_type=kt)
    _check(oracle, keys, kt, seq.sorted_keys.cpu().numpy().view(np.uint64), vals, seq.sorted_values.cpu().numpy())


def test_wide_large_properties(gs, cuda):
    """2^27 u64 keys with enumerated u32 values: sorted on the device, keys a permutation (checksum), values a permutation."""
    from gpu_sort_amd.msb import rdxsrt_unstable_sort_wide
    n = 1 << 27
    g = torch.Generator(device=cuda)
    g.manual_seed(1)
    keys = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=cuda, generator=g)
    ksum = int(keys.sum().item())
    vals = torch.arange(n, dtype=torch.int32, device=cuda)
    orig = keys.clone()
    seq, _ = rdxsrt_unstable_sort_wide(keys, vals, n, torch.empty_like(keys), torch.empty_like(vals), key_type=GS_KEY_I64)
    sk, sv = seq.sorted_keys, seq.sorted_values
    assert bool((sk[1:] >= sk[:-1]).all()) and int(sk.sum().item()) == ksum
    assert bool((orig[sv.long()] == sk).all()) and int(sv.long().sum().item()) == n * (n - 1) // 2


@pytest.mark.parametrize("pairs", [False, True])
@pytest.mark.parametrize("kt", [GS_KEY_F64, GS_KEY_I64, GS_KEY_U64])
@pytest.mark.parametrize("n", [5000, 70000, 120000])
def test_wide_and_of_draws_levels(gs, oracle, cuda, n, kt, pairs):
    """The reference's entropy levels (AND of 1-10 draws, msb/tests: Entropy_UINT64 / Entropy_DOUBLE) as raw bit patterns.  As doubles
    they contain repeated -0.0: after the order-preserving transform a key whose sorted bits are all ones, i.e. indistinguishable
    from a pad inside a local sort -- the case that tells whether a task the order-free plan abandons (crowded bin) reaches the
    stable LSD passes in the order they need (round 3: it did not, and a pad was stored in place of a -0.0)."""
    from gpu_sort_amd.msb import rdxsrt_unstable_sort_wide
    rng = np.random.default_rng(n + kt)
    for level in (1, 3, 4, 5, 9, 11, 0):
        keys = rng.integers(0, 2**64, n, dtype=np.uint64)
        for _ in range(max(level - 1, 0)):
            keys &= rng.integers(0, 2**64, n, dtype=np.uint64)
        if level == 0:
            keys[:] = 0
        if kt == GS_KEY_F64:                      # NaN patterns have no place in the reference's order; keep the rest (incl. -0.0)
            nan = ((keys >> np.uint64(52)) & np.uint64(0x7ff)) == np.uint64(0x7ff)
            keys[nan] &= np.uint64(0x800fffffffffffff)
        dk = torch.from_numpy(keys.view(np.int64).copy()).to(cuda)
        if kt == GS_KEY_F64:
            dk = dk.view(torch.float64)
        vals = np.arange(n, dtype=np.int64) if pairs else None
        dv = torch.from_numpy(vals.copy()).to(cuda) if pairs else None
        seq, _ = rdxsrt_unstable_sort_wide(dk, dv, n, torch.empty_like(dk), torch.empty_like(dv) if pairs else None, key_type=kt)
        out_k = seq.sorted_keys.view(torch.int64).cpu().numpy().view(np.uint64)
        # the expected order from the transform itself (bit patterns, so that -0.0 and +0.0 stay distinct)
        if kt == GS_KEY_F64:
            tw = np.where((keys >> np.uint64(63)) != 0, ~keys, keys | np.uint64(1 << 63))
        elif kt == GS_KEY_I64:
            tw = keys ^ np.uint64(1 << 63)
        else:
            tw = keys
        order = np.argsort(tw, kind="stable")
        assert np.array_equal(out_k, keys[order]), f"level {level}"
        if pairs:
            out_v = seq.sorted_values.cpu().numpy()
            assert np.array_equal(keys[out_v], out_k) and np.array_equal(np.sort(out_v), vals), f"level {level}"
