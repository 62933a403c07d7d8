"""CPU tests: the oracle against the reference's MT19937, the committed golden
fixtures and independent numpy sorts (the parity pin, DESIGN.md "Oracle")."""
import ctypes as C
import os

import numpy as np
import pytest


def test_mt19937_known_answer(oracle):
    # canonical MT19937 init_by_array{0x123,0x234,0x345,0x456} stream
    oracle.lib().orc_mt_init_cub_default()
    got = [oracle.lib().orc_mt_genrand_int32() for _ in range(4)]
    assert got == [1067595299, 955945823, 477289528, 4107218783]


def test_mt19937_matches_reference_build(oracle):
    R = oracle.ref_mersenne()
    if R is None:
        pytest.skip("oracle/_ref not built (reference tree absent)")
    init = (C.c_uint32 * 4)(0x123, 0x234, 0x345, 0x456)
    R.ref_mt_init_by_array(init, 4)
    oracle.lib().orc_mt_init_cub_default()
    for _ in range(5000):
        assert R.ref_mt_genrand_int32() == oracle.lib().orc_mt_genrand_int32()
    R.ref_mt_init_genrand(12345)
    oracle.lib().orc_mt_init_genrand(12345)
    for _ in range(2000):
        assert R.ref_mt_genrand_int32() == oracle.lib().orc_mt_genrand_int32()


def test_random_bits_matches_golden_inputs(oracle, golden):
    # golden inputs were produced from the reference-built MT19937 + numpy AND
    for er in (0, 3):
        want = golden[f"keys_er{er}"]
        got = oracle.cub_random_keys(want.size, entropy_reduction=er)
        assert np.array_equal(got, want)
    assert list(golden["mt_first8"][:4]) == [1067595299, 955945823, 477289528, 4107218783]


def test_random_bits_bit_range(oracle):
    k = oracle.cub_random_keys(1000, 0, 4, 20)
    assert (k & ~np.uint32(((1 << 16) - 1) << 4)).max() == 0
    assert oracle.cub_random_keys(10, -1).max() == 0


def _parse(tag):
    er, n, b, e, d = tag.split("_")
    return int(er[2:]), int(n[1:]), int(b[1:]), int(e), int(d[1:])


def test_oracle_matches_golden(oracle, golden):
    for tag in golden["cases"]:
        er, n, bb, eb, desc = _parse(str(tag))
        keys = golden[f"keys_er{er}"][:n]
        want = golden["v_" + str(tag)]
        ranks = oracle.lsb_reference_ranks(keys, bb, eb, bool(desc))
        assert np.array_equal(ranks, want), tag
        ko, vo = oracle.lsb_sort_pairs(keys, np.arange(n, dtype=np.uint32), bb, eb, bool(desc))
        assert np.array_equal(vo, want) and np.array_equal(ko, keys[want]), tag
        # the 8-bit LSD restatement (north_star formulation) gives the same answer
        k2, v2 = oracle.lsd_radix_sort(keys, np.arange(n, dtype=np.uint32), bb, eb, bool(desc))
        assert np.array_equal(v2, want) and np.array_equal(k2, keys[want]), tag


@pytest.mark.parametrize("n", [0, 1, 2, 1000, 8192, 100003])
def test_oracle_vs_numpy(oracle, n):
    rng = np.random.default_rng(n)
    keys = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    if n > 10:
        keys[: n // 3] &= 0xFF  # plenty of duplicates
    vals = rng.integers(0, 1 << 32, n, dtype=np.uint64).astype(np.uint32)
    assert np.array_equal(oracle.lsb_sort_keys(keys), np.sort(keys))
    order = np.argsort(keys, kind="stable")
    ko, vo = oracle.lsb_sort_pairs(keys, vals)
    assert np.array_equal(ko, keys[order]) and np.array_equal(vo, vals[order])
    order = np.argsort(~keys, kind="stable")
    ko, vo = oracle.lsb_sort_pairs(keys, vals, descending=True)
    assert np.array_equal(ko, keys[order]) and np.array_equal(vo, vals[order])


def test_per_kernel_goldens_compose(oracle):
    # upsweep -> scan -> (block, digit) bases reproduce the stable pass
    n, tile, tpc, grid = 50000, 8192, 2, 4
    keys = oracle.gen_uniform(n, seed=3)
    counts = oracle.upsweep(keys, 8, 8, tile, tpc, grid)
    assert counts.sum() == n
    assert np.array_equal(counts.reshape(256, grid).sum(1), np.bincount((keys >> 8) & 0xFF, minlength=256))
    scanned = oracle.exclusive_scan(counts)
    assert scanned[0] == 0 and scanned[-1] + counts[-1] == n
    ko, _ = oracle.downsweep(keys, None, 8, 8)
    assert np.array_equal(ko, keys[np.argsort((keys >> 8) & 0xFF, kind="stable")])
    # block b's first key of digit d lands at scanned[d*grid+b]
    nt = (n + tile - 1) // tile
    for b in range(grid):
        lo, hi = oracle.chunk_tiles(nt, tpc, b)
        seg = keys[lo * tile: min(hi * tile, n)]
        for d in (0, 17, 255):
            idx = np.nonzero(((seg >> 8) & 0xFF) == d)[0]
            if idx.size:
                assert ko[scanned[d * grid + b]] == seg[idx[0]]


def test_generators(oracle):
    u = oracle.gen_uniform(1 << 16)
    assert np.array_equal(u[100:200], oracle.gen_uniform(100, start=100))  # counter-based
    assert abs(np.unpackbits(u.view(np.uint8)).mean() - 0.5) < 0.01
    z = oracle.gen_zipf(1 << 16)
    assert np.unique(z).size < z.size * 0.7  # heavy duplicates
    assert oracle.gen_entropy_and(100, 0).max() == 0
    e3 = oracle.gen_entropy_and(1 << 16, 3)
    assert abs(np.unpackbits(e3.view(np.uint8)).mean() - 0.125) < 0.01
    e1 = oracle.gen_entropy_and(1 << 10, 1)
    assert np.array_equal(e1, oracle.gen_uniform(1 << 10))
    assert np.array_equal(oracle.gen_enumerated(5, 3), np.arange(3, 8, dtype=np.uint32))


def test_twiddles(oracle):
    f = np.array([-np.inf, -1.5, -0.0, 0.0, 1e-30, 2.5, np.inf], dtype=np.float32)
    tw = np.array([oracle.lib().orc_twiddle_in_f32(int(x)) for x in f.view(np.uint32)], dtype=np.uint32)
    assert np.all(np.diff(tw.astype(np.int64)) > 0)
    back = np.array([oracle.lib().orc_twiddle_out_f32(int(x)) for x in tw], dtype=np.uint32)
    assert np.array_equal(back, f.view(np.uint32))
    i = np.array([-2**31, -5, 0, 7, 2**31 - 1], dtype=np.int32)
    tw = np.array([oracle.lib().orc_twiddle_in_i32(int(x)) for x in i.view(np.uint32)], dtype=np.uint32)
    assert np.all(np.diff(tw.astype(np.int64)) > 0)


def test_msb_checkers(oracle):
    n = 5000
    keys = oracle.gen_entropy_and(n, 4, seed=1)      # many duplicate keys
    vals = oracle.gen_enumerated(n)
    order = np.argsort(keys, kind="stable")
    ks, vs = keys[order], vals[order]
    assert oracle.msb_check_keys(keys, ks) == 0
    assert oracle.msb_check_pairs_enumerated(keys, ks, vs) == 0
    assert oracle.msb_check_pairs(keys, vals, ks, vs) == 0
    # an unstable but valid result: reverse values inside every equal-key run
    vs2 = vs.copy()
    start = 0
    for i in range(1, n + 1):
        if i == n or ks[i] != ks[start]:
            vs2[start:i] = vs2[start:i][::-1]
            start = i
    assert not np.array_equal(vs, vs2)
    assert oracle.msb_check_pairs(keys, vals, ks, vs2) == 0
    assert oracle.msb_check_pairs_enumerated(keys, ks, vs2) == 0
    # broken results are caught
    bad = ks.copy(); bad[[10, 11]] = bad[[11, 10]] if bad[10] != bad[11] else (bad[10] + 1, bad[11])
    assert oracle.msb_check_keys(keys, bad) != 0
    vbad = vs.copy(); vbad[0] = vbad[-1]
    assert oracle.msb_check_pairs_enumerated(keys, ks, vbad) != 0
    assert oracle.msb_check_pairs(keys, vals, ks, vbad) != 0


def test_properties(oracle):
    k = oracle.gen_uniform(10000, seed=9)
    s = np.sort(k)
    assert oracle.multiset_checksum(k) == oracle.multiset_checksum(s)
    assert oracle.count_inversions_adjacent(s) == 0
    assert oracle.count_inversions_adjacent(s[::-1].copy(), descending=True) == 0
    assert oracle.count_inversions_adjacent(k) > 0
    k2 = k.copy(); k2[5] ^= 1
    assert oracle.multiset_checksum(k2) != oracle.multiset_checksum(k)


def test_cpu_baseline_sorts(oracle):
    k = oracle.gen_uniform(1 << 16, seed=2)
    dt, s = oracle.time_std_sort(k)
    assert dt > 0 and np.array_equal(s, np.sort(k))
    dt, used, s = oracle.time_std_sort_mt(k, 4)
    assert used == 4 and np.array_equal(s, np.sort(k))
    v = oracle.gen_enumerated(k.size)
    dt, ks, vs = oracle.time_std_stable_sort_pairs(k & 0xFF, v)
    order = np.argsort(k & 0xFF, kind="stable")
    assert np.array_equal(vs, v[order])


@pytest.mark.parametrize("kt,view", [(3, np.uint64), (4, np.int64), (5, np.float64)])
def test_reference_ranks_64bit_match_numpy_stable_sort(oracle, kt, view):
    rng = np.random.default_rng(kt)
    k = rng.integers(0, 2**64, size=20011, dtype=np.uint64) & rng.integers(0, 2**64, size=20011, dtype=np.uint64)
    if kt == 5:
        k = k[~np.isnan(k.view(np.float64))]
    assert np.array_equal(oracle.lsb_reference_ranks_u64(k, kt), np.argsort(k.view(view), kind="stable"))
    r = oracle.lsb_reference_ranks_u64(k, kt, descending=True)           # reverse / stable sort / reverse
    kk = k.view(view)[r]
    assert np.all(kk[1:] <= kk[:-1])
    ties = kk[1:] == kk[:-1]
    assert np.all(r[1:][ties] > r[:-1][ties])                            # equal keys keep input order
    m = oracle.lsb_reference_ranks_u64(k, 3, 31, 33)
    assert np.array_equal(m, np.argsort((k >> np.uint64(31)) & np.uint64(3), kind="stable"))


def test_twiddles_64bit_are_order_preserving(oracle):
    L = oracle.lib()
    f = np.array([-np.inf, -1e300, -2.5, -5e-324, -0.0, 0.0, 5e-324, 1.0, 1e300, np.inf])
    t = [L.orc_twiddle_in_f64(int(x)) for x in f.view(np.uint64)]
    assert t == sorted(t) and len(set(t)) == len(t)
    assert [L.orc_twiddle_out_f64(x) for x in t] == [int(x) for x in f.view(np.uint64)]
    i = np.array([-2**63, -5, -1, 0, 1, 2**63 - 1], dtype=np.int64)
    t = [L.orc_twiddle_in_i64(int(x)) for x in i.view(np.uint64)]
    assert t == sorted(t)
    assert L.orc_twiddle_in_u64(12345) == 12345


def test_msb_classification_oracle_invariants(oracle):
    """oracle.msb_level_lists (the CPU restatement of row M4 the GPU classification is compared with): at every level the
    buckets passed on and the tasks emitted tile exactly the keys of the level's buckets; every task fits its class and no
    smaller one; a merged task (more bits than the level leaves) is below the merge threshold; a passed-on bucket exceeds
    the largest local capacity; the heavy-hitter rule only fires on buckets one value dominates."""
    caps = oracle.MSB_CLASS_CAPS
    n = (1 << 21) + 123
    for keys in (oracle.gen_uniform(n, seed=3), oracle.gen_zipf(n, seed=4), oracle.gen_entropy_and(n, 4, seed=5)):
        covered_prev = n
        for level in range(3):
            rb = 24 - 8 * level
            buckets, tasks = oracle.msb_level_lists(keys, level, pivot=False)
            in_tasks = sum(s for c in tasks.values() for _, s, _ in c)
            in_buckets = sum(s for _, s in buckets)
            assert in_tasks + in_buckets == covered_prev            # nothing lost, nothing counted twice
            for c, ts in tasks.items():
                for off, size, bits in ts:
                    assert size <= caps[c] and (c == 0 or size > caps[c - 1])
                    assert bits in (rb, rb + 8)
                    if bits == rb + 8:
                        assert size < oracle.MSB_MERGE
            assert all(s > caps[-1] for _, s in buckets)
            ranges = sorted([(o, s) for o, s in buckets] + [(o, s) for c in tasks.values() for o, s, _ in c])
            assert all(a[0] + a[1] <= b[0] for a, b in zip(ranges, ranges[1:]))   # disjoint
            covered_prev = in_buckets
            if not buckets:
                break
    # heavy-hitter rule: a bucket with one value on 97 % of its keys is taken (the three-sample candidate misses a 97 %
    # value with probability 0.003; the seed below does not), one with 30 % is not, whatever the samples say
    rng = np.random.default_rng(1)
    base = (oracle.gen_uniform(60000, seed=9) & np.uint32(0x0000ffff)) | np.uint32(0x12340000)
    hot = base.copy(); hot[rng.random(60000) < 0.97] = np.uint32(0x12345678)
    cold = base.copy(); cold[rng.random(60000) < 0.3] = np.uint32(0x12345678)
    got = oracle.msb_heavy_hitter(hot)
    assert got is not None and got[0] == 0x12345678 and got[1] + got[2] <= 60000 and got[2] == int((hot == 0x12345678).sum())
    assert oracle.msb_heavy_hitter(cold) is None


def test_msb_list_capacities_hold_against_adversarial_size_patterns():
    """the workspace's device-side lists (buckets of a level, local-sort tasks per class, tile records)
    are sized by an argument (gs_msb.hip, msb_max_*); here the bounds gs_msb_capacities reports are held against the
    classification rule (oracle.msb_classify_counts, the restatement of cuda_radix_sort.h:1084-1087,1241-1247) on the
    sub-bucket size patterns that stress them: every sub-bucket one key short of merging, 1-key sub-buckets between
    buckets one key over the largest local sort, and so on.  CPU only: the library is loaded, no kernel runs."""
    import ctypes as C
    import gpu_sort_amd as gs
    from oracle import oracle as O
    cap = O.MSB_CLASS_CAPS[-1]
    merge = O.MSB_MERGE

    def caps_of(n):
        mb, mt, ml = C.c_uint32(), C.c_uint32(), C.c_uint32()
        gs.lib.gs_msb_capacities(n, 0, C.byref(mb), C.byref(mt), C.byref(ml))
        return mb.value, mt.value, ml.value

    patterns = {
        "one short of merging": [merge - 1],                 # nothing merges: every sub-bucket is a task
        "merge limit pairs": [merge - 1, 1],                 # (merge - 1) + 1 is not < merge: two tasks per `merge` keys
        "1 between large": [1, cap + 1],                     # alternating: a 1-key task, a next-level bucket
        "just large": [cap + 1],                             # every sub-bucket a next-level bucket of the smallest size
        "ones": [1],                                         # long merged runs
        "large then ragged": [cap + 1, merge - 1, 1, 1],
    }
    for n in (1 << 20, (1 << 24) + 12345, 1 << 30, (1 << 32) - 1):
        max_b, max_t, max_l = caps_of(n)
        for name, pat in patterns.items():
            # one LEVEL made of buckets that each repeat the pattern over their sub-buckets (up to 256 of them, fewer when
            # the array is too small for that), as many such buckets as n keys allow
            counts = np.zeros(256, dtype=np.int64)
            left = n
            for i in range(256):
                c = pat[i % len(pat)]
                if c > left:
                    break
                counts[i] = c
                left -= c
            per_bucket = int(counts.sum())
            if per_bucket <= cap:                # such a range is a local-sort task, never a bucket of a level
                continue
            nb_real = n // per_bucket
            bk, tk = O.msb_classify_counts(counts, 0, 1)
            next_buckets = nb_real * len(bk)
            tasks_per_class = np.bincount([t[0] for t in tk], minlength=4) * nb_real
            tiles_next = nb_real * sum(-(-size // 8192) for _, size in bk)
            assert nb_real <= max_b and next_buckets <= max_b, (n, name, next_buckets, max_b)
            assert tasks_per_class.max() <= max_t, (n, name, tasks_per_class, max_t)
            assert tiles_next <= max_l, (n, name, tiles_next, max_l)
        # the bound's own argument, checked numerically: at most 2n / merge tasks + 256 per bucket
        assert max_t >= 2 * n // merge + 256 and max_b >= n // cap
