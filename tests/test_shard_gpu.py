"""GPU tests of the shard helpers (gs_shard_histogram_u32 / gs_shard_partition_u32) and of the
single-rank ShardedSorter path.  The multi-rank exchange itself is covered on CPU over gloo
(tests/test_sharded_cpu.py); RCCL needs one GPU per rank and this box has one."""
import numpy as np
import pytest
import torch

from conftest import to_dev, to_u32

pytestmark = pytest.mark.gpu


@pytest.mark.parametrize("n", [0, 1, 1000, 8192, 100003, (1 << 22) + 5])
@pytest.mark.parametrize("dist_kind", ["uniform", "zipf"])
def test_histogram_and_partition(gs, cuda, oracle, n, dist_kind):
    from gpu_sort_amd import sharded
    ops = sharded.DeviceOps(cuda)
    keys = (oracle.gen_uniform if dist_kind == "uniform" else oracle.gen_zipf)(n, seed=2)
    vals = oracle.gen_enumerated(n)
    bits, world = sharded.SHARD_BITS, 8
    dk, dv = to_dev(keys, cuda), to_dev(vals, cuda)
    hist = ops.histogram(dk, n, bits).cpu().numpy()
    want = np.bincount(keys >> np.uint32(32 - bits), minlength=1 << bits)
    assert np.array_equal(hist, want)
    # pretend 8 ranks hold the same shard
    dest, per_rank = sharded.compute_splits(np.tile(hist, (world, 1)), world)
    temp = torch.empty(ops.temp_bytes(max(n, 1), True), dtype=torch.uint8, device=cuda)
    ko, vo = ops.empty(n), ops.empty(n)
    counts = ops.partition(dk, dv, n, bits, dest, world, temp, ko, vo, bin_hist=ops.histogram(dk, n, bits)).cpu().numpy()
    torch.cuda.synchronize()
    d = dest[keys >> np.uint32(32 - bits)] if n else np.zeros(0, np.uint8)
    assert np.array_equal(counts, np.bincount(d, minlength=world))
    gk, gv = to_u32(ko)[:n], to_u32(vo)[:n]
    assert np.array_equal(keys[gv], gk) if n else True            # pairs stay together
    assert np.array_equal(np.sort(gv), vals)                       # a permutation
    gd = dest[gk >> np.uint32(32 - bits)] if n else d
    assert np.all(np.diff(gd.astype(int)) >= 0)                    # grouped by destination rank
    # keys-only form
    ko2 = ops.empty(n)
    counts2 = ops.partition(dk, None, n, bits, dest, world, temp, ko2, None).cpu().numpy()
    assert np.array_equal(counts2, counts)
    assert np.array_equal(np.sort(to_u32(ko2)[:n]), np.sort(keys))


@pytest.mark.parametrize("pairs", [False, True])
@pytest.mark.parametrize("algo", ["lsb", "msb"])
def test_sharded_sorter_single_rank(gs, cuda, oracle, pairs, algo):
    from gpu_sort_amd import sharded
    n = 300007
    keys = oracle.gen_zipf(n, seed=1)
    vals = oracle.gen_enumerated(n) if pairs else None
    srt = sharded.ShardedSorter(n, pairs, cuda, local_algo=algo)
    dk = to_dev(keys, cuda)
    dv = to_dev(vals, cuda) if pairs else None
    chk = srt.input_checksum(dk)
    sk, sv, cnt = srt.sort(dk, dv)
    torch.cuda.synchronize()
    assert cnt == n and np.array_equal(to_u32(sk)[:n], np.sort(keys))
    ok, _ = srt.verify(sk, cnt, chk)
    assert ok
    if pairs:
        assert oracle.msb_check_pairs_enumerated(keys, to_u32(sk)[:n], to_u32(sv)[:n]) == 0
