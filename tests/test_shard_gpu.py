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
@pytest.mark.parametrize("algo", ["lsb", "msb", "pipeline-msb"])
def test_sharded_sorter_single_rank(gs, cuda, oracle, pairs, algo):
    from gpu_sort_amd import sharded
    n = 300007
    keys = oracle.gen_zipf(n, seed=1)
    vals = oracle.gen_enumerated(n) if pairs else None
    if algo == "pipeline-msb":
        srt = sharded.ShardedSorter(n, pairs, cuda, pipeline="msb")
    else:
        srt = sharded.ShardedSorter(n, pairs, cuda, local_algo=algo, pipeline="partition")
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
    assert srt.last["pipeline"] == ("msb" if algo == "pipeline-msb" else "partition")


@pytest.mark.parametrize("n", [0, 1, 777, 8192, 100003, (1 << 21) + 11])
def test_first_pass_groups_by_top_byte(gs, cuda, oracle, n):
    """gs_msb_first_pass_u32 = level 0 of the MSB sort on its own: stable partition on the top byte + sizes."""
    from gpu_sort_amd import sharded
    ops = sharded.DeviceOps(cuda)
    keys, vals = oracle.gen_uniform(n, seed=9), oracle.gen_enumerated(n)
    temp = torch.empty(ops.temp_bytes(max(n, 1), True), dtype=torch.uint8, device=cuda)
    ko, vo = ops.empty(n), ops.empty(n)
    counts = ops.first_pass(to_dev(keys, cuda), to_dev(vals, cuda), n, temp, ko, vo).cpu().numpy()
    torch.cuda.synchronize()
    assert np.array_equal(counts, np.bincount(keys >> np.uint32(24), minlength=256))
    order = np.argsort(keys >> np.uint32(24), kind="stable")
    assert np.array_equal(to_u32(ko)[:n], keys[order]) and np.array_equal(to_u32(vo)[:n], vals[order])


def _emulate_receive(oracle, ops, cuda, shards, byte_lo, byte_hi, pairs):
    """What rank `r` owning top bytes [byte_lo, byte_hi) receives from the given source shards."""
    rk, rv, pieces, base = [], [], [], 0
    for keys in shards:
        n = keys.size
        vals = np.arange(base, base + n, dtype=np.uint32)
        temp = torch.empty(ops.temp_bytes(max(n, 1), pairs), dtype=torch.uint8, device=cuda)
        ko, vo = ops.empty(n), ops.empty(n)
        counts = ops.first_pass(to_dev(keys, cuda), to_dev(vals, cuda) if pairs else None, n, temp, ko,
                                vo if pairs else None).cpu().numpy()
        torch.cuda.synchronize()
        start = int(counts[:byte_lo].sum()); stop = int(counts[:byte_hi].sum())
        rk.append(to_u32(ko)[start:stop].copy())
        if pairs:
            rv.append(to_u32(vo)[start:stop].copy())
        c = np.zeros(256, np.uint64); c[byte_lo:byte_hi] = counts[byte_lo:byte_hi]
        pieces.append(c)
        base += n
    return np.concatenate(rk), (np.concatenate(rv) if pairs else None), np.stack(pieces)


@pytest.mark.parametrize("pairs", [False, True])
@pytest.mark.parametrize("case", ["uniform8", "zipf3", "tiny", "one_byte", "uneven", "empty_range", "deep"])
def test_finish_on_received_pieces(gs, cuda, oracle, pairs, case):
    """gs_msb_finish_u32: buckets arriving in one piece per source rank are sorted where they lie."""
    from gpu_sort_amd import sharded
    ops = sharded.DeviceOps(cuda)
    if case == "uniform8":
        shards = [oracle.gen_uniform(150001 + 1000 * i, seed=i) for i in range(8)]; lo, hi = 64, 96
    elif case == "zipf3":
        shards = [oracle.gen_zipf(400000, seed=0, start=i * 400000) for i in range(3)]; lo, hi = 0, 128
    elif case == "tiny":
        shards = [oracle.gen_uniform(300 + i, seed=20 + i) for i in range(4)]; lo, hi = 0, 256
    elif case == "one_byte":                      # a single hot bucket spread over the sources
        shards = [(oracle.gen_uniform(120000, seed=30 + i) & np.uint32(0x00FFFFFF)) | np.uint32(0x7B000000) for i in range(5)]
        lo, hi = 0x7B, 0x7C
    elif case == "uneven":                        # sources of very different size, one of them empty
        shards = [oracle.gen_uniform(m, seed=40 + i) for i, m in enumerate([500000, 0, 17, 90001])]; lo, hi = 10, 200
    elif case == "deep":                          # two 8M-key buckets in 4 pieces each: a second partition level is needed
        shards = [oracle.gen_uniform(1 << 22, seed=60 + i) & np.uint32(0x01FFFFFF) for i in range(4)]; lo, hi = 0, 2
    else:
        shards = [oracle.gen_uniform(50000, seed=50 + i) & np.uint32(0x0FFFFFFF) for i in range(2)]; lo, hi = 128, 256
    all_keys = np.concatenate(shards)
    rk, rv, pieces = _emulate_receive(oracle, ops, cuda, shards, lo, hi, pairs)
    m = rk.size
    mine = all_keys[(all_keys >> np.uint32(24) >= lo) & (all_keys >> np.uint32(24) < hi)]
    assert m == mine.size == int(pieces.sum())
    temp = torch.empty(ops.temp_bytes(max(m, 1), pairs, len(shards)), dtype=torch.uint8, device=cuda)
    dk, out_k = to_dev(rk, cuda) if m else ops.empty(0), ops.empty(m)
    dv = (to_dev(rv, cuda) if m else ops.empty(0)) if pairs else None
    out_v = ops.empty(m) if pairs else None
    sk, sv = ops.finish(dk, dv, m, out_k, out_v, pieces, temp)
    torch.cuda.synchronize()
    assert np.array_equal(to_u32(sk)[:m], np.sort(mine))
    if pairs:
        got_v = to_u32(sv)[:m]
        assert np.array_equal(all_keys[got_v], to_u32(sk)[:m])        # every value still names its key
        assert np.unique(got_v).size == m


def test_finish_rejects_inconsistent_piece_table(gs, cuda):
    from gpu_sort_amd import sharded
    ops = sharded.DeviceOps(cuda)
    temp = torch.empty(ops.temp_bytes(1000, False, 2), dtype=torch.uint8, device=cuda)
    pieces = np.zeros((2, 256), np.uint64); pieces[0, 3] = 400; pieces[1, 3] = 500     # 900 != 1000
    with pytest.raises(gs.GpuSortError):
        ops.finish(ops.empty(1000), None, 1000, ops.empty(1000), None, pieces, temp)


def _gpu_rank_worker(rank, world, port, n, dist_kind, pairs, pipeline, out_dir, groups=4):
    """One rank of a multi-rank sort whose compute runs on the (shared) GPU through the real DeviceOps; the
    exchange goes over gloo (device buffers staged through the host: RCCL needs one GPU per rank)."""
    import os, sys
    import torch.distributed as dist
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    sys.path.insert(0, root)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    import gpu_sort_amd as gs
    from gpu_sort_amd import sharded
    dev = torch.device("cuda:0")
    torch.cuda.set_device(dev)
    gen = gs.generate_uniform_keys if dist_kind == "uniform" else gs.generate_zipf_keys
    keys = gen(n, seed=0, start=rank * n, device=dev)
    vals = gs.generate_enumerated_values(n, start=rank * n, device=dev) if pairs else None
    srt = sharded.ShardedSorter(n, pairs, dev, pipeline=pipeline, local_algo="lsb", groups=groups)
    chk = srt.input_checksum(keys)
    if groups == 1 and pipeline == "msb":
        srt.stage_times = {}                 # the serialised, timed variant bench.py uses for its stage breakdown
    sk, sv, cnt = srt.sort(keys, vals)
    torch.cuda.synchronize()
    if groups == 1 and pipeline == "msb":
        assert set(srt.stage_times) == {"first_pass_ms", "exchange_ms", "finish_ms"}
    if pipeline == "msb":
        assert srt.last["groups"] == groups and sum(srt.last["group_counts"]) == cnt
    ok, _ = srt.verify(sk, cnt, chk)
    assert ok, "sharded result fails the global properties"
    np.save(os.path.join(out_dir, f"k{rank}.npy"), sk[:cnt].cpu().numpy().view(np.uint32))
    if pairs:
        np.save(os.path.join(out_dir, f"v{rank}.npy"), sv[:cnt].cpu().numpy().view(np.uint32))
    with open(os.path.join(out_dir, f"p{rank}.txt"), "w") as f:
        f.write(srt.last["pipeline"])
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,dist_kind,pairs,pipeline,groups",
                         [(2, "uniform", False, "msb", 4), (4, "uniform", True, "msb", 3), (3, "zipf", False, "msb", 4),
                          (2, "uniform", True, "msb", 1), (2, "uniform", True, "partition", 4)])
def test_multi_rank_on_one_gpu_over_gloo(tmp_path, cuda, oracle, world, dist_kind, pairs, pipeline, groups, n=400003):
    """world ranks (processes) share this GPU: the real kernels run for every rank, with pieces arriving from
    every other rank; only the transport differs from the 8-GPU run (gloo + host staging instead of RCCL)."""
    import socket
    import torch.multiprocessing as mp
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    mp.spawn(_gpu_rank_worker, args=(world, port, n, dist_kind, pairs, pipeline, str(tmp_path), groups), nprocs=world,
             join=True)
    gen = {"uniform": oracle.gen_uniform, "zipf": oracle.gen_zipf}[dist_kind]
    all_keys = np.concatenate([gen(n, 0, r * n) for r in range(world)])
    got = np.concatenate([np.load(tmp_path / f"k{r}.npy") for r in range(world)])
    assert np.array_equal(got, np.sort(all_keys))
    if pairs:
        gv = np.concatenate([np.load(tmp_path / f"v{r}.npy") for r in range(world)])
        assert oracle.msb_check_pairs_enumerated(all_keys, got, gv) == 0
    if n > 1000:
        assert {open(tmp_path / f"p{r}.txt").read() for r in range(world)} == {pipeline}


@pytest.mark.parametrize("n,world,pairs,groups", [(1, 2, True, 1), (5, 3, False, 4)])
def test_multi_rank_tiny_shards_on_the_gpu(tmp_path, cuda, oracle, n, world, pairs, groups):
    """a handful of keys per rank through the real kernels: empty receives, empty buckets, empty groups"""
    test_multi_rank_on_one_gpu_over_gloo(tmp_path, cuda, oracle, world, "uniform", pairs, "msb", groups, n=n)


def test_exchange_path_on_real_rccl_with_one_rank(tmp_path, cuda):
    """The one-GPU box cannot run two RCCL ranks, but a ONE-rank RCCL group runs the N > 1 code path as it is: the
    size all_gather, the list-form all_to_all issued as 4 asynchronous collectives on RCCL's stream, the compute
    stream waiting on each work handle, one finish per group -- keys and pairs, verified by the global properties."""
    import os
    import subprocess
    import sys
    code = r'''
import os, sys, torch, torch.distributed as dist
sys.path.insert(0, %r)
os.environ.setdefault("MASTER_ADDR", "127.0.0.1"); os.environ.setdefault("MASTER_PORT", "%d")
torch.cuda.set_device(0)
dist.init_process_group("nccl", rank=0, world_size=1, device_id=torch.device("cuda:0"))
import gpu_sort_amd as gs
from gpu_sort_amd import sharded
dev = torch.device("cuda:0")
n = 3000017
for pairs in (False, True):
    for groups in (4, 1):
        srt = sharded.ShardedSorter(n, pairs, dev, groups=groups, force_exchange=True)
        for seed in (1, 2):
            keys = gs.generate_uniform_keys(n, seed=seed, device=dev)
            vals = gs.generate_enumerated_values(n, device=dev) if pairs else None
            chk = srt.input_checksum(keys)
            sk, sv, cnt = srt.sort(keys, vals)
            torch.cuda.synchronize()
            ok, _ = srt.verify(sk, cnt, chk)
            assert ok and cnt == n and srt.last["pipeline"] == "msb" and srt.last["groups"] == groups, (pairs, groups, srt.last)
            if pairs:
                assert gs.check_pairs_enumerated(keys, sk, sv, n)[0] == 0
for pairs, algo in ((False, "lsb"), (True, "msb")):      # the fallback pipeline through the same collectives
    srt = sharded.ShardedSorter(n, pairs, dev, pipeline="partition", local_algo=algo, force_exchange=True)
    keys = gs.generate_zipf_keys(n, seed=7, device=dev)
    vals = gs.generate_enumerated_values(n, device=dev) if pairs else None
    chk = srt.input_checksum(keys)
    sk, sv, cnt = srt.sort(keys, vals)
    torch.cuda.synchronize()
    ok, _ = srt.verify(sk, cnt, chk)
    assert ok and cnt == n and srt.last["pipeline"] == "partition", (pairs, algo, srt.last)
    if pairs:
        assert gs.check_pairs_enumerated(keys, sk, sv, n)[0] == 0
dist.destroy_process_group()
print("rccl one-rank ok")
'''
    import socket
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        port = s.getsockname()[1]
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    out = subprocess.run([sys.executable, "-c", code % (root, port)], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0 and "rccl one-rank ok" in out.stdout, out.stdout[-2000:] + out.stderr[-3000:]
