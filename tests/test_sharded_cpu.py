"""CPU tests of the multi-GPU host logic: split computation, exchange plan and the
all-to-all plumbing of gpu_sort_amd.sharded over gloo with world_size 2 and 4.
The compute steps (histogram / partition / local sort) are replaced by a numpy
stand-in defined HERE (test infrastructure; the product's DeviceOps calls the HIP
library and has no CPU fallback)."""
import os
import socket
import sys

import numpy as np
import pytest
import torch
import torch.distributed as dist
import torch.multiprocessing as mp

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


class NumpyOps:
    def __init__(self):
        from oracle import oracle as O
        self.O = O

    @staticmethod
    def _u32(t, n):
        return t[:n].numpy().view(np.uint32)

    def histogram(self, keys, n, bits):
        h = np.bincount(self._u32(keys, n) >> np.uint32(32 - bits), minlength=1 << bits).astype(np.int64)
        return torch.from_numpy(h)

    def partition(self, keys, vals, n, bits, dest_np, world, temp, keys_out, vals_out, bin_hist=None):
        k = self._u32(keys, n)
        d = dest_np[k >> np.uint32(32 - bits)]
        order = np.argsort(d, kind="stable")
        keys_out[:n] = torch.from_numpy(k[order].view(np.int32))
        if vals is not None:
            vals_out[:n] = torch.from_numpy(self._u32(vals, n)[order].view(np.int32))
        return torch.from_numpy(np.bincount(d, minlength=world).astype(np.int64))

    def temp_bytes(self, n, pairs, world=1):
        return 256

    def first_pass(self, keys, vals, n, temp, keys_out, vals_out):
        k = self._u32(keys, n)
        order = np.argsort(k >> np.uint32(24), kind="stable")
        keys_out[:n] = torch.from_numpy(k[order].view(np.int32))
        if vals is not None:
            vals_out[:n] = torch.from_numpy(self._u32(vals, n)[order].view(np.int32))
        return torch.from_numpy(np.bincount(k >> np.uint32(24), minlength=256).astype(np.int64))

    def finish(self, keys, vals, m, keys_out, vals_out, piece_counts, temp):
        # what gs_msb_finish_u32 is told must describe the buffer: source after source, top bytes ascending
        k = self._u32(keys, m)
        assert int(piece_counts.sum()) == m
        pos = 0
        for s in range(piece_counts.shape[0]):
            for b in np.nonzero(piece_counts[s])[0]:
                c = int(piece_counts[s][b])
                assert np.all(k[pos:pos + c] >> np.uint32(24) == b), "piece table does not match the received buffer"
                pos += c
        return self.local_sort(keys, vals, m, keys_out, vals_out, temp)

    def local_sort(self, keys, vals, n, keys_alt, vals_alt, temp, algo="lsb"):
        k = self._u32(keys, n)
        if vals is None:
            keys_alt[:n] = torch.from_numpy(self.O.lsb_sort_keys(k).view(np.int32))
            return keys_alt, None
        ko, vo = self.O.lsb_sort_pairs(k, self._u32(vals, n))
        keys_alt[:n] = torch.from_numpy(ko.view(np.int32))
        vals_alt[:n] = torch.from_numpy(vo.view(np.int32))
        return keys_alt, vals_alt

    def empty(self, n):
        return torch.empty(max(n, 1), dtype=torch.int32)

    def check_sorted(self, keys, count):
        k = self._u32(keys, count)
        s, x = self.O.multiset_checksum(k)
        return self.O.count_inversions_adjacent(k), s, x


def _free_port():
    with socket.socket() as s:
        s.bind(("127.0.0.1", 0))
        return s.getsockname()[1]


def _worker(rank, world, port, n, dist_kind, pairs, out_dir, pipeline="msb", groups=4, max_msg=None):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpu_sort_amd import sharded
    from oracle import oracle as O
    if max_msg:
        sharded.MAX_MSG = max_msg            # tiny messages: the multi-round paths of the exchange
    gen = {"uniform": O.gen_uniform, "zipf": O.gen_zipf}[dist_kind]
    keys = gen(n, 0, rank * n) if dist_kind != "const" else np.full(n, 7, np.uint32)
    vals = (O.gen_enumerated(n, rank * n)) if pairs else None
    srt = sharded.ShardedSorter(n, pairs, torch.device("cpu"), ops=NumpyOps(), pipeline=pipeline, groups=groups)
    tk = torch.from_numpy(keys.view(np.int32).copy())
    tv = torch.from_numpy(vals.view(np.int32).copy()) if pairs else None
    chk = srt.input_checksum(tk)
    sk, sv, cnt = srt.sort(tk, tv)
    ok, glob = srt.verify(sk, cnt, chk)
    assert ok, "sharded result fails the global properties"
    assert int(srt.last["send"].sum()) == n and int(srt.last["recv"].sum()) == cnt
    if srt.last["pipeline"] == "msb" and world > 1:
        assert srt.last["groups"] == groups and sum(srt.last["group_counts"]) == cnt
    with open(os.path.join(out_dir, f"p{rank}.txt"), "w") as f:
        f.write(srt.last["pipeline"])
    np.save(os.path.join(out_dir, f"k{rank}.npy"), sk[:cnt].numpy().view(np.uint32))
    if pairs:
        np.save(os.path.join(out_dir, f"v{rank}.npy"), sv[:cnt].numpy().view(np.uint32))
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,dist_kind,pairs,pipeline,groups,max_msg",
                         [(2, "uniform", False, "msb", 4, None), (2, "zipf", True, "msb", 4, None),
                          (4, "uniform", True, "msb", 3, None), (3, "uniform", False, "msb", 1, None),
                          (2, "uniform", True, "msb", 200, None),     # more groups than buckets
                          (2, "uniform", True, "partition", 4, None), (4, "zipf", False, "partition", 4, None),
                          # messages capped at 1500 elements: every exchange goes out in several rounds
                          (2, "uniform", True, "msb", 4, 1500), (3, "uniform", False, "msb", 1, 1500),
                          (2, "zipf", True, "partition", 4, 1500),
                          # the rank count of BASELINE configs[4]
                          (8, "uniform", False, "msb", 4, None), (8, "zipf", True, "msb", 2, 1500)])
def test_sharded_sort_over_gloo(tmp_path, oracle, world, dist_kind, pairs, pipeline, groups, max_msg, n=50000):
    mp.spawn(_worker, args=(world, _free_port(), n, dist_kind, pairs, str(tmp_path), pipeline, groups, max_msg), nprocs=world,
             join=True)
    used = {open(tmp_path / f"p{r}.txt").read() for r in range(world)}
    assert len(used) == 1                                             # every rank took the same pipeline
    if dist_kind == "uniform":
        assert used == {pipeline}
    gen = {"uniform": oracle.gen_uniform, "zipf": oracle.gen_zipf}[dist_kind]
    all_keys = np.concatenate([gen(n, 0, r * n) for r in range(world)])
    got = np.concatenate([np.load(tmp_path / f"k{r}.npy") for r in range(world)])
    assert np.array_equal(got, np.sort(all_keys))                     # concatenation in rank order = global sort
    if pairs:
        gv = np.concatenate([np.load(tmp_path / f"v{r}.npy") for r in range(world)])
        assert oracle.msb_check_pairs_enumerated(all_keys, got, gv) == 0   # values are global indices
    sizes = [np.load(tmp_path / f"k{r}.npy").size for r in range(world)]
    if dist_kind == "uniform":
        assert max(sizes) < 1.1 * n                                   # balanced to within a few bins


def test_group_bins_properties():
    """Every rank's run of bins is cut into monotone groups of about equal totals; all keys are covered once."""
    sys.path.insert(0, ROOT)
    from gpu_sort_amd.sharded import compute_splits, group_bins
    rng = np.random.default_rng(1)
    for world in (2, 3, 8):
        hist = rng.integers(0, 1000, size=(world, 256)).astype(np.int64)
        hist[:, 17] = 0                                               # an empty bucket
        dest, per_rank = compute_splits(hist, world)
        for groups in (1, 2, 4, 300):
            grp = group_bins(hist, dest, world, groups)
            assert grp.min() >= 0 and grp.max() < groups
            tot = hist.sum(axis=0)
            for r in range(world):
                sel = dest == r
                assert np.all(np.diff(grp[sel]) >= 0)                 # contiguous runs of bins per (rank, group)
                per_group = np.array([tot[sel & (grp == g)].sum() for g in range(groups)])
                assert per_group.sum() == per_rank[r]
                if groups <= 4:
                    assert per_group.max() <= per_rank[r] / groups + tot[sel].max()   # balanced to within one bucket
    assert np.all(group_bins(np.zeros((2, 256), np.int64), np.zeros(256, np.uint8), 2, 4) == 0)    # empty input


def test_compute_splits_properties():
    sys.path.insert(0, ROOT)
    from gpu_sort_amd.sharded import compute_splits, exchange_plan
    rng = np.random.default_rng(0)
    for world in (1, 2, 3, 8):
        hist = rng.integers(0, 1000, size=(world, 4096)).astype(np.int64)
        hist[:, 100] += 200000                                        # one heavy bin
        dest, per_rank = compute_splits(hist, world)
        assert dest.dtype == np.uint8 and dest.min() == 0 and dest.max() <= world - 1
        assert np.all(np.diff(dest.astype(int)) >= 0)                 # contiguous key ranges per rank
        assert per_rank.sum() == hist.sum()
        tot_send = np.zeros(world, np.int64)
        for r in range(world):
            send, recv = exchange_plan(hist, dest, r, world)
            assert send.sum() == hist[r].sum() and recv.sum() == per_rank[r]
            tot_send += send
        assert np.array_equal(tot_send, per_rank)
    dest, per_rank = compute_splits(np.zeros((2, 16), np.int64), 2)   # empty input
    assert per_rank.sum() == 0


@pytest.mark.parametrize("n,world,pipeline,groups", [(1, 2, "msb", 1), (3, 3, "msb", 4), (7, 2, "partition", 1)])
def test_tiny_shards(tmp_path, oracle, n, world, pipeline, groups):
    """a handful of keys per rank: most ranks receive nothing, most buckets and groups are empty"""
    test_sharded_sort_over_gloo(tmp_path, oracle, world, "zipf", True, pipeline, groups, None, n=n)


def _selftest_worker(rank, world, port, elements, corrupt):
    sys.path.insert(0, ROOT)
    os.environ["MASTER_ADDR"] = "127.0.0.1"
    os.environ["MASTER_PORT"] = str(port)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    from gpu_sort_amd import sharded
    if corrupt:
        # a collective that loses the second half of every message (what RCCL of ROCm 7.2 was seen to do to a
        # >= 2 GiB self-send, tools/rccl_selfcopy.py): the self-test must notice
        real = dist.all_to_all_single

        def lossy(out, inp, recv, send, group=None):
            real(out, inp, recv, send, group=group)
            out[out.numel() // 2:] = 0
        dist.all_to_all_single = lossy
        with pytest.raises(RuntimeError, match="self-test failed"):
            sharded.communicator_selftest(torch.device("cpu"), elements=elements)
        dist.all_to_all_single = real
    else:
        assert sharded.communicator_selftest(torch.device("cpu"), elements=elements) == elements
    dist.barrier()
    dist.destroy_process_group()


@pytest.mark.parametrize("world,corrupt", [(1, False), (2, False), (3, False), (2, True)])
def test_communicator_selftest_over_gloo(world, corrupt):
    """The start-up check bench.py runs on the real communicator (one big message around a ring, compared word for word):
    passes on a working collective for 1, 2 and 3 ranks, raises on one that truncates."""
    mp.spawn(_selftest_worker, args=(world, _free_port(), 100003, corrupt), nprocs=world, join=True)
