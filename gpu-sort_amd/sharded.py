"""Multi-GPU bucket-sharded sort: one process per GPU, ONE all-to-all (RCCL over xGMI).

The reference is single-GPU; this is the north_star's 8-GPU path, the MSB sort
cut at its first digit ("msb" pipeline, the default):
  1. every rank runs the MSB path's first digit pass on its shard: keys grouped by top byte
     + the 256 bucket sizes                                             (gs_msb_first_pass_u32)
  2. all ranks exchange the bucket sizes (all_gather, 2 KiB)            (torch.distributed)
  3. every rank computes the same monotone bucket -> rank map with balanced totals; a rank's
     share is then ONE contiguous slice of every grouped shard
  4. one all-to-all with uneven splits moves every top-level bucket to its owner (no single message
     above MAX_MSG: bigger ones go out in rounds)
  5. every rank finishes the MSB sort on the buckets it received, picking the pieces up where
     they lie (no regrouping pass)                                      (gs_msb_finish_u32)
Rank r then holds the r-th slice of the globally sorted sequence.  Opt-in (`groups` > 1): the exchange
as `groups` collectives, each carrying one run of every owner's buckets, all enqueued at once on RCCL's
stream, and one finish per group enqueued behind its own collective only (see ShardedSorter).

When 256 buckets cannot balance the ranks (a heavy top byte), the "partition" pipeline is used
instead (decided identically on every rank from the gathered sizes): 4096-bin histogram of the
top SHARD_BITS bits (gs_shard_histogram_u32), group-by-destination (gs_shard_partition_u32), the
same single all-to-all, then a full local sort (gs_lsb_sort_u32 / gs_msb_sort_u32).

The compute steps go through `ops` (DeviceOps = the HIP library).  The host logic --
split computation and exchange plan -- is backend-independent, so the CPU test-suite
drives it over gloo with a numpy stand-in for `ops` (tests/test_sharded_cpu.py).
"""
import ctypes as C

import numpy as np
import torch
import torch.distributed as dist

SHARD_BITS = 12
# Largest single send/recv handed to the collective library, in 4-byte elements (768 MiB).  Measured on the MI355X box
# (ROCm 7.2 RCCL under torch 2.10): a rank's send to ITSELF silently delivers only the first half of a message of 2 GiB
# or more (tools/rccl_selfcopy.py); 1 GiB is still whole.  Bigger messages are cut into rounds below this size; the
# 512 MiB messages of 8 ranks x 2^30 keys (a little more on some pairs, the buckets being what they are) stay whole.
MAX_MSG = 3 << 26


def compute_splits(hist_all, world):
    """hist_all: (world, nbins) counts.  Returns (dest_of_bin uint8[nbins], recv_total per rank).

    Bins are assigned to ranks in key order; bin b goes to rank floor(world * keys_before_b / N),
    so every rank owns a contiguous key range and totals are balanced to within one bin.  The quotient is
    taken in exact integer arithmetic (Python ints), bit for bit what the C++ host computes with 128-bit
    integers (gs_sharded_compute_splits, csrc_rccl/gs_sharded_rccl.cpp): two hosts of one exchange must never
    disagree by a bucket, which a float64 quotient could do next to an exact multiple."""
    tot = [int(x) for x in np.asarray(hist_all).sum(axis=0, dtype=np.uint64)]
    n = sum(tot)
    dest = np.zeros(len(tot), np.uint8)
    per_rank = np.zeros(world, dtype=np.int64)
    before, prev = 0, 0
    for b, c in enumerate(tot):
        d = min(before * world // n, world - 1) if n else 0
        d = max(d, prev)                             # monotone by construction; kept so for the last rank's clamp
        prev = d
        dest[b] = d
        per_rank[d] += c
        before += c
    return dest, per_rank


def group_bins(hist_all, dest, world, groups):
    """group_of_bin: every rank's run of bins cut into `groups` runs of about equal key totals (monotone
    inside a rank, so a (rank, group) pair is again one contiguous run of bins).  A function of the gathered
    sizes only: every rank computes the same map."""
    tot = hist_all.sum(axis=0).astype(np.float64)
    grp = np.zeros(tot.size, np.int64)
    if groups <= 1:
        return grp
    for r in range(world):
        sel = np.nonzero(dest == r)[0]
        if sel.size == 0:
            continue
        t = tot[sel]
        total = t.sum()
        if total == 0:
            continue
        before = np.cumsum(t) - t
        g = np.minimum((before * groups / total).astype(np.int64), groups - 1)
        grp[sel] = np.maximum.accumulate(g)
    return grp


def biggest_message(hist_all, dest, world):
    """The largest (source, destination) message of the exchange, in elements: known to every rank alike."""
    return max(int(hist_all[src][dest == r].sum()) for src in range(world) for r in range(world)) if hist_all.size else 0


def exchange_plan(hist_all, dest, rank, world):
    """send_counts[r] = my keys going to rank r; recv_counts[r] = keys rank r sends to me."""
    send = np.array([int(hist_all[rank][dest == r].sum()) for r in range(world)], dtype=np.int64)
    recv = np.array([int(hist_all[r][dest == rank].sum()) for r in range(world)], dtype=np.int64)
    return send, recv


class DeviceOps:
    """Compute steps on the GPU through the C ABI (libgpusort.so)."""

    def __init__(self, device):
        from . import _lib
        from .lsb import _stream_ptr
        self.lib, self.check, self._lib, self._sp = _lib.lib, _lib.check, _lib, _stream_ptr
        self.device = device

    def histogram(self, keys, n, bits):
        hist = torch.empty(1 << bits, dtype=torch.int64, device=self.device)
        self.check(self.lib.gs_shard_histogram_u32(keys.data_ptr(), n, bits, hist.data_ptr(), self._lib.GS_KEY_U32,
                                                   self._sp(None)), "gs_shard_histogram_u32")
        return hist

    def partition(self, keys, vals, n, bits, dest_np, world, temp, keys_out, vals_out, bin_hist=None):
        dest = torch.from_numpy(dest_np).to(self.device)
        counts = torch.empty(world, dtype=torch.int64, device=self.device)
        self.check(self.lib.gs_shard_partition_u32(temp.data_ptr(), temp.numel(), keys.data_ptr(), keys_out.data_ptr(),
                                                   vals.data_ptr() if vals is not None else None,
                                                   vals_out.data_ptr() if vals is not None else None, n, bits,
                                                   dest.data_ptr(), world,
                                                   bin_hist.data_ptr() if bin_hist is not None else None,
                                                   counts.data_ptr(), self._lib.GS_KEY_U32,
                                                   self._sp(None)), "gs_shard_partition_u32")
        return counts

    def temp_bytes(self, n, pairs, world=1):
        return max(self.lib.gs_msb_temp_bytes(n, int(pairs)), self.lib.gs_lsb_temp_bytes(n, int(pairs)),
                   self.lib.gs_msb_finish_temp_bytes(n, int(pairs), world), 256)

    def first_pass(self, keys, vals, n, temp, keys_out, vals_out):
        """MSB level 0 on its own: grouped keys (values) + the 256 bucket sizes (int64, on the device)."""
        counts = torch.empty(256, dtype=torch.int64, device=self.device)
        self.check(self.lib.gs_msb_first_pass_u32(temp.data_ptr(), temp.numel(), keys.data_ptr(), keys_out.data_ptr(),
                                                  vals.data_ptr() if vals is not None else None,
                                                  vals_out.data_ptr() if vals is not None else None, n,
                                                  self._lib.GS_KEY_U32, counts.data_ptr(), self._sp(None)),
                   "gs_msb_first_pass_u32")
        return counts

    def finish(self, keys, vals, m, keys_out, vals_out, piece_counts, temp):
        """The rest of the MSB sort on received buckets; piece_counts: (num_src, 256) host array."""
        pc = np.ascontiguousarray(piece_counts, dtype=np.uint64)
        self.check(self.lib.gs_msb_finish_u32(temp.data_ptr(), temp.numel(), keys.data_ptr(),
                                              vals.data_ptr() if vals is not None else None, keys_out.data_ptr(),
                                              vals_out.data_ptr() if vals is not None else None, m,
                                              pc.ctypes.data_as(C.c_void_p), pc.shape[0], self._lib.GS_KEY_U32,
                                              self._sp(None), 0), "gs_msb_finish_u32")
        return keys_out, vals_out

    def local_sort(self, keys, vals, n, keys_alt, vals_alt, temp, algo="lsb"):
        from . import DoubleBuffer, DeviceRadixSort, rdxsrt_unstable_sort, GS_KEY_U32
        if n == 0:
            return keys, vals
        if algo == "msb":
            seq = rdxsrt_unstable_sort(keys, vals, n, keys_alt, vals_alt, pre_allocated_dm=temp, synchronize=False)
            return seq.sorted_keys, seq.sorted_values
        dk = DoubleBuffer(keys, keys_alt)
        if vals is not None:
            dv = DoubleBuffer(vals, vals_alt)
            DeviceRadixSort.SortPairs(temp, temp.numel(), dk, dv, n, key_type=GS_KEY_U32)
            return dk.Current(), dv.Current()
        DeviceRadixSort.SortKeys(temp, temp.numel(), dk, n, key_type=GS_KEY_U32)
        return dk.Current(), None

    def empty(self, n):
        return torch.empty(max(n, 1), dtype=torch.int32, device=self.device)

    def check_sorted(self, keys, count):
        from . import check_sorted
        return check_sorted(keys, count) if count else (0, 0, 0)


def _all_to_all(out, inp, recv_counts, send_counts, group, bound=None):
    """The one exchange.  RCCL moves device buffers directly; under the gloo backend (CPU tests, or two test
    ranks sharing one GPU) device buffers are staged through host memory, because gloo has no device all-to-all.
    bound: an upper bound on ANY rank's largest message that every rank knows (the shard size); when it exceeds
    MAX_MSG the exchange goes out in ceil(bound / MAX_MSG) rounds of the list form, each message cut at multiples of
    MAX_MSG (sender and receiver cut the same message at the same places)."""
    big = max(max(recv_counts, default=0), max(send_counts, default=0)) if bound is None else bound
    if big > MAX_MSG:
        rounds = -(-int(big) // MAX_MSG)
        so = np.concatenate(([0], np.cumsum(send_counts))).astype(np.int64)
        ro = np.concatenate(([0], np.cumsum(recv_counts))).astype(np.int64)
        for q in range(rounds):
            ins = [inp[int(so[r]) + min(q * MAX_MSG, c):int(so[r]) + min((q + 1) * MAX_MSG, c)] for r, c in enumerate(send_counts)]
            outs = [out[int(ro[r]) + min(q * MAX_MSG, c):int(ro[r]) + min((q + 1) * MAX_MSG, c)] for r, c in enumerate(recv_counts)]
            h = _all_to_all_lists(outs, ins, group)
            if h is not None:
                h.wait()
        return
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        h_out = torch.empty(out.numel(), dtype=out.dtype)
        dist.all_to_all_single(h_out, inp.cpu(), recv_counts, send_counts, group=group)
        out.copy_(h_out)
        return
    dist.all_to_all_single(out, inp, recv_counts, send_counts, group=group)


def _all_gather(out, inp, group):
    if inp.is_cuda and dist.get_backend(group) == "gloo":
        h = torch.empty(out.numel(), dtype=out.dtype)
        dist.all_gather_into_tensor(h, inp.cpu(), group=group)
        out.copy_(h)
        return
    dist.all_gather_into_tensor(out, inp, group=group)


def _all_to_all_lists(outs, ins, group):
    """All-to-all of per-peer tensors (views into the grouped shard / the receive buffer).  RCCL: asynchronous on
    its own stream, returns the work handle.  gloo: one all_to_all_single of host copies, returns None."""
    if dist.get_backend(group) == "gloo":
        h_in = torch.cat([t.cpu() for t in ins])
        h_out = torch.empty(sum(o.numel() for o in outs), dtype=h_in.dtype)
        dist.all_to_all_single(h_out, h_in, [o.numel() for o in outs], [t.numel() for t in ins], group=group)
        pos = 0
        for o in outs:
            o.copy_(h_out[pos:pos + o.numel()])
            pos += o.numel()
        return None
    return dist.all_to_all(outs, ins, group=group, async_op=True)


def communicator_selftest(device, group=None, elements=None):
    """Start-up check of the exchange path on the communicator at hand: every rank sends ONE message of `elements`
    32-bit words (default MAX_MSG = 768 MiB, the largest message the sorter ever emits) to the next rank of a ring
    (to itself when there is one rank) and checks, on the device, that every word arrived.  Raises RuntimeError on a
    truncated or altered message, so that a sort never runs on a communicator that drops data at the sizes it uses.
    Returns the number of words checked.  (Why the cap exists: on the MI355X box, ROCm 7.2's RCCL under torch 2.10
    delivers only the first half of a one-rank group's self-send of 1 GiB + 4 bytes -- this very test with
    elements = 2^28 + 1, bench.py records that probe as `message_of_1GiB_plus_4B_whole` -- and of 2 GiB,
    tools/rccl_selfcopy.py; 1 GiB and 768 MiB arrive whole.)"""
    if elements is None:
        elements = MAX_MSG
    world = dist.get_world_size(group)
    rank = dist.get_rank(group)
    nxt, prv = (rank + 1) % world, (rank - 1) % world
    idx = torch.arange(elements, dtype=torch.int32, device=device)
    src = idx * 747796405 + (rank + 1)                       # int32 wrap-around: a different stream per rank
    dst = torch.zeros(elements, dtype=torch.int32, device=device)
    send = [0] * world
    recv = [0] * world
    send[nxt] = elements
    recv[prv] = elements
    if src.is_cuda and dist.get_backend(group) == "gloo":
        h = torch.empty(elements, dtype=torch.int32)
        dist.all_to_all_single(h, src.cpu(), recv, send, group=group)
        dst.copy_(h)
    else:
        dist.all_to_all_single(dst, src, recv, send, group=group)
    expect = idx * 747796405 + (prv + 1)
    bad = int((dst != expect).sum().item())
    if bad:
        first = int(torch.nonzero(dst != expect)[0].item())
        raise RuntimeError(f"communicator self-test failed on rank {rank}: {bad} of {elements} words of a "
                           f"{elements * 4} byte message from rank {prv} are wrong (first at word {first}): the "
                           "collective library truncates or alters large messages on this machine")
    return elements


class ShardedSorter:
    """Sorts a key array that is sharded over the ranks of the default process group."""

    def __init__(self, keys_per_rank, pairs, device, ops=None, local_algo="lsb", slack=1.25, group=None, pipeline="msb",
                 max_imbalance=1.2, groups=1, force_exchange=False):
        """pipeline: "msb" (exchange after the first digit pass, falls back when 256 buckets leave a rank with
        more than max_imbalance x its fair share) or "partition" (12-bit group-by-destination + full local sort,
        local_algo = "lsb" | "msb").  groups: collectives the exchange of the "msb" pipeline is cut into (the
        finish of one group is enqueued behind its own collective only, so it can run while the next groups are
        still in flight).  Default 1 = one exchange, then one finish: on the one-GPU box, where a one-rank RCCL
        group turns the exchange into a device-local copy, the overlapped form is SLOWER (21.9 ms against 14.4 ms
        per 2^30 keys: copy and finish kernels fight for HBM and the collective's kernels run 3x longer), and
        whether xGMI transfers overlap better could not be measured there."""
        self.n, self.pairs, self.device, self.group = keys_per_rank, pairs, device, group
        self.pipeline, self.max_imbalance, self.groups = pipeline, max_imbalance, max(1, int(groups))
        self.stage_times = None        # set to {} to get host-synchronised stage times of the next "msb" sort
        # tests: go through the collectives even with one rank (a one-rank RCCL group on the one-GPU box exercises
        # the N > 1 code path -- list all_to_all, async work handles, stream waits -- on the real backend)
        self.force_exchange = bool(force_exchange) and dist.is_initialized()
        self.world = dist.get_world_size(group) if dist.is_initialized() else 1
        self.rank = dist.get_rank(group) if dist.is_initialized() else 0
        self.ops = ops if ops is not None else DeviceOps(device)
        self.local_algo = local_algo
        # receive capacity: balanced splits give ~n per rank; `slack` covers coarse bins / skew.
        # A heavier rank (one key value owning more than a bin can split) grows the buffers on demand.
        self.cap = int(keys_per_rank * slack) + 4096
        self._alloc(self.cap)
        self.part_k = self.ops.empty(self.n)
        self.part_v = self.ops.empty(self.n) if pairs else None
        self.last = None

    def _alloc(self, cap):
        self.cap = cap
        self.recv_k, self.alt_k = self.ops.empty(cap), self.ops.empty(cap)
        self.recv_v = self.ops.empty(cap) if self.pairs else None
        self.alt_v = self.ops.empty(cap) if self.pairs else None
        self.temp = torch.empty(self.ops.temp_bytes(max(cap, self.n), self.pairs, self.world), dtype=torch.uint8,
                                device=self.device)

    def sort(self, keys, vals=None):
        """keys (and vals): this rank's shard (device tensors of self.n elements).
        Returns (sorted_keys, sorted_vals, count): this rank's slice of the global order."""
        n, world, rank = self.n, self.world, self.rank
        if self.pipeline == "msb":
            out = self._sort_msb(keys, vals)
            if out is not None:
                return out
        hist = self.ops.histogram(keys, n, SHARD_BITS)
        exchange = world > 1 or self.force_exchange
        if exchange:
            gathered = torch.empty(world * hist.numel(), dtype=hist.dtype, device=hist.device)
            _all_gather(gathered, hist, self.group)
            hist_all = gathered.cpu().numpy().reshape(world, -1)
        else:
            hist_all = hist.cpu().numpy().reshape(1, -1)
        dest, per_rank = compute_splits(hist_all, world)
        send, recv = exchange_plan(hist_all, dest, rank, world)
        m = int(recv.sum())
        if m > self.cap:
            self._alloc(int(m * 1.1) + 4096)
        self.ops.partition(keys, vals, n, SHARD_BITS, dest, world, self.temp, self.part_k, self.part_v, bin_hist=hist)
        if exchange:
            _all_to_all(self.recv_k[:m], self.part_k[:n], recv.tolist(), send.tolist(), self.group, bound=biggest_message(hist_all, dest, world))
            if self.pairs:
                _all_to_all(self.recv_v[:m], self.part_v[:n], recv.tolist(), send.tolist(), self.group, bound=biggest_message(hist_all, dest, world))
            rk, rv = self.recv_k, self.recv_v
        else:
            rk, rv = self.part_k, self.part_v
            if m > rk.numel():
                raise RuntimeError("internal: single-rank receive exceeds shard size")
        sk, sv = self.ops.local_sort(rk, rv, m, self.alt_k if exchange else self.recv_k,
                                     self.alt_v if exchange else self.recv_v, self.temp, self.local_algo)
        self.last = dict(count=m, send=send, recv=recv, per_rank=per_rank, dest=dest, pipeline="partition")
        return sk, sv, m

    def _gather_counts(self, counts):
        if self.world > 1 or self.force_exchange:
            gathered = torch.empty(self.world * counts.numel(), dtype=counts.dtype, device=counts.device)
            _all_gather(gathered, counts, self.group)
            return gathered.cpu().numpy().reshape(self.world, -1)
        return counts.cpu().numpy().reshape(1, -1)

    def _sort_msb(self, keys, vals):
        """Exchange after the first digit pass.  Returns None when top-byte buckets cannot balance the ranks
        (every rank takes the same decision: it is a function of the gathered sizes only)."""
        n, world, rank = self.n, self.world, self.rank
        timed = self.stage_times is not None
        if timed:
            import time
            self._sync()
            t0 = time.perf_counter()
        counts = self.ops.first_pass(keys, vals, n, self.temp, self.part_k, self.part_v)
        hist_all = self._gather_counts(counts)                       # (world, 256)
        dest, per_rank = compute_splits(hist_all, world)
        total = int(hist_all.sum())
        if world > 1 and total and per_rank.max() > self.max_imbalance * total / world:
            return None
        send, recv = exchange_plan(hist_all, dest, rank, world)
        m = int(recv.sum())
        if m > self.cap:
            self._alloc(int(m * 1.1) + 4096)
        if timed:
            self._sync()
            t1 = time.perf_counter()
        if world == 1 and not self.force_exchange:
            pieces = np.where(dest[None, :] == rank, hist_all, 0)
            self.ops.finish(self.part_k, self.part_v, m, self.alt_k, self.alt_v, pieces, self.temp)
            if timed:
                self._sync()
                self.stage_times.update(first_pass_ms=(t1 - t0) * 1e3, exchange_ms=0.0, finish_ms=(time.perf_counter() - t1) * 1e3)
            self.last = dict(count=m, send=send, recv=recv, per_rank=per_rank, dest=dest, pipeline="msb", groups=1)
            return self.alt_k, self.alt_v, m
        # bucket order = key order, dest and grp are monotone: what goes to (rank r, group g) is one contiguous
        # slice of the grouped shard.  The receive buffer is group-major, source-major inside a group.
        G = 1 if timed else self.groups
        grp = group_bins(hist_all, dest, world, G)
        mine = hist_all[rank].astype(np.int64)
        my_off = np.cumsum(mine) - mine
        own = [(dest == rank) & (grp == g) for g in range(G)]
        recv_g = [np.array([int(hist_all[s][own[g]].sum()) for s in range(world)], dtype=np.int64) for g in range(G)]
        m_g = [int(x.sum()) for x in recv_g]
        goff = np.concatenate(([0], np.cumsum(m_g))).astype(np.int64)
        works = []
        for g in range(G):
            if G == 1:
                # one collective: the slices are adjacent in rank order, so the plain all_to_all_single does it
                # (in rounds when a message could exceed MAX_MSG)
                _all_to_all(self.recv_k[:m], self.part_k[:n], recv.tolist(), send.tolist(), self.group, bound=biggest_message(hist_all, dest, world))
                if self.pairs:
                    _all_to_all(self.recv_v[:m], self.part_v[:n], recv.tolist(), send.tolist(), self.group, bound=biggest_message(hist_all, dest, world))
                works.append([])
                break
            # the largest (source, destination) message of this group, known to every rank from the gathered sizes
            biggest = max(int(hist_all[src][(dest == r) & (grp == g)].sum()) for src in range(world) for r in range(world))
            rounds = max(1, -(-biggest // MAX_MSG))
            w = []
            for q in range(rounds):
                ins_k, outs_k, ins_v, outs_v = [], [], [], []
                o = int(goff[g])
                for r in range(world):
                    sel = np.nonzero((dest == r) & (grp == g))[0]
                    start = int(my_off[sel[0]]) if sel.size else 0
                    cnt = int(mine[sel].sum()) if sel.size else 0
                    c = int(recv_g[g][r])
                    s0, s1 = min(q * MAX_MSG, cnt), min((q + 1) * MAX_MSG, cnt)        # my message to r, cut at MAX_MSG
                    r0, r1 = min(q * MAX_MSG, c), min((q + 1) * MAX_MSG, c)            # r's message to me, same cuts
                    ins_k.append(self.part_k[start + s0:start + s1])
                    outs_k.append(self.recv_k[o + r0:o + r1])
                    if self.pairs:
                        ins_v.append(self.part_v[start + s0:start + s1])
                        outs_v.append(self.recv_v[o + r0:o + r1])
                    o += c
                # an error here propagates: a rank that changed its collective sequence on its own (a retry in another
                # form) would leave its peers waiting in this one
                w.append(_all_to_all_lists(outs_k, ins_k, self.group))
                if self.pairs:
                    w.append(_all_to_all_lists(outs_v, ins_v, self.group))
            works.append(w)
        if timed:
            for w in works:
                for h in w:
                    if h is not None:
                        h.wait()
            self._sync()
            t2 = time.perf_counter()
        for g in range(G):
            for h in works[g]:
                if h is not None:
                    h.wait()                      # the compute stream waits; the host does not
            if m_g[g] == 0:
                continue
            o = int(goff[g])
            pieces = np.where(own[g][None, :], hist_all, 0)     # what every source sent me in this group, per top byte
            self.ops.finish(self.recv_k[o:], self.recv_v[o:] if self.pairs else None, m_g[g], self.alt_k[o:],
                            self.alt_v[o:] if self.pairs else None, pieces, self.temp)
        if timed:
            self._sync()
            self.stage_times.update(first_pass_ms=(t1 - t0) * 1e3, exchange_ms=(t2 - t1) * 1e3,
                                    finish_ms=(time.perf_counter() - t2) * 1e3)
        self.last = dict(count=m, send=send, recv=recv, per_rank=per_rank, dest=dest, pipeline="msb", groups=G,
                         group_counts=m_g)
        return self.alt_k, self.alt_v, m

    def _sync(self):
        if self.device.type == "cuda":
            torch.cuda.synchronize(self.device)

    def verify(self, sorted_keys, count, input_checksum=None):
        """Global correctness from size-independent properties: every rank's slice is sorted, slices are
        ordered across ranks, and the key multiset (count, sum and xor of a hash) is preserved."""
        inv, s, x = self.ops.check_sorted(sorted_keys, count)
        first = int(sorted_keys[0].item()) & 0xFFFFFFFF if count else None
        last = int(sorted_keys[count - 1].item()) & 0xFFFFFFFF if count else None
        info = [None] * self.world
        mine = (inv, s, x, count, first, last)
        if self.world > 1:
            dist.all_gather_object(info, mine, group=self.group)
        else:
            info = [mine]
        ok = all(i[0] == 0 for i in info)
        prev_last = None
        for i in info:
            if i[3] == 0:
                continue
            if prev_last is not None and i[4] < prev_last:
                ok = False
            prev_last = i[5]
        total = sum(i[3] for i in info)
        ssum = sum(i[1] for i in info) % (1 << 64)
        sxor = 0
        for i in info:
            sxor ^= i[2]
        if input_checksum is not None:
            ok = ok and (total, ssum, sxor) == input_checksum
        return ok, (total, ssum, sxor)

    def input_checksum(self, keys):
        """(count, hash sum, hash xor) of the whole sharded input."""
        _, s, x = self.ops.check_sorted(keys, self.n)
        info = [None] * self.world
        if self.world > 1:
            dist.all_gather_object(info, (s, x, self.n), group=self.group)
        else:
            info = [(s, x, self.n)]
        sxor = 0
        for i in info:
            sxor ^= i[1]
        return sum(i[2] for i in info), sum(i[0] for i in info) % (1 << 64), sxor
