"""Randomised parity campaign on the GPU: LSB / MSB / segmented sorts of random sizes, key types, bit ranges,
directions and key distributions, and the 64-bit configurations of the wide LSB sort, against torch's stable sort on the same device (an implementation independent
of both the library and the oracle).  usage: python tools/fuzz.py [iterations] [seed]
Exit code 1 and the failing case on the first mismatch."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_sort_amd as gs

iters = int(sys.argv[1]) if len(sys.argv) > 1 else 500
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
dev = torch.device("cuda:0")
KT = {"u32": gs.GS_KEY_U32, "i32": gs.GS_KEY_I32, "f32": gs.GS_KEY_F32}
EDGES = [0, 1, 2, 63, 64, 65, 255, 256, 257, 2047, 2048, 2049, 4607, 4608, 4609, 6911, 6912, 6913, 8191, 8192, 8193,
         9215, 9216, 9217, 16383, 16384, 17407, 17408, 17409, 65535, 65536, 65537, 8192 * 9 - 1, 8192 * 17 + 5,
         1 << 18, (1 << 20) + 3, 8192 * 2048, 8192 * 2048 + 1, (1 << 22) + 12345]


BIG = os.environ.get("FUZZ_BIG") == "1"      # few iterations, sizes up to 2^28 + (32-bit byte offsets, many tiles)


def pick_n():
    if BIG:
        return int(rng.choice([1 << 26, (1 << 27) + 12345, (1 << 28) + 7, 200_000_003, 100_000_000 + int(rng.integers(0, 9000))]))
    r = rng.random()
    if r < 0.35:
        return int(rng.choice(EDGES)) + int(rng.integers(-3, 4)) * int(rng.random() < 0.3)
    if r < 0.7:
        return int(rng.integers(0, 40000))
    if r < 0.95:
        return int(rng.integers(40000, 3_000_000))
    return int(rng.integers(3_000_000, 20_000_000))


def make_keys(n, kind):
    g = torch.Generator(device=dev); g.manual_seed(int(rng.integers(0, 2**31)))
    def rnd():
        return torch.randint(-2**31, 2**31, (n,), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
    if kind == "uniform":
        k = rnd()
    elif kind.startswith("and"):
        k = rnd()
        for _ in range(int(kind[3:])):
            k &= rnd()
    elif kind == "few":
        vals = rnd()[:max(1, int(rng.integers(1, 9)))] if n else rnd()
        k = vals[torch.randint(0, max(1, vals.numel()), (n,), device=dev, generator=g)] if n else rnd()
    elif kind == "const":
        k = torch.full((n,), int(rng.integers(-2**31, 2**31)), dtype=torch.int32, device=dev)
    elif kind == "ones_heavy":
        k = rnd()
        k[torch.rand(n, device=dev, generator=g) < 0.5] = -1
    elif kind == "sorted":
        k = torch.sort(rnd())[0]
    elif kind == "reverse":
        k = torch.sort(rnd(), descending=True)[0]
    elif kind == "zipf":
        k = gs.generate_zipf_keys(n, seed=int(rng.integers(0, 1 << 30)), device=dev) if n else rnd()
    elif kind == "clustered":
        # buckets of two fixed top bytes holding a few thousand keys with few distinct 16-bit tails each (the MSB local sorts' plan
        # for few distinct values), one of them far too large for a local sort (so that the level shows skew)
        if n == 0:
            return rnd()
        P = max(1, n // int(rng.integers(3000, 16000)))
        pref = torch.randint(0, 1 << 16, (P,), device=dev, generator=g)
        which = torch.randint(0, P, (n,), device=dev, generator=g)
        which[torch.rand(n, device=dev, generator=g) < 0.15] = 0
        D = int(rng.choice([1, 2, 7, 100, 255, 256, 1000, 2047, 2048, 2049, 5000]))
        tail = (torch.randint(0, D, (n,), device=dev, generator=g) * 40503 + which * 977) & 0xFFFF
        k = ((pref[which] << 16) | tail).to(torch.int64)
        k = torch.where(k >= 2**31, k - 2**32, k).to(torch.int32)
    elif kind == "low_bytes":
        k = rnd() & 0xFFFF
    else:   # hot top byte
        k = (rnd() & 0x00FFFFFF) | (int(rng.integers(0, 128)) << 24)
    return k.contiguous()


def twiddled(keys, kt):
    """order-preserving unsigned image of the keys, as int64"""
    b = keys.to(torch.int64) & 0xFFFFFFFF
    if kt == "u32":
        return b
    if kt == "i32":
        return b ^ 0x80000000
    neg = keys < 0                                  # f32 bit patterns: flip all bits of negatives, the sign of the rest
    return torch.where(neg, b ^ 0xFFFFFFFF, b ^ 0x80000000)


def ref_perm(keys, kt, begin, end, desc, seg_id=None):
    width = end - begin
    d = (twiddled(keys, kt) >> begin) & ((1 << width) - 1) if width > 0 else torch.zeros_like(keys, dtype=torch.int64)
    if desc:
        d = ((1 << width) - 1) - d if width > 0 else d
    if seg_id is not None:
        d = d + (seg_id << 32)
    return torch.sort(d, stable=True)[1]


def fail(msg, **case):
    print("MISMATCH:", msg, case, flush=True)
    sys.exit(1)


MIN64 = -(1 << 63)


def wide_case(it):
    """gs_lsb_sort_wide: 64-bit keys (u64 / i64 / f64) with no, 32-bit or 64-bit values, or i32 keys with i64 values"""
    n = min(max(0, pick_n()), 4_000_000)
    g = torch.Generator(device=dev); g.manual_seed(int(rng.integers(0, 2**31)))
    combo = str(rng.choice(["k64", "k64v32", "k64v64", "k32v64"]))
    kt = str(rng.choice(["u64", "i64", "f64"])) if combo != "k32v64" else str(rng.choice(["u32", "i32"]))
    kbits = 64 if combo != "k32v64" else 32
    def rnd64():
        hi = torch.randint(-2**31, 2**31, (n,), dtype=torch.int64, device=dev, generator=g)
        lo = torch.randint(0, 2**32, (n,), dtype=torch.int64, device=dev, generator=g)
        return (hi << 32) | lo
    if kbits == 64:
        keys = rnd64()
        for _ in range(int(rng.choice([0, 0, 1, 3, 8]))):
            keys &= rnd64()
        if rng.random() < 0.15:
            keys = keys & 0xFFFF                                  # many duplicates, constant high digits
        if kt == "f64" and n:
            f = keys.view(torch.float64)
            keys = torch.where(torch.isnan(f), torch.zeros_like(keys), keys)
    else:
        keys = make_keys(n, str(rng.choice(KINDS)))
    keys = keys.contiguous()
    vdt = {"k64": None, "k64v32": torch.int32, "k64v64": torch.int64, "k32v64": torch.int64}[combo]
    vals = None if vdt is None else (torch.arange(n, device=dev).to(vdt) * (3 if vdt == torch.int64 else 1))
    desc = bool(rng.random() < 0.4)
    if rng.random() < 0.5:
        begin, end = 0, kbits
    else:
        begin = int(rng.integers(0, kbits)); end = int(rng.integers(begin, kbits + 1))
    case = dict(it=it, algo="wide", combo=combo, n=n, kt=kt, desc=desc, begin=begin, end=end, seed=seed)
    ktid = {"u32": gs.GS_KEY_U32, "i32": gs.GS_KEY_I32, "u64": 3, "i64": 4, "f64": 5}[kt]
    dk = gs.DoubleBuffer(keys.clone(), torch.empty_like(keys))
    dv = gs.DoubleBuffer(vals.clone(), torch.empty_like(vals)) if vals is not None else None
    R = gs.DeviceRadixSort
    fn = (R.SortPairsDescending if desc else R.SortPairs) if dv is not None else (R.SortKeysDescending if desc else R.SortKeys)
    args = (dk, dv, n) if dv is not None else (dk, n)
    nb = fn(None, 0, *args)
    temp = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
    fn(temp, nb, *args, begin, end, key_type=ktid)
    if n == 0:
        return
    # order-preserving image of the keys whose SIGNED order is the key order
    if kbits == 32:
        img = twiddled(keys, kt)                                  # 0 .. 2^32-1 as int64
        U = img
    else:
        b = keys
        if kt == "u64":
            U = b
        elif kt == "i64":
            U = b ^ MIN64
        else:
            U = torch.where(b < 0, ~b, b ^ MIN64)
        img = U ^ MIN64
    width = end - begin
    if width == kbits:
        d = img
        if desc:
            d = ~d if kbits == 64 else ((1 << 32) - 1) - d
    elif width == 0:
        d = torch.zeros_like(img)
    else:
        d = (U >> begin) & ((1 << width) - 1)
        if desc:
            d = ((1 << width) - 1) - d
    perm = torch.sort(d, stable=True)[1]
    if not torch.equal(dk.Current()[:n], keys[perm]):
        fail("wide keys", **case)
    if dv is not None and not torch.equal(dv.Current()[:n], vals[perm]):
        fail("wide values (stability)", **case)
    # the same keys through the MSB path's wide kernel set (gs_msb_sort_wide: all key bits, ascending, unstable: values
    # may be permuted inside runs of equal keys only)
    from gpu_sort_amd.msb import rdxsrt_unstable_sort_wide
    mk, mv = keys.clone(), (vals.clone() if vals is not None else None)
    seq, _ = rdxsrt_unstable_sort_wide(mk, mv, n, torch.empty_like(mk), torch.empty_like(mv) if mv is not None else None, key_type=ktid)
    full = torch.sort(img, stable=True)[1]
    if not torch.equal(seq.sorted_keys[:n], keys[full]):
        fail("wide MSB keys", **case)
    if mv is not None:
        def canon(order_img, v):      # values sorted inside every run of equal keys
            i1 = torch.sort(v.to(torch.int64), stable=True)[1]
            i2 = torch.sort(order_img[i1], stable=True)[1]
            return v[i1][i2]
        if not torch.equal(canon(img[full], seq.sorted_values[:n]), canon(img[full], vals[full])):
            fail("wide MSB values", **case)


def any_case(it):
    """gs_lsb_sort_any: 8- / 16-bit keys (and 32-bit keys with odd value sizes) with values of 0 / 1 / 2 / 4 / 8 / 16 bytes, against
    torch's stable sort of the keys' order-preserving images"""
    n = min(max(0, pick_n()), 2_000_000)
    kt = str(rng.choice(["u8", "i8", "i16", "u16", "bool", "u32", "i32"]))
    bits = {"u8": 8, "i8": 8, "bool": 8, "i16": 16, "u16": 16, "u32": 32, "i32": 32}[kt]
    gtype = {"u8": gs.GS_KEY_U8, "bool": gs.GS_KEY_U8, "i8": gs.GS_KEY_I8, "i16": gs.GS_KEY_I16, "u16": gs.GS_KEY_U16,
             "u32": gs.GS_KEY_U32, "i32": gs.GS_KEY_I32}[kt]
    vb = int(rng.choice([0, 1, 2, 4, 8, 16])) if bits < 32 else int(rng.choice([1, 2, 16]))
    g = torch.Generator(device=dev); g.manual_seed(int(rng.integers(0, 2**31)))
    raw = torch.randint(0, 1 << 31, (max(n, 1),), device=dev, generator=g, dtype=torch.int64)[:n]
    if rng.random() < 0.4:
        raw = raw & torch.randint(0, 1 << 31, (max(n, 1),), device=dev, generator=g, dtype=torch.int64)[:n]
    img = raw & ((1 << bits) - 1)                       # unsigned bit pattern of the key
    if kt == "bool":
        img = img & 1
    tdt = {8: torch.uint8, 16: torch.int16, 32: torch.int32}[bits]
    wrap = torch.where(img >= (1 << (bits - 1)), img - (1 << bits), img) if bits > 8 else img
    keys = wrap.to(tdt).contiguous()
    if kt == "i8":
        keys = keys.view(torch.int8)
    elif kt == "bool":
        keys = keys.view(torch.bool)
    order_img = img ^ (1 << (bits - 1)) if kt in ("i8", "i16", "i32") else img
    desc = bool(rng.random() < 0.4)
    begin, end = (0, bits) if rng.random() < 0.5 else (lambda b: (b, int(rng.integers(b, bits + 1))))(int(rng.integers(0, bits)))
    width = end - begin
    d = (order_img >> begin) & ((1 << width) - 1) if width > 0 else torch.zeros_like(order_img)
    if desc and width > 0:
        d = ((1 << width) - 1) - d
    perm = torch.sort(d, stable=True)[1]
    vals = None
    if vb:
        vals = {1: lambda: torch.randint(0, 256, (n,), device=dev, generator=g).to(torch.uint8),
                2: lambda: torch.randint(-2**15, 2**15, (n,), device=dev, generator=g).to(torch.int16),
                4: lambda: torch.arange(n, dtype=torch.int32, device=dev),
                8: lambda: torch.arange(n, dtype=torch.int64, device=dev) * 0x100000001,
                16: lambda: torch.randint(-2**31, 2**31 - 1, (n, 4), device=dev, generator=g).to(torch.int32)}[vb]()
    case = dict(it=it, algo="any", n=n, kt=kt, vb=vb, desc=desc, begin=begin, end=end, seed=seed)
    dk = gs.DoubleBuffer(keys.clone(), torch.zeros_like(keys))
    dv = gs.DoubleBuffer(vals.clone(), torch.zeros_like(vals)) if vb else None
    R = gs.DeviceRadixSort
    fn = (R.SortPairsDescending if desc else R.SortPairs) if vb else (R.SortKeysDescending if desc else R.SortKeys)
    args = (dk, dv, n) if vb else (dk, n)
    nb = fn(None, 0, *args, key_type=gtype)
    temp = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
    fn(temp, nb, *args, begin, end, key_type=gtype)
    if n == 0:
        return
    if not torch.equal(dk.Current().view(torch.uint8), keys[perm].contiguous().view(torch.uint8)):
        fail("any keys", **case)
    if vb and not torch.equal(dv.Current(), vals[perm]):
        fail("any values (stability)", **case)


def shard_case(it):
    """The multi-GPU pipeline's kernels with every rank emulated in this process: gs_msb_first_pass_u32 on W random
    shards, the host's split / group maps, then for one random rank the receive buffer it would get (group-major,
    source-major, pieces where the exchange would put them) and gs_msb_finish_u32 per group."""
    from gpu_sort_amd import sharded
    W = int(rng.integers(1, 9)); G = int(rng.integers(1, 5)); pairs = bool(rng.random() < 0.5)
    sizes = [int(rng.choice([0, 1, 777, 8192, 50000, 200000, 400003])) for _ in range(W)]
    kind = str(rng.choice(KINDS))
    case = dict(it=it, algo="shard", W=W, G=G, pairs=pairs, sizes=sizes, kind=kind, seed=seed)
    ops = sharded.DeviceOps(dev)
    nmax = max(max(sizes), 1)
    temp = torch.empty(ops.temp_bytes(sum(sizes) + 4096, pairs, W), dtype=torch.uint8, device=dev)
    keys, vals, part_k, part_v, hist = [], [], [], [], []
    base = 0
    for n_s in sizes:
        k = make_keys(n_s, kind) if n_s else torch.empty(0, dtype=torch.int32, device=dev)
        v = (torch.arange(n_s, device=dev) + base).to(torch.int32)
        base += n_s
        pk, pv = ops.empty(n_s), (ops.empty(n_s) if pairs else None)
        if n_s:
            c = ops.first_pass(k, v if pairs else None, n_s, temp, pk, pv)
            hist.append(c.cpu().numpy())
        else:
            hist.append(np.zeros(256, np.int64))
        keys.append(k); vals.append(v); part_k.append(pk); part_v.append(pv)
    hist_all = np.stack(hist).astype(np.int64)
    if hist_all.sum() == 0:
        return
    dest, per_rank = sharded.compute_splits(hist_all, W)
    grp = sharded.group_bins(hist_all, dest, W, G)
    r = int(rng.integers(0, W))
    m = int(per_rank[r])
    if m == 0:
        return
    recv_k, out_k = ops.empty(m), ops.empty(m)
    recv_v, out_v = (ops.empty(m), ops.empty(m)) if pairs else (None, None)
    off = np.cumsum(hist_all, axis=1) - hist_all                     # offset of bin b in source s's grouped shard
    o = 0
    for g in range(G):
        own = (dest == r) & (grp == g)
        o0 = o
        for sidx in range(W):
            sel = np.nonzero(own)[0]
            cnt = int(hist_all[sidx][own].sum())
            if cnt:
                first = sel[np.nonzero(hist_all[sidx][sel])[0][0]]
                st = int(off[sidx][first])
                recv_k[o:o + cnt] = part_k[sidx][st:st + cnt]
                if pairs:
                    recv_v[o:o + cnt] = part_v[sidx][st:st + cnt]
                o += cnt
        if o > o0:
            pieces = np.where(own[None, :], hist_all, 0)
            ops.finish(recv_k[o0:], recv_v[o0:] if pairs else None, o - o0, out_k[o0:], out_v[o0:] if pairs else None, pieces, temp)
    allk = torch.cat(keys); allv = torch.cat(vals)
    mine = torch.from_numpy(dest == r).to(dev)[(allk.to(torch.int64) >> 24) & 0xFF]
    exp = torch.sort(allk[mine].to(torch.int64) & 0xFFFFFFFF)[0]
    if not torch.equal(out_k[:m].to(torch.int64) & 0xFFFFFFFF, exp):
        fail("sharded finish keys", **case)
    if pairs:
        got = ((out_k[:m].to(torch.int64) & 0xFFFFFFFF) << 32) | (out_v[:m].to(torch.int64) & 0xFFFFFFFF)
        want = ((allk[mine].to(torch.int64) & 0xFFFFFFFF) << 32) | (allv[mine].to(torch.int64) & 0xFFFFFFFF)
        if not torch.equal(torch.sort(got)[0], torch.sort(want)[0]):
            fail("sharded finish values", **case)


KINDS = ["uniform", "and1", "and3", "and6", "and10", "few", "const", "ones_heavy", "sorted", "reverse", "low_bytes", "hot", "zipf", "clustered"]
counts = {}
for it in range(iters):
    algo = str(rng.choice(["lsb", "lsb", "msb", "msb", "seg", "wide", "any"]))
    if algo == "any":
        counts[algo] = counts.get(algo, 0) + 1
        any_case(it)
        continue
    if algo == "wide":
        counts[algo] = counts.get(algo, 0) + 1
        wide_case(it)
        continue
    if rng.random() < 0.08:
        counts["shard"] = counts.get("shard", 0) + 1
        shard_case(it)
        continue
    pairs = bool(rng.random() < 0.5)
    n = max(0, pick_n())
    kind = str(rng.choice(KINDS))
    kt = str(rng.choice(["u32", "u32", "i32", "f32"]))
    keys = make_keys(n, kind)
    if kt == "f32" and n:                            # no NaNs (the reference's tests generate none): clear them
        f = keys.view(torch.float32)
        keys = torch.where(torch.isnan(f), torch.zeros_like(keys), keys).contiguous()
    vals = torch.randperm(n, device=dev).to(torch.int32) if (pairs and n and rng.random() < 0.5) else torch.arange(n, dtype=torch.int32, device=dev)
    case = dict(it=it, algo=algo, pairs=pairs, n=n, kind=kind, kt=kt, seed=seed)
    counts[algo] = counts.get(algo, 0) + 1
    a, b = keys.clone(), torch.full_like(keys, 0x3C3C3C3C)
    va, vb = (vals.clone(), torch.full_like(vals, 0x3C3C3C3C)) if pairs else (None, None)
    if algo == "msb":
        seq = gs.rdxsrt_unstable_sort(a, va, n, b, vb, key_type=KT[kt])
        if n == 0:
            continue
        perm = ref_perm(keys, kt, 0, 32, False)
        if not torch.equal(seq.sorted_keys[:n], keys[perm]):
            fail("msb keys", **case)
        if pairs:                                   # unstable: compare (key, value) pairs as multisets per key
            got = (twiddled(seq.sorted_keys[:n], kt) << 32) | (seq.sorted_values[:n].to(torch.int64) & 0xFFFFFFFF)
            exp = (twiddled(keys, kt) << 32) | (vals.to(torch.int64) & 0xFFFFFFFF)
            if not torch.equal(torch.sort(got)[0], torch.sort(exp)[0]):
                fail("msb values", **case)
        continue
    desc = bool(rng.random() < 0.4)
    if rng.random() < 0.5:
        begin, end = 0, 32
    else:
        begin = int(rng.integers(0, 32)); end = int(rng.integers(begin, 33))
    case.update(desc=desc, begin=begin, end=end)
    dk = gs.DoubleBuffer(a, b)
    dv = gs.DoubleBuffer(va, vb) if pairs else None
    if algo == "lsb" and rng.random() < 0.3:
        # the plain-pointer overloads (CUB's no-overwrite mode): input untouched, result in *_out
        R = gs.DeviceRadixSort
        counts["lsb_copy"] = counts.get("lsb_copy", 0) + 1
        ko, vo = torch.full_like(keys, 0x3C3C3C3C), (torch.full_like(vals, 0x3C3C3C3C) if pairs else None)
        if pairs:
            nb = R.SortPairsCopy(None, 0, a, ko, va, vo, n)
            temp = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
            R.SortPairsCopy(temp, nb, a, ko, va, vo, n, begin, end, key_type=KT[kt], descending=desc)
        else:
            nb = R.SortKeysCopy(None, 0, a, ko, n)
            temp = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
            R.SortKeysCopy(temp, nb, a, ko, n, begin, end, key_type=KT[kt], descending=desc)
        if n == 0:
            continue
        perm = ref_perm(keys, kt, begin, end, desc)
        if not torch.equal(a, keys) or (pairs and not torch.equal(va, vals)):
            fail("lsb copy: input modified", **case)
        if not torch.equal(ko[:n], keys[perm]) or (pairs and not torch.equal(vo[:n], vals[perm])):
            fail("lsb copy result", **case)
        continue
    if algo == "lsb":
        R = gs.DeviceRadixSort
        fn = (R.SortPairsDescending if desc else R.SortPairs) if pairs else (R.SortKeysDescending if desc else R.SortKeys)
        args = (dk, dv, n) if pairs else (dk, n)
        nb = fn(None, 0, *args)
        temp = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
        fn(temp, nb, *args, begin, end, key_type=KT[kt])
        if n == 0:
            continue
        if os.environ.get("FUZZ_SELFTEST") and n > 2:       # negative control: the campaign must notice this
            dk.Current()[n // 2] ^= 1
        perm = ref_perm(keys, kt, begin, end, desc)
        if not torch.equal(dk.Current()[:n], keys[perm]):
            fail("lsb keys", **case)
        if pairs and not torch.equal(dv.Current()[:n], vals[perm]):
            fail("lsb values (stability)", **case)
        continue
    # segmented
    if n == 0 or n >= (1 << 31):
        continue
    nseg = int(rng.choice([1, 2, 7, 100, 1000, max(1, n // 50)]))
    cuts = np.sort(rng.integers(0, n + 1, size=nseg - 1)) if nseg > 1 else np.zeros(0, np.int64)
    offs = np.concatenate([[0], cuts, [n]]).astype(np.int64)
    begin_offs, end_offs = offs[:-1].copy(), offs[1:].copy()
    if rng.random() < 0.3 and nseg > 2:              # leave gaps: shrink some segments
        shrink = rng.random(nseg) < 0.3
        end_offs = np.where(shrink, begin_offs + (end_offs - begin_offs) // 2, end_offs)
    case.update(nseg=nseg)
    ob = torch.from_numpy(begin_offs.astype(np.int32)).to(dev); oe = torch.from_numpy(end_offs.astype(np.int32)).to(dev)
    S = gs.DeviceSegmentedRadixSort
    fn = (S.SortPairsDescending if desc else S.SortPairs) if pairs else (S.SortKeysDescending if desc else S.SortKeys)
    args = (dk, dv, n, nseg, ob, oe) if pairs else (dk, n, nseg, ob, oe)
    nb = fn(None, 0, *args)
    temp = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
    fn(temp, nb, *args, begin, end, key_type=KT[kt])
    # expected: positions inside a segment take the segment's stable order, the others are not part of the result
    seg_id = torch.full((n,), -1, dtype=torch.int64, device=dev)
    inside = torch.zeros(n, dtype=torch.bool, device=dev)
    lens = torch.from_numpy(end_offs - begin_offs).to(dev)
    starts = torch.from_numpy(begin_offs).to(dev)
    ids = torch.repeat_interleave(torch.arange(nseg, device=dev), lens)
    pos = torch.repeat_interleave(starts, lens) + (torch.arange(int(lens.sum()), device=dev) - torch.repeat_interleave(torch.cumsum(lens, 0) - lens, lens))
    seg_id[pos] = ids; inside[pos] = True
    width = end - begin
    d = (twiddled(keys, kt) >> begin) & ((1 << width) - 1) if width > 0 else torch.zeros(n, dtype=torch.int64, device=dev)
    if desc and width > 0:
        d = ((1 << width) - 1) - d
    comp = torch.where(inside, (seg_id << 32) + d, torch.full_like(d, 1 << 62))
    perm = torch.sort(comp, stable=True)[1][: int(inside.sum())]        # elements of segment 0 in order, then 1, ...
    order = torch.sort(torch.where(inside, seg_id, torch.full_like(seg_id, 1 << 40)), stable=True)[1][: int(inside.sum())]
    exp_k = keys.clone(); exp_k[order] = keys[perm]
    got_k = dk.Current()[:n]
    if not torch.equal(got_k[inside], exp_k[inside]):
        fail("segmented keys", **case)
    if pairs:
        exp_v = vals.clone(); exp_v[order] = vals[perm]
        if not torch.equal(dv.Current()[:n][inside], exp_v[inside]):
            fail("segmented values", **case)
    if (it + 1) % 100 == 0:
        print(f"{it + 1} cases ok {counts}", flush=True)
torch.cuda.synchronize()
print(f"ALL {iters} CASES OK (seed {seed}) {counts}", flush=True)
