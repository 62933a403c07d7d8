"""One-off check near the index-type limit: n = 2^32 - 3*8192 - 7 u32 keys (16 GiB per buffer), LSB and MSB,
keys only and pairs, verified by device-side sortedness + multiset checksum (+ enumerated values mod 2^32)."""
import os, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
dev = torch.device("cuda:0")
n = (1 << 32) - 3 * 8192 - 7
args = sys.argv[1:] or ["lsb", "msb", "lsb_pairs", "msb_pairs"]
const = "--max-const" in args          # n = 2^32 - 1 equal keys: one bucket of (almost) 2^32 keys at every MSB level
if const:
    args.remove("--max-const")
    n = (1 << 32) - 1
for algo in args:
    pairs = algo.endswith("pairs")
    keys = torch.full((n,), 7, dtype=torch.int32, device=dev) if const else gs.generate_uniform_keys(n, seed=5, device=dev)
    _, s0, x0 = gs.check_sorted(keys)
    alt = torch.empty_like(keys)
    vals = gs.generate_enumerated_values(n, device=dev) if pairs else None
    valt = torch.empty_like(keys) if pairs else None
    orig = keys.clone() if pairs else None
    torch.cuda.synchronize(); t0 = time.perf_counter()
    if algo.startswith("msb"):
        seq = gs.rdxsrt_unstable_sort(keys, vals, n, alt, valt)
        out_k, out_v = seq.sorted_keys, seq.sorted_values
    else:
        dk = gs.DoubleBuffer(keys, alt)
        if pairs:
            dv = gs.DoubleBuffer(vals, valt)
            nb = gs.DeviceRadixSort.SortPairs(None, 0, dk, dv, n)
            temp = torch.empty(nb, dtype=torch.uint8, device=dev)
            gs.DeviceRadixSort.SortPairs(temp, nb, dk, dv, n, key_type=gs.GS_KEY_U32)
            out_v = dv.Current()
        else:
            nb = gs.DeviceRadixSort.SortKeys(None, 0, dk, n)
            temp = torch.empty(nb, dtype=torch.uint8, device=dev)
            gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, key_type=gs.GS_KEY_U32)
        out_k = dk.Current()
    torch.cuda.synchronize(); ms = (time.perf_counter() - t0) * 1e3
    inv, s1, x1 = gs.check_sorted(out_k)
    ok = inv == 0 and (s1, x1) == (s0, x0)
    extra = ""
    if pairs:
        bad = gs.check_pairs_enumerated(orig, out_k, out_v)
        ok = ok and bad[0] == 0
        extra = f" pair check {bad}"
    print(f"{algo}: n={n} {ms:.1f} ms (first call, host-inclusive) inversions={inv} multiset={'same' if (s1, x1) == (s0, x0) else 'DIFFERENT'}{extra} -> {'OK' if ok else 'FAIL'}", flush=True)
    del keys, alt, vals, valt, orig, out_k
    torch.cuda.empty_cache()
    if not ok:
        sys.exit(1)
