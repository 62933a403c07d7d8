"""One-off check of gs_lsb_sort_wide above 2^31 elements (byte offsets beyond 2^34): i64 keys, and (i32 keys, i64
values = enumerated).  Sortedness + wrapping sum on the device, in chunks."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
dev = torch.device("cuda:0")
n = (1 << 31) + 12345
def sorted_and_sum(t):
    ok, tot, step = True, 0, 1 << 28
    for lo in range(0, n, step):
        c = t[lo:min(n, lo + step + 1)]
        ok = ok and bool((c[1:] >= c[:-1]).all())
        tot = (tot + int(t[lo:min(n, lo + step)].sum())) & ((1 << 64) - 1)
    return ok, tot
# 1. i64 keys
g = torch.Generator(device=dev); g.manual_seed(1)
keys = torch.randint(-2**62, 2**62, (n,), dtype=torch.int64, device=dev, generator=g)
_, s0 = sorted_and_sum(keys)
dk = gs.DoubleBuffer(keys, torch.empty_like(keys))
nb = gs.DeviceRadixSort.SortKeys(None, 0, dk, n)
temp = torch.empty(nb, dtype=torch.uint8, device=dev)
gs.DeviceRadixSort.SortKeys(temp, nb, dk, n)
torch.cuda.synchronize()
ok, s1 = sorted_and_sum(dk.Current())
print(f"i64 keys n={n}: sorted={ok} sum {'same' if s0 == s1 else 'DIFFERENT'}", flush=True)
good = ok and s0 == s1
del keys, dk, temp; torch.cuda.empty_cache()
# 2. i32 keys + i64 values (value = 2 * original index + 1): keys sorted, and key[original index of pair] == key
k32 = torch.randint(-2**31, 2**31 - 1, (n,), dtype=torch.int32, device=dev, generator=g)
orig = k32.clone()
vals = torch.arange(n, dtype=torch.int64, device=dev)
dk = gs.DoubleBuffer(k32, torch.empty_like(k32)); dv = gs.DoubleBuffer(vals, torch.empty_like(vals))
nb = gs.DeviceRadixSort.SortPairs(None, 0, dk, dv, n)
temp = torch.empty(nb, dtype=torch.uint8, device=dev)
gs.DeviceRadixSort.SortPairs(temp, nb, dk, dv, n)
torch.cuda.synchronize()
ko, vo = dk.Current(), dv.Current()
ok = True
step = 1 << 28
for lo in range(0, n, step):
    hi = min(n, lo + step)
    c = ko[lo:min(n, hi + 1)]
    ok = ok and bool((c[1:] >= c[:-1]).all()) and bool((orig[vo[lo:hi]] == ko[lo:hi]).all())
    # stability: among equal keys the original indices ascend
    same = ko[lo + 1:hi] == ko[lo:hi - 1]
    ok = ok and bool((vo[lo + 1:hi][same] > vo[lo:hi - 1][same]).all())
print(f"(i32, i64) pairs n={n}: sorted, values follow, stable = {ok}", flush=True)
sys.exit(0 if (good and ok) else 1)
