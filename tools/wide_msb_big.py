"""One-off check of gs_msb_sort_wide beyond 4 GiB of keys (byte offsets > 2^32): 2^30 + 12345 i64 keys, and
2^29 + 7 (i64, i64) pairs: sorted on the device, multiset (sum) kept, values follow their keys."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
from gpu_sort_amd.msb import rdxsrt_unstable_sort_wide
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(5)
n = (1 << 30) + 12345
k = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=dev, generator=g)
ksum = int(k.sum().item())
seq, _ = rdxsrt_unstable_sort_wide(k, None, n, torch.empty_like(k), None, key_type=gs.GS_KEY_I64)
ok1 = bool((seq.sorted_keys[1:] >= seq.sorted_keys[:-1]).all()) and int(seq.sorted_keys.sum().item()) == ksum
print("2^30+12345 i64 keys:", "ok" if ok1 else "FAIL")
del k, seq
torch.cuda.empty_cache()
n = (1 << 29) + 7
k = torch.randint(-2**63, 2**63 - 1, (n,), dtype=torch.int64, device=dev, generator=g)
orig = k.clone()
v = torch.arange(n, dtype=torch.int64, device=dev)
seq, _ = rdxsrt_unstable_sort_wide(k, v, n, torch.empty_like(k), torch.empty_like(v), key_type=gs.GS_KEY_I64)
sk, sv = seq.sorted_keys, seq.sorted_values
ok2 = bool((sk[1:] >= sk[:-1]).all()) and bool((orig[sv] == sk).all()) and int(sv.sum().item()) == n * (n - 1) // 2
print("2^29+7 (i64,i64) pairs:", "ok" if ok2 else "FAIL")
sys.exit(0 if ok1 and ok2 else 1)
