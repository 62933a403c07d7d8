"""Per-level census of an MSB sort (buckets, tiles, local-sort tasks per class, flagged tasks) read back from
the workspace after the call.  python tools/msb_census.py [log2n] [uniform|zipf] [pairs]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_sort_amd as gs
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dist = sys.argv[2] if len(sys.argv) > 2 else "uniform"
pairs = len(sys.argv) > 3 and sys.argv[3] == "pairs"
n = 1 << logn
dev = "cuda:0"
gen = gs.generate_uniform_keys if dist == "uniform" else gs.generate_zipf_keys
a = gen(n, device=dev); b = torch.empty_like(a)
va = gs.generate_enumerated_values(n, device=dev) if pairs else None
vb = torch.empty_like(a) if pairs else None
nb = gs.lib.gs_msb_temp_bytes(n, int(pairs))
temp = torch.zeros(nb, dtype=torch.uint8, device=dev)
with gs.KernelProfile() as prof:
    gs.rdxsrt_unstable_sort(a, va, n, b, vb, pre_allocated_dm=temp)
    torch.cuda.synchronize()
print({k: round(v[0], 3) for k, v in prof.read().items()})
from gpu_sort_amd.msb import msb_census, msb_algorithmic_bytes
census = msb_census(temp, n, pairs)
for L, c in enumerate(census):
    print(f"level {L}: buckets {c['buckets']:7d}  tiles {c['tiles']:8d}  keys {c['keys']:11d} ({c['keys'] / n:6.1%})  "
          f"heavy-hitter buckets {c['pivot_buckets']} ({c['pivot_keys'] / n:6.1%} of the keys)  "
          f"tasks per class {c['tasks']} ({c['task_keys'] / n:6.1%} of the keys)  "
          f"tasks left to the general plan: {'yes' if c['flagged'] else 'no'}")
print("algorithmic bytes:", msb_algorithmic_bytes(census, n, pairs))
