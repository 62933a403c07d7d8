"""Per-kernel-group times of the wide MSB sort (gs_msb_sort_wide): python tools/wide_kprof.py [log2n] [none|u32|u64]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
from gpu_sort_amd.msb import rdxsrt_unstable_sort_wide
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 28)
vk = sys.argv[2] if len(sys.argv) > 2 else "none"
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
info = torch.iinfo(torch.int64)
src = torch.randint(info.min, info.max, (n,), dtype=torch.int64, device=dev, generator=g)
k, ka = src.clone(), torch.empty_like(src)
v = None if vk == "none" else torch.arange(n, dtype=torch.int32 if vk == "u32" else torch.int64, device=dev)
va = None if v is None else torch.empty_like(v)
_, dm = rdxsrt_unstable_sort_wide(k, v, n, ka, va, key_type=gs.GS_KEY_I64)
tot = {}
for r in range(4):
    k.copy_(src)
    with gs.KernelProfile() as prof:
        rdxsrt_unstable_sort_wide(k, v, n, ka, va, key_type=gs.GS_KEY_I64, dm=dm)
        torch.cuda.synchronize()
    if r: tot = {kk: tot.get(kk, 0) + vv[0] / 3 for kk, vv in prof.read().items()}
print(os.path.basename(gs.LIB_PATH), vk, {kk: round(vv, 3) for kk, vv in tot.items()}, "sum %.3f" % sum(tot.values()), "sorted", bool((k[1:] >= k[:-1]).all()))
