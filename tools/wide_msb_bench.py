"""python tools/wide_msb_bench.py [log2n] -- the wide MSB sort (gs_msb_sort_wide) next to the wide LSB sort (gs_lsb_sort_wide)
for 64-bit keys without / with 32-bit / with 64-bit values and 32-bit keys with 64-bit values, uniform keys."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
from gpu_sort_amd.msb import rdxsrt_unstable_sort_wide
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << logn
dev = torch.device("cuda:0")
g = torch.Generator(device=dev); g.manual_seed(0)
def timed(f, reps=5):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(); f(); b.record(); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best
for kd, vd in ((torch.int64, None), (torch.int64, torch.int32), (torch.int64, torch.int64), (torch.int32, torch.int64)):
    info = torch.iinfo(kd)
    src = torch.randint(info.min, info.max, (n,), dtype=kd, device=dev, generator=g)
    k, ka = src.clone(), torch.empty_like(src)
    v = torch.arange(n, dtype=vd, device=dev) if vd else None
    va = torch.empty_like(v) if vd is not None else None
    kt = gs.GS_KEY_I64 if kd == torch.int64 else gs.GS_KEY_I32
    _, dm = rdxsrt_unstable_sort_wide(k, v, n, ka, va, key_type=kt)
    def msb():
        k.copy_(src)
        rdxsrt_unstable_sort_wide(k, v, n, ka, va, key_type=kt, dm=dm)
    tcopy = timed(lambda: k.copy_(src))
    tm = timed(msb) - tcopy
    ok = bool((k[1:] >= k[:-1]).all())
    def lsb():
        k.copy_(src)
        dk = gs.DoubleBuffer(k, ka)
        if vd is None: gs.DeviceRadixSort.SortKeys(temp, nb, dk, n)
        else: gs.DeviceRadixSort.SortPairs(temp, nb, dk, gs.DoubleBuffer(v, va), n)
    nb = gs.lib.gs_lsb_wide_temp_bytes(n, k.element_size(), v.element_size() if vd else 0)
    temp = torch.empty(max(nb, 1), dtype=torch.uint8, device=dev)
    try:
        tl = timed(lsb) - tcopy
    except Exception as e:
        tl = float("nan")
    print(f"keys {kd} vals {vd}: n=2^{logn}  MSB wide {tm:7.2f} ms ({n / tm / 1e6:6.1f} G/s)   LSB wide {tl:7.2f} ms   sorted={ok}")
