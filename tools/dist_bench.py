"""LSB and MSB sort times at 2^28 u32 keys for adversarial key distributions (constant, few values, sorted,
reversed, half all-ones, 16-bit range, one hot top byte, AND-reduced entropy) next to uniform keys."""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
dev = torch.device("cuda:0")
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << logn
g = torch.Generator(device=dev); g.manual_seed(1)
def rnd():
    return torch.randint(-2**31, 2**31, (n,), dtype=torch.int64, device=dev, generator=g).to(torch.int32)
def make(kind):
    if kind == "uniform": return rnd()
    if kind == "const": return torch.full((n,), 123456789, dtype=torch.int32, device=dev)
    if kind == "few4": return rnd()[:4][torch.randint(0, 4, (n,), device=dev, generator=g)]
    if kind == "sorted": return torch.sort(rnd())[0]
    if kind == "reverse": return torch.sort(rnd(), descending=True)[0]
    if kind == "ones_half":
        k = rnd(); k[torch.rand(n, device=dev, generator=g) < 0.5] = -1; return k
    if kind == "low16": return rnd() & 0xFFFF
    if kind == "hot_top_byte": return (rnd() & 0x00FFFFFF) | (0x5A << 24)
    if kind.startswith("and"):
        k = rnd()
        for _ in range(int(kind[3:])): k &= rnd()
        return k
a, b = torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)
nb = max(gs.lib.gs_lsb_temp_bytes(n, 0), gs.lib.gs_msb_temp_bytes(n, 0))
temp = torch.empty(nb, dtype=torch.uint8, device=dev)
def timed(fn, src):
    ts = []
    for _ in range(6):
        a.copy_(src); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); out = fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    inv = gs.check_sorted(out)[0]
    return sorted(ts[1:])[2], inv
def lsb():
    dk = gs.DoubleBuffer(a, b); gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, key_type=gs.GS_KEY_U32); return dk.Current()
def msb():
    return gs.rdxsrt_unstable_sort(a, None, n, b, None, pre_allocated_dm=temp, synchronize=False).sorted_keys
print(f"2^{logn} u32 keys: distribution | LSB ms | MSB ms")
for kind in ("uniform", "const", "few4", "sorted", "reverse", "ones_half", "low16", "hot_top_byte", "and2", "and5", "and10"):
    src = make(kind).contiguous()
    tl, il = timed(lsb, src); tm, im = timed(msb, src)
    print(f"{kind:14s} | {tl:8.3f} | {tm:8.3f} {'' if il == 0 and im == 0 else 'NOT SORTED'}", flush=True)
