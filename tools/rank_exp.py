"""Timing experiment: one 8-bit LSB pass on (a) uniform keys, (b) keys whose digit is constant per
aligned 64-key group (every wave register holds one digit: runs are 256-B multiples).
python tools/rank_exp.py [log2n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << logn
dev = torch.device("cuda:0")
uni = gs.generate_uniform_keys(n, device=dev)
grp = gs.generate_uniform_keys(n // 64, seed=5, device=dev).repeat_interleave(64)
grp = (uni & ~0xff) | (grp & 0xff)
a, b = torch.empty_like(uni), torch.empty_like(uni)
nb = gs.lib.gs_lsb_temp_bytes(n, 0)
temp = torch.empty(nb, dtype=torch.uint8, device=dev)
for name, src in (("uniform", uni), ("grouped64", grp)):
    prof = gs.KernelProfile()
    for r in range(6):
        a.copy_(src)
        dk = gs.DoubleBuffer(a, b)
        if r == 1: prof.__enter__()
        gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, 0, 8, key_type=gs.GS_KEY_U32)
    prof.__exit__()
    torch.cuda.synchronize()
    res = prof.read()
    print(name, os.environ.get("GS_EXP_UNSTABLE_ALL"), os.environ.get("GS_LIB_PATH", "default")[-20:], {k: round(v[0] / v[1], 4) for k, v in res.items()}, flush=True)
