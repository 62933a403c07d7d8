import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 30
bitlist = [int(x) for x in sys.argv[2].split(",")] if len(sys.argv) > 2 else [1, 2, 4, 5, 6, 7, 8]
n = 1 << logn
dev = torch.device("cuda:0")
src = gs.generate_uniform_keys(n, device=dev)
a, b = torch.empty_like(src), torch.empty_like(src)
nb = gs.lib.gs_lsb_temp_bytes(n, 0)
temp = torch.empty(nb, dtype=torch.uint8, device=dev)
for bits in bitlist:
    prof = gs.KernelProfile()
    for r in range(6):
        a.copy_(src)
        dk = gs.DoubleBuffer(a, b)
        if r == 1: prof.__enter__()
        gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, 0, bits, key_type=gs.GS_KEY_U32)
    prof.__exit__()
    torch.cuda.synchronize()
    res = prof.read()
    print(bits, {k: round(v[0] / v[1], 4) for k, v in res.items()}, flush=True)
