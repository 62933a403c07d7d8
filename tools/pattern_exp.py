"""Does the LSB downsweep slow down when its input is already grouped by the top byte (the
situation of an MSB level-1 partition: every resident tile writes into one 16 MiB window)?
python tools/pattern_exp.py [log2n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << logn
dev = torch.device("cuda:0")
src = gs.generate_uniform_keys(n, device=dev)
a, b = torch.empty_like(src), torch.empty_like(src)
nb = gs.lib.gs_lsb_temp_bytes(n, 0)
temp = torch.empty(nb, dtype=torch.uint8, device=dev)
def one(keys_in, lo, hi, tag):
    for r in range(3):
        a.copy_(keys_in)
        dk = gs.DoubleBuffer(a, b)
        torch.cuda.synchronize()
        with gs.KernelProfile() as prof:
            gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, lo, hi, key_type=gs.GS_KEY_U32)
            torch.cuda.synchronize()
    print(tag, {k: round(v[0] / v[1], 4) for k, v in prof.read().items()}, flush=True)
    return dk.Current().clone()
one(src, 16, 24, "uniform input, pass on bits 16..23      ")
g = one(src, 24, 32, "uniform input, pass on bits 24..31      ")
one(g, 16, 24, "grouped by top byte, pass on bits 16..23")
g2 = one(g, 16, 24, "same again                              ")
one(g2, 8, 16, "grouped by top 2 bytes, bits 8..15      ")
