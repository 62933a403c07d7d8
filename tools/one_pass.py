"""python tools/one_pass.py LOG2N BEGIN_BIT END_BIT [reps] -- LSB sort restricted to a bit range (profiling aid)"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
logn, bb, eb = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3])
reps = int(sys.argv[4]) if len(sys.argv) > 4 else 2
n = 1 << logn
dev = torch.device("cuda:0")
src = gs.generate_uniform_keys(n, device=dev)
a, b = torch.empty_like(src), torch.empty_like(src)
nb = gs.lib.gs_lsb_temp_bytes(n, 0)
temp = torch.empty(nb, dtype=torch.uint8, device=dev)
for r in range(reps):
    a.copy_(src)
    dk = gs.DoubleBuffer(a, b)
    gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, bb, eb, key_type=gs.GS_KEY_U32)
torch.cuda.synchronize()
