"""One segmented sort of 2^28 keys in equal segments (to put under rocprofv3): python tools/seg_one.py [segment length]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
n = 1 << 28; seglen = int(sys.argv[1]) if len(sys.argv) > 1 else 256
dev = "cuda:0"
src = gs.generate_uniform_keys(n, device=dev)
a, b = torch.empty_like(src), torch.empty_like(src)
nseg = n // seglen
offs = torch.arange(0, nseg + 1, dtype=torch.int64, device=dev).mul_(seglen).to(torch.int32)
dk = gs.DoubleBuffer(a, b)
nb = gs.DeviceSegmentedRadixSort.SortKeys(None, 0, dk, n, nseg, offs[:-1], offs[1:])
temp = torch.empty(nb, dtype=torch.uint8, device=dev)
for it in range(2):
    a.copy_(src); dk.selector = 0
    gs.DeviceSegmentedRadixSort.SortKeys(temp, nb, dk, n, nseg, offs[:-1], offs[1:], key_type=gs.GS_KEY_U32)
torch.cuda.synchronize()
