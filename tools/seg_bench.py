"""Timing of the segmented sort for a few segmentations of 2^log2n uniform keys (tools only).
python tools/seg_bench.py [log2n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_sort_amd as gs
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << logn
dev = "cuda:0"
src = gs.generate_uniform_keys(n, device=dev)
a, b = torch.empty_like(src), torch.empty_like(src)
for seglen in (1 << 5, 1 << 8, 1 << 9, 1 << 10, 1 << 11, 1 << 12, 1 << 14, 1 << 16, 1 << 20, 1 << 24, n):
    nseg = n // seglen
    offs = torch.arange(0, nseg + 1, dtype=torch.int64, device=dev).mul_(seglen).to(torch.int32)
    dk = gs.DoubleBuffer(a, b)
    nb = gs.DeviceSegmentedRadixSort.SortKeys(None, 0, dk, n, nseg, offs[:-1], offs[1:])
    temp = torch.empty(nb, dtype=torch.uint8, device=dev)
    best = 1e9
    for it in range(3):
        a.copy_(src); dk.selector = 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); gs.DeviceSegmentedRadixSort.SortKeys(temp, nb, dk, n, nseg, offs[:-1], offs[1:], key_type=gs.GS_KEY_U32); e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print(f"2^{logn} keys in {nseg} segments of {seglen}: {best:.2f} ms  {n / best / 1e6:.1f} Gkeys/s  temp {nb / 2**20:.0f} MiB", flush=True)
