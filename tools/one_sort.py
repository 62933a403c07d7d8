"""Run a few sorts for profiling: python tools/one_sort.py LOG2N [pairs] [reps]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 28
pairs = len(sys.argv) > 2 and sys.argv[2] == "pairs"
reps = int(sys.argv[3]) if len(sys.argv) > 3 else 2
algo = sys.argv[4] if len(sys.argv) > 4 else "lsb"
dist = sys.argv[5] if len(sys.argv) > 5 else "uniform"
n = 1 << logn
dev = torch.device("cuda:0")
gen = gs.generate_uniform_keys if dist == "uniform" else gs.generate_zipf_keys
src = gen(n, device=dev)
a, b = torch.empty_like(src), torch.empty_like(src)
va = gs.generate_enumerated_values(n, device=dev) if pairs else None
vb = torch.empty_like(src) if pairs else None
nb = max(gs.lib.gs_lsb_temp_bytes(n, int(pairs)), gs.lib.gs_msb_temp_bytes(n, int(pairs)), 1)
temp = torch.empty(nb, dtype=torch.uint8, device=dev)
for r in range(reps):
    a.copy_(src)
    if algo == "lsb":
        dk = gs.DoubleBuffer(a, b)
        if pairs:
            gs.DeviceRadixSort.SortPairs(temp, nb, dk, gs.DoubleBuffer(va, vb), n, key_type=gs.GS_KEY_U32)
        else:
            gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, key_type=gs.GS_KEY_U32)
        res = dk.Current()
    else:
        res = gs.rdxsrt_unstable_sort(a, va, n, b, vb, pre_allocated_dm=temp).sorted_keys
torch.cuda.synchronize()
print("inv", gs.check_sorted(res)[0])
