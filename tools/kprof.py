"""Per-kernel times via the library's event hook: python tools/kprof.py LOG2N [pairs] [algo] [dist]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 30
pairs = len(sys.argv) > 2 and sys.argv[2] == "pairs"
algo = sys.argv[3] if len(sys.argv) > 3 else "lsb"
dist = sys.argv[4] if len(sys.argv) > 4 else "uniform"
n = 1 << logn
dev = torch.device("cuda:0")
gen = gs.generate_uniform_keys if dist == "uniform" else gs.generate_zipf_keys
src = gen(n, device=dev)
a, b = torch.empty_like(src), torch.empty_like(src)
va = gs.generate_enumerated_values(n, device=dev) if pairs else None
vb = torch.empty_like(src) if pairs else None
nb = max(gs.lib.gs_lsb_temp_bytes(n, int(pairs)), gs.lib.gs_msb_temp_bytes(n, int(pairs)), 1)
temp = torch.empty(nb, dtype=torch.uint8, device=dev)
def run():
    a.copy_(src)
    if algo == "lsb":
        dk = gs.DoubleBuffer(a, b)
        if pairs: gs.DeviceRadixSort.SortPairs(temp, nb, dk, gs.DoubleBuffer(va, vb), n, key_type=gs.GS_KEY_U32)
        else: gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, key_type=gs.GS_KEY_U32)
        return dk.Current()
    return gs.rdxsrt_unstable_sort(a, va, n, b, vb, pre_allocated_dm=temp, synchronize=False).sorted_keys
run(); torch.cuda.synchronize()
prof = gs.KernelProfile()
reps = 5
with prof:
    for _ in range(reps): res = run()
torch.cuda.synchronize()
tot = 0
for k, (ms, cnt) in prof.read().items():
    print(f"{k:16s} {ms/cnt:8.4f} ms/launch x {cnt//reps}/sort"); tot += ms / reps
print(f"sum {tot:.3f} ms/sort  -> {n/tot/1e6:.2f} Gkeys/s   inv={gs.check_sorted(res)[0]} GS_DEBUG={os.environ.get('GS_DEBUG')}")
