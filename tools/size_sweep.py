"""Throughput over sizes 2^10 .. 2^30 for the LSB and MSB sorts (keys only and pairs), median of 7 device-timed
calls each; the table that sits in profiles/ as r01_v8_size_sweep.txt."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
dev = torch.device("cuda:0")

def timed(fn, prep, trials=7):
    ts = []
    for _ in range(trials + 2):
        prep(); torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[2:])
    return ts[len(ts) // 2]

print("log2n | LSB keys ms (Gkeys/s) | LSB pairs ms (Gpairs/s) | MSB keys ms (Gkeys/s) | MSB pairs ms (Gpairs/s)")
for logn in (10, 12, 14, 16, 18, 20, 22, 24, 26, 28, 30):
    n = 1 << logn
    src = gs.generate_uniform_keys(n, device=dev)
    a, b = torch.empty_like(src), torch.empty_like(src)
    vsrc = gs.generate_enumerated_values(n, device=dev)
    va, vb = torch.empty_like(src), torch.empty_like(src)
    nb = max(gs.lib.gs_lsb_temp_bytes(n, 1), gs.lib.gs_msb_temp_bytes(n, 1), 256)
    temp = torch.empty(nb, dtype=torch.uint8, device=dev)
    def prep():
        a.copy_(src); va.copy_(vsrc)
    row = [f"{logn:5d}"]
    for algo, pairs in (("lsb", False), ("lsb", True), ("msb", False), ("msb", True)):
        if algo == "lsb":
            def fn():
                dk = gs.DoubleBuffer(a, b)
                if pairs: gs.DeviceRadixSort.SortPairs(temp, nb, dk, gs.DoubleBuffer(va, vb), n, key_type=gs.GS_KEY_U32)
                else: gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, key_type=gs.GS_KEY_U32)
        else:
            def fn():
                gs.rdxsrt_unstable_sort(a, va if pairs else None, n, b, vb if pairs else None, pre_allocated_dm=temp, synchronize=False)
        ms = timed(fn, prep)
        row.append(f"{ms:9.4f} ({n / ms / 1e6:7.2f})")
    print(" | ".join(row), flush=True)
