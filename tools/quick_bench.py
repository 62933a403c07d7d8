"""Quick LSB timing (development aid; bench.py is the contract)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs

def run(logn, pairs, trials=5):
    n = 1 << logn
    dev = torch.device("cuda:0")
    src = gs.generate_uniform_keys(n, device=dev)
    a, b = torch.empty_like(src), torch.empty_like(src)
    if pairs:
        va, vb = gs.generate_enumerated_values(n, device=dev), torch.empty_like(src)
    nb = gs.lib.gs_lsb_temp_bytes(n, int(pairs))
    temp = torch.empty(nb, dtype=torch.uint8, device=dev)
    times = []
    for t in range(trials + 1):
        a.copy_(src)
        dk = gs.DoubleBuffer(a, b)
        dv = gs.DoubleBuffer(va, vb) if pairs else None
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if pairs: gs.DeviceRadixSort.SortPairs(temp, nb, dk, dv, n, key_type=gs.GS_KEY_U32)
        else: gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, key_type=gs.GS_KEY_U32)
        e1.record(); e1.synchronize()
        times.append(e0.elapsed_time(e1))
    inv, _, _ = gs.check_sorted(dk.Current())
    ms = sorted(times[1:])[len(times[1:]) // 2]
    bpk = 80 if pairs else 48
    print(f"n=2^{logn} pairs={pairs} median {ms:.3f} ms  {n/ms/1e6:.2f} Gkeys/s  {bpk*n/ms/1e6:.0f} GB/s algorithmic  inv={inv}", flush=True)

if __name__ == "__main__":
    for logn in (24, 28, 30):
        run(logn, False)
    run(28, True)
    run(30, True)
