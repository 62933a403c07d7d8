"""Ragged sizes and misaligned inputs (tools only): LSB keys at n = 2^28 + r, and from a pointer that is
4-byte but not 16-byte aligned.  python tools/ragged_bench.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
dev = "cuda:0"
base = gs.generate_uniform_keys((1 << 28) + 16384, device=dev)
def run(n, off=0, algo="lsb"):
    src = base[off:off + n]
    a = torch.empty(n + 4, dtype=torch.int32, device=dev)[off:off + n]
    b = torch.empty(n + 4, dtype=torch.int32, device=dev)[off:off + n]
    nb = max(gs.lib.gs_lsb_temp_bytes(n, 0), gs.lib.gs_msb_temp_bytes(n, 0))
    temp = torch.empty(nb, dtype=torch.uint8, device=dev)
    best = 1e9
    for it in range(4):
        a.copy_(src)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        if algo == "lsb":
            dk = gs.DoubleBuffer(a, b)
            gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, key_type=gs.GS_KEY_U32)
            res = dk.Current()
        else:
            res = gs.rdxsrt_unstable_sort(a, None, n, b, None, pre_allocated_dm=temp, synchronize=False).sorted_keys
        e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    inv = gs.check_sorted(res.contiguous() if off else res)[0]
    print(f"{algo} n=2^28+{n - (1 << 28):5d} offset {off}: {best:.3f} ms  {n / best / 1e6:.1f} Gkeys/s  inv={inv}", flush=True)
for r in (0, 5, 4000, 8191):
    run((1 << 28) + r)
run(1 << 28, off=1)
run((1 << 28) + 4000, off=3)
for r in (0, 4000):
    run((1 << 28) + r, algo="msb")
