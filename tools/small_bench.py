"""Latency of small sorts (tools only): python tools/small_bench.py"""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
dev = "cuda:0"
for n in (1000, 8000, 17408, 17409, 100000, 1 << 20):
    src = gs.generate_uniform_keys(n, device=dev)
    a, b = src.clone(), torch.empty_like(src)
    nb = gs.lib.gs_lsb_temp_bytes(n, 0)
    temp = torch.empty(nb, dtype=torch.uint8, device=dev)
    reps = 200
    dk = gs.DoubleBuffer(a, b)
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            dk.selector = 0
            gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, key_type=gs.GS_KEY_U32)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    print(f"n={n:8d}: {dt * 1e6:8.1f} us per sort (back-to-back, host-inclusive)", flush=True)

# the same sorts replayed from a captured HIP graph (the library makes no host-side decisions during a call,
# so a sort on fixed buffers can be captured once and replayed)
print("--- replayed from a HIP graph")
for n in (1000, 17408, 17409, 100000, 1 << 20, 1 << 24):
    src = gs.generate_uniform_keys(n, device=dev)
    a, b = src.clone(), torch.empty_like(src)
    nb = gs.lib.gs_lsb_temp_bytes(n, 0)
    temp = torch.empty(nb, dtype=torch.uint8, device=dev)
    dk = gs.DoubleBuffer(a, b)
    side = torch.cuda.Stream()
    with torch.cuda.stream(side):
        gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, key_type=gs.GS_KEY_U32)      # warm-up on the capture stream
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    dk.selector = 0
    with torch.cuda.graph(g, stream=side):
        a.copy_(src)
        gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, key_type=gs.GS_KEY_U32)
    res_sel = dk.selector
    reps = 200
    for it in range(3):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(reps):
            g.replay()
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    out = dk.d_buffers[res_sel]
    ok = bool((out[1:].to(torch.int64).bitwise_and(0xFFFFFFFF) >= out[:-1].to(torch.int64).bitwise_and(0xFFFFFFFF)).all()) if n > 1 else True
    print(f"n={n:8d}: {dt * 1e6:8.1f} us per replay (copy-in + sort)  sorted={ok}", flush=True)
