"""What the box reports for HBM and what plain streaming reaches on it: device-to-device copy,
read-only and write-only streams of 4 GiB, median of 9 device-timed runs each."""
import torch, subprocess
dev = torch.device("cuda:0")
p = torch.cuda.get_device_properties(dev)
print("device:", p.name, "| CUs:", p.multi_processor_count, "| memory:", round(p.total_memory / 2**30, 1), "GiB")
for attr in ("memory_clock_rate", "memory_bus_width", "clock_rate", "L2_cache_size", "gcnArchName"):
    if hasattr(p, attr):
        print(f"  {attr} = {getattr(p, attr)}")
if hasattr(p, "memory_clock_rate") and hasattr(p, "memory_bus_width"):
    print(f"  memory_clock_rate x 2 (DDR) x bus_width / 8 = {p.memory_clock_rate * 1e3 * 2 * p.memory_bus_width / 8 / 1e12:.2f} TB/s")
n = 1 << 30
a = torch.empty(n, dtype=torch.int32, device=dev).random_()
b = torch.empty_like(a)
def med(fn, reps=9):
    ts = []
    for _ in range(reps + 2):
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); fn(); e1.record(); e1.synchronize()
        ts.append(e0.elapsed_time(e1))
    ts = sorted(ts[2:]); return ts[len(ts) // 2]
t = med(lambda: b.copy_(a)); print(f"D2D copy 4 GiB -> 4 GiB: {t:.3f} ms = {2 * 4 * n / t / 1e9:.2f} TB/s (read + write)")
t = med(lambda: a.sum()); print(f"read-only (sum of 4 GiB): {t:.3f} ms = {4 * n / t / 1e9:.2f} TB/s")
t = med(lambda: b.fill_(7)); print(f"write-only (fill 4 GiB): {t:.3f} ms = {4 * n / t / 1e9:.2f} TB/s")
try:
    out = subprocess.run(["rocm-smi", "--showclocks"], capture_output=True, text=True, timeout=30).stdout
    print("\n".join(l for l in out.splitlines() if "mclk" in l.lower() or "sclk" in l.lower())[:600])
except Exception as e:
    print("rocm-smi:", e)
