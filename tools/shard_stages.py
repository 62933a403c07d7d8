"""Stage-by-stage timing of the single-rank sharded pipeline (development aid)."""
import sys, os, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_sort_amd as gs
from gpu_sort_amd import sharded
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 30
world = int(sys.argv[2]) if len(sys.argv) > 2 else 8      # pretend-world for the split (kernels only)
n = 1 << logn
dev = torch.device("cuda:0")
ops = sharded.DeviceOps(dev)
keys = gs.generate_uniform_keys(n, device=dev)
temp = torch.empty(ops.temp_bytes(n, False), dtype=torch.uint8, device=dev)
out = ops.empty(n)
def T():
    torch.cuda.synchronize(); return time.perf_counter()
for rep in range(3):
    t0 = T(); hist = ops.histogram(keys, n, sharded.SHARD_BITS); t1 = T()
    h = hist.cpu().numpy(); hist_all = np.tile(h, (world, 1)); t2 = T()
    dest, per_rank = sharded.compute_splits(hist_all, world); send, recv = sharded.exchange_plan(hist_all, dest, 0, world); t3 = T()
    cnt = ops.partition(keys, None, n, sharded.SHARD_BITS, dest, world, temp, out, None, bin_hist=hist); t4 = T()
    print(f"hist {1e3*(t1-t0):.3f} ms | d2h {1e3*(t2-t1):.3f} | splits {1e3*(t3-t2):.3f} | partition {1e3*(t4-t3):.3f}", flush=True)
