"""python tools/overlap_exp.py [LOG2N] -- how much do the LSB upsweep and downsweep gain from running side by side?
Timing only (the downsweep uses the spine of an earlier upsweep; the concurrent upsweep writes a second workspace):
serial = upsweep then downsweep on one stream; overlap = the same two kernels on two streams."""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
from gpu_sort_amd._lib import check as _check
gs.check = _check
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << logn
dev = torch.device("cuda:0")
src = gs.generate_uniform_keys(n, device=dev)
out = torch.empty_like(src)
nb = gs.lib.gs_lsb_temp_bytes(n, 0)
t1 = torch.empty(nb, dtype=torch.uint8, device=dev)
t2 = torch.empty(nb, dtype=torch.uint8, device=dev)
L = gs.lib
def up(temp, stream): gs.check(L.gs_lsb_upsweep_u32(temp.data_ptr(), nb, src.data_ptr(), n, 0, 8, 0, 0, stream.cuda_stream), "up")
def scan(temp, stream): gs.check(L.gs_lsb_scan_spine(temp.data_ptr(), nb, n, stream.cuda_stream), "scan")
def down(temp, stream): gs.check(L.gs_lsb_downsweep_u32(temp.data_ptr(), nb, src.data_ptr(), out.data_ptr(), None, None, n, 0, 8, 0, 0, 0, stream.cuda_stream), "down")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
up(t1, s1); scan(t1, s1); torch.cuda.synchronize()
def timed(f, reps=6):
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        a.record(s1); f(); s1.wait_stream(s2); b.record(s1); torch.cuda.synchronize()
        best = min(best, a.elapsed_time(b))
    return best
print("upsweep alone      %.3f ms" % timed(lambda: up(t2, s1)))
print("downsweep alone    %.3f ms" % timed(lambda: down(t1, s1)))
print("serial up;down     %.3f ms" % timed(lambda: (up(t2, s1), down(t1, s1))))
def both():
    s2.wait_stream(s1)
    down(t1, s1); up(t2, s2)
print("overlap down||up   %.3f ms" % timed(both))
def both2():
    s2.wait_stream(s1)
    up(t2, s2); down(t1, s1)
print("overlap up||down   %.3f ms" % timed(both2))
