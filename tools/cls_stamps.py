"""Experiment build (-DGS_EXP_CLS): where block 0 of every classification launch spends its time (100 MHz stamps)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_sort_amd as gs
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 24
n = 1 << logn
dev = torch.device("cuda:0")
raw = C.CDLL(gs.LIB_PATH)
src = gs.generate_uniform_keys(n, device=dev)
a, b = torch.empty_like(src), torch.empty_like(src)
temp = torch.empty(gs.lib.gs_msb_temp_bytes(n, 0), dtype=torch.uint8, device=dev)
for _ in range(3):
    a.copy_(src); gs.rdxsrt_unstable_sort(a, None, n, b, None, pre_allocated_dm=temp)
torch.cuda.synchronize()
out = np.zeros(16 * 8, dtype=np.uint64)
raw.gs_exp_cls_stamps(out.ctypes.data_as(C.c_void_p))
st = out.reshape(8, 16).astype(np.int64)
names = ["start", "counts loaded", "scan 1", "lists reset", "merge loop", "block scans", "atomics", "end"]
for L in range(4):
    r = st[L]
    if r[0] == 0: continue
    print("level", L, " ".join("%s +%.2f us" % (names[k], (r[k] - r[0]) / 100.0) for k in range(1, 8) if r[k] >= r[0]))
