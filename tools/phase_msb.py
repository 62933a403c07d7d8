"""Experiment build only (-DGS_EXP_PHASES): phases of the first local-sort task of every block.
GS_LIB_PATH=.../libgpusort_phases.so python tools/phase_msb.py [log2n]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_sort_amd as gs
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << logn
dev = torch.device("cuda:0")
raw = C.CDLL(gs.LIB_PATH)
src = gs.generate_uniform_keys(n, device=dev)
a, b = src.clone(), torch.empty_like(src)
nb = gs.lib.gs_msb_temp_bytes(n, 0)
temp = torch.empty(nb, dtype=torch.uint8, device=dev)
for r in range(2):
    a.copy_(src)
    with gs.KernelProfile() as prof:
        gs.rdxsrt_unstable_sort(a, None, n, b, None, pre_allocated_dm=temp)
        torch.cuda.synchronize()
print({k: round(v[0], 3) for k, v in prof.read().items()})
blocks = 16384
out = np.zeros(blocks * 32, dtype=np.uint32)
raw.gs_exp_msb_phases(out.ctypes.data_as(C.c_void_p), blocks)
m = out.reshape(blocks, 32).astype(np.float64)
m = m[m[:, 16] > 0]
names = {20: "record+drain", 21: "load issue", 0: "load wait", 1: "p1 -", 2: "p1 rank", 3: "p1 bar", 4: "p1 scan", 5: "p1 base+bar", 6: "p1 scatter+bar", 8: "p2 readback+bar",
         9: "p2 rank", 10: "p2 bar", 11: "p2 scan", 12: "p2 base+bar", 13: "p2 scatter+bar", 16: "store issue", 17: "final bar"}
print("blocks with data:", len(m), " total clocks/task:", round(m.sum(1).mean()))
for k, v in names.items():
    print(f"  {v:18s} {m[:, k].mean():9.0f}")
