"""Upsweep timing only (experiment builds write their results in another layout): python tools/ups_time.py [log2n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << logn
dev = torch.device("cuda:0")
src = gs.generate_uniform_keys(n, device=dev)
nb = gs.lib.gs_lsb_temp_bytes(n, 0)
temp = torch.empty(nb, dtype=torch.uint8, device=dev)
ms = []
for r in range(8):
    a, b = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    a.record()
    gs._lib.check(gs.lib.gs_lsb_upsweep_u32(temp.data_ptr(), nb, src.data_ptr(), n, 8, 8, 0, gs.GS_KEY_U32, None), "upsweep")
    b.record()
    torch.cuda.synchronize()
    ms.append(a.elapsed_time(b))
ms = sorted(ms[1:])
print(f"{os.path.basename(gs.LIB_PATH):28s} upsweep 2^{logn}: median {ms[len(ms)//2]:.4f} ms  min {ms[0]:.4f} ms")
