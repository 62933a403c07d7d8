"""Experiment build only (-DGS_EXP_PHASES): where a downsweep block's lifetime goes.
GS_LIB_PATH=.../libgpusort_phases.so python tools/phase_exp.py [log2n]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 30
n = 1 << logn
dev = torch.device("cuda:0")
raw = C.CDLL(gs.LIB_PATH)
uni = gs.generate_uniform_keys(n, device=dev)
grp = gs.generate_uniform_keys(n // 64, seed=5, device=dev).repeat_interleave(64)
grp = (uni & ~0xff) | (grp & 0xff)
a, b = torch.empty_like(uni), torch.empty_like(uni)
nb = gs.lib.gs_lsb_temp_bytes(n, 0)
temp = torch.empty(nb, dtype=torch.uint8, device=dev)
names = ["load issue", "load wait", "rank", "barrier1", "scan+bar2", "lds scatter", "barrier3", "store issue", "store drain"]
import numpy as np
tiles = n // 8192
out = np.zeros(tiles * 16, dtype=np.uint32)
for name, src in (("uniform", uni), ("grouped64", grp)):
    for r in range(4):
        a.copy_(src)
        dk = gs.DoubleBuffer(a, b)
        torch.cuda.synchronize()
        with gs.KernelProfile() as prof:
            gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, 0, 8, key_type=gs.GS_KEY_U32)
            torch.cuda.synchronize()
    raw.gs_exp_phases(out.ctypes.data_as(C.c_void_p), tiles)
    m = out.reshape(tiles, 16)[:, :9].astype(np.float64)
    print(name, {k: round(v[0] / v[1], 4) for k, v in prof.read().items()})
    print("   mean clocks/tile:", {names[i]: round(m[:, i].mean()) for i in range(9)}, "total", round(m.sum(1).mean()))
    full = out.reshape(tiles, 16).astype(np.float64)
    print("   shader clocks per 100 MHz tick: %.2f  -> %.0f MHz; block lifetime %.2f us" % (full[:, 9].sum() / full[:, 10].sum(), 100 * full[:, 9].sum() / full[:, 10].sum(), full[:, 10].mean() / 100))
    print("   median          :", {names[i]: round(float(np.median(m[:, i]))) for i in range(9)})
