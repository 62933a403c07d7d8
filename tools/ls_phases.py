"""Experiment build only (-DGS_EXP_LS_PHASES): where a task of the MSB local sort spends its time, per plan and size class
(wave 0 of every task stamps the shader clock after each phase).
GS_LIB_PATH=.../gsvariant_lsp.so python tools/ls_phases.py [log2n] [uniform|zipf] [pairs]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_sort_amd as gs
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dist = sys.argv[2] if len(sys.argv) > 2 else "uniform"
pairs = len(sys.argv) > 3 and sys.argv[3] == "pairs"
n = 1 << logn
dev = torch.device("cuda:0")
raw = C.CDLL(gs.LIB_PATH)
raw.gs_exp_ls_phases.argtypes = [C.c_void_p]
src = (gs.generate_zipf_keys if dist == "zipf" else gs.generate_uniform_keys)(n, device=dev)
a, b = torch.empty_like(src), torch.empty_like(src)
va = gs.generate_enumerated_values(n, device=dev) if pairs else None
vb = torch.empty_like(src) if pairs else None
dm = torch.empty(gs.lib.gs_msb_temp_bytes(n, int(pairs)), dtype=torch.uint8, device=dev)
for r in range(3):
    a.copy_(src)
    torch.cuda.synchronize()
    if r == 2: assert raw.gs_exp_ls_phases(None) == 0
    with gs.KernelProfile() as prof:
        gs.rdxsrt_unstable_sort(a, va, n, b, vb, pre_allocated_dm=dm)
        torch.cuda.synchronize()
print({k: round(v[0], 3) for k, v in prof.read().items()})
NT, NC = 65536, 4
out = np.zeros(3 * NC * NT * 16, dtype=np.uint32)
assert raw.gs_exp_ls_phases(out.ctypes.data_as(C.c_void_p)) == 0
m = out.reshape(3, NC, NT, 16).astype(np.float64)
names = {0: ["next task record", "wait for keys", "zero counters (1st task)", "count: fetch-adds", "barrier", "word sums + wave scan", "barrier",
             "bases over words", "barrier", "lookup", "barrier", "keys -> buffer", "barrier", "request next keys", "read buffer + stores", "TOTAL"],
         1: ["next task record", "wait for keys", "zero histogram + barrier", "1st pass fetch-adds", "barrier", "scan bins", "barrier",
             "base lookup", "barrier", "keys -> buffer", "barrier", "stable passes", "-", "request next keys", "read buffer + stores", "TOTAL"],
         2: ["next task record", "wait for keys", "zero map + counters + barrier", "mark present values", "barrier", "map scan (1 barrier)",
             "prefixes written + barrier", "count per distinct value", "barrier", "counter scan (2 barriers)", "base lookup", "barrier + keys -> buffer",
             "barrier", "request next keys", "read buffer + stores", "TOTAL"]}
from gpu_sort_amd.msb import msb_census
cens = msb_census(dm, n, pairs)
print("census tasks per level:", [c["tasks"] for c in cens], "task keys:", [c["task_keys"] for c in cens])
for plan in (0, 1, 2):
    for cls in range(NC):
        t = m[plan, cls]
        live = t[:, 15] > 0
        if not live.any(): continue
        k = int(live.sum())
        tot = t[live, 15]
        print("\nplan %s class %d: %d tasks stamped (first %d of the list), mean %.0f clocks per task (p10 %.0f p90 %.0f)" %
              (("one-pass", "general", "few distinct values")[plan], cls, k, NT, tot.mean(), np.percentile(tot, 10), np.percentile(tot, 90)))
        for i in range(15):
            v = t[live, i].mean()
            if v > 0: print("   %-28s %8.0f clocks  %5.1f %%" % (names[plan][i], v, 100 * v / tot.mean()))
