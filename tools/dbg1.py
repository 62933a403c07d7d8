import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch, numpy as np
import gpu_sort_amd as gs
dev = torch.device("cuda:0")
for logn in (20, 24, 26, 27):
    n = 1 << logn
    keys = gs.generate_uniform_keys(n, device=dev)
    orig = keys.clone()
    vals = gs.generate_enumerated_values(n, device=dev)
    print("vals head", vals[:5].tolist(), vals[-3:].tolist())
    dk = gs.DoubleBuffer(keys, torch.empty_like(keys)); dv = gs.DoubleBuffer(vals, torch.empty_like(keys))
    nb = gs.DeviceRadixSort.SortPairs(None, 0, dk, dv, n)
    temp = torch.empty(nb, dtype=torch.uint8, device=dev)
    gs.DeviceRadixSort.SortPairs(temp, nb, dk, dv, n, key_type=gs.GS_KEY_U32)
    torch.cuda.synchronize()
    k = dk.Current().cpu().numpy().view(np.uint32); v = dv.Current().cpu().numpy().view(np.uint32)
    o = orig.cpu().numpy().view(np.uint32)
    print(logn, "sel", dk.selector, dv.selector, "v head", v[:5], "max v", v.max(), "ok", np.array_equal(o[v], k), "bad cnt", (o[v] != k).sum())
    print(gs.check_pairs_enumerated(orig, dk.Current(), dv.Current()), n*(n-1)//2)
