import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_sort_amd as gs
from oracle import oracle as O
n = 3000017
keys = (O.gen_uniform(n, seed=8) & 0x00FFFFFF) | 0x5A000000
dev = "cuda:0"
def run(pairs):
    k = torch.from_numpy(keys.view(np.int32).copy()).to(dev)
    alt = torch.empty_like(k)
    v = va = None
    if pairs:
        v = torch.arange(n, dtype=torch.int32, device=dev); va = torch.empty_like(v)
    seq = gs.rdxsrt_unstable_sort(k, v, n, alt, va)
    torch.cuda.synchronize()
    out = seq.sorted_keys.cpu().numpy().view(np.uint32)
    want = np.sort(keys)
    bad = np.nonzero(out != want)[0]
    print("pairs" if pairs else "keys", "mismatches", bad.size, "first", bad[:5], "last", bad[-5:] if bad.size else None)
    if bad.size:
        i = bad[0]
        print(" out ", [hex(x) for x in out[i:i+6]]); print(" want", [hex(x) for x in want[i:i+6]])
        # multiset check
        print(" multiset equal:", np.array_equal(np.sort(out), want))
        d = np.diff(out.astype(np.int64)) < 0
        print(" inversions:", d.sum(), "at", np.nonzero(d)[0][:10])
    if pairs:
        vo = seq.sorted_values.cpu().numpy().view(np.uint32)
        print(" value check:", O.msb_check_pairs_enumerated(keys, out, vo))
run(False); run(True)
