"""Per-level census of an MSB sort (buckets, tiles, local-sort tasks per class, flagged tasks) read back from
the workspace after the call.  python tools/msb_census.py [log2n] [uniform|zipf] [pairs]"""
import sys, os, struct
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_sort_amd as gs
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 30
dist = sys.argv[2] if len(sys.argv) > 2 else "uniform"
pairs = len(sys.argv) > 3 and sys.argv[3] == "pairs"
n = 1 << logn
dev = "cuda:0"
gen = gs.generate_uniform_keys if dist == "uniform" else gs.generate_zipf_keys
a = gen(n, device=dev); b = torch.empty_like(a)
va = gs.generate_enumerated_values(n, device=dev) if pairs else None
vb = torch.empty_like(a) if pairs else None
nb = gs.lib.gs_msb_temp_bytes(n, int(pairs))
temp = torch.zeros(nb, dtype=torch.uint8, device=dev)
with gs.KernelProfile() as prof:
    gs.rdxsrt_unstable_sort(a, va, n, b, vb, pre_allocated_dm=temp)
    torch.cuda.synchronize()
off = (gs.lib.gs_lsb_temp_bytes(n, int(pairs)) + 255) // 256 * 256
raw = temp[off: off + 5 * 48].cpu().numpy().tobytes()
print({k: round(v[0], 3) for k, v in prof.read().items()})
for L in range(4):
    packed, t0, t1, t2, t3, flagged, pb, pk, _ = struct.unpack_from("<Q6IQQ", raw, L * 48)
    print(f"level {L}: buckets {packed >> 32:7d}  tiles {packed & 0xffffffff:8d} ({(packed & 0xffffffff) * 8192 / n:6.1%} of the keys)  "
          f"heavy-hitter buckets {pb} ({pk / n:6.1%} of the keys)  "
          f"tasks per class {[t0, t1, t2, t3]}  tasks left to the general plan: {'yes' if flagged else 'no'}")
