"""What a bucket dominated by ONE key value costs each MSB kernel: keys = CONST with probability p, uniform otherwise.
[GS_MSB_PIVOT=0] python tools/hot_bucket_exp.py [log2n] [p ...]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 28
ps = [float(x) for x in sys.argv[2:]] or [0.0, 0.5, 0.9, 0.99]
n = 1 << logn
dev = torch.device("cuda:0")
uni = gs.generate_uniform_keys(n, device=dev)
sel = gs.generate_uniform_keys(n, seed=77, device=dev).to(torch.int64) & 0xffffffff
nb = gs.lib.gs_msb_temp_bytes(n, 0)
temp = torch.empty(nb, dtype=torch.uint8, device=dev)
for p in ps:
    src = torch.where(sel < int(p * 2**32), torch.full_like(uni, 0x5A3C7E19 - 2**32 if 0x5A3C7E19 >= 2**31 else 0x5A3C7E19), uni)
    a, b = torch.empty_like(src), torch.empty_like(src)
    for mode in ("nopivot" if os.environ.get("GS_MSB_PIVOT") == "0" else "pivot",):   # the switch is read once per process
        a.copy_(src); gs.rdxsrt_unstable_sort(a, None, n, b, None, pre_allocated_dm=temp); torch.cuda.synchronize()
        with gs.KernelProfile() as prof:
            for _ in range(3):
                a.copy_(src)
                res = gs.rdxsrt_unstable_sort(a, None, n, b, None, pre_allocated_dm=temp, synchronize=False).sorted_keys
            torch.cuda.synchronize()
        t = {k: round(v[0] / 3, 3) for k, v in prof.read().items()}
        print(f"p={p:4.2f} {mode:8s} total {sum(t.values()):7.3f} ms  {t}  inv={gs.check_sorted(res)[0]}")
