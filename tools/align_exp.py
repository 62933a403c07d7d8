"""Does the MSB level-1 upsweep / scatter care whether its tiles start on 32 KiB boundaries?  Uniform keys (buckets of the top byte end
anywhere) against keys whose top byte holds exactly n/256 keys each (every level-1 tile aligned like an LSB tile).
python tools/align_exp.py [log2n]"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
n = 1 << (int(sys.argv[1]) if len(sys.argv) > 1 else 30)
dev = torch.device("cuda:0")
uni = gs.generate_uniform_keys(n, device=dev)
# exactly n/256 keys per top byte: top byte = index mod 256, the rest from the uniform keys
idx = torch.arange(n, device=dev, dtype=torch.int64)
al = (((idx & 255) << 24) | (uni.to(torch.int64) & 0xFFFFFF))
al = torch.where(al >= 2**31, al - 2**32, al).to(torch.int32)
del idx
work, alt = torch.empty_like(uni), torch.empty_like(uni)
dm = torch.empty(gs.lib.gs_msb_temp_bytes(n, 0), dtype=torch.uint8, device=dev)
# exactly n/65536 keys per 16-bit prefix: every local-sort task of level 1 starts on a multiple of 16384 keys
idx = torch.arange(n, device=dev, dtype=torch.int64)
al2 = (((idx & 65535) << 16) | (uni.to(torch.int64) & 0xFFFF))
al2 = torch.where(al2 >= 2**31, al2 - 2**32, al2).to(torch.int32)
del idx
for name, src in (("uniform", uni), ("aligned buckets", al), ("aligned tasks", al2), ("uniform", uni), ("aligned buckets", al), ("aligned tasks", al2)):
    tot = {}
    for r in range(4):
        work.copy_(src)
        with gs.KernelProfile() as prof:
            gs.rdxsrt_unstable_sort(work, None, n, alt, None, pre_allocated_dm=dm)
            torch.cuda.synchronize()
        if r: tot = {k: tot.get(k, 0) + v[0] / 3 for k, v in prof.read().items()}
    print(f"{name:16s}", {k: round(v, 3) for k, v in tot.items()}, "sorted", bool((work[1:] >= work[:-1]).all()))
