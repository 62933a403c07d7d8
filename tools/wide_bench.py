#!/usr/bin/env python3
"""Time gs_lsb_sort_wide for the 64-bit configurations (not a headline number; tools only).
usage: python tools/wide_bench.py [log2n]"""
import os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs

log2n = int(sys.argv[1]) if len(sys.argv) > 1 else 28
n = 1 << log2n
dev = "cuda:0"
for kdt, vdt in ((torch.int64, None), (torch.int64, torch.int32), (torch.int64, torch.int64), (torch.int32, torch.int64),
                 (torch.float64, None)):
    if kdt.is_floating_point:
        src = torch.randn(n, dtype=kdt, device=dev)
    else:
        info = torch.iinfo(kdt)
        src = torch.randint(info.min, info.max, (n,), dtype=kdt, device=dev)
    dk = gs.DoubleBuffer(src.clone(), torch.empty_like(src))
    dv = None
    if vdt is not None:
        v = torch.arange(n, device=dev).to(vdt)
        dv = gs.DoubleBuffer(v, torch.empty_like(v))
    args = (dk, dv, n) if dv is not None else (dk, n)
    fn = gs.DeviceRadixSort.SortPairs if dv is not None else gs.DeviceRadixSort.SortKeys
    nb = fn(None, 0, *args)
    temp = torch.empty(nb, dtype=torch.uint8, device=dev)
    best = 1e9
    for it in range(4):
        dk.d_buffers[0].copy_(src); dk.selector = 0
        if dv is not None: dv.selector = 0
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        with gs.KernelProfile() as prof:
            e0.record(); fn(temp, nb, *args); e1.record()
            torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1))
    kb, vb = src.element_size(), (0 if vdt is None else torch.empty(0, dtype=vdt).element_size())
    passes = kb
    alg = n * passes * (kb + 2 * (kb + vb))
    k = prof.read()
    print(f"keys {kdt} vals {vdt}: n=2^{log2n} {best:.2f} ms  {n / best / 1e6:.1f} Gkeys/s  "
          f"algorithmic {alg / best / 1e6:.0f} GB/s  kernels {({a: round(b[0] / b[1], 3) for a, b in k.items()})}")
