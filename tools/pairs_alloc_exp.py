"""Pairs downsweep time per fresh allocation of its four arrays inside ONE process (hipMalloc through torch, cache emptied in
between; sometimes with a dummy allocation held to move the arrays).  python tools/pairs_alloc_exp.py"""
import sys, os
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
dev = torch.device("cuda:0")
n = 1 << 30
nb = gs.lib.gs_lsb_temp_bytes(n, 1)
temp = torch.empty(nb, dtype=torch.uint8, device=dev)
src = gs.generate_uniform_keys(n, device=dev)
hold = []
for trial in range(8):
    if trial in (2, 4, 6):
        hold.append(torch.empty((3 + trial) << 28, dtype=torch.uint8, device=dev))      # shift what the next arrays get
    a, b, va, vb = (torch.empty(n, dtype=torch.int32, device=dev) for _ in range(4))
    prof = gs.KernelProfile()
    for r in range(4):
        a.copy_(src)
        gs.generate_enumerated_values(n, device=dev, out=va)
        if r == 1:
            prof.__enter__()
        gs.DeviceRadixSort.SortPairs(temp, nb, gs.DoubleBuffer(a, b), gs.DoubleBuffer(va, vb), n, key_type=gs.GS_KEY_U32)
    prof.__exit__()
    torch.cuda.synchronize()
    k = prof.read()
    # plain streaming through the same four arrays: a -> b and va -> vb copies, timed with events
    ev = [torch.cuda.Event(enable_timing=True) for _ in range(3)]
    b.copy_(a); vb.copy_(va)
    ev[0].record(); b.copy_(a); ev[1].record(); vb.copy_(va); ev[2].record()
    torch.cuda.synchronize()
    cp = f"copy a->b {ev[0].elapsed_time(ev[1]):.3f} ms  va->vb {ev[1].elapsed_time(ev[2]):.3f} ms"
    print(f"trial {trial}: a={a.data_ptr():#x} b={b.data_ptr():#x} va={va.data_ptr():#x} vb={vb.data_ptr():#x}  "
          f"downsweep {k['lsb_downsweep'][0] / k['lsb_downsweep'][1]:.3f} ms  upsweep {k['lsb_upsweep'][0] / k['lsb_upsweep'][1]:.3f}  {cp}", flush=True)
    del a, b, va, vb
    torch.cuda.empty_cache()
