"""Does the LSB pairs downsweep depend on where the value arrays lie RELATIVE to the key arrays?  The same GPU was seen to run
it in 3.3, 3.7 and 4.2 ms in different processes (tools/box_probe.sh) with identical clocks, TLB counters and micro-patterns.
All four arrays are cut out of ONE allocation at chosen byte offsets from each other.  python tools/pairs_offset_exp.py"""
import sys, os, json
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
dev = torch.device("cuda:0")
n = 1 << 30
B = n * 4
MiB = 1 << 20
pool = torch.empty(4 * B + 4096 * MiB, dtype=torch.uint8, device=dev)       # 16 GiB + 4 GiB of slack
nb = gs.lib.gs_lsb_temp_bytes(n, 1)
temp = torch.empty(nb, dtype=torch.uint8, device=dev)
src = gs.generate_uniform_keys(n, device=dev)
print("pool base mod 2 MiB:", hex(pool.data_ptr() % (2 * MiB)), flush=True)


def view(off):
    return pool[off:off + B].view(torch.int32)


def run(offs, label):
    a, b, va, vb = (view(o) for o in offs)
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
    print(f"{label:58s} downsweep {k['lsb_downsweep'][0] / k['lsb_downsweep'][1]:.3f} ms", flush=True)


G = B
for skew, label in ((0, "values exactly 8 GiB behind the keys (a, b, va, vb back to back)"),
                    (256, "+256 B"), (4096, "+4 KiB"), (64 * 1024 + 256, "+64 KiB + 256 B"), (MiB + 4352, "+1 MiB + 4352 B"),
                    (17 * MiB + 8448, "+17 MiB + 8448 B"), (1024 * MiB, "+1 GiB"), (2048 * MiB + 2 * MiB, "+2 GiB + 2 MiB")):
    run((0, G, 2 * G + skew, 3 * G + skew), label)
# also the alternate key buffer skewed against the first
run((0, G + 17 * MiB + 8448, 2 * G + 34 * MiB + 12544, 3 * G + 51 * MiB + 4352), "all four arrays at mutually odd offsets")
run((0, 2 * G, G, 3 * G), "order a, va, b, vb")
