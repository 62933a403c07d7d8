"""python tools/pipe_check.py [log2 of the largest size] -- pipelined LSB passes against torch's stable sort on the device,
with the threshold lowered so that small arrays take them too (GS_LSB_PIPE_MIN_TILES=0 must be set by the caller)."""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
from gpu_sort_amd._lib import check
dev = torch.device("cuda:0")
top = int(sys.argv[1]) if len(sys.argv) > 1 else 26
sizes = [8192, 8193, 16384 + 5, 65536, 100003, 8192 * 48 * 8, 8192 * 48 * 8 + 8191, (1 << 22) + 77, (1 << top) + 12345]
bad = 0
for n in sizes:
    src = gs.generate_uniform_keys(n, device=dev)
    for pairs in (False, True):
        for (bb, eb, desc) in ((0, 32, False), (0, 32, True), (3, 29, False), (8, 24, True), (0, 16, False), (5, 14, False)):
            a = src.clone(); b = torch.empty_like(a)
            dk = gs.DoubleBuffer(a, b)
            nb = gs.lib.gs_lsb_temp_bytes(n, int(pairs))
            temp = torch.empty(nb, dtype=torch.uint8, device=dev)
            if pairs:
                va = gs.generate_enumerated_values(n, device=dev); dv = gs.DoubleBuffer(va, torch.empty_like(va))
                fn = gs.DeviceRadixSort.SortPairsDescending if desc else gs.DeviceRadixSort.SortPairs
                fn(temp, nb, dk, dv, n, bb, eb, key_type=gs.GS_KEY_U32)
            else:
                fn = gs.DeviceRadixSort.SortKeysDescending if desc else gs.DeviceRadixSort.SortKeys
                fn(temp, nb, dk, n, bb, eb, key_type=gs.GS_KEY_U32)
            st = C.c_uint32(0)
            check(gs.lib.gs_lsb_pipe_status(temp.data_ptr(), n, C.byref(st), None), "status")
            k64 = src.to(torch.int64) & 0xffffffff
            mask = ((1 << (eb - bb)) - 1)
            dig = (k64 >> bb) & mask
            order = torch.sort(dig, stable=True, descending=desc).indices if not desc else None
            if desc:   # CUB: reverse, stable ascending, reverse
                order = torch.sort(dig.flip(0), stable=True).indices
                order = (n - 1 - order).flip(0)
            exp = src[order]
            ok = torch.equal(dk.Current(), exp)
            if pairs: ok = ok and torch.equal(dv.Current().to(torch.int64) & 0xffffffff, order)
            if not ok or st.value:
                bad += 1
                print(f"FAIL n={n} pairs={pairs} bits=[{bb},{eb}) desc={desc} status={st.value} mismatches={(dk.Current() != exp).sum().item()}")
print("pipe_check:", "all ok" if not bad else f"{bad} failures", "mode", os.environ.get("GS_LSB_MODE"), "min_tiles", os.environ.get("GS_LSB_PIPE_MIN_TILES"))
sys.exit(1 if bad else 0)
