"""Experiment build only (-DGS_EXP_PHASES [-DGS_EXP_PERSIST=768]): timeline of the downsweep blocks of one pass —
who ran where and when, how long each phase took, how evenly the CUs were served.
GS_LIB_PATH=.../gsvariant_ph768.so python tools/phase_timeline.py [log2n] [pairs]"""
import sys, os, ctypes as C
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import numpy as np, torch
import gpu_sort_amd as gs
logn = int(sys.argv[1]) if len(sys.argv) > 1 else 30
pairs = len(sys.argv) > 2 and sys.argv[2] == "pairs"
n = 1 << logn
dev = torch.device("cuda:0")
raw = C.CDLL(gs.LIB_PATH)
uni = gs.generate_uniform_keys(n, device=dev)
a, b = torch.empty_like(uni), torch.empty_like(uni)
nb = gs.lib.gs_lsb_temp_bytes(n, int(pairs))
va = gs.generate_enumerated_values(n, device=dev) if pairs else None
vb = torch.empty_like(uni) if pairs else None
temp = torch.empty(nb, dtype=torch.uint8, device=dev)
names = ["load issue", "load wait", "rank", "barrier1", "scan+bar2", "lds scatter", "barrier3", "store issue", "store drain"]
tiles = n // 8192
out = np.zeros(tiles * 16, dtype=np.uint32)
for r in range(3):
    a.copy_(uni)
    dk = gs.DoubleBuffer(a, b)
    torch.cuda.synchronize()
    with gs.KernelProfile() as prof:
        if pairs: gs.DeviceRadixSort.SortPairs(temp, nb, dk, gs.DoubleBuffer(va, vb), n, 0, 8, key_type=gs.GS_KEY_U32)
        else: gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, 0, 8, key_type=gs.GS_KEY_U32)
        torch.cuda.synchronize()
raw.gs_exp_phases(out.ctypes.data_as(C.c_void_p), tiles)
m = out.reshape(tiles, 16).astype(np.int64)
print({k: round(v[0] / v[1], 4) for k, v in prof.read().items()})
ph = m[:, :9].astype(np.float64)
clk = m[:, 9].sum() / m[:, 10].sum()           # shader clocks per 10 ns
print("shader clock %.0f MHz; tile time %.2f us mean" % (100 * clk, m[:, 10].mean() / 100))
print("mean us per phase:", {names[i]: round(ph[:, i].mean() / clk / 100, 2) for i in range(9)})
start = (m[:, 11] - m[:, 11].min()) & 0xffffffff
end = start + m[:, 10]
print("pass length by the tiles' own clocks: %.3f ms" % (end.max() / 1e5))
blk = m[:, 12]
cu = (m[:, 14] & 0xf) * 1000 + ((m[:, 13] >> 13) & 7) * 100 + ((m[:, 13] >> 12) & 1) * 50 + ((m[:, 13] >> 8) & 0xf)
ucu = np.unique(cu)
print("distinct (xcc, se, sh, cu):", len(ucu))
per_cu = np.array([np.sum(cu == c) for c in ucu])
print("tiles per CU: min %d  p10 %d  median %d  p90 %d  max %d" % (per_cu.min(), np.percentile(per_cu, 10), np.median(per_cu), np.percentile(per_cu, 90), per_cu.max()))
nblk = int(blk.max()) + 1
if nblk <= 4096:                                  # persistent build: when did each block finish, how long were its tiles
    fin = np.array([end[blk == k].max() for k in range(nblk)]) / 1e5
    print("block finish times (ms): min %.3f p10 %.3f median %.3f p90 %.3f max %.3f" % (fin.min(), np.percentile(fin, 10), np.median(fin), np.percentile(fin, 90), fin.max()))
    k = 5
    sel = np.where(blk == k)[0]
    order = sel[np.argsort(start[sel])]
    print("block 5, first tiles: start(us), length(us):", [(round(start[i] / 100, 1), round(m[i, 10] / 100, 1)) for i in order[:8]])
# occupancy over time: tiles in flight per 50 us
T = int(end.max() // 5000) + 1
occ = np.zeros(T)
for t0, t1 in zip(start, end):
    a0, a1 = int(t0 // 5000), int(t1 // 5000)
    occ[a0:a1 + 1] += 1
print("tiles in flight (sampled per 50 us bin, counts of tiles touching the bin):", [int(x) for x in occ[:: max(1, T // 16)]])
# evolution: per 200 us bin of start time: tiles started, mean tile length, mean load wait, spread of tile indices in flight
idx = np.arange(tiles)
bins = (start // 20000).astype(int)
print("bin(0.2ms)  tiles  len_us  loadwait_us  store_issue_us  tile-index spread of the tiles started in the bin")
for bb in range(bins.max() + 1):
    sel = bins == bb
    if sel.sum() == 0: continue
    print("  %2d  %6d  %6.2f  %6.2f  %6.2f   %d" % (bb, sel.sum(), m[sel, 10].mean() / 100, ph[sel, 1].mean() / clk / 100, ph[sel, 7].mean() / clk / 100,
          int(np.percentile(idx[sel], 99) - np.percentile(idx[sel], 1))))
