"""A/B of library builds INSIDE one process on the SAME buffers (the placement of the arrays changes a kernel's time by several
percent from one allocation to the next, profiles/README.md round 3): every build is loaded with ctypes and runs the LSB sort of
2^30 keys (or pairs) alternately, per-kernel times from each build's own event hook.
python tools/ab_inproc.py [pairs|msb|msbzipf] libA.so libB.so ...   (names inside gpu-sort_amd/lib)"""
import ctypes as C, os, sys
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
import gpu_sort_amd as gs
args = sys.argv[1:]
mode = args[0] if args and args[0] in ("pairs", "msb", "msbzipf") else "keys"
if mode != "keys":
    args = args[1:]
pairs = mode == "pairs"
msb = mode.startswith("msb")
libdir = os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "gpu-sort_amd", "lib")
dev = torch.device("cuda:0")
n = 1 << int(os.environ.get('LOG2N', '30'))
# CHURN=k: perturb where the arrays land (a fresh process usually gets a "fast" placement): k rounds of odd-sized allocations, every
# other one freed again (with empty_cache, so the driver gets the memory back) before the arrays are allocated
hold = []
for r in range(int(os.environ.get("CHURN", "0"))):
    tmp = [torch.empty(int((0.3 + 0.37 * ((7 * r + 3 * j) % 11)) * 2**30), dtype=torch.uint8, device=dev) for j in range(8)]
    hold += tmp[::2]
    del tmp
    torch.cuda.empty_cache()
src = (gs.generate_zipf_keys if mode == "msbzipf" else gs.generate_uniform_keys)(n, device=dev)
if os.environ.get("SLAB"):
    # the four arrays carved from ONE allocation (the reliably slow placement of the pairs sort, profiles/README.md round 3)
    mode = os.environ["SLAB"]
    pad = int(os.environ.get("SLAB_PAD", "0")) // 4            # elements between the carved arrays
    if not pairs:                                               # keys only: a and b carved from one allocation
        slab = torch.empty(2 * (n + pad), dtype=torch.int32, device=dev)
        a, b = slab[0:n], slab[n + pad: 2 * n + pad]
    elif mode in ("1", "2"):
        slab = torch.empty(4 * (n + pad), dtype=torch.int32, device=dev)
        parts = [slab[i * (n + pad): i * (n + pad) + n] for i in range(4)]
        a, va, b, vb = parts if mode == "1" else (parts[0], parts[2], parts[1], parts[3])    # 2: a, b, va, vb
    elif mode == "3":                                           # inputs in one slab, outputs in another
        s1, s2 = torch.empty(2 * (n + pad), dtype=torch.int32, device=dev), torch.empty(2 * (n + pad), dtype=torch.int32, device=dev)
        a, va, b, vb = s1[0:n], s1[n + pad: 2 * n + pad], s2[0:n], s2[n + pad: 2 * n + pad]
    elif mode == "4":                                           # keys in one slab, values in another
        s1, s2 = torch.empty(2 * (n + pad), dtype=torch.int32, device=dev), torch.empty(2 * (n + pad), dtype=torch.int32, device=dev)
        a, b, va, vb = s1[0:n], s1[n + pad: 2 * n + pad], s2[0:n], s2[n + pad: 2 * n + pad]
    else:                                                       # 5: a and va in one slab, b and vb separate allocations
        s1 = torch.empty(2 * (n + pad), dtype=torch.int32, device=dev)
        a, va = s1[0:n], s1[n + pad: 2 * n + pad]
        b, vb = torch.empty(n, dtype=torch.int32, device=dev), torch.empty(n, dtype=torch.int32, device=dev)
    if pairs: gs.generate_enumerated_values(n, device=dev, out=va)
    else: va = vb = None
else:
    a, b = torch.empty_like(src), torch.empty_like(src)
    va = gs.generate_enumerated_values(n, device=dev) if pairs else None
    vb = torch.empty_like(src) if pairs else None
nb = max(gs.lib.gs_lsb_temp_bytes(n, int(pairs)), gs.lib.gs_msb_temp_bytes(n, 0))
temp = torch.empty(nb, dtype=torch.uint8, device=dev)
libs = []
for name in args:
    L = C.CDLL(os.path.join(libdir, name))
    L.gs_profile_create.restype = C.c_void_p
    L.gs_lsb_sort_u32.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.POINTER(C.c_int), C.c_uint64, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p]
    L.gs_msb_sort_u32.argtypes = [C.c_void_p, C.c_size_t, C.c_void_p, C.c_void_p, C.c_uint64, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int]
    L.gs_profile_begin.argtypes = [C.c_void_p]
    L.gs_profile_read.argtypes = [C.c_void_p, C.POINTER(C.c_double), C.POINTER(C.c_uint64)]
    libs.append((name, L, C.c_void_p(L.gs_profile_create())))
res = {f"{i}:{name}": [] for i, (name, _, _) in enumerate(libs)}
stream = C.c_void_p(torch.cuda.current_stream().cuda_stream)
for rnd in range(5):
    for i, (name, L, prof) in enumerate(libs):
        a.copy_(src)
        if pairs:
            gs.generate_enumerated_values(n, device=dev, out=va)
        keys = (C.c_void_p * 2)(a.data_ptr(), b.data_ptr())
        vals = (C.c_void_p * 2)(va.data_ptr(), vb.data_ptr()) if pairs else None
        sel = C.c_int(0)
        ms0, c0 = (C.c_double * 10)(), (C.c_uint64 * 10)()
        L.gs_profile_read(prof, ms0, c0)
        L.gs_profile_begin(prof)
        if msb:
            e = L.gs_msb_sort_u32(temp.data_ptr(), nb, a.data_ptr(), None, n, b.data_ptr(), None, None, None, 0, stream, 0)
        else:
            e = L.gs_lsb_sort_u32(temp.data_ptr(), nb, keys, vals, C.byref(sel), n, 0, 32, 0, 0, stream)
        L.gs_profile_end()
        assert e == 0, e
        torch.cuda.synchronize()
        ms1, c1 = (C.c_double * 10)(), (C.c_uint64 * 10)()
        L.gs_profile_read(prof, ms1, c1)
        if rnd:
            if msb:
                res[f"{i}:{name}"].append(tuple(ms1[k] - ms0[k] for k in (0, 2, 3, 4, 5, 6)))
            else:
                res[f"{i}:{name}"].append(((ms1[0] - ms0[0]) / 4, (ms1[2] - ms0[2]) / 4))
med = lambda xs: sorted(xs)[len(xs) // 2]
for name, v in res.items():
    if msb:
        cols = [med([x[k] for x in v]) for k in range(6)]
        print(f"{name:28s} per sort: lsb_up {cols[0]:.3f} lsb_down {cols[1]:.3f} msb_hist {cols[2]:.3f} classify {cols[3]:.3f} partition {cols[4]:.3f} local {cols[5]:.3f}  sum {sum(cols):.3f} ms", flush=True)
        continue
    ups = sorted(x[0] for x in v); dss = sorted(x[1] for x in v)
    print(f"{name:28s} upsweep median {ups[len(ups)//2]:.4f}  downsweep median {dss[len(dss)//2]:.4f} min {dss[0]:.4f} max {dss[-1]:.4f} ms/launch", flush=True)
