"""One line per call: what this box is (GPU unique id, clocks, power cap, partition modes) next to the times that were seen to
differ between boxes -- the LSB downsweep for keys and for pairs at 2^30 ('slow kind of box').
python tools/box_probe.py >> gpurun_out/box_probe.jsonl"""
import json, os, subprocess, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))


def sh(cmd):
    try:
        return subprocess.run(cmd, shell=True, capture_output=True, text=True, timeout=30).stdout.strip()
    except Exception as e:
        return "error: %r" % e


facts = {"time": time.strftime("%H:%M:%S"),
         "unique_id": sh("rocm-smi --showuniqueid 2>/dev/null | grep -i 'unique' | head -2"),
         "clocks": sh("rocm-smi --showclocks 2>/dev/null | grep -E 'sclk|mclk|fclk|socclk' | head -8"),
         "power": sh("rocm-smi --showmaxpower --showpower 2>/dev/null | grep -E 'Power|power' | head -6"),
         "perf": sh("rocm-smi --showperflevel 2>/dev/null | grep -i perf | head -2"),
         "partition": sh("rocm-smi --showcomputepartition --showmemorypartition 2>/dev/null | grep -i partition | head -4"),
         "temp": sh("rocm-smi --showtemp 2>/dev/null | grep -i 'junction\\|hbm\\|edge' | head -6"),
         "cpu": sh("grep -m1 'model name' /proc/cpuinfo"), "kernel": sh("uname -r")}
import torch
import gpu_sort_amd as gs
dev = torch.device("cuda:0")
n = 1 << 30
out = {}
for pairs in (False, True):
    src = gs.generate_uniform_keys(n, device=dev)
    a, b = torch.empty_like(src), torch.empty_like(src)
    va = gs.generate_enumerated_values(n, device=dev) if pairs else None
    vb = torch.empty_like(src) if pairs else None
    nb = gs.lib.gs_lsb_temp_bytes(n, int(pairs))
    temp = torch.empty(nb, dtype=torch.uint8, device=dev)
    prof = gs.KernelProfile()
    for r in range(4):
        a.copy_(src)
        if r == 1:
            prof.__enter__()
        dk = gs.DoubleBuffer(a, b)
        if pairs:
            gs.DeviceRadixSort.SortPairs(temp, nb, dk, gs.DoubleBuffer(va, vb), n, key_type=gs.GS_KEY_U32)
        else:
            gs.DeviceRadixSort.SortKeys(temp, nb, dk, n, key_type=gs.GS_KEY_U32)
    prof.__exit__()
    torch.cuda.synchronize()
    k = prof.read()
    out["pairs" if pairs else "keys"] = {g: round(v[0] / v[1], 4) for g, v in k.items()}
    out[("pairs" if pairs else "keys") + "_ptr_mod_2MiB"] = [hex(t.data_ptr() % (2 << 20)) for t in ([a, b] + ([va, vb] if pairs else []))]
    del src, a, b, va, vb, temp
    torch.cuda.empty_cache()
facts["ms_per_launch"] = out
facts["clocks_after"] = sh("rocm-smi --showclocks 2>/dev/null | grep -E 'sclk|mclk' | head -4")
print(json.dumps(facts))
