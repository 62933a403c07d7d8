"""Build profiles/pmc_traffic.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of bench.py.

HBM bytes per launch of the dominant kernel, corrected as MI355X_MICROARCH.md (HBM / rocprofv3)
prescribes: separate --pmc passes; FETCH_SIZE is in KiB and on gfx950 reports half of a streaming
read (calibrated here on lsb_upsweep, a pure read of 4*n bytes); WRITE_SIZE (KiB) is exact.
usage: python tools/make_pmc_traffic.py FETCH_DIR WRITE_DIR LOG2N [pairs] [kernel-substring] [tag]
"""
import csv, glob, json, sys, collections

def per_kernel(d, counter):
    acc = collections.defaultdict(list)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return acc

fetch_dir, write_dir, log2n = sys.argv[1], sys.argv[2], int(sys.argv[3])
pairs = len(sys.argv) > 4 and sys.argv[4] == "pairs"
want = sys.argv[5] if len(sys.argv) > 5 else "lsb_downsweep"
tag = sys.argv[6] if len(sys.argv) > 6 else "r01"
F, W = per_kernel(fetch_dir, "FETCH_SIZE"), per_kernel(write_dir, "WRITE_SIZE")
def pick(acc, sub):
    for k, v in acc.items():
        if sub in k and "true, true" not in k and "false, true" not in k:
            return k, sum(v) / len(v), len(v)
    return None, 0.0, 0
kname, fkb, fl = pick(F, want)
_, wkb, wl = pick(W, want)
_, up_fkb, _ = pick(F, "lsb_upsweep")
n = 1 << log2n
calib = (4.0 * n) / (up_fkb * 1024.0) if up_fkb else 2.0       # upsweep reads exactly 4n bytes
hbm = calib * fkb * 1024.0 + wkb * 1024.0
alg = (16 if pairs else 8) * n
out = {"kernel": want, "kernel_name": kname, "log2n": log2n, "pairs": pairs,
       "fetch_size_kib_per_launch": fkb, "write_size_kib_per_launch": wkb, "launches_sampled": [fl, wl],
       "fetch_correction": round(calib, 4),
       "fetch_correction_note": "gfx950 FETCH_SIZE under-reports streaming reads; factor calibrated on lsb_upsweep (reads exactly 4n bytes) in the same run",
       "hbm_bytes_per_launch": hbm, "algorithmic_bytes_per_launch": alg, "traffic_over_algorithmic": round(hbm / alg, 4),
       "collected_with": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --steps 2 --warmup 1 --no-cpu-baseline" + (" --pairs" if pairs else "")}
name = "profiles/pmc_traffic.json" if not pairs else "profiles/pmc_traffic_pairs.json"
json.dump(out, open(name, "w"), indent=1)
print(json.dumps(out))
