"""Build profiles/pmc_traffic_msb.json from two rocprofv3 --pmc passes (FETCH_SIZE, WRITE_SIZE) of `bench.py --algo msb`
(uniform and Zipf keys): HBM bytes per SORT for every kernel group of the MSB path, corrected as MI355X_MICROARCH.md
prescribes (separate passes; FETCH_SIZE in KiB, doubled on gfx950 by the calibration on lsb_upsweep -- a pure read of
4n bytes in the same run; WRITE_SIZE exact).
usage: python tools/make_pmc_traffic_msb.py TAG FETCH_DIR WRITE_DIR SORTS_IN_RUN [TAG FETCH_DIR WRITE_DIR SORTS ...]   (TAG: uniform | zipf)"""
import csv, glob, json, sys, collections

GROUPS = (("lsb_upsweep", "lsb_upsweep_kernel"), ("lsb_downsweep", "lsb_downsweep_kernel"), ("msb_histogram", "msb_upsweep_kernel"),
          ("msb_partition", "msb_scatter_kernel"), ("msb_local_sort", "msb_local_sort_kernel"))


def per_group(d, counter):
    acc = collections.defaultdict(float)
    cnt = collections.defaultdict(int)
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        for r in csv.DictReader(open(f)):
            if r["Counter_Name"] != counter:
                continue
            for g, sub in GROUPS:
                if sub in r["Kernel_Name"]:
                    acc[g] += float(r["Counter_Value"])
                    cnt[g] += 1
    return acc, cnt


out = {"log2n": 30, "collected_with": "rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE (separate passes) --kernel-trace -- python3 bench.py --algo msb [--dist zipf] --steps 2 --warmup 1 --no-cpu-baseline --no-also",
       "fetch_correction_note": "gfx950 FETCH_SIZE under-reports streaming reads; factor calibrated on lsb_upsweep (reads exactly 4n bytes per launch) in the same run"}
a = sys.argv[1:]
for i in range(0, len(a), 4):
    tag, fd, wd, sorts = a[i], a[i + 1], a[i + 2], int(a[i + 3])
    F, fc = per_group(fd, "FETCH_SIZE")
    W, wc = per_group(wd, "WRITE_SIZE")
    n = 1 << 30
    calib = (4.0 * n * fc["lsb_upsweep"]) / (F["lsb_upsweep"] * 1024.0) if F["lsb_upsweep"] else 2.0
    out[tag] = {"fetch_correction": round(calib, 4), "sorts_sampled": sorts, "hbm_bytes_per_sort": {}, "launches_per_sort": {}}
    for g, _ in GROUPS:
        if fc[g] == 0 and wc[g] == 0:
            continue
        out[tag]["hbm_bytes_per_sort"][g] = (calib * F[g] * 1024.0 + W[g] * 1024.0) / sorts
        out[tag]["launches_per_sort"][g] = fc[g] / sorts
json.dump(out, open("profiles/pmc_traffic_msb.json", "w"), indent=1)
print(json.dumps(out))
