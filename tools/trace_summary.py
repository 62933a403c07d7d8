"""Per-dispatch kernel durations from a rocprofv3 --kernel-trace run, in launch order.
python tools/trace_summary.py <dir> [skip_first_n]"""
import csv, glob, sys, re
d = sys.argv[1]
files = glob.glob(d + "/**/*kernel_trace.csv", recursive=True)
rows = []
for f in files:
    with open(f) as fh:
        for r in csv.DictReader(fh):
            rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"], r.get("Grid_Size", "")))
rows.sort()
def short(n):
    n = re.sub(r"\(.*", "", n)
    n = n.replace("void gs::", "").replace("gs::", "")
    return n[:70]
skip = int(sys.argv[2]) if len(sys.argv) > 2 else 0
for s, e, n, g in rows[skip:]:
    if "gs::" in n or "gs" in n[:8]:
        print(f"{(e - s) / 1e3:10.1f} us  grid {g:>10}  {short(n)}")
