"""Timeline of the LAST sharded sort of a rocprofv3 --kernel-trace run: every kernel (ours and the collective
library's) with start and end relative to the step's first kernel, so that overlap is visible.
python tools/trace_timeline.py <dir>"""
import csv, glob, sys, re
rows = []
for f in glob.glob(sys.argv[1] + "/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        rows.append((int(r["Start_Timestamp"]), int(r["End_Timestamp"]), r["Kernel_Name"]))
rows.sort()
# the last first-pass upsweep marks the start of the last step
starts = [i for i, r in enumerate(rows) if "lsb_upsweep_kernel" in r[2]]
i0 = starts[int(sys.argv[2]) if len(sys.argv) > 2 else -1]
i1 = starts[starts.index(i0) + 1] if starts.index(i0) + 1 < len(starts) else len(rows)
t0 = rows[i0][0]
def short(n):
    n = re.sub(r"\(.*", "", n).replace("void gs::", "").replace("gs::", "")
    return n[:60]
for s, e, n in rows[i0:i1]:
    if (e - s) > 20000:       # > 20 us
        print(f"{(s - t0) / 1e6:8.3f} -> {(e - t0) / 1e6:8.3f} ms  ({(e - s) / 1e6:6.3f})  {short(n)}")
