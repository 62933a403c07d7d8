#!/bin/bash
# One sample of "what kind of box is this": tools/box_probe.py (facts + LSB keys / pairs kernel times) and one rocprofv3 --pmc pass
# over a pairs sort (address-translation counters of the pairs downsweep).  Writes gpurun_out/box_probe_<time>.json / .pmc.txt.
mkdir -p gpurun_out
export TMPDIR=/tmp
R=${GRAFT_REPO_ROOT:-$PWD}
T=$(date +%H%M%S)
timeout -k 10 200 python3 $R/tools/box_probe.py 2>/dev/null | tail -1 > $R/gpurun_out/box_probe_$T.json
rm -rf $R/gpurun_out/pmc_probe
timeout -k 10 150 rocprofv3 --pmc TCP_UTCL1_TRANSLATION_MISS_sum TCP_UTCL1_TRANSLATION_MISS_UNDER_MISS_sum TCP_UTCL1_REQUEST_sum TCP_UTCL1_STALL_UTCL2_REQ_OUT_OF_CREDITS_sum \
    --output-format csv -d $R/gpurun_out/pmc_probe -- python3 $R/tools/one_sort.py 30 pairs 1 > /dev/null 2>&1
( cd $R && python3 tools/pmc_summary.py gpurun_out/pmc_probe | grep downsweep ) > $R/gpurun_out/box_probe_$T.pmc.txt 2>&1
rm -rf $R/gpurun_out/pmc_probe
( timeout -k 10 120 $R/tools/micro/lsb_floor 30 8 probe 2>&1 | grep -E "pattern|real" ) > $R/gpurun_out/box_probe_$T.patterns.txt
( timeout -k 10 60 $R/tools/micro/stream 2>&1 | tail -12 ) > $R/gpurun_out/box_probe_$T.stream.txt
cat $R/gpurun_out/box_probe_$T.patterns.txt | cut -c1-150
cat $R/gpurun_out/box_probe_$T.pmc.txt
python3 - <<PY
import json
d = json.load(open("$R/gpurun_out/box_probe_$T.json"))
print("box", d["unique_id"].split("Unique ID:")[-1].strip(), "keys ds", d["ms_per_launch"]["keys"]["lsb_downsweep"], "pairs ds", d["ms_per_launch"]["pairs"]["lsb_downsweep"])
PY
