python tools/kprof.py 30 keys msb zipf 2>&1 | grep -E "sum|histogram|upsweep"
python tools/kprof.py 30 keys msb uniform 2>&1 | grep -E "sum|histogram|upsweep"
python tools/kprof.py 30 keys lsb zipf 2>&1 | grep -E "sum|upsweep"
python tools/kprof.py 30 keys lsb uniform 2>&1 | grep -E "sum|upsweep"
python tools/dist_bench.py 28 2>&1 | grep -v amdgpu.ids
