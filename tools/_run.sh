python tools/kprof.py 30 2>&1 | grep -E "sum"
python tools/kprof.py 30 pairs 2>&1 | grep -E "sum"
python tools/kprof.py 30 keys msb 2>&1 | grep -E "sum|partition"
python tools/kprof.py 30 pairs msb 2>&1 | grep -E "sum|partition"
python tools/kprof.py 30 keys msb zipf 2>&1 | grep -E "sum"
python tools/wide_bench.py 2>&1 | grep -v amdgpu | tail -5
timeout -k 10 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -3
