set -e
python tools/msb_census.py 30 zipf 2>&1 | grep -v amdgpu.ids | grep level
python tools/kprof.py 30 keys msb zipf 2>&1 | grep -v amdgpu.ids
GS_MSB_PIVOT=0 python tools/kprof.py 30 keys msb zipf 2>&1 | grep -v amdgpu.ids | tail -1
python tools/kprof.py 30 keys lsb zipf 2>&1 | grep -v amdgpu.ids | tail -1
python tools/kprof.py 30 keys msb uniform 2>&1 | grep -v amdgpu.ids | tail -1
timeout -k 10 600 python -m pytest tests/test_msb_gpu.py tests/test_fuzz_gpu.py -x -q 2>&1 | tail -3
python tools/dist_bench.py 28 2>&1 | grep -v amdgpu.ids
