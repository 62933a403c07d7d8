timeout -k 10 900 python -m pytest tests/test_msb_gpu.py -x -q -k "config4 or large" 2>&1 | tail -3
python bench.py --algo msb --dist zipf --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null
python bench.py --algo msb --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null
python bench.py --dist zipf --no-cpu-baseline --steps 10 --warmup 2 2>/dev/null
