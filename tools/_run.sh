python bench.py --gpus 2 --rehearse-on-one-gpu --log2n 26 --steps 3 --warmup 1 --verify --no-cpu-baseline 2>&1 | grep -E "^\{" | cut -c1-700
echo rc=$?
python bench.py --force-sharded --one-rank-rccl --log2n 28 --steps 3 --warmup 1 --verify --no-cpu-baseline 2>&1 | grep -E "^\{" | cut -c1-400
