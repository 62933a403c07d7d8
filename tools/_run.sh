cd gpu-sort_amd/drivers
timeout -k 10 120 ./msb_sharded --log2n 24 --reps 2; echo rc=$?
timeout -k 10 120 ./msb_sharded --log2n 24 --reps 2 --pairs; echo rc=$?
timeout -k 10 200 ./msb_sharded --log2n 30 --reps 3; echo rc=$?
cd ../.. && timeout -k 10 600 python -m pytest tests/test_drivers_gpu.py -x -q 2>&1 | tail -3
