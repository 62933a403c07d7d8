set -e
export TMPDIR=/tmp
O=gpurun_out/${GS_MEASURE_TAG:-v9}; mkdir -p $O
python bench.py > $O/bench.json 2> $O/bench.err
echo bench-done
python bench.py --pairs --no-cpu-baseline > $O/bench_pairs.json 2>> $O/bench.err
python bench.py --algo msb --no-cpu-baseline > $O/bench_msb.json 2>> $O/bench.err
python bench.py --algo msb --dist zipf --no-cpu-baseline > $O/bench_msb_zipf.json 2>> $O/bench.err
python bench.py --algo msb --pairs --no-cpu-baseline > $O/bench_msb_pairs.json 2>> $O/bench.err
python bench.py --dist zipf --no-cpu-baseline > $O/bench_lsb_zipf.json 2>> $O/bench.err
python bench.py --force-sharded --verify --no-cpu-baseline > $O/bench_sharded1.json 2>> $O/bench.err
echo variants-done
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_lsb -- python3 $OLDPWD/bench.py --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$O/profiled_bench.json 2> $OLDPWD/$O/prof_lsb.err)
echo prof-lsb-done
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_msb -- python3 $OLDPWD/bench.py --algo msb --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$O/profiled_bench_msb.json 2> $OLDPWD/$O/prof_msb.err)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_sh -- python3 $OLDPWD/bench.py --force-sharded --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$O/profiled_bench_sharded1.json 2> $OLDPWD/$O/prof_sh.err)
(cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $OLDPWD/$O/prof_msbp -- python3 $OLDPWD/bench.py --algo msb --pairs --steps 3 --warmup 1 --no-cpu-baseline > $OLDPWD/$O/profiled_bench_msb_pairs.json 2> $OLDPWD/$O/prof_msbp.err)
echo prof-done
find $O -name "*kernel_trace.csv" -delete
ls $O
