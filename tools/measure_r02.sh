# The whole measurement set of a build (round 2): bench lines of every configuration, rocprofv3 kernel stats of the
# LSB / pairs / MSB-Zipf benches, PMC traffic (separate FETCH_SIZE / WRITE_SIZE passes) for keys and pairs.
# usage (on the GPU box): GS_MEASURE_TAG=r02_v1 bash tools/measure_r02.sh
set -e
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${GS_MEASURE_TAG:-r02_v1}; mkdir -p $O
python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
echo bench-done
python bench.py --pairs --no-cpu-baseline > $O/bench_pairs.json 2>> $O/bench.err
python bench.py --algo msb --no-cpu-baseline > $O/bench_msb.json 2>> $O/bench.err
python bench.py --algo msb --dist zipf --no-cpu-baseline > $O/bench_msb_zipf.json 2>> $O/bench.err
python bench.py --algo msb --pairs --no-cpu-baseline > $O/bench_msb_pairs.json 2>> $O/bench.err
python bench.py --dist zipf --no-cpu-baseline > $O/bench_lsb_zipf.json 2>> $O/bench.err
python bench.py --force-sharded --verify --no-cpu-baseline > $O/bench_sharded1.json 2>> $O/bench.err
(cd gpu-sort_amd/drivers && ./msb_sharded --log2n 30 --reps 3 2>/dev/null | tail -1) > $O/msb_sharded_cpp_one_rank.json
echo variants-done
prof() { # name, bench args...
  local name=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 $R/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline > $O/profiled_bench_$name.json 2> $O/prof_$name.err)
}
prof lsb
prof pairs --pairs
prof msb_zipf --algo msb --dist zipf
prof msb --algo msb
echo prof-done
pmc() { # name, counter, bench args...
  local name=$1; shift; local ctr=$1; shift
  (cd /tmp && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$name -- python3 $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline > /dev/null 2> $O/pmc_$name.err)
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc fetch_pairs FETCH_SIZE --pairs
pmc write_pairs WRITE_SIZE --pairs
python tools/make_pmc_traffic.py $O/pmc_fetch $O/pmc_write 30 keys lsb_downsweep r02 > $O/pmc_traffic.txt
python tools/make_pmc_traffic.py $O/pmc_fetch_pairs $O/pmc_write_pairs 30 pairs lsb_downsweep r02 >> $O/pmc_traffic.txt
cp profiles/pmc_traffic.json profiles/pmc_traffic_pairs.json $O/
echo pmc-done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
ls $O
