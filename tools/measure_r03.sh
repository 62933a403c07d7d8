# The whole measurement set of a build (round 3): the driver's bench line (with `also`), its rocprofv3 kernel stats,
# PMC traffic (separate FETCH_SIZE / WRITE_SIZE passes) for LSB keys, LSB pairs, MSB uniform and MSB Zipf.
# usage (on the GPU box): GS_MEASURE_TAG=r03_v1 bash tools/measure_r03.sh
set -e
export TMPDIR=/tmp
R=$PWD
O=$R/gpurun_out/${GS_MEASURE_TAG:-r03_v1}; mkdir -p $O
python bench.py --steps 20 --warmup 3 > $O/bench.json 2> $O/bench.err
echo bench-done
prof() { # name, bench args...
  local name=$1; shift
  (cd /tmp && rocprofv3 --kernel-trace --stats --output-format csv -d $O/prof_$name -- python3 $R/bench.py "$@" --steps 3 --warmup 1 --no-cpu-baseline --no-also > $O/profiled_bench_$name.json 2> $O/prof_$name.err)
}
prof lsb
prof pairs --pairs
prof msb_zipf --algo msb --dist zipf
prof msb --algo msb
echo prof-done
pmc() { # name, counter, bench args...
  local name=$1; shift; local ctr=$1; shift
  (cd /tmp && rocprofv3 --pmc $ctr --kernel-trace --output-format csv -d $O/pmc_$name -- python3 $R/bench.py "$@" --steps 2 --warmup 1 --no-cpu-baseline --no-also > /dev/null 2> $O/pmc_$name.err)
}
pmc fetch FETCH_SIZE
pmc write WRITE_SIZE
pmc fetch_pairs FETCH_SIZE --pairs
pmc write_pairs WRITE_SIZE --pairs
pmc fetch_msb FETCH_SIZE --algo msb
pmc write_msb WRITE_SIZE --algo msb
pmc fetch_msb_zipf FETCH_SIZE --algo msb --dist zipf
pmc write_msb_zipf WRITE_SIZE --algo msb --dist zipf
python tools/make_pmc_traffic.py $O/pmc_fetch $O/pmc_write 30 keys lsb_downsweep r03 > $O/pmc_traffic.txt
python tools/make_pmc_traffic.py $O/pmc_fetch_pairs $O/pmc_write_pairs 30 pairs lsb_downsweep r03 >> $O/pmc_traffic.txt
python tools/make_pmc_traffic_msb.py uniform $O/pmc_fetch_msb $O/pmc_write_msb 3 zipf $O/pmc_fetch_msb_zipf $O/pmc_write_msb_zipf 3 >> $O/pmc_traffic.txt
cp profiles/pmc_traffic.json profiles/pmc_traffic_pairs.json profiles/pmc_traffic_msb.json $O/
echo pmc-done
find $O -name "*kernel_trace.csv" -delete; find $O -name "*counter_collection.csv" -delete; find $O -name "*agent_info.csv" -delete
ls $O
