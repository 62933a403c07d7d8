#!/bin/bash
# usage: tools/pmc.sh TAG "counters..." -- args for one_sort.py     (one rocprofv3 --pmc pass)
TAG=$1; shift; CTRS=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_$TAG
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d gpurun_out/pmc_$TAG -- python3 tools/one_sort.py "$@" > gpurun_out/pmc_$TAG.log 2>&1
