#!/bin/bash
# usage: tools/pmc2.sh TAG "counters..." script args...   (one rocprofv3 --pmc pass of an arbitrary tools/ script)
TAG=$1; shift; CTRS=$1; shift
export TMPDIR=/tmp
mkdir -p gpurun_out/pmc_$TAG
rocprofv3 --pmc $CTRS --kernel-trace --output-format csv -d gpurun_out/pmc_$TAG -- python3 "$@" > gpurun_out/pmc_$TAG.log 2>&1
