#!/bin/bash
# A/B two builds of the library in ONE gpurun call (box-to-box variance is +-5-10 %):
#   tools/ab.sh "<kprof args>" libA.so libB.so [rounds]
args="$1"; A="$2"; B="$3"; R="${4:-2}"
for r in $(seq 1 $R); do
  for L in "$A" "$B"; do
    echo "== $(basename $L) :: $args"
    GS_LIB_PATH=$PWD/gpu-sort_amd/lib/$L python tools/kprof.py $args || exit 1
  done
done
