#!/bin/bash
# Builds gpu-sort_amd/lib/gsvariant_<name>.so (not libgpusort_*: that prefix belongs to libgpusort_rccl.so) from the current sources with extra -D flags, in its own
# object directory (for tools/ab.sh):  tools/build_variant.sh <name> "<flags>"
name="$1"; flags="$2"
cd "$(dirname "$0")/../gpu-sort_amd/csrc" || exit 1
obj=/tmp/gs_variant_$name; mkdir -p $obj
rm -f $obj/*.o          # a failed compile must not link a stale object
for f in *.hip; do
  /opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -I../../include -I. -Wno-unused-result -Wno-unused-value $flags -c $f -o $obj/${f%.hip}.o &
done
wait
/opt/rocm/bin/hipcc --offload-arch=gfx950 -shared -fPIC -o ../lib/gsvariant_$name.so $obj/*.o && echo built gsvariant_$name.so
