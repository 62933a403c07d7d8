#!/bin/bash
# usage: tools/sweep_env.sh "v1 v2 ..." -- runs kprof with GS_DEBUG=value for each
for v in $1; do echo "GS_VALU_ROUNDS=$v"; GS_VALU_ROUNDS=$v python tools/kprof.py 30 2>&1 | grep -E "downsweep|sum"; done
