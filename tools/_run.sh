GS_LIB_PATH=$PWD/gpu-sort_amd/lib/libgpusort_smallup.so python tools/kprof.py 30 2>&1 | grep -E "upsweep|other|sum"
