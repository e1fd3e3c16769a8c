#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( for mode in "" "--noisy"; do for lvl in 1 6; do FH_TRACE=1 timeout 600 python tools/gz_parallel_file.py $mode --level $lvl 2>&1 | grep "threads:\|text as\|read .*ms\|text pump" | awk '!seen[$0]++'; done; done ) > gpurun_out/r02s_gz_parallel.txt
grep "threads:\|text as" gpurun_out/r02s_gz_parallel.txt
grep -B4 "16 threads" gpurun_out/r02s_gz_parallel.txt | grep "pump\|read" | tail -12
( time timeout 600 python -m pytest tests/test_gpu_bgzf_device.py -q -x ) 2>&1 | tail -4
