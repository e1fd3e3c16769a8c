#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 600 python -m pytest tests/test_gpu_bgzf_device.py tests/test_gpu_host_layer.py -q -x ) 2>&1 | tail -6
( for mode in "" "--noisy"; do for lvl in 1 6; do FH_TRACE=1 timeout 600 python tools/gz_parallel_file.py $mode --level $lvl 2>&1 | grep "threads:\|text as\|read .*ms\|text pump" | awk '!seen[$0]++'; done; done ) | tee gpurun_out/r02q_gz_parallel.txt | grep "threads:\|text as"
