#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( time python bench.py --steps 5 --warmup 1 ) > gpurun_out/r02g_bench.log 2>&1
grep '^{"metric' gpurun_out/r02g_bench.log > gpurun_out/r02g_bench_10G.json
tail -3 gpurun_out/r02g_bench.log
FH_FULL_GBASES_C3=2 FH_FULL_GBASES=1 FH_FULL_GBASES_C4=1 FH_C5_FILES=64 timeout 900 python -m pytest tests/test_gpu_full_size.py -m gpu -x -q 2>&1 | tail -3
