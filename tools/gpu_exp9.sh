#!/bin/bash
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/c3_stats -o stats --output-format csv -- python $R/bench.py --steps 3 --warmup 1 --no-cpu-baseline --k 31 --n 2000000 > $R/gpurun_out/c3_stats.log 2>&1
cd $R
tail -1 gpurun_out/c3_stats.log | cut -c1-300
head -14 gpurun_out/c3_stats/stats_kernel_stats.csv | cut -c1-200
