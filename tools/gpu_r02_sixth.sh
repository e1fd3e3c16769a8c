#!/bin/bash
# bench with extras + kernel timelines of the 2 M-hash sketch with and without the sampling pre-pass
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$GRAFT_REPO_ROOT
( time python bench.py --steps 5 --warmup 1 ) > gpurun_out/r02f_bench.log 2>&1
tail -4 gpurun_out/r02f_bench.log | head -1 > gpurun_out/r02f_bench_10G.json
cd /tmp
for v in sample nosample; do
  if [ $v = nosample ]; then export FH_NO_SAMPLE=1; else unset FH_NO_SAMPLE; fi
  rocprofv3 --kernel-trace -d $R/gpurun_out/r02f_c3_$v -o t --output-format csv -- python $R/bench.py --steps 1 --warmup 1 --no-cpu-baseline --no-extras --k 31 --n 2000000 > $R/gpurun_out/r02f_c3_$v.log 2>&1
  python $R/tools/kernel_timeline.py $R/gpurun_out/r02f_c3_$v > $R/gpurun_out/r02f_c3_timeline_$v.txt 2>&1
  tail -30 $R/gpurun_out/r02f_c3_timeline_$v.txt
done
