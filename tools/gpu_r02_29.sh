#!/bin/bash
# kernel timeline of the CLI-default oversketch (k = 21, n = 200 000), parity tail
export TMPDIR=/tmp
mkdir -p gpurun_out
R=$PWD
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py -q -x -k "not c4_50gbase and not c5_batch" ) 2>&1 | tail -2
cd /tmp
rocprofv3 --kernel-trace -d $R/gpurun_out/r02y_n200k -o t --output-format csv -- python $R/bench.py --no-cpu-baseline --no-extras --k 21 --n 200000 --steps 2 --warmup 1 > $R/gpurun_out/r02y_n200k.log 2>&1
cd $R
python tools/kernel_timeline.py gpurun_out/r02y_n200k --min-ms 0.02 | tail -60 | tee gpurun_out/r02y_n200k_timeline.txt
FH_TRACE=1 python bench.py --no-cpu-baseline --no-extras --k 21 --n 200000 --steps 1 --warmup 1 2>&1 | tail -40 | cut -c1-200
