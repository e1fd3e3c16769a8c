#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests/test_gpu_parity.py tests/test_gpu_errors.py tests/test_gpu_fuzz.py tests/test_gpu_host_layer.py -m gpu -x -q --durations=8 ) > gpurun_out/r02c_pytest.log 2>&1
tail -25 gpurun_out/r02c_pytest.log
for k in 21 31 33 48 64; do python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-extras --k $k 2>&1 | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('k=$k', round(d['value']/1e9,1), 'Gbases/s', d['ms_per_step'], 'ms', d['roofline']['achieved'], 'GB/s kernel')"; done 2>&1 | tee gpurun_out/r02c_wide_bench.txt
