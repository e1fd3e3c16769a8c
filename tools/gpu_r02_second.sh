#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests/test_gpu_host_layer.py tests/test_gpu_multirank.py -m gpu -x -q --durations=8 ) > gpurun_out/r02b_pytest.log 2>&1
tail -15 gpurun_out/r02b_pytest.log
bash tools/gpu_bench_full.sh r02a c2_k21_n1000 > gpurun_out/r02a_full.log 2>&1; tail -4 gpurun_out/r02a_full.log
bash tools/gpu_bench_full.sh r02a_k31 c2_k31_n1000 --k 31 > gpurun_out/r02a_k31_full.log 2>&1; tail -4 gpurun_out/r02a_k31_full.log
