#!/bin/bash
# round 2, first GPU call: the whole -m gpu suite (incl. the new full-size C3/C4/C5 and sharded-input tests) and the bench
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 3000 python -m pytest tests -m gpu -x -q --durations=15 ) > gpurun_out/r02a_pytest.log 2>&1
tail -30 gpurun_out/r02a_pytest.log
( time python bench.py --steps 5 --warmup 1 ) > gpurun_out/r02a_bench.log 2>&1
tail -3 gpurun_out/r02a_bench.log
nproc; free -g | head -2; df -h /dev/shm /tmp | tail -2
