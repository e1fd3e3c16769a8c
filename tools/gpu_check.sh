#!/bin/bash
# one GPU-box session: all GPU tests, smoke, default bench.  Logs -> gpurun_out/
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== pytest gpu =="
timeout 1500 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 | tee gpurun_out/pytest_gpu.log
echo "== smoke =="
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -3 | tee gpurun_out/smoke.log
echo "== bench =="
timeout 900 python bench.py 2>&1 | tail -1 | tee gpurun_out/bench.log
