#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 ) > gpurun_out/r02e_pytest.log 2>&1
tail -22 gpurun_out/r02e_pytest.log
python tools/e2e_gz.py 2>&1 | tee gpurun_out/r02e_e2e_gz.txt
python tools/e2e_files.py 2>&1 | tail -8 | tee gpurun_out/r02e_e2e_files.txt
