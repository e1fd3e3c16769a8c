#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 600 python -m pytest tests/test_gpu_bgzf_device.py -q -x ) > gpurun_out/r02n_pytest.log 2>&1
tail -5 gpurun_out/r02n_pytest.log
for mode in "" "--noisy"; do
  for lvl in 1 6; do
    FH_TRACE=1 timeout 300 python tools/bgzf_device_file.py $mode --level $lvl --reps 3 2>&1 | grep -v "^\[fh\]" | tail -3
  done
done | tee gpurun_out/r02n_bgzf.txt
