#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_bgzf_device.py -q -x --durations=5 ) > gpurun_out/r02l_pytest.log 2>&1
tail -8 gpurun_out/r02l_pytest.log
for mode in "" "--noisy"; do
  for lvl in 1 6; do
    FH_TRACE=1 python tools/bgzf_device_file.py $mode --level $lvl --reps 3 2>&1 | grep -v "^\[fh\]" | tail -4
    FINCH_DEVICE_INFLATE=0 python tools/bgzf_device_file.py $mode --level $lvl --reps 3 2>&1 | tail -1
  done
done | tee gpurun_out/r02l_bgzf.txt
