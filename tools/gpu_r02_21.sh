#!/bin/bash
# lane-parallel symbol loop of the device-side BGZF inflate: tests, then A/B against the serial loop
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 600 python -m pytest tests/test_gpu_bgzf_device.py -q -x ) > gpurun_out/r02r_pytest.log 2>&1
tail -8 gpurun_out/r02r_pytest.log
for mode in "" "--noisy"; do
  for lvl in 1 6; do
    timeout 300 python tools/bgzf_device_file.py $mode --level $lvl --reps 3 2>&1 | tail -1
    FH_BGZF_SERIAL=1 timeout 300 python tools/bgzf_device_file.py $mode --level $lvl --reps 3 2>&1 | tail -1 | sed 's/^/  serial loop: /'
  done
done | tee gpurun_out/r02r_bgzf_par.txt
