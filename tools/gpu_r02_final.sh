#!/bin/bash
# end-of-round evidence: full -m gpu suite, the driver's bench line (with extras) + rocprofv3 kernel stats + PMC passes,
# compressed-file rates
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1800 python -m pytest tests -m gpu -q -x --durations=6 ) > gpurun_out/r02z_pytest.log 2>&1
tail -12 gpurun_out/r02z_pytest.log
timeout 1500 bash tools/gpu_bench_full.sh r02z 2>&1 | tail -5
( for mode in "" "--noisy"; do timeout 600 python tools/gz_parallel_file.py $mode --level 6 2>&1 | grep "threads:\|text as"; done ) | tee gpurun_out/r02z_gz_parallel.txt
( for mode in "" "--noisy"; do timeout 300 python tools/bgzf_device_file.py $mode --level 6 --reps 3 2>&1 | tail -1; FINCH_DEVICE_INFLATE=0 timeout 300 python tools/bgzf_device_file.py $mode --level 6 --reps 3 2>&1 | tail -1; done ) | tee gpurun_out/r02z_bgzf.txt
