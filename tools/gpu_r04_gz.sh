#!/bin/bash
# round 4: plain gzip inflated on the device -- tests, then timings and a kernel trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_gzip_device.py tests/test_gpu_bgzf_device.py -x -q -m gpu 2>&1 | tail -30 | tee gpurun_out/r04_gz_tests.txt
( for i in 1 2; do timeout 600 python tools/gz_bench.py 1000000 1 | tail -2; done
  GZ_ONLY=device timeout 600 python tools/gz_bench.py 1000000 6 | tail -1
  GZ_ONLY=device timeout 600 python tools/gz_bench.py 1000000 9 | tail -1
  FINCH_DEVICE_GZIP=1 timeout 600 python tools/gz_bench.py 4000000 1 noisy 2>&1 | tail -3
  FINCH_DEVICE_GZIP=1 timeout 600 python tools/gz_bench.py 4000000 6 noisy 2>&1 | tail -3
  FH_TRACE=1 GZ_ONLY=device timeout 600 python tools/gz_bench.py 1000000 1 2>&1 | grep "gzip" | tail -4 ) 2>&1 | tee gpurun_out/r04_gz_bench.txt
rm -rf gpurun_out/gz_trace; GZ_ONLY=device timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gz_trace -o gz -- python tools/gz_bench.py 1000000 1 > /dev/null 2>&1
python tools/trace_busy.py gpurun_out/gz_trace --tail 0.25 --chain 34 2>&1 | grep -v "copyBuffer\|fillBuffer" | tee gpurun_out/r04_gz_chain.txt
rm -rf gpurun_out/gz_trace
