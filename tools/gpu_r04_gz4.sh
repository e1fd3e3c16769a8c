#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
( GZ_ONLY=device timeout 600 python tools/gz_bench.py 1000000 1
  FINCH_GZIP_PIECE=4194304 GZ_ONLY=device timeout 600 python tools/gz_bench.py 1000000 1
  GZ_ONLY=device timeout 600 python tools/gz_bench.py 1000000 6
  FINCH_DEVICE_GZIP=1 GZ_ONLY=device timeout 600 python tools/gz_bench.py 4000000 1 noisy 2>&1 | tail -3 ) 2>&1 | tee gpurun_out/r04_gz_bench4.txt
rm -rf gpurun_out/gz_trace; GZ_ONLY=device timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gz_trace -o gz -- python tools/gz_bench.py 1000000 1 > /dev/null 2>&1
python tools/trace_busy.py gpurun_out/gz_trace --tail 0.25 --chain 14 2>&1 | grep -v "copyBuffer\|fillBuffer" | tee gpurun_out/r04_gz_chain4.txt
rm -rf gpurun_out/gz_trace
