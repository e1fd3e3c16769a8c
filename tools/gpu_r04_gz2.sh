#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
GZ_ONLY=device GZ_TRACE=1 timeout 600 python tools/gz_bench.py 1000000 1 2>&1 | tail -60 > gpurun_out/r04_gz_fhtrace.txt
rm -rf gpurun_out/gz_trace; GZ_ONLY=device timeout 600 rocprofv3 --kernel-trace --output-format csv -d gpurun_out/gz_trace -o gz -- python tools/gz_bench.py 1000000 1 > /dev/null 2>&1
python tools/trace_busy.py gpurun_out/gz_trace --tail 0.25 --chain 60 > gpurun_out/r04_gz_chain.txt 2>&1
rm -rf gpurun_out/gz_trace
