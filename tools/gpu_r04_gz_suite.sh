#!/bin/bash
# round 4, with plain gzip on the device: the whole GPU suite, the host-layer fuzzers (their .gz route is the device's now), a gzip fuzzer
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 2700 python -m pytest tests -x -q -m gpu 2>&1 | tail -6 | tee gpurun_out/r04y_pytest_gpu_tail.txt
( timeout 600 python tools/gz_bench.py 1000000 1 const 0 | tail -3 ) 2>&1 | tee gpurun_out/r04_gz_spec_after.txt
rm -rf gpurun_out/gz_trace; GZ_ONLY=device timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gz_trace -o gz -- python tools/gz_bench.py 1000000 1 const 0 > /dev/null 2>&1
python tools/trace_busy.py gpurun_out/gz_trace --tail 0.25 --chain 40 2>&1 | grep -v "copyBuffer\|fillBuffer" | tee -a gpurun_out/r04_gz_spec_after.txt
rm -rf gpurun_out/gz_trace
