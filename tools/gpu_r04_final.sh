#!/bin/bash
# round 4, final state (plain gzip inflated on the device in): the evidence set (tag r04z), the gzip timings, the fuzzers
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
bash tools/gpu_r04_evidence.sh r04z
( echo "# tools/gz_bench.py <reads> <zlib level> [noisy]: one gzip file through finch_sketch_files, best of 4 calls; device = inflated on the device, host = FINCH_DEVICE_GZIP=0"
  for args in "1000000 1" "1000000 6" "1000000 9" "4000000 1 noisy" "4000000 6 noisy"; do timeout 600 python tools/gz_bench.py $args 2>&1 | tail -3; done
  echo "# the reader's own account of one file (FH_TRACE=1)"
  FH_TRACE=1 GZ_ONLY=device GZ_REPS=3 timeout 600 python tools/gz_bench.py 1000000 1 2>&1 | grep "complete\|text there\|gzip batch:\|gzip on the device" | tail -4 ) 2>&1 | tee gpurun_out/r04_gz_bench.txt
rm -rf gpurun_out/gz_trace; GZ_ONLY=device timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gz_trace -o gz -- python tools/gz_bench.py 1000000 1 > /dev/null 2>&1
python tools/trace_busy.py gpurun_out/gz_trace --tail 0.25 --chain 34 2>&1 | grep -v "copyBuffer\|fillBuffer" | tee gpurun_out/r04_gz_chain.txt
rm -rf gpurun_out/gz_trace
( timeout 900 python tools/fuzz_params.py 250 970001 2>&1 | tail -1
  timeout 1200 python tools/fuzz_gzip.py 300 970002 2>&1 | tail -1 ) | tee gpurun_out/r04y_fuzz.txt
