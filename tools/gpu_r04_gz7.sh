#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gzip_device.py -x -q -m gpu 2>&1 | tail -2
for i in 1 2 3; do FH_TRACE=1 GZ_ONLY=device GZ_REPS=4 timeout 600 python tools/gz_bench.py 1000000 1 2>&1 | grep "complete\|text there\|^device" | tail -5; done 2>&1 | tee gpurun_out/r04_gz_backoff.txt
GZ_ONLY=device timeout 600 python tools/gz_bench.py 1000000 6 | tail -1
GZ_ONLY=device timeout 600 python tools/gz_bench.py 4000000 1 noisy | tail -1
