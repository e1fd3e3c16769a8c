#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
FH_TRACE=1 GZ_ONLY=device GZ_REPS=2 timeout 600 python tools/gz_bench.py 4000000 1 noisy 2>&1 | grep "complete\|text there\|^device\|gzip on the device" | tail -28 | tee gpurun_out/r04_gz_noisy_trace.txt
