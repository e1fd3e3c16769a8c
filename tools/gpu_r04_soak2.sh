#!/bin/bash
# round 4, the shipped library (wide shift for K = 23..28, opaque D words): the kernel fuzzer and the parameter fuzzer on fresh seeds
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
( FH_FUZZ_CASES=1500 FH_FUZZ_SEED=990001 timeout 250 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -1
  timeout 200 python tools/fuzz_params.py 120 990002 2>&1 | tail -1 ) | tee gpurun_out/r04zzz_fuzz.txt
