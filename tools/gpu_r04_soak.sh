#!/bin/bash
# round 4, final state: soak of the fuzzers on fresh seeds
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
( timeout 1500 python tools/fuzz_gzip.py 900 980001 2>&1 | tail -1
  FH_FUZZ_CASES=2000 FH_FUZZ_SEED=980002 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -1
  FUZZ_FILES=1 timeout 900 python tools/fuzz_device_text.py 300 980003 2>&1 | tail -1 ) | tee gpurun_out/r04_soak.txt
