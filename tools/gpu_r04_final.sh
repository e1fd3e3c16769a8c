#!/bin/bash
# round 4, after the two fixes the kernel fuzzer led to: the evidence set again (tag r04z) and the host-layer fuzzers on fresh seeds
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
bash tools/gpu_r04_evidence.sh r04z
( timeout 1200 python tools/fuzz_params.py 500 950001 2>&1 | tail -2
  FUZZ_FILES=1 timeout 900 python tools/fuzz_device_text.py 400 950002 2>&1 | tail -2
  FUZZ_SHARDED=1 timeout 900 python tools/fuzz_device_text.py 400 950003 2>&1 | tail -2 ) | tee gpurun_out/r04_fuzz_after_fixes.txt
