#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
( for front in 0 1; do echo "== FINCH_GZ_FRONT=$front"
  FINCH_GZ_FRONT=$front timeout 900 python tools/gz_batch.py 64 40000 | tail -2
  FINCH_GZ_FRONT=$front timeout 900 python tools/gz_batch.py 256 4000 | tail -2
  FINCH_GZ_FRONT=$front timeout 900 python tools/gz_batch.py 16 400000 | tail -2; done ) 2>&1 | tee gpurun_out/r04_gz_batch.txt
