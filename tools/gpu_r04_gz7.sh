#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for piece in 4194304 16777216 33554432 16777216 4194304; do
  echo "== piece $piece"
  FINCH_GZIP_PIECE=$piece FH_TRACE=1 GZ_ONLY=device GZ_REPS=4 timeout 600 python tools/gz_bench.py 1000000 1 2>&1 | grep "complete\|text there\|^device" | tail -5
done 2>&1 | tee gpurun_out/r04_gz_ab_piece.txt
FINCH_GZIP_PIECE=16777216 GZ_ONLY=device timeout 600 python tools/gz_bench.py 1000000 6 | tail -1
FINCH_GZIP_PIECE=16777216 GZ_ONLY=device timeout 600 python tools/gz_bench.py 4000000 1 noisy | tail -1
