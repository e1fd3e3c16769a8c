#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for ck in 0 16384 20480 0 16384; do
  echo "== FH_GZ_CHUNK=$ck"
  if [ $ck = 0 ]; then GZ_ONLY=device timeout 600 python tools/gz_bench.py 4000000 1 noisy | tail -1; GZ_ONLY=device timeout 600 python tools/gz_bench.py 4000000 6 noisy | tail -1
  else FH_GZ_CHUNK=$ck GZ_ONLY=device timeout 600 python tools/gz_bench.py 4000000 1 noisy | tail -1; FH_GZ_CHUNK=$ck GZ_ONLY=device timeout 600 python tools/gz_bench.py 4000000 6 noisy | tail -1; fi
done 2>&1 | tee gpurun_out/r04_gz_ab_chunk_big.txt
