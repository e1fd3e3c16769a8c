#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
( echo "default"; for i in 1 2; do GZ_ONLY=device timeout 600 python tools/gz_bench.py 1000000 1 | tail -1; done
  echo "chunk 8192"; for i in 1 2; do FH_GZ_CHUNK=8192 GZ_ONLY=device timeout 600 python tools/gz_bench.py 1000000 1 | tail -1; done
  echo "chunk 8192 piece 4M"; FINCH_GZIP_PIECE=4194304 FH_GZ_CHUNK=8192 GZ_ONLY=device timeout 600 python tools/gz_bench.py 1000000 1 | tail -1
  echo "piece 2M"; FINCH_GZIP_PIECE=2097152 GZ_ONLY=device timeout 600 python tools/gz_bench.py 1000000 1 | tail -1
  echo "chunk 12288 piece 4M"; FINCH_GZIP_PIECE=4194304 FH_GZ_CHUNK=12288 GZ_ONLY=device timeout 600 python tools/gz_bench.py 1000000 1 | tail -1
  echo "level 6 chunk 8192"; FH_GZ_CHUNK=8192 GZ_ONLY=device timeout 600 python tools/gz_bench.py 1000000 6 | tail -1 ) 2>&1 | tee gpurun_out/r04_gz_bench5.txt
