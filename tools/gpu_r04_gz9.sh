#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_gzip_device.py tests/test_gpu_bgzf_device.py -x -q -m gpu 2>&1 | tail -3
bash tools/gpu_r04_gz6.sh 2>&1 | grep "k_gz_chunks"
for lib in finch_rs_amd/libfinch_hip.so build/ab/gz_nowin.so; do
  FH_LIB=$lib python bench.py --steps 2 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "
import json,sys
d=json.loads(sys.stdin.read().strip().splitlines()[-1]); e=d['extras']['compressed_fastq']; print('$lib', {k:v for k,v in e.items() if k.endswith('gbases_per_s') or k.endswith('on_device')})"
done
