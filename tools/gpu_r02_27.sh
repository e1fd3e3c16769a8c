#!/bin/bash
# prefilter on the sum of the two fmix states: A/B on one box, then parity
export TMPDIR=/tmp
mkdir -p gpurun_out
for rep in 1 2; do
for lib in tools/ab/libfinch_old.so finch_rs_amd/libfinch_hip.so; do
  for k in 21 31; do
    FH_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-extras --k $k --steps 10 --warmup 2 2>/dev/null | grep '^{"metric' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$lib k=$k', round(d['value']/1e9,1), d['roofline']['frac'], d['roofline'].get('kernel_ms'))"
  done
done
done
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py -q -x -k "not c4_50gbase and not c5_batch" ) 2>&1 | tail -5
