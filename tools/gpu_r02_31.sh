#!/bin/bash
# K = 25..32 as one 16-wave workgroup per CU with the 4-byte tables replicated 32-fold: parity, then A/B
export TMPDIR=/tmp
mkdir -p gpurun_out
( timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py -q -x -k "not c4_50gbase and not c5_batch" ) 2>&1 | tail -3
for lib in tools/ab/libfinch_c2.so finch_rs_amd/libfinch_hip.so; do
  for k in 31 32 30 28 25 21; do
    FH_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-extras --k $k --steps 10 --warmup 2 2>/dev/null | grep '^{"metric' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$lib k=$k', round(d['value']/1e9,1), d['roofline']['frac'])"
  done
  FH_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-extras --k 31 --n 2000000 --steps 5 --warmup 1 2>/dev/null | grep '^{"metric' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$lib c3 sketch', round(d['value']/1e9,1), d['ms_per_step'])"
done
