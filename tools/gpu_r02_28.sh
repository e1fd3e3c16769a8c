#!/bin/bash
# three-input xor / high-word add of the k >= 25 kernels: parity, then k = 31 and 27 against the build before
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py -q -x -k "not c4_50gbase and not c5_batch" ) 2>&1 | tail -4
for lib in tools/ab/libfinch_old.so finch_rs_amd/libfinch_hip.so; do
  for k in 31 27 24 21; do
    FH_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-extras --k $k --steps 10 --warmup 2 2>/dev/null | grep '^{"metric' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$lib k=$k', round(d['value']/1e9,1), d['roofline']['frac'])"
  done
done
