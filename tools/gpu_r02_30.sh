#!/bin/bash
# experiment: the sketch kernel without the k2 words' B lookup (wrong hashes; what that LDS read costs)
export TMPDIR=/tmp
for rep in 1 2; do
for lib in tools/ab/libfinch_nob2.so tools/ab/libfinch_bcast.so finch_rs_amd/libfinch_hip.so; do
  for k in 21 31; do
    FH_LIB=$PWD/$lib timeout 300 python bench.py --no-cpu-baseline --no-extras --k $k --steps 10 --warmup 2 2>/dev/null | grep '^{"metric' | python -c "
import sys,json
d=json.loads(sys.stdin.read()); print('$lib k=$k', round(d['value']/1e9,1), d['roofline']['frac'])"
  done
done
done
