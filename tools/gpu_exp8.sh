#!/bin/bash
export TMPDIR=/tmp
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["value"]/1e9,1), d["roofline"]["achieved"], d["roofline"]["launches"], d["ms_per_step"])'
B="python bench.py --steps 3 --warmup 1 --no-cpu-baseline"
echo "== base"; $B 2>/dev/null | python -c "$P"
for f in finch_rs_amd/libfinch_hip_un*.so; do echo "== $f"; FH_LIB=$PWD/$f $B 2>/dev/null | python -c "$P"; done
for w in 20 24 28 32 40; do echo "== waves/CU $w"; FH_WAVES_PER_CU=$w $B 2>/dev/null | python -c "$P"; done
