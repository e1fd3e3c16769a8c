#!/bin/bash
export TMPDIR=/tmp
B="python bench.py --gbases 2 --steps 3 --warmup 1 --no-cpu-baseline"
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["value"]/1e9,1), d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"])'
echo "== base"; $B 2>/dev/null | python -c "$P"
for v in NOHASH NOLDS NOWINDOW NOWINLDS; do echo "== $v"; FH_LIB=$PWD/finch_rs_amd/libfinch_hip_abl_$v.so $B 2>/dev/null | python -c "$P"; done
