#!/bin/bash
export TMPDIR=/tmp
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["value"]/1e9,1), d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"], d["ms_per_step"])'
for ml in 16777216 67108864 268435456; do echo "== max_launch $ml"; python bench.py --steps 5 --warmup 1 --no-cpu-baseline --max-launch $ml 2>/dev/null | python -c "$P"; done
for w in 8 12 16 24 32; do echo "== waves/CU $w"; FH_WAVES_PER_CU=$w python bench.py --steps 3 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$P"; done
