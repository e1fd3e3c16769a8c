#!/bin/bash
export TMPDIR=/tmp
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["value"]/1e9,1), d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"], d["roofline"]["launches"], d["ms_per_step"])'
B="python bench.py --gbases 4 --steps 2 --warmup 1 --no-cpu-baseline"
echo "== default"; $B 2>/dev/null | python -c "$P"
echo "== range 64M"; FH_MAX_RANGE=67108864 $B 2>/dev/null | python -c "$P"
echo "== range 256M"; FH_MAX_RANGE=268435456 $B 2>/dev/null | python -c "$P"
echo "== waves 16"; FH_WAVES_PER_CU=16 $B 2>/dev/null | python -c "$P"
echo "== waves 24"; FH_WAVES_PER_CU=24 $B 2>/dev/null | python -c "$P"
echo "== waves 8"; FH_WAVES_PER_CU=8 $B 2>/dev/null | python -c "$P"
python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k collisions 2>&1 | tail -30
