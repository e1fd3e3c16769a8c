#!/bin/bash
export TMPDIR=/tmp
python -m pytest tests -m gpu -q 2>&1 | tail -40
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["value"]/1e9,1), d["roofline"]["achieved"], d["roofline"]["avg_launch_ms"], d["roofline"]["launches"], d["ms_per_step"])'
python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$P"
