#!/bin/bash
export TMPDIR=/tmp
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["value"]/1e9,1), d["roofline"]["achieved"], d["roofline"]["launches"], d["ms_per_step"], d["sketch_check"])'
echo "== k31 n=2M (C3 sketch part)"; python bench.py --steps 2 --warmup 1 --no-cpu-baseline --k 31 --n 2000000 2>&1 | tail -1 | python -c "$P"
echo "== k21 n=200000 (CLI default oversketch)"; python bench.py --steps 2 --warmup 1 --no-cpu-baseline --k 21 --n 200000 2>&1 | tail -1 | python -c "$P"
echo "== k21 n=1000 2 Gbase"; python bench.py --steps 3 --warmup 1 --no-cpu-baseline --gbases 2 2>&1 | tail -1 | python -c "$P"
