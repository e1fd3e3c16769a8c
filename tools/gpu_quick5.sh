#!/bin/bash
export TMPDIR=/tmp
python -m pytest tests -m gpu -q -x --deselect tests/test_gpu_full_size.py 2>&1 | tail -4
P='import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(round(d["value"]/1e9,1), d["roofline"]["achieved"], d["roofline"]["launches"], d["ms_per_step"])'
echo "== main"; python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | python -c "$P"
echo "== k31 n=2M"; python bench.py --steps 2 --warmup 1 --no-cpu-baseline --k 31 --n 2000000 2>/dev/null | python -c "$P"
echo "== k21 n=200K"; python bench.py --steps 2 --warmup 1 --no-cpu-baseline --n 200000 2>/dev/null | python -c "$P"
