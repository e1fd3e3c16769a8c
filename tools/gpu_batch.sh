#!/bin/bash
# batch-of-files rates (SURVEY config C5 shape, scaled down) + default bench
export TMPDIR=/tmp
python tools/e2e_files.py 2>&1 | tail -9
python bench.py --steps 5 --warmup 1 --no-cpu-baseline 2>/dev/null | tail -1 | python -c 'import sys,json; d=json.loads(sys.stdin.read()); print(round(d["value"]/1e9,1), d["roofline"]["achieved"], d["roofline"]["launches"], d["ms_per_step"])'
