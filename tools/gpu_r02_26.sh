#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_fuzz.py tests/test_gpu_full_size.py -q -x -k "not c4_50gbase and not c5_batch" ) 2>&1 | tail -5
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/r02x_bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02x_bench.json').read())
print(d['value']/1e9)
for k in ('k21_n200000','c3'): print(k, {a:b for a,b in d['extras'][k].items() if a not in ('what','pmc')})
PY
