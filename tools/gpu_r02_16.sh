#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 600 python tools/e2e_gz.py 2>&1 | tail -12 | tee gpurun_out/r02p_e2e_gz.txt
( time timeout 600 python -m pytest tests/test_gpu_parity.py -q -x -k "large_sketch_takes or handle_cache" ) 2>&1 | tail -5
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/r02p_bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02p_bench.json').read())
print(d['value']/1e9, d['roofline']['frac'])
for k,v in d['extras'].items(): print(k, {a:b for a,b in v.items() if a not in ('what','pmc')})
PY
