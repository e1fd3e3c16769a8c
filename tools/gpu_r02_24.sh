#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_host_layer.py tests/test_gpu_full_size.py tests/test_abi.py -q -x -m "gpu or not gpu" -k "not c4_50gbase and not c5_batch" ) 2>&1 | tail -5
timeout 900 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/r02v_bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02v_bench.json').read())
print(d['value']/1e9)
print('c3', {a:b for a,b in d['extras']['c3'].items() if a not in ('what','pmc')})
PY
