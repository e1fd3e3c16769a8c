#!/bin/bash
# full -m gpu suite, the driver's bench line with extras, gzip / BGZF file rates
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1500 python -m pytest tests -m gpu -q -x --durations=8 ) > gpurun_out/r02o_pytest.log 2>&1
tail -14 gpurun_out/r02o_pytest.log
timeout 900 python bench.py 2>/dev/null | grep '^{"metric' > gpurun_out/r02o_bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02o_bench.json').read())
print(d['value']/1e9, d['roofline']['frac'], d['cpu_baseline']['value'])
for k,v in d['extras'].items(): print(k, {a:b for a,b in v.items() if a not in ('what','pmc')})
PY
timeout 600 python tools/e2e_gz.py 2>&1 | tail -12 | tee gpurun_out/r02o_e2e_gz.txt
