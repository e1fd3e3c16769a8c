#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 900 python -m pytest tests/test_gpu_host_layer.py tests/test_gpu_bgzf_device.py -q -x ) 2>&1 | tail -5
timeout 300 python tools/bgzf_device_file.py --level 6 --reps 3 2>&1 | tail -1
timeout 300 python tools/bgzf_device_file.py --noisy --level 6 --reps 3 2>&1 | tail -1
cd /tmp; rm -rf /tmp/st_bz; timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st_bz -o s --output-format csv -- python $GRAFT_REPO_ROOT/tools/bgzf_device_file.py --level 6 --reps 2 > /tmp/st.log 2>&1; head -8 /tmp/st_bz/*kernel_stats.csv | cut -c1-60,200-330
cd $GRAFT_REPO_ROOT; timeout 600 python bench.py --no-cpu-baseline 2>/dev/null | grep '^{"metric' > gpurun_out/r02w_bench.json
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r02w_bench.json').read())
for k in ('end_to_end_fastq_text','compressed_fastq','c5_batch_1gpu'): print(k, {a:b for a,b in d['extras'][k].items() if a not in ('what','pmc')})
PY
