#!/bin/bash
export TMPDIR=/tmp
mkdir -p gpurun_out
( time timeout 1200 python -m pytest tests/test_gpu_parity.py tests/test_gpu_host_layer.py -m gpu -x -q --durations=5 ) > gpurun_out/r02d_pytest.log 2>&1
tail -12 gpurun_out/r02d_pytest.log
for v in "" "FH_NO_SAMPLE=1"; do
for cfg in "--k 31 --n 2000000" "--k 21 --n 200000" "--k 21 --n 1000"; do
  env $v FH_TRACE=${TRACE:-} python bench.py --steps 4 --warmup 1 --no-cpu-baseline --no-extras $cfg 2>/dev/null | tail -1 | python -c "
import sys,json; d=json.loads(sys.stdin.read()); print('$v $cfg:', round(d['value']/1e9,1), 'Gbases/s', d['ms_per_step'], 'ms/step  kernel', d['roofline']['achieved'], 'GB/s  launches/step', d['roofline']['launches']/4)"
done; done 2>&1 | tee gpurun_out/r02d_sample_ab.txt
FH_TRACE=1 python bench.py --steps 1 --warmup 0 --no-cpu-baseline --no-extras --k 31 --n 2000000 2>&1 | grep "\[fh\]" | head -30 | tee gpurun_out/r02d_trace_c3.txt
