#!/bin/bash
# round 4: plain gzip inflated on the device -- tests, then timings and a kernel trace
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_gzip_device.py -x -q -m gpu 2>&1 | tail -30 | tee gpurun_out/r04_gz_tests.txt
( timeout 600 python tools/gz_bench.py 1000000 1
  FH_GZ_CHUNK=8192 timeout 600 python tools/gz_bench.py 1000000 1
  FINCH_GZIP_PIECE=4194304 timeout 600 python tools/gz_bench.py 1000000 1
  timeout 600 python tools/gz_bench.py 1000000 6
  FINCH_DEVICE_GZIP=1 GZ_ONLY=device FH_TRACE=1 timeout 600 python tools/gz_bench.py 4000000 1 noisy 2>&1 | grep -v "^\[fh\] launch\|range\|tile\|prune\|finish" | tail -30 ) 2>&1 | tee gpurun_out/r04_gz_bench.txt
rm -rf gpurun_out/gz_trace; GZ_ONLY=device timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gz_trace -o gz -- python tools/gz_bench.py 1000000 1 > /dev/null 2>&1
python - <<'PY' | tee gpurun_out/r04_gz_kernels.txt
import csv, glob, collections
f = glob.glob("gpurun_out/gz_trace/**/*kernel_stats.csv", recursive=True)
for row in list(csv.DictReader(open(f[0])))[:14]:
    print("%-70s calls %6s total %10.3f ms avg %9.1f us" % (row["Name"][:70], row["Calls"], float(row["TotalDurationNs"]) / 1e6, float(row["AverageNs"]) / 1e3))
PY
python tools/trace_busy.py gpurun_out/gz_trace --tail 0.25 --chain 70 > gpurun_out/r04_gz_chain.txt 2>&1
rm -rf gpurun_out/gz_trace
