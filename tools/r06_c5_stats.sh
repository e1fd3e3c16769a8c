#!/bin/bash
# `bench.py --workload c5` under rocprofv3 --kernel-trace --stats: the batch kernel's average launch duration by the profiler next to the
# HIP-event figure of the same run's bench line (roofline.avg_launch_ms).   gpurun -- 'bash tools/r06_c5_stats.sh <tag>'
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; T=${1:-r06}; export TMPDIR=/tmp
rm -rf /tmp/c5s; cd /tmp
rocprofv3 --kernel-trace --stats -d /tmp/c5s -o stats --output-format csv -- python $R/bench.py --workload c5 --steps 2 --warmup 1 --no-cpu-baseline --no-live-pmc > $R/gpurun_out/${T}_c5_stats_bench.json 2> /tmp/c5s.err
cd $R
cp $(find /tmp/c5s -name "*kernel_stats.csv" | head -1) gpurun_out/${T}_c5_kernel_stats.csv
head -5 gpurun_out/${T}_c5_kernel_stats.csv
python - <<PY
import json
r = json.loads([l for l in open("gpurun_out/${T}_c5_stats_bench.json") if l.startswith("{")][-1])
print(r["config"]["files_per_s"], "files/s;", "roofline:", {k: r["roofline"][k] for k in ("launches", "avg_launch_ms", "kernel_ms_total", "achieved", "frac")})
PY
