#!/bin/bash
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
for lib in finch_rs_amd/libfinch_hip.so build/ab/coop8.so build/ab/coop12.so build/ab/coop48.so; do
  for args in "1000000 1" "1000000 6" "4000000 1 noisy"; do
    rm -rf gpurun_out/gz_trace
    FH_LIB=$lib FINCH_GZIP_PIECE=1000000000 GZ_ONLY=device GZ_REPS=3 timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/gz_trace -o gz -- python tools/gz_bench.py $args > /tmp/gzb.txt 2>&1
    python - "$lib" "$args" <<'PY'
import csv, glob, sys
f = glob.glob("gpurun_out/gz_trace/**/*kernel_stats.csv", recursive=True)
for row in csv.DictReader(open(f[0])):
    if "k_gz_chunks" in row["Name"]:
        print("%-30s %-18s k_gz_chunks: %s calls, total %.2f ms, min %.2f ms" % (sys.argv[1], sys.argv[2], row["Calls"], float(row["TotalDurationNs"]) / 1e6, float(row["MinNs"]) / 1e6))
PY
  done
done 2>&1 | grep k_gz_chunks | tee gpurun_out/r04_gz_ab_coop.txt
rm -rf gpurun_out/gz_trace
