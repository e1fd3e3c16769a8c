#!/bin/bash
# BGZF inflate with the LDS ring against without: tests, kernel times, bench extras
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 1500 python -m pytest tests/test_gpu_gzip_device.py tests/test_gpu_bgzf_device.py -x -q -m gpu 2>&1 | tail -3
for lib in finch_rs_amd/libfinch_hip.so build/ab/old.so finch_rs_amd/libfinch_hip.so build/ab/old.so; do
  rm -rf gpurun_out/bz_trace
  FH_LIB=$lib timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/bz_trace -o c -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/b.json 2>/dev/null
  python - "$lib" <<'PY'
import csv, glob, json, sys
f = glob.glob("gpurun_out/bz_trace/**/*kernel_stats.csv", recursive=True)
for row in csv.DictReader(open(f[0])):
    if "k_bgzf_inflate" in row["Name"] or "k_gz_chunks" in row["Name"]:
        print("%-32s %-30s calls %4s total %8.2f ms" % (sys.argv[1], row["Name"][:30], row["Calls"], float(row["TotalDurationNs"]) / 1e6))
d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])["extras"]["compressed_fastq"]
print("   ", {k: v for k, v in d.items() if k.endswith("gbases_per_s")})
PY
done 2>&1 | tee gpurun_out/r04_bz_ring_ab.txt
rm -rf gpurun_out/bz_trace
