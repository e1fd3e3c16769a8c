#!/bin/bash
# the CRC-32 kernels (BGZF members, gzip slices): a lane per 64 bytes of a 4 KiB block against a lane per sixty-fourth of a member
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
timeout 900 python -m pytest tests/test_gpu_gzip_device.py tests/test_gpu_bgzf_device.py -x -q -m gpu 2>&1 | tail -2
for lib in finch_rs_amd/libfinch_hip.so; do
  rm -rf gpurun_out/crc_trace
  FH_LIB=$lib timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d gpurun_out/crc_trace -o c -- python bench.py --steps 1 --warmup 1 --no-cpu-baseline > /tmp/b.json 2>/dev/null
  python - "$lib" <<'PY'
import csv, glob, json, sys
f = glob.glob("gpurun_out/crc_trace/**/*kernel_stats.csv", recursive=True)
for row in csv.DictReader(open(f[0])):
    if "crc" in row["Name"]:
        print("%-32s %-40s calls %4s avg %8.1f us" % (sys.argv[1], row["Name"][:40], row["Calls"], float(row["AverageNs"]) / 1e3))
d = json.loads(open("/tmp/b.json").read().strip().splitlines()[-1])["extras"]["compressed_fastq"]
print("   ", {k: v for k, v in d.items() if k.endswith("gbases_per_s")})
PY
done 2>&1 | tee gpurun_out/r04_crc_ab.txt
rm -rf gpurun_out/crc_trace
