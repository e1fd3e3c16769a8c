#!/bin/bash
# counters of the chunk kernel of the device-side gzip inflate (and of k_bgzf_inflate for comparison): one file each, GZ_ONLY=device
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
R=gpurun_out
rm -rf $R/gzp_a $R/gzp_b
GZ_ONLY=device GZ_REPS=1 timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SMEM --kernel-trace -d $R/gzp_a -o p --output-format csv -- python tools/gz_bench.py 1000000 1 > /dev/null 2>&1
GZ_ONLY=device GZ_REPS=1 timeout 600 rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS --kernel-trace -d $R/gzp_b -o p --output-format csv -- python tools/gz_bench.py 1000000 1 > /dev/null 2>&1
python - <<'PY' | tee gpurun_out/r04_gz_pmc.txt
import csv, glob, collections
for d in ("gpurun_out/gzp_a", "gpurun_out/gzp_b"):
    for f in glob.glob(d + "/**/*counter_collection.csv", recursive=True):
        acc = collections.defaultdict(float); n = collections.Counter()
        for r in csv.DictReader(open(f)):
            k = r["Kernel_Name"]
            if "k_gz_chunks" in k or "k_gz_text" in k or "k_gz_win" in k:
                acc[(k.split("(")[0], r["Counter_Name"])] += float(r["Counter_Value"]); n[(k.split("(")[0], r["Counter_Name"])] += 1
        for (k, c), v in sorted(acc.items()):
            print("%-24s %-22s %18.0f  (%d dispatches)" % (k, c, v, n[(k, c)]))
PY
rm -rf $R/gzp_a $R/gzp_b
