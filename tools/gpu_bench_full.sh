#!/bin/bash
# full-size bench (configs[1]: 10 Gbase) + rocprofv3 kernel stats + PMC passes for the roofline block
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python bench.py --steps 5 --warmup 1 2>&1 | tail -1 | tee gpurun_out/bench_full.json
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/stats -o stats --output-format csv -- python $R/bench.py --steps 2 --warmup 1 --no-cpu-baseline > $R/gpurun_out/stats.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --kernel-trace -d $R/gpurun_out/pmc_fetch -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU SQ_WAVE_CYCLES SQ_BUSY_CU_CYCLES --kernel-trace -d $R/gpurun_out/pmc_write -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/pmc_write.log 2>&1
cd $R
ls -R gpurun_out/stats | head; 
python - <<'PY'
import csv,glob,collections
for f in glob.glob("gpurun_out/stats/**/*kernel_stats.csv", recursive=True):
    print(open(f).read()[:3000])
for d in ["gpurun_out/pmc_fetch","gpurun_out/pmc_write"]:
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(float); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            key=(r["Kernel_Name"][:40], r["Counter_Name"])
            acc[key]+=float(r["Counter_Value"]); n[key]+=1
        for k in sorted(acc): print(d,k,acc[k],n[k])
    for f in glob.glob(d+"/**/*kernel_trace.csv", recursive=True):
        tot=collections.defaultdict(float); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            tot[r["Kernel_Name"][:40]]+= (int(r["End_Timestamp"])-int(r["Start_Timestamp"])); n[r["Kernel_Name"][:40]]+=1
        for k in tot: print(d,"dur_ns",k,tot[k],n[k])
PY
