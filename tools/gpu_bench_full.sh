#!/bin/bash
# full-size bench (configs[1]: 10 Gbase) + rocprofv3 kernel stats + PMC passes for the roofline block
# usage: bash tools/gpu_bench_full.sh <tag>   -> gpurun_out/<tag>_*
TAG=${1:-r01}
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python bench.py --steps 5 --warmup 1 2>&1 | tail -1 | tee gpurun_out/${TAG}_bench_10G.json
cd /tmp
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats -o stats --output-format csv -- python $R/bench.py --steps 5 --warmup 1 --no-cpu-baseline > $R/gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --kernel-trace -d $R/gpurun_out/${TAG}_pmc_fetch -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d $R/gpurun_out/${TAG}_pmc_write -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/${TAG}_pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $R/gpurun_out/${TAG}_pmc_sq -o p --output-format csv -- python $R/bench.py --steps 1 --warmup 0 --no-cpu-baseline > $R/gpurun_out/${TAG}_pmc_sq.log 2>&1
cd $R
python - <<PY
import csv,glob,collections,json
tag="$TAG"
out={}
for d in ["gpurun_out/%s_pmc_fetch"%tag,"gpurun_out/%s_pmc_write"%tag,"gpurun_out/%s_pmc_sq"%tag]:
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(float); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            if "k2_sketch" in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
        for k in acc: out[k]={"sum":acc[k],"dispatches":n[k]}
    for f in glob.glob(d+"/**/*kernel_trace.csv", recursive=True):
        tot=0; cnt=0
        for r in csv.DictReader(open(f)):
            if "k2_sketch" in r["Kernel_Name"]:
                tot+=int(r["End_Timestamp"])-int(r["Start_Timestamp"]); cnt+=1
        out.setdefault("k2_duration_ns",{})[d.split("_pmc_")[-1]]={"sum":tot,"dispatches":cnt}
json.dump(out,open("gpurun_out/%s_pmc_k2.json"%tag,"w"),indent=1)
print(json.dumps(out,indent=1))
PY
head -6 gpurun_out/${TAG}_stats/stats_kernel_stats.csv
