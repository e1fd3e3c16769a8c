#!/bin/bash
# full-size bench + rocprofv3 kernel stats + PMC passes for the roofline block
# usage: bash tools/gpu_bench_full.sh <tag> [<pmc key> [bench args...]]   -> gpurun_out/<tag>_*
#   default: configs[3] (50 Gbase, k=21, n=1000), key c4_k21_n1000
#   e.g.     bash tools/gpu_bench_full.sh r03_k31 c2_k31_n1000 --workload c2 --k 31
TAG=${1:-r03}; KEY=${2:-c4_k21_n1000}; shift; shift
ARGS="$@"
mkdir -p gpurun_out; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
python bench.py --steps 5 --warmup 1 --no-extras --no-cpu-baseline --no-live-pmc $ARGS 2>&1 | tail -1 | tee gpurun_out/${TAG}_bench.json
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-extras --no-live-pmc $ARGS"
rocprofv3 --kernel-trace --stats -d $R/gpurun_out/${TAG}_stats -o stats --output-format csv -- $B --steps 5 --warmup 1 > $R/gpurun_out/${TAG}_stats.log 2>&1
rocprofv3 --pmc GRBM_GUI_ACTIVE FETCH_SIZE --kernel-trace -d $R/gpurun_out/${TAG}_pmc_fetch -o p --output-format csv -- $B --steps 1 --warmup 0 > $R/gpurun_out/${TAG}_pmc_fetch.log 2>&1
rocprofv3 --pmc WRITE_SIZE SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d $R/gpurun_out/${TAG}_pmc_write -o p --output-format csv -- $B --steps 1 --warmup 0 > $R/gpurun_out/${TAG}_pmc_write.log 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $R/gpurun_out/${TAG}_pmc_sq -o p --output-format csv -- $B --steps 1 --warmup 0 > $R/gpurun_out/${TAG}_pmc_sq.log 2>&1
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
                out.setdefault("kernel_names", [])
                if r["Kernel_Name"] not in out["kernel_names"]: out["kernel_names"].append(r["Kernel_Name"])
                acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
        for k in acc: out[k]={"sum":acc[k],"dispatches":n[k]}
    for f in glob.glob(d+"/**/*kernel_trace.csv", recursive=True):
        tot=0; cnt=0
        for r in csv.DictReader(open(f)):
            if "k2_sketch" in r["Kernel_Name"]:
                tot+=int(r["End_Timestamp"])-int(r["Start_Timestamp"]); cnt+=1
        out.setdefault("k2_duration_ns",{})[d.split("_pmc_")[-1]]={"sum":tot,"dispatches":cnt}
# positions the profiled launches covered: from the bench line of the same run
for line in open("gpurun_out/%s_pmc_sq.log"%tag):
    if line.startswith("{"):
        r=json.loads(line)["roofline"]; out["positions"]=r["alg_bytes_per_launch"]*r["launches"]; out["launches"]=r["launches"]
json.dump(out,open("gpurun_out/%s_pmc_k2.json"%tag,"w"),indent=1)
print(json.dumps(out,indent=1))
PY
head -6 gpurun_out/${TAG}_stats/stats_kernel_stats.csv
cp gpurun_out/${TAG}_stats/stats_kernel_stats.csv gpurun_out/${TAG}_kernel_stats.csv
POS=$(python -c "import json;print(json.load(open('gpurun_out/${TAG}_pmc_k2.json')).get('positions'))")
echo "KEY=$KEY positions=$POS"
# per-wave-iteration summary of this set -> gpurun_out/${TAG}_pmc_summary.json (merge into profiles/pmc_summary.json with
#   python tools/pmc_summary.py $KEY profiles/${TAG}_pmc_k2.json $POS   after copying the counter file to profiles/)
python tools/pmc_summary.py $KEY gpurun_out/${TAG}_pmc_k2.json $POS --out gpurun_out/${TAG}_pmc_summary.json
