#!/bin/bash
# PMC counters of the k2_sketch dispatches of an arbitrary command (summed over its launches)
# usage (GPU box): bash tools/pmc_k2.sh <tag> <command...>   -> gpurun_out/<tag>_pmc.json
TAG=$1; shift
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_SALU SQ_INSTS_LDS --kernel-trace -d $R/gpurun_out/${TAG}_a -o p --output-format csv -- "$@" > /dev/null 2>&1
rocprofv3 --pmc SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE --kernel-trace -d $R/gpurun_out/${TAG}_b -o p --output-format csv -- "$@" > /dev/null 2>&1
cd $R
python - <<PY
import csv,glob,collections,json
out={}
for d in ["gpurun_out/${TAG}_a","gpurun_out/${TAG}_b"]:
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(float)
        for r in csv.DictReader(open(f)):
            if "k2_sketch" in r["Kernel_Name"]: acc[r["Counter_Name"]]+=float(r["Counter_Value"])
        out.update(acc)
    for f in glob.glob(d+"/**/*kernel_trace.csv", recursive=True):
        out["k2_ms_"+d[-1]]=sum(int(r["End_Timestamp"])-int(r["Start_Timestamp"]) for r in csv.DictReader(open(f)) if "k2_sketch" in r["Kernel_Name"])/1e6
json.dump(out,open("gpurun_out/${TAG}_pmc.json","w"),indent=1)
print("${TAG}", json.dumps(out))
PY
