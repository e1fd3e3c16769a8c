#!/bin/bash
# round 4, the shipped kernel of configs[3] (opaque D words in): rocprofv3 --kernel-trace --stats of the default workload and
# ONE counter pass (instruction counts) -> tag r04zzz
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD; export TMPDIR=/tmp; O=$R/gpurun_out; T=r04zzz
cd /tmp
B="python $R/bench.py --no-cpu-baseline --no-extras"
timeout 120 rocprofv3 --kernel-trace --stats -d $O/${T}_stats -o stats --output-format csv -- $B --steps 5 --warmup 1 > $O/${T}_stats.log 2>&1
timeout 120 rocprofv3 --pmc SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU --kernel-trace -d $O/${T}_pmc_insts -o p --output-format csv -- $B --steps 1 --warmup 0 > $O/${T}_pmc_insts.log 2>&1
cd $R
cp $O/${T}_stats/stats_kernel_stats.csv $O/${T}_c4_kernel_stats.csv
python - <<PY
import csv,glob,collections,json
acc=collections.defaultdict(float)
for f in glob.glob("$O/${T}_pmc_insts/**/*counter_collection.csv", recursive=True):
    for r in csv.DictReader(open(f)):
        if "k2_sketch" in r["Kernel_Name"]: acc[r["Counter_Name"]]+=float(r["Counter_Value"])
pos=None
for line in open("$O/${T}_pmc_insts.log"):
    if line.startswith("{"):
        r=json.loads(line)["roofline"]; pos=r["alg_bytes_per_launch"]*r["launches"]
out={"counters":dict(acc),"positions":pos}
if pos:
    wi=pos/64.0
    out.update({k.lower().replace("sq_insts_","")+"_per_wave_iter":round(v/wi,2) for k,v in acc.items()})
json.dump(out,open("$O/${T}_c4_pmc_insts.json","w"),indent=1); print(json.dumps(out))
PY
head -3 $O/${T}_c4_kernel_stats.csv
rm -rf $O/${T}_stats $O/${T}_pmc_insts
