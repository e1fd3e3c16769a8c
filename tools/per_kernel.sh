#!/bin/bash
# Per-kernel breakdown of an arbitrary command: time (rocprofv3 --kernel-trace --stats) and HBM traffic per kernel name
# (two --pmc passes: FETCH_SIZE | WRITE_SIZE; FETCH_SIZE doubled per MI355X_MICROARCH.md's gfx950 correction).
# usage (GPU box): bash tools/per_kernel.sh <tag> <command...>   -> gpurun_out/<tag>_per_kernel.txt
TAG=$1; shift
R=$GRAFT_REPO_ROOT; export TMPDIR=/tmp; cd /tmp
rm -rf /tmp/pk_$TAG; mkdir -p /tmp/pk_$TAG
rocprofv3 --kernel-trace -d /tmp/pk_$TAG/t -o p --output-format csv -- "$@" > /tmp/pk_$TAG/out_t.txt 2>/tmp/pk_$TAG/err_t.txt
rocprofv3 --pmc FETCH_SIZE --kernel-trace -d /tmp/pk_$TAG/f -o p --output-format csv -- "$@" > /dev/null 2>&1
rocprofv3 --pmc WRITE_SIZE --kernel-trace -d /tmp/pk_$TAG/w -o p --output-format csv -- "$@" > /dev/null 2>&1
cd $R
python - <<PY > gpurun_out/${TAG}_per_kernel.txt
import csv,glob,collections
def name(n): return n.split("(")[0].replace("void ","")[:64]
t=collections.defaultdict(lambda:[0,0.0]); f=collections.defaultdict(float); w=collections.defaultdict(float)
for fn in glob.glob("/tmp/pk_${TAG}/t/**/*kernel_trace.csv", recursive=True):
    for r in csv.DictReader(open(fn)):
        k=name(r["Kernel_Name"]); t[k][0]+=1; t[k][1]+=(int(r["End_Timestamp"])-int(r["Start_Timestamp"]))/1e6
for d,acc,c in (("f",f,"FETCH_SIZE"),("w",w,"WRITE_SIZE")):
    for fn in glob.glob("/tmp/pk_${TAG}/%s/**/*counter_collection.csv"%d, recursive=True):
        for r in csv.DictReader(open(fn)):
            if r["Counter_Name"]==c: acc[name(r["Kernel_Name"])]+=float(r["Counter_Value"])
print(open("/tmp/pk_${TAG}/out_t.txt").read().strip())
print("%-64s %7s %10s %12s %12s" % ("kernel","calls","ms","HBM read MB","HBM write MB"))
for k,(c,ms) in sorted(t.items(), key=lambda kv:-kv[1][1]):
    print("%-64s %7d %10.3f %12.1f %12.1f" % (k,c,ms,2*f[k]*1024/1e6,w[k]*1024/1e6))
print("%-64s %7d %10.3f %12.1f %12.1f" % ("TOTAL",sum(c for c,_ in t.values()),sum(ms for _,ms in t.values()),sum(2*v*1024/1e6 for v in f.values()),sum(v*1024/1e6 for v in w.values())))
PY
cat gpurun_out/${TAG}_per_kernel.txt
