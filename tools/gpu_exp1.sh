#!/bin/bash
mkdir -p gpurun_out; export TMPDIR=/tmp
B="python bench.py --gbases 2 --steps 3 --warmup 1 --no-cpu-baseline"
for w in 8 16 24 32; do echo "== waves/CU=$w unroll=32"; FH_WAVES_PER_CU=$w $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value']/1e9, d['roofline']['achieved'])"; done
for u in 4 8 16; do for w in 16 32; do echo "== waves/CU=$w unroll=$u"; FH_LIB=$PWD/finch_rs_amd/libfinch_hip_u$u.so FH_WAVES_PER_CU=$w $B 2>/dev/null | python -c "import sys,json; d=json.loads(sys.stdin.readlines()[-1]); print(d['value']/1e9, d['roofline']['achieved'])"; done; done
echo "== rocprof pmc"
cd /tmp; 
rocprofv3 --pmc SQ_WAVES SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc1 -o pmc1 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --gbases 1 --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc1.log 2>&1
rocprofv3 --pmc SQ_INSTS_LDS SQ_ACTIVE_INST_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM SQ_IFETCH SQ_INSTS_SALU --kernel-trace -d $GRAFT_REPO_ROOT/gpurun_out/pmc2 -o pmc2 --output-format csv -- python $GRAFT_REPO_ROOT/bench.py --gbases 1 --steps 1 --warmup 0 --no-cpu-baseline > $GRAFT_REPO_ROOT/gpurun_out/pmc2.log 2>&1
cd $GRAFT_REPO_ROOT; ls -R gpurun_out | head -30
python - <<'PY'
import csv,glob,collections
for d in ["gpurun_out/pmc1","gpurun_out/pmc2"]:
    for f in glob.glob(d+"/**/*counter_collection.csv", recursive=True):
        acc=collections.defaultdict(float); n=collections.Counter()
        for r in csv.DictReader(open(f)):
            if "k2_sketch" in r["Kernel_Name"]:
                acc[r["Counter_Name"]]+=float(r["Counter_Value"]); n[r["Counter_Name"]]+=1
        for k in acc: print(d,k,acc[k],n[k])
PY
