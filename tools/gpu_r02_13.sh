#!/bin/bash
# PMC counters of k_bgzf_inflate (noisy-quality file: literal-heavy members)
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
i=0
for set in "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE" \
           "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_WAIT_INST_LDS SQ_IFETCH SQ_INSTS_BRANCH" \
           "GRBM_GUI_ACTIVE FETCH_SIZE WRITE_SIZE TCP_TOTAL_CACHE_ACCESSES_sum TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  rm -rf /tmp/pmc_bz
  rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_bz -o p --output-format csv -- python $R/tools/bgzf_device_file.py --noisy --reps 1 > /tmp/pmc_bz_$i.log 2>&1
  f=$(ls /tmp/pmc_bz/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then
    python - "$f" <<'PY' | tee -a $R/gpurun_out/r02m_bgzf_pmc.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); n = collections.Counter()
seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if "bgzf" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"])
    seen.add((k, r["Dispatch_Id"]))
for k in acc:
    nd = len([1 for kk, d in seen if kk == k])
    print(k, "dispatches", nd, {c: round(v) for c, v in acc[k].items()})
PY
  else
    tail -5 /tmp/pmc_bz_$i.log | tee -a $R/gpurun_out/r02m_bgzf_pmc.txt
  fi
done
