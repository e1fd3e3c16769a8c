#!/bin/bash
# rocprofv3 evidence for the lane-parallel k_bgzf_inflate: kernel stats of one bgzip'd FASTQ, then two PMC passes
export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
mkdir -p $R/gpurun_out
cd /tmp
for mode in "" "--noisy"; do
  tag=const; [ -n "$mode" ] && tag=noisy
  rm -rf /tmp/st_bz
  timeout 300 rocprofv3 --kernel-trace --stats -d /tmp/st_bz -o s --output-format csv -- python $R/tools/bgzf_device_file.py $mode --level 6 --reps 2 > /tmp/st_bz_$tag.log 2>&1
  f=$(ls /tmp/st_bz/*kernel_stats.csv 2>/dev/null | head -1)
  [ -n "$f" ] && head -8 "$f" > $R/gpurun_out/r02t_bgzf_kernel_stats_$tag.csv
done
rm -f $R/gpurun_out/r02t_bgzf_inflate_pmc.txt
i=0
for set in "SQ_WAVES SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_INSTS_SMEM SQ_INSTS_VMEM_WR SQ_INSTS_VMEM_RD" \
           "SQ_WAVE_CYCLES SQ_BUSY_CYCLES SQ_WAIT_INST_ANY SQ_WAIT_ANY SQ_ACTIVE_INST_ANY SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_INSTS_BRANCH"; do
  i=$((i+1))
  rm -rf /tmp/pmc_bz
  timeout 400 rocprofv3 --pmc $set --kernel-trace -d /tmp/pmc_bz -o p --output-format csv -- python $R/tools/bgzf_device_file.py --noisy --level 6 --reps 1 > /tmp/pmc_bz_$i.log 2>&1
  f=$(ls /tmp/pmc_bz/*counter_collection.csv 2>/dev/null | head -1)
  if [ -n "$f" ]; then
    python - "$f" <<'PY' | tee -a $R/gpurun_out/r02t_bgzf_inflate_pmc.txt
import csv, sys, collections
acc = collections.defaultdict(lambda: collections.defaultdict(float)); seen = set()
for r in csv.DictReader(open(sys.argv[1])):
    k = r["Kernel_Name"].split("(")[0]
    if "bgzf" not in k: continue
    acc[k][r["Counter_Name"]] += float(r["Counter_Value"]); seen.add((k, r["Dispatch_Id"]))
for k in acc:
    print(k, "dispatches", len([1 for kk, d in seen if kk == k]), {c: round(v) for c, v in acc[k].items()})
PY
  else
    tail -3 /tmp/pmc_bz_$i.log | tee -a $R/gpurun_out/r02t_bgzf_inflate_pmc.txt
  fi
done
cat $R/gpurun_out/r02t_bgzf_kernel_stats_const.csv $R/gpurun_out/r02t_bgzf_kernel_stats_noisy.csv
