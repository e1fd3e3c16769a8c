# usage: bash tools/sample_pass_exp.sh   (on the GPU box: what the sampling pre-pass of a large sketch costs by run length;
#        default = one run per resident wave, FH_DEBUG=sample_run_tiles=8 = the fixed run length until round 5)
cd /tmp; export TMPDIR=/tmp
R=$GRAFT_REPO_ROOT
run() { # label k n env...
  lbl=$1; k=$2; n=$3; shift; shift; shift
  rm -rf /tmp/se_$lbl
  env FH_DEBUG="trace ${DBG:-}" "$@" PASSES=6 rocprofv3 --kernel-trace --stats -d /tmp/se_$lbl -o s --output-format csv -- python $R/tools/oversketch_trace.py $k $n > /tmp/se_$lbl.out 2>/tmp/se_$lbl.err
  f=$(find /tmp/se_$lbl -name "s_kernel_stats.csv" | head -1)
  echo "== $lbl (k=$k n=$n): $(grep 'sample:' /tmp/se_$lbl.err | tail -1 | cut -c1-150)"
  python - "$f" <<'PY'
import csv, sys
for r in csv.DictReader(open(sys.argv[1])):
    if "k2_sketch" in r["Name"]:
        print("   %-44s calls %3s  avg %8.3f ms  min %8.3f" % (r["Name"][:44], r["Calls"], float(r["AverageNs"]) / 1e6, float(r["MinNs"]) / 1e6))
PY
  tail -2 /tmp/se_$lbl.out | cut -c1-120
}
run adaptive_k21 21 200000 X=1
DBG=sample_run_tiles=8 run rt8_k21 21 200000
run adaptive_k31 31 2000000 X=1
DBG=sample_run_tiles=8 run rt8_k31 31 2000000
run adaptive_k21_b 21 200000 X=1
DBG=sample_run_tiles=8 run rt8_k21_b 21 200000
