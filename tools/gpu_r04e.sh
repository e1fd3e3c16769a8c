#!/bin/bash
# round 4, fifth GPU call: folded epilogue + two-level select (tests), the hot loop in rounds of 16 / 8 for every K (A/B: steady
# state and the start-up of short launches), the batch in steady state
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r04e
mkdir -p $O
timeout 1500 python -m pytest tests/test_gpu_fast_path.py tests/test_gpu_parity.py -x -q > $O/pytest.txt 2>&1; tail -3 $O/pytest.txt
L=cur=finch_rs_amd/libfinch_hip.so,r16=build/ab/r16.so,r8=build/ab/r8.so
timeout 900 python tools/ab_k.py --libs $L --ks 17,21,22,24,31 --gbases 10 2>&1 | tee $O/ab_rounds_10g.txt
timeout 900 python tools/ab_k.py --libs $L --ks 21,31 --gbases 0.004 --steps 20 2>&1 | tee $O/ab_rounds_4mb.txt
for lib in finch_rs_amd/libfinch_hip.so build/ab/r16.so build/ab/r8.so; do
  echo "== $lib"; for t in 12 16; do FH_LIB=$R/$lib python tools/batch_trace.py 1024 $t; done
done | tee $O/c5_libs.txt
cd /tmp
for lib in finch_rs_amd/libfinch_hip.so build/ab/r8.so; do
  n=$(basename $lib .so)
  FH_LIB=$R/$lib rocprofv3 --kernel-trace --stats -d $O/c5_$n -o t --output-format csv -- python $R/tools/batch_trace.py 1024 12 > $O/c5_trace_$n.txt 2> $O/c5_trace_$n.err
  python $R/tools/trace_busy.py $O/c5_$n --tail 0.45 > $O/c5_busy_$n.txt; cat $O/c5_trace_$n.txt; head -16 $O/c5_busy_$n.txt
  rm -rf $O/c5_$n
done
