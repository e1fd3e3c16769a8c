#!/bin/bash
# The round's closing fuzz campaign on the GPU box: the kernel fuzzer and the parity suites with every block forced through the
# segment kernels at four strides (any stride must give the oracle's sketch) and unforced, then the parameter / text / gzip fuzzers.
#   gpurun --timeout 3600 -- 'bash tools/fuzz_campaign.sh [<out name> [<seed offset>]]'   -> gpurun_out/<out name>.txt (r05_fuzz_final)
cd $GRAFT_REPO_ROOT; export TMPDIR=/tmp
O=gpurun_out/${1:-r05_fuzz_final}.txt; : > $O
SO=${2:-0}
IFS=";" read -ra CFGS <<< "${FUZZ_CFGS:-40 616161 1200;100 626262 1200;151 636363 2000;168 646464 1200;61 666666 800;0 656565 1500}"
for cfg in "${CFGS[@]}"; do
  set -- $cfg
  set -- $1 $(($2 + SO)) $3
  # (stride R: every block without a stride through the work-item form of the segment kernels, SEG_RAGGED)
  if [ "$1" = "R" ]; then export FH_DEBUG=seg_ragged=1; elif [ "$1" != "0" ]; then export FH_DEBUG=seg_stride=$1; else unset FH_DEBUG; fi
  echo "== FH_DEBUG=${FH_DEBUG:-unset} seed $2 cases $3" >> $O
  # (tests/test_gpu_segments.py asserts which stride a block went by: only where none is forced)
  SEGT=""; [ "$1" = "0" ] && SEGT=tests/test_gpu_segments.py
  FH_FUZZ_CASES=$3 FH_FUZZ_SEED=$2 timeout 900 python -m pytest tests/test_gpu_fuzz.py tests/test_gpu_parity.py tests/test_gpu_fast_path.py tests/test_gpu_process.py $SEGT -x -q 2>&1 | tail -3 >> $O
done
unset FH_DEBUG
echo "== fuzz_params" >> $O; timeout 600 python tools/fuzz_params.py 2>&1 | tail -3 >> $O
echo "== fuzz_device_text" >> $O; timeout 600 python tools/fuzz_device_text.py 2>&1 | tail -3 >> $O
echo "== fuzz_gzip" >> $O; timeout 600 python tools/fuzz_gzip.py 2>&1 | tail -3 >> $O
echo "== fuzz_batch" >> $O; timeout 900 python tools/fuzz_batch.py 2500 ${SO}7 2>&1 | tail -3 >> $O
echo "== fuzz_files" >> $O; timeout 900 python tools/fuzz_files.py 2500 ${SO}9 2>&1 | tail -3 >> $O
cat $O
