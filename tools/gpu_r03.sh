#!/bin/bash
# round-3 GPU sessions, one stage per invocation:  bash tools/gpu_r03.sh <stage> [args]   (outputs -> gpurun_out/r03_<stage>*)
# (the stages are listed in profiles/README.md next to the files they produced)
STAGE=$1; shift
mkdir -p gpurun_out; export TMPDIR=/tmp
case $STAGE in
ab_rounds)   # parity of the round structure + A/B of the variants
  timeout 900 python -m pytest tests/test_gpu_parity.py -x -q -m gpu -k "all_k or palindrom or golden or longer_sequence" 2>&1 | tail -5 | tee gpurun_out/r03_ab_rounds_pytest.txt
  timeout 900 python tools/ab_k.py --libs base=build/ab/base.so,r8=finch_rs_amd/libfinch_hip.so,r16=build/ab/r16.so,r8w5=build/ab/r8w5.so,r8from22=build/ab/r8from22.so \
      --ks 21,22,24,25,28,31,32 2>&1 | tee gpurun_out/r03_ab_rounds.txt
  ;;
*) echo "unknown stage $STAGE"; exit 2;;
esac
