#!/bin/bash
# one case of tests/test_gpu_fuzz.py under the A/B knobs of the small-sketch path: which piece makes it fail?
# usage (GPU box): bash tools/fuzz_one.sh <seed> <case>
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
S=$1; C=$2
for env in "" "FH_DEBUG=no_hist" "FH_DEBUG=no_fast" "FH_DEBUG=no_static_units" "FH_DEBUG=no_reset_fold" "FH_DEBUG=no_spec"; do
  r=$(env $env FH_FUZZ_CASES=$((C+1)) FH_FUZZ_SEED=$S timeout 300 python -m pytest "tests/test_gpu_fuzz.py::test_random_configuration[$C]" -x -q -m gpu 2>&1 | grep -E "passed|failed|'mode'|rep" | tail -3 | tr '\n' ' ')
  echo "[$env] $r"
done
