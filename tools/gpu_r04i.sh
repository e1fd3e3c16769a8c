#!/bin/bash
# round 4, the hot loop's cycles taken apart: measurement-only builds that leave one part of a
# wave-iteration out each (WRONG sketches, timing only) against the shipped build, on one box; then the GPU suite on the
# shipped build.
#   NO_PHASE_A: tiles behind a range's second are never loaded or classified   NO_ADMIT: a bound no hash is under (the branch is there, never taken)
#   NO_LDS: the table records are made of their offsets, no table is read      A_ADMIT / ALL3: the first two / all three
#   FWD_ONLY: the forward strand's word taken as the canonical one             ALL4: all four
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r04i
mkdir -p $O
L=cur=finch_rs_amd/libfinch_hip.so,NO_PHASE_A=build/ab/NO_PHASE_A.so,NO_ADMIT=build/ab/NO_ADMIT.so,NO_LDS=build/ab/NO_LDS.so,FWD_ONLY=build/ab/FWD_ONLY.so,A_ADMIT=build/ab/A_ADMIT.so,ALL3=build/ab/ALL3.so,ALL4=build/ab/ALL4.so,cur_again=finch_rs_amd/libfinch_hip.so
timeout 1200 python tools/ab_k.py --libs $L --ks 21,31 --gbases 10 2>&1 | tee $O/ab_parts_10g.txt
[ "$1" = pmc ] || exit 0
# evidence hygiene: the configs[1] and k = 33 counter sets from THIS round's kernels
bash tools/gpu_bench_full.sh r04i_c2 c2_k21_n1000 --workload c2 > $O/c2_full.log 2>&1; tail -3 $O/c2_full.log
bash tools/gpu_bench_full.sh r04i_k33 c2_k33_n1000 --workload c2 --k 33 > $O/k33_full.log 2>&1; tail -3 $O/k33_full.log
rm -rf gpurun_out/r04i_*_stats gpurun_out/r04i_*_pmc_fetch gpurun_out/r04i_*_pmc_write gpurun_out/r04i_*_pmc_sq
