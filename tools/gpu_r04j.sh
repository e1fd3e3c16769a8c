#!/bin/bash
# round 4: the canonical word shifted to the top of its two registers (pre_shift = 64 - 2K for K = 17..32: no mask on the upper
# half of either strand's window) -- OLD = the smallest shift for every K (the code as it was), W23 = wide for K >= 23 (the
# kernels that run in two rounds of 16), W17R17 = wide for K >= 17 with K = 17..22 in two rounds as well (one pass of 32
# spills 10-23 registers with the wide shift)
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r04j
mkdir -p $O
L=OLD=build/ab/OLD.so,W23=build/ab/W23.so,W17R17=build/ab/W17R17.so,OLD_again=build/ab/OLD.so
timeout 1200 python tools/ab_k.py --libs $L --ks 17,21,22,23,24,25,28,31 --gbases 10 2>&1 | tee $O/ab_pre_wide_10g.txt
