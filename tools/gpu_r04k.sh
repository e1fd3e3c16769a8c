#!/bin/bash
# round 4: the digit-reversed string's words made opaque to the compiler (it folds the upper-half bit field of a forward window
# into a 64-bit shift of the pair pairrev64 leaves: v_lshrrev_b64 + v_and_b32 where one v_bfe_u32 does; k = 21 ISA 58.8 -> 58.0
# VALU per position, 127 VGPRs) -- DOPQ against the shipped build, twice each
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r04k
mkdir -p $O
L=cur=finch_rs_amd/libfinch_hip.so,DOPQ=build/ab/DOPQ.so,cur_again=finch_rs_amd/libfinch_hip.so,DOPQ_again=build/ab/DOPQ.so
timeout 900 python tools/ab_k.py --libs $L --ks 17,21,22,24,31 --gbases 10 2>&1 | tee $O/ab_d_opaque_10g.txt
