#!/bin/bash
# round 4, second GPU call: the whole GPU suite on the new path
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r04b
mkdir -p $O
nproc > $O/box.txt
timeout 2400 python -m pytest tests -m gpu -x -q --durations=12 > $O/pytest_gpu.txt 2>&1; echo "rc=$?" >> $O/pytest_gpu.txt
tail -30 $O/pytest_gpu.txt
