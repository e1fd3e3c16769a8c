#!/bin/bash
# round 4: fuzz campaign over the paths the round touched (fused epilogues, deferred speculation, host-packed small FASTA,
# fh_process, shared-table kernels), fresh seeds
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
( timeout 1500 python tools/fuzz_params.py 700 940001 2>&1 | tail -2
  FUZZ_FILES=1 timeout 1200 python tools/fuzz_device_text.py 700 940002 2>&1 | tail -2
  FUZZ_SHARDED=1 timeout 1200 python tools/fuzz_device_text.py 700 940003 2>&1 | tail -2
  FH_FUZZ_CASES=1500 FH_FUZZ_SEED=41414 timeout 1500 python -m pytest tests/test_gpu_fuzz.py -x -q -m gpu 2>&1 | tail -2
  timeout 900 python -m pytest tests/test_gpu_host_layer.py tests/test_gpu_process.py tests/test_gpu_full_size.py::test_c1_ecoli_sized_fasta_through_sketch_files -x -q 2>&1 | tail -2 ) | tee gpurun_out/r04_fuzz_campaign.txt
