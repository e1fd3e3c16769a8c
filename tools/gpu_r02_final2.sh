#!/bin/bash
# end-of-round evidence after the multiply-add changes of the sketch kernel: everything of gpu_r02_final.sh plus the
# k = 31 kernel stats / PMC passes
bash tools/gpu_r02_final.sh
timeout 900 bash tools/gpu_bench_full.sh r02z_k31 c2_k31_n1000 --k 31 2>&1 | tail -3
