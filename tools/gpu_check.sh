#!/bin/bash
# one GPU-box session: parity tests, smoke, instruction-rate microbench, short bench.  Logs -> gpurun_out/
mkdir -p gpurun_out
export TMPDIR=/tmp
echo "== rocminfo ==" ; rocminfo 2>/dev/null | grep -E "Marketing Name|Compute Unit|Max Clock" | head -8
echo "== pytest gpu ==" 
timeout 900 python -m pytest tests -m gpu -x -q 2>&1 | tail -30 | tee gpurun_out/pytest_gpu.log
echo "== smoke =="
timeout 300 python __graft_entry__.py smoke 2>&1 | tail -5 | tee gpurun_out/smoke.log
echo "== ubench =="
timeout 300 ./tools/ubench 2.4 2>&1 | tee gpurun_out/ubench.log
echo "== bench (1 Gbase quick) =="
timeout 600 python bench.py --gbases 1 --steps 3 --warmup 1 --cpu-sample-mbases 60 2>&1 | tail -5 | tee gpurun_out/bench_quick.log
