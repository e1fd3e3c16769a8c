#!/bin/bash
# round 4: fh_process + trait path, configs[0] test, host-packed small FASTA with scratch, default threads
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
R=$PWD
export TMPDIR=/tmp
O=$R/gpurun_out/r04h
mkdir -p $O
timeout 1800 python -m pytest tests/test_gpu_process.py tests/test_gpu_fast_path.py tests/test_gpu_parity.py tests/test_gpu_host_layer.py "tests/test_gpu_full_size.py::test_c1_ecoli_sized_fasta_through_sketch_files" tests/test_gpu_full_size.py::test_c5_batch_of_fastas_through_sketch_files_scaled -x -q > $O/pytest.txt 2>&1; tail -12 $O/pytest.txt
for t in 0 12 16; do python tools/batch_trace.py 1024 $t; done | tee $O/c5_threads.txt
timeout 900 python bench.py --steps 10 --warmup 2 --no-cpu-baseline > $O/bench.json 2> $O/bench.err; echo "bench rc=$?"
python - <<PY
import json
d = json.load(open("$O/bench.json"))
print("default: %.1f Gbases/s %.3f ms" % (d["value"] / 1e9, d["ms_per_step"]))
for k, v in d.get("extras", {}).items():
    print("  ", k, {kk: vv for kk, vv in v.items() if kk not in ("what", "pmc", "sketch_check")}, (v.get("sketch_check") or {}).get("matches_golden"))
PY
