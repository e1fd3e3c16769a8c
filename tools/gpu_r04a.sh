#!/bin/bash
# round 4, first GPU call: the single-synchronisation path -- its own tests, the parity suite, and a first bench line
cd "$GRAFT_REPO_ROOT" 2>/dev/null || cd /root/repo
export TMPDIR=/tmp
O=gpurun_out/r04a
mkdir -p $O
rocm-smi --showproductname 2>/dev/null | head -8 > $O/box.txt; nproc >> $O/box.txt
timeout 1500 python -m pytest tests/test_gpu_fast_path.py -x -q --durations=8 > $O/pytest_fast.txt 2>&1; echo "fast rc=$?" >> $O/pytest_fast.txt
tail -25 $O/pytest_fast.txt
timeout 1500 python -m pytest tests/test_gpu_parity.py tests/test_gpu_errors.py -x -q --durations=5 > $O/pytest_parity.txt 2>&1; echo "parity rc=$?" >> $O/pytest_parity.txt
tail -8 $O/pytest_parity.txt
timeout 600 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline > $O/bench_c4.json 2> $O/bench_c4.err; echo "bench rc=$?"
cat $O/bench_c4.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e9, d['ms_per_step'], d['roofline'], d['sketch_check']['matches_golden'])"
timeout 600 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --workload c2 > $O/bench_c2.json 2> $O/bench_c2.err
cat $O/bench_c2.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print(d['value']/1e9, d['ms_per_step'], d['roofline']['launches'], d['sketch_check']['matches_golden'])"
FH_NO_FAST=1 timeout 600 python bench.py --steps 10 --warmup 2 --no-extras --no-cpu-baseline --workload c2 > $O/bench_c2_nofast.json 2> $O/bench_c2_nofast.err
cat $O/bench_c2_nofast.json | python -c "import sys,json; d=json.loads(sys.stdin.read()); print('nofast', d['value']/1e9, d['ms_per_step'], d['roofline']['launches'], d['sketch_check']['matches_golden'])"
FH_TRACE=1 timeout 300 python bench.py --steps 1 --warmup 1 --no-extras --no-cpu-baseline --workload c2 2>&1 | grep "^\[fh\]" | tail -20 > $O/trace_c2.txt
cat $O/trace_c2.txt
